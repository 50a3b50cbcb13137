"""``TransformedPredictor`` and ``OutOfSampleModel`` (external/fv3fit/fv3fit/_shared/models.py:279-337, 340-440),
registered under the reference's names.  Their array work -- the named data transforms, the novelty scores, the taper
values and the tapered tendencies -- runs on the device (``data_transform.py``, ``novelty.py``)."""
import dataclasses
import os
from typing import Callable, Optional, Sequence

import yaml

from .. import ops
from ..cubedsphere._device import like_input, on_device
from ..xr_compat import DataArray, Dataset, from_compat, merge, to_compat
from . import io
from .data_transform import ChainedDataTransform, DataTransform
from .models import _load_yaml
from .novelty import NoveltyDetector, get_taper_function, taper_mask
from .predictor import Predictor


@io.register("output_transformed_model")
class TransformedPredictor(Predictor):
    _CONFIG_FILENAME = "output_transformed_model.yaml"
    _BASE_MODEL_SUBDIR = "base_model_data"

    def __init__(self, base_model: Predictor, transforms: Sequence[DataTransform]):
        """``transforms`` are applied in order to the base model's predictions (merged with the inputs); their outputs
        join the prediction."""
        self.base_model = base_model
        self.transforms = list(transforms)
        self.output_transform = ChainedDataTransform(self.transforms)
        inputs_for_derived = self.output_transform.input_variables
        derived_outputs = self.output_transform.output_variables
        input_variables = (set(base_model.input_variables) | set(inputs_for_derived)) - set(base_model.output_variables)
        output_variables = set(base_model.output_variables) | set(derived_outputs)
        super().__init__(sorted(input_variables), sorted(output_variables))

    def predict(self, X):
        x = to_compat(X)
        prediction = to_compat(self.base_model.predict(X))
        # xr.merge([prediction, X], compat="override"): a name in both is the prediction's
        transform_inputs = merge([x, prediction])
        transformed = self.output_transform.apply(transform_inputs)
        outputs = Dataset({name: transformed[name] for name in self.output_transform.output_variables})
        return from_compat(merge([prediction, outputs]), X)

    def dump(self, path: str):
        base_model_path = os.path.join(path, self._BASE_MODEL_SUBDIR)
        os.makedirs(path, exist_ok=True)
        io.dump(self.base_model, base_model_path)
        with open(os.path.join(path, self._CONFIG_FILENAME), "w") as f:
            yaml.safe_dump({"base_model": base_model_path, "transforms": [dataclasses.asdict(t) for t in self.transforms]}, f)

    @classmethod
    def load(cls, path: str) -> "TransformedPredictor":
        config = _load_yaml(path, cls._CONFIG_FILENAME)
        base_model = io.load(os.path.join(path, cls._BASE_MODEL_SUBDIR))
        return cls(base_model, [DataTransform(name=t["name"], kwargs=dict(t.get("kwargs") or {})) for t in config["transforms"]])


def _scale_by_columns(da: DataArray, taper: DataArray) -> DataArray:
    """``da * taper`` with ``taper`` over a subset of ``da``'s dims (the horizontal ones), in float64 as xarray's product
    of a float32 field and an integer / float64 taper is."""
    extra = [d for d in da.dims if d not in taper.dims]
    if set(taper.dims) - set(da.dims):
        raise ValueError(f"taper values over {taper.dims} cannot scale an output over {da.dims}")
    import torch

    shared = list(taper.dims)
    a = ops.cast(on_device(da.transpose(*extra, *shared).data), torch.float64).contiguous()
    b = ops.cast(on_device(taper.data), torch.float64).contiguous()
    n_extra = 1
    for d in extra:
        n_extra *= da.sizes[d]
    res = ops.ew("mul", a.reshape(n_extra, 1, -1), b.reshape(1, -1)).reshape(a.shape)
    out = DataArray(like_input(res, da.data), dims=tuple(extra) + tuple(shared), coords=da.coords, attrs={})
    return out.transpose(*da.dims)


@io.register("out_of_sample")
class OutOfSampleModel(Predictor):
    _TAPER_VALUES_OUTPUT_VAR = "taper_values"
    _CONFIG_FILENAME = "out_of_sample_model.yaml"

    def __init__(self, base_model: Predictor, novelty_detector: NoveltyDetector, cutoff: float = 0,
                 taper: Optional[Callable[[DataArray], DataArray]] = None):
        """``base_model``'s outputs, scaled column by column by ``taper(centred novelty score)``; the default taper
        suppresses them entirely where the score exceeds ``cutoff``."""
        self.base_model = base_model
        self.novelty_detector = novelty_detector
        self.cutoff = cutoff
        self.taper = taper or get_taper_function(taper_mask.__name__, {"cutoff": cutoff})
        inputs = set(base_model.input_variables) | set(novelty_detector.input_variables)
        outputs = set(base_model.output_variables) | set(novelty_detector.output_variables) | {self._TAPER_VALUES_OUTPUT_VAR}
        super().__init__(input_variables=tuple(sorted(inputs)), output_variables=tuple(sorted(outputs)))

    def predict(self, X):
        base_predict = to_compat(self.base_model.predict(X))
        centered_scores, diagnostics = self.novelty_detector.predict_novelties(X, cutoff=self.cutoff)
        diagnostics = to_compat(diagnostics)
        taper_values = to_compat(self.taper(to_compat(centered_scores)))
        diagnostics[self._TAPER_VALUES_OUTPUT_VAR] = taper_values
        tapered = Dataset(attrs=base_predict.attrs)
        for name in self.base_model.output_variables:
            tapered[name] = _scale_by_columns(base_predict[name], taper_values)
        return from_compat(merge([tapered, diagnostics]), X)

    def dump(self, path):
        raise NotImplementedError("no dump method yet for this class, you can define one manually using instructions at "
                                  "http://vulcanclimatemodeling.com/docs/fv3fit/composite-models.html")

    @classmethod
    def load(cls, path: str) -> "OutOfSampleModel":
        config = _load_yaml(path, cls._CONFIG_FILENAME)
        base_model = io.load(config["base_model_path"])
        novelty_detector = io.load(config["novelty_detector_path"])
        cutoff = config.get("cutoff", 0)
        assert isinstance(novelty_detector, NoveltyDetector)
        tapering_config = {
            "name": taper_mask.__name__, "cutoff": cutoff, "ramp_min": cutoff,
            "ramp_max": 1 if cutoff == 0 else max(cutoff * 2, cutoff / 2), "threshold": cutoff,
            **config.get("tapering_function", {}),
        }
        taper = get_taper_function(tapering_config["name"], tapering_config)
        return cls(base_model, novelty_detector, cutoff=cutoff, taper=taper)
