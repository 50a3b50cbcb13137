"""The composite predictors of fv3fit (external/fv3fit/fv3fit/_shared/models.py:19-107, 223-276,
442-520; configs _shared/config.py:11-24, 116-142), registered under the reference's names so that
model directories written for them load here.  Their array work -- vertical tapering, ensemble
mean / median, output squashing -- runs on the device on the base models' predictions.
``DerivedModel`` (models.py:110-220) lives in ``derived.py`` with the part of ``vcm.DerivedMapping`` it needs."""
import dataclasses
import os
from typing import Hashable, Iterable, Mapping, Sequence, Set

import numpy as np
import yaml

from .. import ops
from ..cubedsphere._device import like_input, on_device
from ..xr_compat import DataArray, Dataset, from_compat, merge, to_compat
from . import io
from .predictor import Predictor

_NO_DUMP = ("no dump method yet for this class, you can define one manually using instructions at "
            "http://vulcanclimatemodeling.com/docs/fv3fit/composite-models.html")


def _load_yaml(path: str, filename: str) -> dict:
    with open(os.path.join(path, filename)) as f:
        return yaml.safe_load(f)


def vertical_tapering_scale_factors(n_levels: int, cutoff: int, rate: float) -> np.ndarray:
    """external/vcm/vcm/calc/calc.py:52-56: exp((z - cutoff) / rate) above ``cutoff``, 1 from there on."""
    z = np.arange(n_levels)
    return np.hstack([np.exp((z[slice(None, cutoff)] - cutoff) / rate), np.ones(n_levels - cutoff)])


@dataclasses.dataclass
class TaperConfig:
    cutoff: int
    rate: float
    taper_dim: str = "z"

    def apply(self, data: DataArray) -> DataArray:
        d = to_compat(data)
        axis = d.get_axis_num(self.taper_dim)
        scaling = vertical_tapering_scale_factors(d.sizes[self.taper_dim], self.cutoff, self.rate)
        res = ops.level_scale(on_device(d.data), on_device(scaling), axis)
        return from_compat(d._replace(data=like_input(res, d.data)), data)


@dataclasses.dataclass
class SquashedOutputConfig:
    squash_by_name: Hashable
    squash_threshold: float
    squash_to: float = 0.0
    additional_squash_target_names: Sequence[Hashable] = ()

    def squash(self, predictions):
        p = to_compat(predictions)
        out = Dataset(attrs=p.attrs)
        for name in p:
            out[name] = p[name]
        by = p[self.squash_by_name]
        keep = ops.ew("gt_s", on_device(by.data), scalar=self.squash_threshold)  # predictions[by] > threshold
        for name in [self.squash_by_name] + list(self.additional_squash_target_names):
            target = p[name].transpose(*by.dims)
            t = on_device(target.data)
            res = ops.ew("where_s", t, keep.to(t.dtype), scalar=self.squash_to)
            out[name] = target._replace(data=like_input(res, target.data)).transpose(*p[name].dims)
        return from_compat(out, predictions)


def _members(models) -> tuple:
    members = tuple(models)
    if not members:
        raise ValueError("at least one model must be given")
    return members


def _union_variables(models):
    """(sorted inputs, sorted outputs) over a group of predictors -- what every composite advertises."""
    inputs = sorted({name for m in models for name in m.input_variables})
    outputs = sorted({name for m in models for name in m.output_variables})
    return tuple(inputs), tuple(outputs)


@io.register("combined_output_model")
class CombinedOutputModel(Predictor):
    _CONFIG_FILENAME = "combined_output_model.yaml"

    def __init__(self, models: Iterable[Predictor]):
        self._models = _members(models)
        # every output name may come from one member only: count the claims, report the names claimed more than once
        claims = {}
        for model in self._models:
            for name in model.output_variables:
                claims[name] = claims.get(name, 0) + 1
        common_outputs = {name for name, count in claims.items() if count > 1}
        if common_outputs:
            raise ValueError(f"All models being combined must have different outputs, got {common_outputs} multiple times.")
        inputs, outputs = _union_variables(self._models)
        super().__init__(input_variables=inputs, output_variables=outputs)

    def predict(self, X):
        """Merge predictions of all models into a single dataset."""
        return from_compat(merge([to_compat(m.predict(X)) for m in self._models]), X)

    def dump(self, path):
        raise NotImplementedError(_NO_DUMP)

    @classmethod
    def load(cls, path: str) -> "CombinedOutputModel":
        config = _load_yaml(path, cls._CONFIG_FILENAME)
        return cls([io.load(p) for p in config["models"]])


@io.register("tapered_model")
class TaperedModel(Predictor):
    _CONFIG_FILENAME = "tapered_model.yaml"

    def __init__(self, model, tapering: Mapping[str, TaperConfig]):
        unknown = [name for name in tapering if name not in model.output_variables]
        if unknown:
            raise KeyError(f"Tapered variable {unknown[0]} not in model output variables.")
        self.model, self.tapering = model, tapering
        inputs, outputs = _union_variables([model])
        super().__init__(input_variables=inputs, output_variables=outputs)

    @classmethod
    def load(cls, path: str) -> "TaperedModel":
        config = _load_yaml(path, cls._CONFIG_FILENAME)
        model = io.load(config["model"])
        return cls(model, {name: TaperConfig(**c) for name, c in config["tapering"].items()})

    def predict(self, X):
        """Predict an output dataset and taper outputs"""
        output = to_compat(self.model.predict(X))
        out = Dataset(attrs=output.attrs)
        for name in output:
            out[name] = self.tapering[name].apply(output[name]) if name in self.tapering else output[name]
        return from_compat(out, X)

    def dump(self, path):
        raise NotImplementedError(_NO_DUMP)


@io.register("ensemble")
class EnsembleModel(Predictor):
    _CONFIG_FILENAME = "ensemble_model.yaml"

    def __init__(self, models: Iterable[Predictor], reduction: str):
        self._models = _members(models)
        if reduction.lower() not in ("mean", "median"):
            raise NotImplementedError(f"Got reduction {reduction}: only mean, median supported")
        self._reduction = reduction
        outputs = set(self._models[0].output_variables)
        odd = next((m for m in self._models[1:] if set(m.output_variables) != outputs), None)
        if odd is not None:
            raise ValueError(f"all models in ensemble must have same outputs, got {outputs} and {set(odd.output_variables)}")
        inputs, all_outputs = _union_variables(self._models)
        super().__init__(input_variables=inputs, output_variables=all_outputs)

    def predict(self, X):
        """Member predictions reduced along a new 'member' dimension."""
        outputs = [to_compat(m.predict(X)) for m in self._models]
        first = outputs[0]
        out = Dataset(attrs=first.attrs)
        for name in first:
            members = [on_device(o[name].transpose(*first[name].dims).data) for o in outputs]
            res = ops.member_reduce(members, "median" if self._reduction == "median" else "mean")
            out[name] = first[name]._replace(data=like_input(res, first[name].data))
        return from_compat(out, X)

    def dump(self, path):
        raise NotImplementedError(_NO_DUMP)

    @classmethod
    def load(cls, path: str) -> "EnsembleModel":
        config = _load_yaml(path, cls._CONFIG_FILENAME)
        return cls([io.load(p) for p in config["models"]], config["reduction"])


@io.register("squashed_output_model")
class SquashedOutputModel(Predictor):
    _CONFIG_FILENAME = "squashed_output_model.yaml"

    def __init__(self, base_model: Predictor, squashing: Sequence[SquashedOutputConfig]):
        self._validate(squashing, base_model.output_variables)
        self._base_model = base_model
        self._squashing = squashing
        super().__init__(input_variables=base_model.input_variables, output_variables=base_model.output_variables)

    def predict(self, X):
        squashed_predictions = self._base_model.predict(X)
        for config in self._squashing:
            squashed_predictions = config.squash(squashed_predictions)
        return squashed_predictions

    @classmethod
    def load(cls, path: str) -> "SquashedOutputModel":
        config = _load_yaml(path, cls._CONFIG_FILENAME)
        base_model = io.load(config["base_model_path"])
        return cls(base_model, [SquashedOutputConfig(**c) for c in config["squashing"]])

    def dump(self, path: str):
        raise NotImplementedError(_NO_DUMP)

    @staticmethod
    def _validate(squashing_configs: Sequence[SquashedOutputConfig], output_variables: Iterable[Hashable]):
        squash_targets: Set[Hashable] = set()
        squash_by_names: Set[Hashable] = set()
        for config in squashing_configs:
            if config.squash_by_name not in output_variables:
                raise ValueError(f"The squash by variable {config.squash_by_name} must among the set of model output variable names.")
            if config.squash_by_name in squash_by_names:
                raise ValueError(f"Only one squashing rule per output variable; {config.squash_by_name} appears twice.")
            target_overlap = set(config.additional_squash_target_names).intersection(squash_targets)
            if len(target_overlap) > 0:
                raise ValueError(f"Each output variable may only be targeted by one squash; {target_overlap} targeted by more than one.")
            squash_targets.update(config.additional_squash_target_names)
            squash_by_names.add(config.squash_by_name)
        squash_by_in_targets = squash_by_names.intersection(squash_targets)
        if len(squash_by_in_targets) > 0:
            raise ValueError(f"Squash by variables may not also be squash targets to avoid order dependence; {squash_by_in_targets} in both.")
