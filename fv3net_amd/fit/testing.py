"""``ConstantOutputPredictor``: the reference's own fake model backend
(external/fv3fit/fv3fit/testing.py:36-132), used to exercise callers of the Predictor API."""
import os
from typing import Hashable, Iterable, Mapping, Sequence, Union

import numpy as np
import yaml

from ..xr_compat import DataArray, Dataset, from_compat, to_compat
from . import io
from .predictor import Predictor
from .novelty import NoveltyDetector
from .stacking import Z_DIM_NAMES, match_prediction_to_input_coords


@io.register("constant-output")
class ConstantOutputPredictor(Predictor):
    """Predicts constant values (zero unless ``set_outputs`` was called); vector outputs take
    their vertical size from the input."""

    _CONFIG_FILENAME = "config.yaml"

    def __init__(self, input_variables: Iterable[Hashable], output_variables: Iterable[Hashable],
                 unstacked_dims: Sequence[str] = ("z",)):
        super().__init__(list(input_variables), list(output_variables))
        self._outputs: Mapping[str, Union[np.ndarray, float]] = {}
        self._unstacked_dims = list(unstacked_dims)

    def set_outputs(self, **outputs: Union[np.ndarray, float]):
        self._outputs.update(outputs)

    def predict(self, X):
        x = to_compat(X)
        first = x[list(self.input_variables)[0]]
        zdims = [d for d in first.dims if d in self._unstacked_dims]
        sample_dims = [d for d in first.dims if d not in zdims]
        out = Dataset()
        for name in self.output_variables:
            value = np.asarray(self._outputs.get(name, 0.0))
            if value.ndim == 0:
                shape = [first.sizes[d] for d in sample_dims]
                out[name] = DataArray(np.full(shape, float(value)), dims=sample_dims)
            else:
                shape = [first.sizes[d] for d in sample_dims] + [value.shape[0]]
                out[name] = DataArray(np.broadcast_to(value, shape).copy(), dims=sample_dims + zdims[:1])
        return from_compat(match_prediction_to_input_coords(x, out), X)

    def dump(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self._CONFIG_FILENAME), "w") as f:
            yaml.safe_dump(
                {"input_variables": list(self.input_variables), "output_variables": list(self.output_variables),
                 "unstacked_dims": self._unstacked_dims,
                 "outputs": {k: np.asarray(v).tolist() for k, v in self._outputs.items()}}, f)

    @classmethod
    def load(cls, path: str) -> "ConstantOutputPredictor":
        with open(os.path.join(path, cls._CONFIG_FILENAME)) as f:
            config = yaml.safe_load(f)
        obj = cls(config["input_variables"], config["output_variables"], config.get("unstacked_dims", ("z",)))
        obj.set_outputs(**{k: np.asarray(v) for k, v in config.get("outputs", {}).items()})
        return obj



@io.register("constant-output-novelty")
class ConstantOutputNoveltyDetector(NoveltyDetector):
    """The reference's fake detector (external/fv3fit/fv3fit/testing.py:135-170): scores are always 0, so the cutoff
    alone decides what counts as a novelty."""

    def __init__(self, input_variables: Iterable[Hashable]):
        super().__init__(input_variables=input_variables)

    def predict(self, data):
        x = to_compat(data)
        first = x[next(iter(self.input_variables))]
        dims = [d for d in first.dims if d not in Z_DIM_NAMES]
        zeros = DataArray(np.zeros([first.sizes[d] for d in dims], dtype=first.dtype), dims=dims,
                          coords={k: v for k, v in first.coords.items() if k in dims})
        out = Dataset()
        out[self._SCORE_OUTPUT_VAR] = zeros
        out[self._CENTERED_SCORE_OUTPUT_VAR] = zeros
        return from_compat(out, data)

    def dump(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "attrs.yaml"), "w") as f:
            yaml.safe_dump({"input_variables": list(self.input_variables)}, f)

    @classmethod
    def load(cls, path: str) -> "ConstantOutputNoveltyDetector":
        with open(os.path.join(path, "attrs.yaml")) as f:
            return cls(**yaml.safe_load(f))
