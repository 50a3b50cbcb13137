"""Novelty detectors and the taper functions ``OutOfSampleModel`` scales a model's tendencies with
(external/fv3fit/fv3fit/_shared/novelty_detector.py:20-58, _shared/taper_function.py:23-73,
sklearn/_min_max_novelty_detector.py:47-164, sklearn/_ocsvm_novelty_detector.py:60-207), scored on the device.

The reference keeps a fitted sklearn object in a joblib pickle; here a detector is its few arrays (``arrays.npz`` +
``metadata.yaml``, nothing executable): ``from_sklearn`` exports a fitted scaler / pipeline where sklearn exists,
``fit`` (min-max only) computes the same arrays from a dataset."""
import os
from typing import Callable, Dict, Hashable, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import yaml

from .. import ops
from ..cubedsphere._device import like_input, on_device
from ..xr_compat import DataArray, Dataset, from_compat, to_compat
from . import io
from .predictor import Predictor
from .stacking import Z_DIM_NAMES, column_sources, match_prediction_to_input_coords

Clip = Mapping[Hashable, Mapping[str, Optional[int]]]  # PackerConfig.clip: name -> {"start", "stop", "step"}


def _slice(clip: Optional[Clip], name) -> slice:
    c = (clip or {}).get(name) or {}
    return slice(c.get("start"), c.get("stop"), c.get("step"))


class NoveltyDetector(Predictor):
    """A predictor of a per-column score of being out of sample (novelty_detector.py:20-58): ``predict`` returns
    ``novelty_score`` and ``centered_score``, ``predict_novelties`` adds the 0 / 1 classification against a cutoff."""

    _NOVELTY_OUTPUT_VAR = "is_novelty"
    _SCORE_OUTPUT_VAR = "novelty_score"
    _CENTERED_SCORE_OUTPUT_VAR = "centered_score"
    _ARRAYS_NAME = "arrays.npz"
    _METADATA_NAME = "metadata.yaml"

    def __init__(self, input_variables: Iterable[Hashable], clip: Optional[Clip] = None):
        super().__init__(list(input_variables), [self._NOVELTY_OUTPUT_VAR, self._SCORE_OUTPUT_VAR, self._CENTERED_SCORE_OUTPUT_VAR])
        self.clip = {k: dict(v) for k, v in (clip or {}).items()}
        self._device_arrays: Dict[str, torch.Tensor] = {}

    def _on_device(self, key: str, values) -> torch.Tensor:
        if key not in self._device_arrays:
            self._device_arrays[key] = on_device(np.ascontiguousarray(np.asarray(values, np.float64)))
        return self._device_arrays[key]

    def _columns(self, data):
        """The input variables as clipped ``[feature, sample]`` device arrays, in input order (``stack`` + ``pack``)."""
        x = to_compat(data)
        sources, sample_dims, sizes, _, host_input = column_sources(x, self.input_variables, Z_DIM_NAMES)
        return x, [sources[name][_slice(self.clip, name)] for name in self.input_variables], sample_dims, sizes, host_input

    def _score_dataset(self, x: Dataset, score: torch.Tensor, centered: torch.Tensor, sample_dims, sizes, host_input, like):
        shape = [sizes[d] for d in sample_dims]
        out = Dataset()
        out[self._SCORE_OUTPUT_VAR] = DataArray(like_input(score.reshape(shape), host_input), dims=tuple(sample_dims))
        out[self._CENTERED_SCORE_OUTPUT_VAR] = DataArray(like_input(centered.reshape(shape), host_input), dims=tuple(sample_dims))
        return from_compat(match_prediction_to_input_coords(x, out), like)

    def predict_novelties(self, X, cutoff: float = 0.0) -> Tuple[DataArray, Dataset]:
        diagnostics = to_compat(self.predict(X))
        centered = diagnostics[self._CENTERED_SCORE_OUTPUT_VAR]
        flag = ops.ew("gt_s", on_device(centered.data).contiguous(), scalar=cutoff).to(torch.int64)  # xr.where(score > cutoff, 1, 0)
        diagnostics[self._NOVELTY_OUTPUT_VAR] = centered._replace(data=like_input(flag, centered.data), name=None)
        return from_compat(centered, X), from_compat(diagnostics, X)

    def _dump(self, path: str, arrays: Mapping[str, np.ndarray], metadata: dict) -> None:
        os.makedirs(path, exist_ok=True)
        np.savez(os.path.join(path, self._ARRAYS_NAME), **{k: np.asarray(v, np.float64) for k, v in arrays.items()})
        with open(os.path.join(path, self._METADATA_NAME), "w") as f:
            yaml.safe_dump({"input_variables": list(self.input_variables), "clip": self.clip, **metadata}, f)

    @classmethod
    def _read(cls, path: str):
        if not os.path.exists(os.path.join(path, cls._ARRAYS_NAME)):
            raise ValueError(f"{path} holds no {cls._ARRAYS_NAME}: a detector pickled by the reference (sklearn + joblib) must be "
                             f"exported with {cls.__name__}.from_sklearn(...).dump(path) where sklearn can unpickle it")
        with open(os.path.join(path, cls._METADATA_NAME)) as f:
            metadata = yaml.safe_load(f)
        with np.load(os.path.join(path, cls._ARRAYS_NAME), allow_pickle=False) as z:
            arrays = {k: z[k] for k in z.files}
        return metadata, arrays


@io.register("minmax")
class MinMaxNoveltyDetector(NoveltyDetector):
    """score = max(0, max_f x' - 1) + max(0, -min_f x') over a column's features scaled to the training range, x' =
    ``MinMaxScaler.transform(x)`` (sklearn/_min_max_novelty_detector.py:94-121); > 0: some feature is out of range."""

    def __init__(self, input_variables: Iterable[Hashable], scale: np.ndarray, offset: np.ndarray, clip: Optional[Clip] = None):
        """``scale`` / ``offset``: the fitted scaler's ``scale_`` / ``min_`` over the packed features."""
        super().__init__(input_variables, clip)
        self.scale_, self.min_ = np.asarray(scale, np.float64), np.asarray(offset, np.float64)
        if self.scale_.shape != self.min_.shape or self.scale_.ndim != 1:
            raise ValueError("scale and offset must be 1-D arrays of one length (the packed features)")

    @classmethod
    def from_sklearn(cls, input_variables, scaler, clip: Optional[Clip] = None) -> "MinMaxNoveltyDetector":
        return cls(input_variables, scaler.scale_, scaler.min_, clip)

    @classmethod
    def fit(cls, input_variables, X, clip: Optional[Clip] = None) -> "MinMaxNoveltyDetector":
        """``MinMaxScaler().fit`` on the packed training columns: scale_ = 1 / (max - min) (1 where they coincide), min_ =
        -min * scale_ (sklearn/preprocessing/_data.py, feature_range (0, 1)).  Training is host work."""
        probe = cls(input_variables, np.ones(1), np.zeros(1), clip)
        _, columns, *_ = probe._columns(X)
        packed = np.concatenate([c.cpu().numpy() for c in columns], axis=0)  # [feature, sample]
        lo, hi = np.nanmin(packed, axis=1).astype(np.float64), np.nanmax(packed, axis=1).astype(np.float64)
        span = hi - lo
        span[span == 0.0] = 1.0
        return cls(input_variables, 1.0 / span, -lo * (1.0 / span), clip)

    def predict(self, data):
        x, columns, sample_dims, sizes, host_input = self._columns(data)
        if sum(int(c.shape[0]) for c in columns) != self.scale_.shape[0]:
            raise ValueError(f"the inputs pack to {sum(int(c.shape[0]) for c in columns)} features, the detector was fitted on {self.scale_.shape[0]}")
        scale, offset = self._on_device("scale", self.scale_), self._on_device("offset", self.min_)
        bounds = np.cumsum([0] + [int(c.shape[0]) for c in columns])
        score = ops.minmax_score(columns, [scale[a:b] for a, b in zip(bounds[:-1], bounds[1:])],
                                 [offset[a:b] for a, b in zip(bounds[:-1], bounds[1:])])
        return self._score_dataset(x, score, score, sample_dims, sizes, host_input, data)

    def dump(self, path: str) -> None:
        self._dump(path, {"scale": self.scale_, "offset": self.min_}, {})

    @classmethod
    def load(cls, path: str) -> "MinMaxNoveltyDetector":
        metadata, arrays = cls._read(path)
        return cls(metadata["input_variables"], arrays["scale"], arrays["offset"], metadata.get("clip"))


@io.register("ocsvm")
class OCSVMNoveltyDetector(NoveltyDetector):
    """score = -``Pipeline(StandardScaler(), OneClassSVM(kernel="rbf")).score_samples``; centred on the largest score of
    the training data (sklearn/_ocsvm_novelty_detector.py:104-160)."""

    def __init__(self, input_variables: Iterable[Hashable], mean: np.ndarray, scale: np.ndarray, support_vectors: np.ndarray,
                 dual_coef: np.ndarray, gamma: float, maximum_training_score: float, clip: Optional[Clip] = None):
        super().__init__(input_variables, clip)
        self.mean_, self.scale_ = np.asarray(mean, np.float64), np.asarray(scale, np.float64)
        self.support_vectors_ = np.atleast_2d(np.asarray(support_vectors, np.float64))
        self.dual_coef_ = np.asarray(dual_coef, np.float64).reshape(-1)
        self.gamma, self.maximum_training_score = float(gamma), float(maximum_training_score)
        nf = self.mean_.shape[0]
        if self.scale_.shape != (nf,) or self.support_vectors_.shape[1:] != (nf,) or self.dual_coef_.shape != self.support_vectors_.shape[:1]:
            raise ValueError("mean / scale [feature], support_vectors [n_sv, feature], dual_coef [n_sv] do not fit together")

    @classmethod
    def from_sklearn(cls, input_variables, pipeline, maximum_training_score: float, clip: Optional[Clip] = None) -> "OCSVMNoveltyDetector":
        scaler, svm = pipeline.steps[0][1], pipeline.steps[-1][1]
        if svm.kernel != "rbf":
            raise ValueError(f"only the rbf kernel is implemented, got {svm.kernel!r}")
        return cls(input_variables, scaler.mean_, scaler.scale_, svm.support_vectors_, svm.dual_coef_, svm._gamma, maximum_training_score, clip)

    def predict(self, data):
        x, columns, sample_dims, sizes, host_input = self._columns(data)
        nf, n = sum(int(c.shape[0]) for c in columns), int(columns[0].shape[1])
        if nf != self.mean_.shape[0]:
            raise ValueError(f"the inputs pack to {nf} features, the detector was fitted on {self.mean_.shape[0]}")
        packed = torch.empty((nf, n), dtype=torch.float64, device=columns[0].device)
        row = 0
        for c in columns:
            packed[row:row + c.shape[0]].copy_(c)
            row += int(c.shape[0])
        score = ops.ocsvm_score(packed, self._on_device("mean", self.mean_), self._on_device("scale", self.scale_),
                                self._on_device("sv", self.support_vectors_), self._on_device("coef", self.dual_coef_), self.gamma)
        centered = ops.ew("add_s", score, scalar=-self.maximum_training_score)
        return self._score_dataset(x, score, centered, sample_dims, sizes, host_input, data)

    def dump(self, path: str) -> None:
        self._dump(path, {"mean": self.mean_, "scale": self.scale_, "support_vectors": self.support_vectors_, "dual_coef": self.dual_coef_},
                   {"gamma": self.gamma, "maximum_training_score": self.maximum_training_score})

    @classmethod
    def load(cls, path: str) -> "OCSVMNoveltyDetector":
        metadata, arrays = cls._read(path)
        return cls(metadata["input_variables"], arrays["mean"], arrays["scale"], arrays["support_vectors"], arrays["dual_coef"],
                   metadata["gamma"], metadata["maximum_training_score"], metadata.get("clip"))


# ---- taper functions (taper_function.py:23-73): novelty score -> the fraction of the tendency that is kept ----
def _ew(op: str, score: DataArray, scalar: float = 0.0) -> DataArray:
    return score._replace(data=like_input(ops.ew(op, on_device(score.data).contiguous(), scalar=scalar), score.data), name=None)


def taper_mask(novelty_score, cutoff: float = 0, **kwargs):
    """0 where the score exceeds the cutoff, else 1."""
    s = to_compat(novelty_score)
    keep = 1 - ops.ew("gt_s", on_device(s.data).contiguous(), scalar=cutoff).to(torch.int64)
    return from_compat(s._replace(data=like_input(keep, s.data), name=None), novelty_score)


def taper_ramp(novelty_score, ramp_min: float = 0, ramp_max: float = 1, **kwargs):
    """clip((ramp_max - score) / (ramp_max - ramp_min), 0, 1)."""
    s = to_compat(novelty_score)
    unclipped = _ew("div_s", _ew("add_s", _ew("mul_s", s, -1.0), ramp_max), ramp_max - ramp_min)
    return from_compat(_ew("clip01", unclipped), novelty_score)


def taper_decay(novelty_score, threshold: float = 0, rate: float = 0.5, **kwargs):
    """minimum(rate ** (score - threshold), 1)."""
    s = to_compat(novelty_score)
    return from_compat(_ew("minimum_s", _ew("pow_base_s", _ew("add_s", s, -threshold), rate), 1.0), novelty_score)


_TAPERS = {f.__name__: f for f in (taper_mask, taper_ramp, taper_decay)}


def get_taper_function(name: str = taper_mask.__name__, config: Optional[dict] = None) -> Callable:
    if name not in _TAPERS:
        raise ValueError("Incorrect tapering name")
    taper_func, config = _TAPERS[name], dict(config or {})
    return lambda x: taper_func(x, **config)
