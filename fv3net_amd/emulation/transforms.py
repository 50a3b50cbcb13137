"""The tensor transforms a hook configuration may carry (``zhao_carr_emulation.model.tensor_transform``,
external/emulation/emulation/config.py:120,145-161): transforms that need no fitting -- ``Difference``,
``CloudWaterDiffPrecpd`` and ``TransformedVariableConfig`` with ``LogTransform`` / ``LimitValueTransform``
(external/fv3fit/fv3fit/emulation/transforms/transforms.py:17-158, factories.py:59-73) -- applied around the model exactly as
``emulation.models.transform_model`` does (models.py:56-65): ``forward`` on the inputs, the model, ``backward`` on inputs and
predictions together.  ``ConditionallyScaled`` needs data to be built (``factory.build({})`` fails in the reference too):
refused when the configuration is read.

The arrays are whatever the hook hands the model: device tensors (``HipEmulator`` keeps the state on the GPU; every
operation here is then one launch of the library's elementwise kernel, ``fv3hip_ew``) or numpy arrays.
"""
import dataclasses
from typing import Dict, List, Mapping, Optional, Sequence, Union

import numpy as np


def _is_device(a) -> bool:
    return hasattr(a, "is_cuda")


def _ew(op: str, a, b=None, scalar: float = 0.0):
    """One elementwise step on device tensors (any strides: a transposed view is handled through its buffer) or numpy."""
    if _is_device(a):
        from .. import ops

        if a.dim() == 2 and not a.is_contiguous() and a.t().is_contiguous():  # the hook's [sample, feature] views
            return _ew(op, a.t(), None if b is None else b.t(), scalar).t()
        return ops.ew(op, a.contiguous(), None if b is None else b.contiguous(), scalar=scalar)
    a = np.asarray(a)
    if op == "sub":
        return a - np.asarray(b)
    if op == "add":
        return a + np.asarray(b)
    if op == "mul_s":
        return a.dtype.type(scalar) * a
    if op == "log_floor_s":
        return np.log(np.where(a < scalar, a.dtype.type(scalar), a))
    if op == "exp":
        return np.exp(a)
    if op == "relu_threshold_s":
        return np.where(a > scalar, a, a.dtype.type(0))
    if op == "below_s":
        return np.where(a < scalar, a, a.dtype.type(0))
    raise ValueError(op)


@dataclasses.dataclass
class Difference:
    """``to = after - before`` (transforms.py:17-58)."""

    to: str
    before: str
    after: str

    def forward(self, x: Mapping) -> Dict:
        x = {**x}
        x[self.to] = _ew("sub", x[self.after], x[self.before])
        return x

    def backward(self, y: Mapping) -> Dict:
        y = {**y}
        y[self.after] = _ew("add", y[self.before], y[self.to])
        return y


@dataclasses.dataclass
class CloudWaterDiffPrecpd:
    """transforms.py:61-108: the cloud after gscond is ``cloud_input - sphum_source``."""

    to: str
    sphum_source: str
    cloud_input: str
    cloud_after_precpd: str

    def _cloud_after_gscond(self, x):
        return _ew("add", x[self.cloud_input], _ew("mul_s", x[self.sphum_source], scalar=-1.0))

    def forward(self, x: Mapping) -> Dict:
        x = {**x}
        x[self.to] = _ew("sub", x[self.cloud_after_precpd], self._cloud_after_gscond(x))
        return x

    def backward(self, y: Mapping) -> Dict:
        y = {**y}
        y[self.cloud_after_precpd] = _ew("add", self._cloud_after_gscond(y), y[self.to])
        return y


@dataclasses.dataclass
class LogTransform:
    """``y = log(max(x, epsilon))``, ``x = exp(y)`` (transforms.py:111-129)."""

    epsilon: float = 1e-30

    def forward(self, x):
        return _ew("log_floor_s", x, scalar=self.epsilon)

    def backward(self, x):
        return _ew("exp", x)


@dataclasses.dataclass
class LimitValueTransform:
    """forward: identity; backward: the value where ``lower < y < upper``, 0 elsewhere (transforms.py:131-158)."""

    lower: Optional[float] = 0.0
    upper: Optional[float] = None

    def forward(self, x):
        return x

    def backward(self, x):
        if self.lower is not None:
            x = _ew("relu_threshold_s", x, scalar=self.lower)
        if self.upper is not None:
            x = _ew("below_s", x, scalar=self.upper)
        return x


@dataclasses.dataclass
class TransformedVariableConfig:
    """factories.py:59-73 / transforms.py:165-190 (``UnivariateTransform``)."""

    source: str
    transform: Union[LogTransform, LimitValueTransform]
    to: Optional[str] = None

    def forward(self, x: Mapping) -> Dict:
        out = {**x}
        out[self.to or self.source] = self.transform.forward(x[self.source])
        return out

    def backward(self, y: Mapping) -> Dict:
        out = {**y}
        out[self.source] = self.transform.backward(y[self.to or self.source])
        return out


class ComposedTransform:
    """transforms.py:227-245: every transform whose inputs are present, forward in order, backward in reverse."""

    def __init__(self, transforms: Sequence):
        self.transforms = list(transforms)

    def forward(self, x: Mapping) -> Dict:
        for t in self.transforms:
            try:
                x = t.forward(x)
            except KeyError:
                pass
        return dict(x)

    def backward(self, y: Mapping) -> Dict:
        for t in self.transforms[::-1]:
            try:
                y = t.backward(y)
            except KeyError:
                pass
        return dict(y)


def _univariate_from_dict(d: Mapping):
    keys = set(d)
    if keys <= {"epsilon"}:
        return LogTransform(**d)
    if keys <= {"lower", "upper"}:
        return LimitValueTransform(**d)
    raise ValueError(f"unknown univariate transform {dict(d)!r}: expected {{epsilon}} (LogTransform) or {{lower, upper}} (LimitValueTransform)")


def transform_from_dict(d: Mapping):
    """One entry of ``tensor_transform`` (the reference lets dacite pick the dataclass whose fields match)."""
    keys = set(d)
    if {"to", "before", "after"} == keys:
        return Difference(**d)
    if {"to", "sphum_source", "cloud_input", "cloud_after_precpd"} == keys:
        return CloudWaterDiffPrecpd(**d)
    if "transform" in keys and "source" in keys and keys <= {"source", "transform", "to"}:
        return TransformedVariableConfig(source=d["source"], transform=_univariate_from_dict(d["transform"]), to=d.get("to"))
    if "condition_on" in keys or "bins" in keys:
        raise NotImplementedError(
            "ConditionallyScaled has to be fitted to data (factories.py:96-157); a hook configuration can carry only "
            "transforms that build from nothing -- the reference's ModelConfig.build fails on it as well.  A saved "
            "emulator carries its conditional scaling in its own spec.yaml.")
    raise ValueError(f"unknown tensor transform {dict(d)!r}")


def transforms_from_config(entries: Optional[List[Mapping]]) -> ComposedTransform:
    return ComposedTransform([transform_from_dict(e) for e in (entries or [])])
