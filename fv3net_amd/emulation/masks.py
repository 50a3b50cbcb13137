"""Generic masks applied to the emulator outputs (external/emulation/emulation/masks.py:1-76), on the
device.  ``TimeMask`` (the online schedule) lives in ``schedule.py``."""
from typing import Callable, Iterable, Optional, Union

import torch

from .. import _lib
from ..cubedsphere._device import like_input, on_device
from ..ops import _ptr, _stream
from .hook import FortranState

Mask = Callable[[FortranState, FortranState], FortranState]
_CODE = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def compose_masks(funcs: Iterable[Mask]) -> Mask:
    """Compose multiple masks; first masks are applied first."""
    func_list = list(funcs)

    def composed(state: FortranState, emulator: FortranState):
        out: FortranState = emulator
        for func in func_list:
            out = func(state, out)
        return out

    return composed


def _float_dev(a):
    t = on_device(a)
    if t.dtype not in _CODE:
        t = t.to(torch.float64)
    return t.contiguous()


class RangeMask:
    def __init__(self, key: str, min: Optional[float] = None, max: Optional[float] = None) -> None:
        self.min = min
        self.max = max
        self.key = key

    def __call__(self, state: FortranState, emulator: FortranState) -> FortranState:
        out = {**emulator}
        if self.min is None and self.max is None:
            return out
        x = out[self.key]
        if not hasattr(x, "shape") or getattr(x, "ndim", 0) == 0:  # plain numbers stay on the host
            if self.min is not None:
                x = max(x, self.min)
            if self.max is not None:
                x = min(x, self.max)
            out[self.key] = x
            return out
        t = _float_dev(x)
        res = torch.empty_like(t)
        _lib.call_on(t.device, "fv3hip_clamp", _ptr(t), _CODE[t.dtype], t.numel(), float(self.min if self.min is not None else 0.0),
                  float(self.max if self.max is not None else 0.0), int(self.min is not None), int(self.max is not None),
                  _ptr(res), _stream(t.device))
        out[self.key] = like_input(res, x)
        return out


class LevelMask:
    """Levels ``[start, stop)`` of the emulator field are replaced by the Fortran state's values (or
    another state field, or a constant); the result is float64 as in the reference."""

    def __init__(self, key: str, start: Optional[int], stop: Optional[int], fill_value: Union[float, str, None] = None):
        self.key = key
        self.start = start
        self.stop = stop
        self.fill_value = fill_value

    def __call__(self, state: FortranState, emulator: FortranState) -> FortranState:
        field = _float_dev(emulator[self.key])
        n0 = int(field.shape[0])
        n1 = int(field.numel() // max(n0, 1))
        start, stop, _ = slice(self.start, self.stop).indices(n0)
        src, fill = None, 0.0
        if self.fill_value is None:
            src = _float_dev(state[self.key])
        elif isinstance(self.fill_value, str):
            src = _float_dev(state[self.fill_value])
        else:
            fill = float(self.fill_value)
        if src is not None and tuple(src.shape) != tuple(field.shape):
            raise ValueError(f"shape mismatch: {tuple(src.shape)} vs {tuple(field.shape)}")
        out = torch.empty(field.shape, dtype=torch.float64, device=field.device)
        _lib.call_on(field.device, "fv3hip_level_fill", _ptr(field), _CODE[field.dtype], _ptr(src), _CODE[src.dtype] if src is not None else 0,
                  fill, n0, n1, start, stop, _ptr(out), _stream(field.device))
        return {**emulator, self.key: like_input(out, emulator[self.key])}
