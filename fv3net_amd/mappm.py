"""Drop-in for the reference's f2py extension module ``mappm``
(external/mappm/mappm/mappm.f90; built by external/mappm/setup.py).

``import fv3net_amd.mappm as mappm; mappm.mappm(p_in, f_in, p_out, 1, n_columns, iv, kord, ptop)``
has the f2py signature the reference calls (external/vcm/vcm/cubedsphere/regridz.py:332-334,
external/vcm/tests/test_mappm.py:13): arrays are ``[n_columns, levels]`` of any float dtype and
either memory order, the result is float32 ``[n_columns, kn]``.
"""
import numpy as np
import torch

from . import ops
from .cubedsphere._device import like_input, on_device


def mappm(p_in, f_in, p_out, i1, i2, iv, kord, ptop):
    """Mass-conserving PPM remap of ``f_in`` (layer means between the interface pressures ``p_in``)
    onto the layers bounded by ``p_out``.  ``i1``/``i2`` select the 1-based column range as in the
    Fortran (``i1=1, i2=n_columns`` for all); ``ptop`` is unused (as in the reference)."""
    p1, f1, p2 = (x if isinstance(x, torch.Tensor) else np.asarray(x) for x in (p_in, f_in, p_out))
    if p1.ndim != 2 or f1.ndim != 2 or p2.ndim != 2:
        raise ValueError("mappm expects 2-d arrays [column, level]")
    i1, i2 = int(i1), int(i2)
    lo, hi = i1 - 1, i2
    if lo < 0 or hi > p1.shape[0] or hi < lo:
        raise ValueError(f"column range i1={i1}, i2={i2} outside 1..{p1.shape[0]}")
    sel = slice(lo, hi)
    res = ops.mappm(on_device(p1[sel]), on_device(f1[sel]), on_device(p2[sel]), iv=int(iv), kord=int(kord), z_axis=-1)
    return like_input(res, f_in if isinstance(f_in, torch.Tensor) else np.empty(0))


def interpolate_2d(xp, x, y, fill_value=np.nan):
    """The other routine of the reference's native module (interpolate_2d.f90:1-28; f2py call at
    external/vcm/vcm/interpolate.py:165-169): ``[m, n]`` arrays, linear interpolation of each row of
    ``y(x)`` onto the row of ``xp``, ``fill_value`` outside the row's range; float64 ``[m, n_out]``."""
    a, b, c = (t if isinstance(t, torch.Tensor) else np.asarray(t) for t in (xp, x, y))
    if a.ndim != 2 or b.ndim != 2 or c.ndim != 2:
        raise ValueError("interpolate_2d expects 2-d arrays [row, point]")
    res = ops.interpolate_2d(on_device(a), on_device(b), on_device(c), fill_value=float(fill_value), z_axis=-1)
    return like_input(res, y if isinstance(y, torch.Tensor) else np.empty(0))
