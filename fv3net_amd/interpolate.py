"""``vcm.interpolate.interpolate_1d`` for spatially varying output levels
(external/vcm/vcm/interpolate.py:105-235), on the device: the field is interpolated along the one
dimension in which ``x`` and ``xp`` differ, in place in whatever layout the arrays have (the reference
swaps that axis last, flattens to [sample, level] and calls ``mappm.interpolate_2d``).

The reference's other branch -- 1-D ``xp``, the same output levels for every column, which
``interpolate_to_pressure_levels`` takes -- goes through ``metpy.interpolate.interpolate_1d``
(metpy==1.1.0, not part of the reference tree): linear interpolation along the axis with NaN outside the
column's range.  Here it runs through the same kernel with the levels broadcast to the columns; it is pinned by
the reference's known answers for that branch (external/vcm/tests/test_interpolate.py:85-96, 135-147), beyond
which it is parity-unpinned against metpy."""
from typing import Optional

import numpy as np

from . import ops
from .cubedsphere._device import like_input, on_device
from .xr_compat import DataArray, Dataset, from_compat, to_compat


def _interpolate_da(xp: DataArray, x: DataArray, y: DataArray) -> DataArray:
    old = set(x.dims) - set(xp.dims)
    new = set(xp.dims) - set(x.dims)
    if len(old) != 1 or len(new) != 1:
        raise ValueError("x and xp must share all dimensions except one")
    old_dim, new_dim = old.pop(), new.pop()
    if set(y.dims) != set(x.dims):
        raise ValueError("the field must share dimensions with x")
    order = list(y.dims)
    xt = x.transpose(*order)
    xpt = xp.transpose(*[new_dim if d == old_dim else d for d in order])
    axis = order.index(old_dim)
    res = ops.interpolate_2d(on_device(xpt.data), on_device(xt.data), on_device(y.data), z_axis=axis)
    dims = tuple(new_dim if d == old_dim else d for d in order)
    coords = {k: v for k, v in y.coords.items() if k != old_dim}
    return DataArray(like_input(res, y.data), dims=dims, coords=coords, name=y.name, attrs=y.attrs)


def interpolate_1d(xp, x, field, dim: Optional[str] = None):
    """Interpolate ``field`` (DataArray, or Dataset: every variable sharing ``x``'s dims) from the
    coordinate ``x`` to the levels ``xp``; NaN outside each column's range."""
    p, c, f = to_compat(xp), to_compat(x), to_compat(field)
    if p.ndim == 1:
        if dim is None:
            raise ValueError("dim argument needed for 1D xp")
        one = lambda da: _interpolate_constant_levels(p, c, da, dim)  # noqa: E731
    else:
        one = lambda da: _interpolate_da(p, c, da)  # noqa: E731
    if isinstance(f, Dataset):
        out = Dataset(attrs=f.attrs)
        for v in f:
            out[v] = one(f[v]) if set(f[v].dims) >= set(c.dims) else f[v]
        return from_compat(out, field)
    return from_compat(one(f), field)


def _interpolate_constant_levels(xp: DataArray, x: DataArray, y: DataArray, dim: str) -> DataArray:
    """interpolate.py:153-179: the same output levels for every column; the output dimension takes ``dim``'s place
    and carries the levels as its coordinate."""
    out_dim = xp.dims[0]
    if not set(y.dims) >= set(x.dims):
        raise ValueError("the field must share dimensions with x")
    order = list(y.dims)
    axis = order.index(dim)
    # x is broadcast against the field along the dims only the field has (a time axis on the field but not on delp),
    # as xr.apply_ufunc does for the reference (interpolate.py:165-179)
    xt = on_device(x.transpose(*[d for d in order if d in x.dims]).data)
    if len(x.dims) != len(order):
        xt = xt.reshape([y.sizes[d] if d in x.dims else 1 for d in order]).expand(*[y.sizes[d] for d in order]).contiguous()
    levels = on_device(xp.data).to(xt.dtype)
    shape = [1] * len(order)
    shape[axis] = levels.numel()
    target = list(xt.shape)
    target[axis] = levels.numel()
    xpt = levels.reshape(shape).expand(*target).contiguous()  # (a copy of the levels per column: memory only)
    res = ops.interpolate_2d(xpt, xt, on_device(y.data), z_axis=axis)
    dims = tuple(out_dim if d == dim else d for d in order)
    coords = {k: v for k, v in y.coords.items() if k != dim}
    coords[out_dim] = np.asarray(xp.values)
    return DataArray(like_input(res, y.data), dims=dims, coords=coords, name=y.name, attrs=y.attrs)


# for use in regridding values to the same vertical grid [Pa]: the levels of the ERA-Interim reanalysis (interpolate.py:30-73)
PRESSURE_GRID = DataArray(np.array([
    300.0, 500.0, 700.0, 1000.0, 2000.0, 3000.0, 5000.0, 7000.0, 10000.0, 12500.0, 15000.0, 17500.0, 20000.0, 22500.0, 25000.0,
    30000.0, 35000.0, 40000.0, 45000.0, 50000.0, 55000.0, 60000.0, 65000.0, 70000.0, 75000.0, 77500.0, 80000.0, 82500.0, 85000.0,
    87500.0, 90000.0, 92500.0, 95000.0, 97500.0, 100000.0]), dims=["pressure"])


def interpolate_to_pressure_levels(field, delp, levels=PRESSURE_GRID, dim: str = "pfull", ptop: float = 300.0):
    """Regrid an atmospheric field on hybrid levels to fixed pressure levels (interpolate.py:77-102): linear in the
    Simmons-Burridge midpoint pressure of each column."""
    from .thermo import pressure_at_midpoint_log

    return interpolate_1d(levels, pressure_at_midpoint_log(delp, toa_pressure=ptop, dim=dim), field, dim=dim)
