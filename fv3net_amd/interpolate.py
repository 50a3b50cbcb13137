"""``vcm.interpolate.interpolate_1d`` for spatially varying output levels
(external/vcm/vcm/interpolate.py:105-235), on the device: the field is interpolated along the one
dimension in which ``x`` and ``xp`` differ, in place in whatever layout the arrays have (the reference
swaps that axis last, flattens to [sample, level] and calls ``mappm.interpolate_2d``).

The reference's other branch -- 1-D ``xp`` -- goes through ``metpy.interpolate.interpolate_1d``, which is
not part of the reference tree; it is not provided here."""
from typing import Optional

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
        raise NotImplementedError("1-D output levels go through metpy in the reference, which is not part of it; "
                                  "broadcast xp to the columns' shape to use the native path")
    if isinstance(f, Dataset):
        out = Dataset(attrs=f.attrs)
        for v in f:
            out[v] = _interpolate_da(p, c, f[v]) if set(f[v].dims) >= set(c.dims) else f[v]
        return from_compat(out, field)
    return from_compat(_interpolate_da(p, c, f), field)
