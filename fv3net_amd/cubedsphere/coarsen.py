"""Horizontal coarse-graining with the interface of ``vcm.cubedsphere.coarsen``
(external/vcm/vcm/cubedsphere/coarsen.py), computed by the HIP kernels in ``libfv3hip.so``.

Same function names, argument meaning and error behaviour as the reference; objects are
``fv3net_amd.xr_compat`` DataArrays / Datasets, or real xarray objects when xarray is installed.
"""
from typing import Callable, Dict, Hashable, List, Mapping, Optional, Union

import numpy as np
import torch

from .. import ops
from ..xr_compat import DataArray, Dataset, from_compat, to_compat
from ._device import float_tensor, horizontal_last, like_input, on_device
from .constants import COORD_X_OUTER, COORD_Y_OUTER

NUM_TILES = 6
SUBTILE_FILE_PATTERN = "{prefix}.tile{tile:d}.nc.{subtile:04d}"
STAGGERED_DIMS = [COORD_X_OUTER, COORD_Y_OUTER]

CoordFunc = Union[str, Callable, Mapping[Hashable, Union[str, Callable]]]


# ---------------------------------------------------------------------------------------------
# coordinates
# ---------------------------------------------------------------------------------------------
def coarsen_coords_coord_func(coordinate: np.ndarray, axis=-1) -> np.ndarray:
    """coarsen.py:109-132: ((first value of each window - 1) // factor + 1) as float32."""
    return ((coordinate[:, 0] - 1) // coordinate.shape[1] + 1).astype(int).astype(np.float32)


def coarsen_coords(coarsening_factor: int, reference_subtile, dims: List[Hashable]) -> Dict[Hashable, np.ndarray]:
    """coarsen.py:83-106."""
    reference_subtile = to_compat(reference_subtile)
    result = {}
    for dim in dims:
        c = np.asarray(reference_subtile.coords[dim])
        result[dim] = ((c[::coarsening_factor] - 1) // coarsening_factor + 1).astype(int).astype(np.float32)
    return result


def add_coordinates(reference_obj, coarsened_obj, coarsening_factor: int, dims: List[Hashable]):
    """coarsen.py:135-161."""
    ref, out = to_compat(reference_obj), to_compat(coarsened_obj)
    coords = coarsen_coords(coarsening_factor, ref, dims)
    if isinstance(out, DataArray):
        result = out.assign_coords(coords)
    else:
        result = Dataset({k: v.assign_coords({d: c for d, c in coords.items() if d in v.dims}) for k, v in out.items()},
                         attrs=out.attrs)
    return from_compat(result, coarsened_obj)


_NP_BY_NAME = {"mean": np.mean, "median": np.median, "max": np.max, "min": np.min, "sum": np.sum}


def _reduce_coord(values: np.ndarray, factor: int, function) -> np.ndarray:
    """Apply an xarray-style coord_func (name or callable) to a 1-d dimension coordinate."""
    if isinstance(function, str):
        n = values.shape[0]
        pad = (-n) % factor  # boundary="pad": pad with NaN and use the nan-skipping reduction
        v = values.astype(np.float64) if pad and not np.issubdtype(values.dtype, np.floating) else values
        if pad:
            v = np.concatenate([v, np.full(pad, np.nan, dtype=v.dtype)])
        fn = getattr(np, "nan" + function) if pad else _NP_BY_NAME[function]
        return fn(v.reshape(-1, factor), axis=-1)
    n = (values.shape[0] // factor) * factor
    tail = values[n:]
    windows = values[:n].reshape(-1, factor)
    if tail.size:  # block_reduce pads the ragged end with cval = NaN
        padded = np.concatenate([tail.astype(np.float64), np.full(factor - tail.size, np.nan)])
        windows = np.concatenate([windows.astype(np.float64), padded[None, :]])
    return function(windows, -1)


def _coord_function_for(coord_func: CoordFunc, dim: Hashable):
    if hasattr(coord_func, "keys") and hasattr(coord_func, "__getitem__"):
        return coord_func.get(dim, "mean")
    return coord_func


def _coarsened_coords(da: DataArray, block_sizes: Mapping[Hashable, int], coord_func: CoordFunc):
    coords = {}
    for dim, c in da.coords.items():
        if dim in block_sizes and dim in da.dims:
            coords[dim] = np.asarray(_reduce_coord(np.asarray(c), block_sizes[dim], _coord_function_for(coord_func, dim)))
        else:
            coords[dim] = c
    return coords


def _propagate_attrs(reference_obj, obj):
    if isinstance(reference_obj, Dataset):
        for variable in reference_obj:
            if variable in obj:
                obj[variable].attrs = dict(reference_obj[variable].attrs)
    obj.attrs = dict(reference_obj.attrs)
    return obj


def _map_dataset(obj, func):
    """Apply a DataArray function to a DataArray, or to every variable of a Dataset."""
    if isinstance(obj, Dataset):
        out = Dataset(attrs=obj.attrs)
        for name, da in obj.items():
            out[name] = func(da)
        return out
    return func(obj)


# ---------------------------------------------------------------------------------------------
# weighted averages
# ---------------------------------------------------------------------------------------------
def _weights_tensor(weights: DataArray, outer: List[Hashable], y_dim, x_dim, what: str):
    extra = [d for d in weights.dims if d not in outer and d not in (y_dim, x_dim)]
    if extra or y_dim not in weights.dims or x_dim not in weights.dims:
        raise ValueError(
            f"{what} dims {weights.dims} must contain {(y_dim, x_dim)} and otherwise be a subset of "
            f"the field's dims {tuple(outer) + (y_dim, x_dim)}"
        )
    w_outer = [d for d in outer if d in weights.dims]
    w = float_tensor(on_device(weights.transpose(*w_outer, y_dim, x_dim).data))
    if w_outer != outer[: len(w_outer)]:
        # not a leading subset of the field's outer dims: align with singleton axes and let the
        # array layer broadcast
        shape = [weights.sizes[d] if d in w_outer else 1 for d in outer] + list(w.shape[-2:])
        w = w.reshape(shape)
    return w


def _weighted_block_average_da(da: DataArray, weights: DataArray, factor: int, x_dim, y_dim, coord_func):
    if x_dim not in da.dims or y_dim not in da.dims:
        raise ValueError(f"field {da.name!r} with dims {da.dims} lacks the horizontal dims {(x_dim, y_dim)}")
    t, outer = horizontal_last(da, y_dim, x_dim)
    w = _weights_tensor(weights, outer, y_dim, x_dim, "weights")
    res = ops.weighted_block_average(float_tensor(t), w, factor)
    out = DataArray(like_input(res, da.data), dims=tuple(outer) + (y_dim, x_dim), name=da.name, attrs=da.attrs,
                    coords=_coarsened_coords(da, {x_dim: factor, y_dim: factor}, coord_func))
    return out.transpose(*da.dims)


def weighted_block_average(
    obj,
    weights,
    coarsening_factor: int,
    x_dim: Hashable = "xaxis_1",
    y_dim: Hashable = "yaxis_2",
    coord_func: CoordFunc = coarsen_coords_coord_func,
):
    """Coarsen a DataArray or Dataset through weighted block averaging (coarsen.py:183-218):
    ``(obj * weights).coarsen(x, y).sum() / weights.coarsen(x, y).sum()`` with NaN-skipping sums."""
    o, w = to_compat(obj), to_compat(weights)
    if isinstance(o, Dataset) and len(o) > 1:
        result = _weighted_block_average_ds(o, w, coarsening_factor, x_dim, y_dim, coord_func)
    else:
        result = _map_dataset(
            o, lambda da: _weighted_block_average_da(da, w, coarsening_factor, x_dim, y_dim, coord_func)
        )
    return from_compat(_propagate_attrs(o, result), obj)


def _weighted_block_average_ds(ds: Dataset, w: DataArray, factor: int, x_dim, y_dim, coord_func) -> Dataset:
    """The variables of a Dataset that share dims, shape and the weights' dtype go four to a launch (the weights -- often a
    whole 3-D masked area, regridz.py:200-220 -- are read once per four fields); the others one by one.  Same results."""
    kinds = {"float32": torch.float32, "float64": torch.float64, "torch.float32": torch.float32, "torch.float64": torch.float64}
    kind = lambda a: kinds.get(str(a.dtype))  # (numpy or torch data)
    groups = {}
    for name, da in ds.items():
        ok = x_dim in da.dims and y_dim in da.dims and kind(da) is not None and kind(da) == kind(w)
        groups.setdefault((da.dims, tuple(da.shape)) if ok else ("single", name), []).append(name)
    done = {}
    for key, names in groups.items():
        if key[0] == "single" or len(names) == 1:
            for n in names:
                done[n] = _weighted_block_average_da(ds[n], w, factor, x_dim, y_dim, coord_func)
            continue
        first, outer = horizontal_last(ds[names[0]], y_dim, x_dim)
        wt = _weights_tensor(w, outer, y_dim, x_dim, "weights")
        tensors = [first] + [horizontal_last(ds[n], y_dim, x_dim)[0] for n in names[1:]]
        for n, res in zip(names, ops.weighted_block_average_multi(tensors, wt, factor)):
            da = ds[n]
            r = DataArray(like_input(res, da.data), dims=tuple(outer) + (y_dim, x_dim), name=da.name, attrs=da.attrs,
                          coords=_coarsened_coords(da, {x_dim: factor, y_dim: factor}, coord_func))
            done[n] = r.transpose(*da.dims)
    out = Dataset(attrs=ds.attrs)
    for name in ds:
        out[name] = done[name]
    return out


def mass_weighted_block_average(
    obj,
    delp,
    area,
    coarsening_factor: int,
    x_dim: Hashable = "xaxis_1",
    y_dim: Hashable = "yaxis_2",
    coord_func: CoordFunc = coarsen_coords_coord_func,
):
    """``weighted_block_average(obj, delp * area, ...)`` -- what the restart pipelines do for W, T, ua, va and the
    non-fraction tracers (coarsen_restarts.py:384-396, 884-891) -- without materialising ``delp * area``: variables that
    share ``delp``'s dims and dtype go through the fused kernel four at a time (``ops.mass_weighted_block_average``), any
    other variable through the product and ``weighted_block_average``.  Same results either way."""
    o, d, a = to_compat(obj), to_compat(delp), to_compat(area)
    single = isinstance(o, DataArray)
    ds = Dataset({o.name or "field": o}) if single else o
    out = Dataset(attrs=ds.attrs)
    fused = [n for n, da in ds.items() if da.dims == d.dims and da.dtype == d.dtype and da.shape == d.shape
             and x_dim in da.dims and y_dim in da.dims]
    if fused:
        dt, outer = horizontal_last(d, y_dim, x_dim)
        w = _weights_tensor(a, outer, y_dim, x_dim, "area")
        if dt.dtype in (torch.float32, torch.float64) and w.dim() <= dt.dim():
            tensors = [horizontal_last(ds[n], y_dim, x_dim)[0] for n in fused]
            results = ops.mass_weighted_block_average(tensors, dt, w, coarsening_factor)
            for n, res in zip(fused, results):
                da = ds[n]
                r = DataArray(like_input(res, da.data), dims=tuple(outer) + (y_dim, x_dim), name=da.name, attrs=da.attrs,
                              coords=_coarsened_coords(da, {x_dim: coarsening_factor, y_dim: coarsening_factor}, coord_func))
                out[n] = r.transpose(*da.dims)
        else:
            fused = []
    rest = [n for n in ds if n not in fused]
    if rest:
        from .coarsen_restarts import _mul  # (the product kernel; import here: coarsen_restarts imports this module)

        weights = _mul(d, a)
        for n in rest:
            out[n] = _weighted_block_average_da(ds[n], to_compat(weights), coarsening_factor, x_dim, y_dim, coord_func)
    ordered = Dataset(attrs=ds.attrs)
    for n in ds:
        ordered[n] = out[n]
    result = _propagate_attrs(ds, ordered)
    return from_compat(result[o.name or "field"] if single else result, obj)


def _coarsen_downsample_coordinate(reference: DataArray, dim, factor, coord_func):
    if dim not in reference.coords:
        return None
    return np.asarray(_reduce_coord(np.asarray(reference.coords[dim]), factor, _coord_function_for(coord_func, dim)))


def _edge_dims(edge, x_dim, y_dim):
    if edge == "x":
        return x_dim, y_dim
    elif edge == "y":
        return y_dim, x_dim
    raise ValueError(f"'edge' most be either 'x' or 'y'; got {edge}.")


def _edge_weighted_da(da: DataArray, spacing: DataArray, factor, x_dim, y_dim, edge, coord_func):
    coarsen_dim, downsample_dim = _edge_dims(edge, x_dim, y_dim)
    t, outer = horizontal_last(da, y_dim, x_dim)
    w = _weights_tensor(spacing, outer, y_dim, x_dim, "spacing")
    res = ops.edge_weighted_block_average(float_tensor(t), w, factor, edge)
    coords = _coarsened_coords(da, {coarsen_dim: factor}, coord_func)
    down = _coarsen_downsample_coordinate(da, downsample_dim, factor, coord_func)
    if down is not None:
        coords[downsample_dim] = down
    out = DataArray(like_input(res, da.data), dims=tuple(outer) + (y_dim, x_dim), name=da.name, attrs=da.attrs, coords=coords)
    return out.transpose(*da.dims)


def edge_weighted_block_average(
    obj,
    spacing,
    coarsening_factor: int,
    x_dim: Hashable = "xaxis_1",
    y_dim: Hashable = "yaxis_1",
    edge: str = "x",
    coord_func: CoordFunc = coarsen_coords_coord_func,
):
    """Coarsen along a block edge (coarsen.py:221-273)."""
    _edge_dims(edge, x_dim, y_dim)
    o, w = to_compat(obj), to_compat(spacing)
    result = _map_dataset(
        o, lambda da: _edge_weighted_da(da, w, coarsening_factor, x_dim, y_dim, edge, coord_func)
    )
    return from_compat(_propagate_attrs(o, result), obj)


# ---------------------------------------------------------------------------------------------
# plain block reductions
# ---------------------------------------------------------------------------------------------
_REDUCTION_NAMES = {np.mean: "mean", np.median: "median", np.sum: "sum", np.min: "min", np.max: "max",
                    np.nanmean: "mean", np.nansum: "sum", np.nanmin: "min", np.nanmax: "max",
                    np.amin: "min", np.amax: "max"}


def _block_reduce_da(da: DataArray, factor_y, factor_x, stride_y, stride_x, method, x_dim, y_dim, block_sizes,
                     coord_func, nan_policy="skip", require_exact=True):
    if x_dim not in da.dims or y_dim not in da.dims:
        return da  # consistent with xarray's coarsen: untouched if the dims are absent
    if require_exact and (da.sizes[y_dim] % factor_y or da.sizes[x_dim] % factor_x):
        raise ValueError(
            f"Could not coarsen a dimension of size {da.sizes[y_dim] if da.sizes[y_dim] % factor_y else da.sizes[x_dim]} "
            f"with window {factor_y if da.sizes[y_dim] % factor_y else factor_x}"
        )
    t, outer = horizontal_last(da, y_dim, x_dim)
    res = ops.block_reduce(t, (factor_y, factor_x), (stride_y, stride_x), op=method, nan_policy=nan_policy)
    out = DataArray(like_input(res, da.data), dims=tuple(outer) + (y_dim, x_dim), name=da.name, attrs=da.attrs,
                    coords=_coarsened_coords(da, block_sizes, coord_func))
    return out.transpose(*da.dims)


def block_coarsen(
    obj,
    coarsening_factor: int,
    x_dim: Hashable = "xaxis_1",
    y_dim: Hashable = "yaxis_1",
    method: str = "sum",
    coord_func: CoordFunc = coarsen_coords_coord_func,
    func_kwargs: Optional[Dict] = None,
):
    """Coarsen by an operation over blocks (coarsen.py:795-840): xarray's coarsen methods
    (sum, mean, min, max; NaN-skipping) plus 'median' and 'mode'."""
    func_kwargs = func_kwargs or {}
    if method == "median":
        return block_median(obj, coarsening_factor, x_dim=x_dim, y_dim=y_dim, coord_func=coord_func, **func_kwargs)
    if method == "mode":
        return _block_mode(obj, coarsening_factor, x_dim=x_dim, y_dim=y_dim, coord_func=coord_func, **func_kwargs)
    if method not in ("sum", "mean", "min", "max"):
        raise AttributeError(f"coarsen objects have no method {method!r}")
    o = to_compat(obj)
    f = coarsening_factor
    result = _map_dataset(
        o, lambda da: _block_reduce_da(da, f, f, f, f, method, x_dim, y_dim, {x_dim: f, y_dim: f}, coord_func)
    )
    if isinstance(result, Dataset):
        result.attrs = {}
    return from_compat(result, obj)


def horizontal_block_reduce(
    obj,
    coarsening_factor: int,
    reduction_function: Callable,
    x_dim: Hashable = "xaxis_1",
    y_dim: Hashable = "yaxis_1",
    coord_func: CoordFunc = coarsen_coords_coord_func,
):
    """coarsen.py:520-554.  The reduction must be one the device implements (numpy's mean, median,
    sum, min, max or ``functools.partial(_mode_reduce, nan_policy=...)``)."""
    return xarray_block_reduce(
        obj, {x_dim: coarsening_factor, y_dim: coarsening_factor}, reduction_function, coord_func=coord_func
    )


def _mode_reduce(arr, axis=(0,), nan_policy: str = "propagate"):
    """Marker for the block-mode reduction (coarsen.py:743-747); evaluated on the device."""
    raise NotImplementedError("_mode_reduce is only a selector for block reductions on the device")


def _resolve_reduction(reduction_function):
    import functools

    if isinstance(reduction_function, functools.partial) and reduction_function.func is _mode_reduce:
        return "mode", reduction_function.keywords.get("nan_policy", "propagate")
    if reduction_function is _mode_reduce:
        return "mode", "propagate"
    try:
        name = _REDUCTION_NAMES[reduction_function]
    except (KeyError, TypeError):
        raise NotImplementedError(
            f"block reduction with {reduction_function!r} is not implemented on the device; supported: "
            "numpy mean/median/sum/min/max and _mode_reduce"
        ) from None
    if name == "median":
        return "median", "propagate"
    # np.mean / np.sum ... propagate NaN in the reference's block_reduce; nan* variants skip
    skipping = reduction_function in (np.nanmean, np.nansum, np.nanmin, np.nanmax)
    return name, ("skip" if skipping else "propagate")


def _xarray_block_reduce_dataarray(
    da, block_sizes: Mapping[Hashable, int], reduction_function: Callable, cval: float = np.nan,
    coord_func: CoordFunc = coarsen_coords_coord_func,
):
    """coarsen.py:393-460, for block sizes over (at most) two dims."""
    da_c = to_compat(da)
    reduction_dims = [d for d in da_c.dims if block_sizes.get(d, 1) != 1 or d in block_sizes]
    reduction_dims = [d for d in reduction_dims if d in block_sizes]
    if not reduction_dims:
        return da
    nontrivial = [d for d in reduction_dims if block_sizes[d] != 1]
    if len(nontrivial) > 2:
        raise NotImplementedError("block reductions over more than two dimensions are not implemented")
    while len(nontrivial) < 2:  # pad with a trivial window over some other dim
        spare = [d for d in da_c.dims if d not in nontrivial]
        if not spare:
            da_c = DataArray(da_c.data[None], dims=("__unit__",) + da_c.dims, coords=da_c.coords, name=da_c.name, attrs=da_c.attrs)
            spare = ["__unit__"]
        nontrivial.append(spare[-1])
    y_dim, x_dim = sorted(nontrivial, key=da_c.dims.index)
    fy, fx = block_sizes.get(y_dim, 1), block_sizes.get(x_dim, 1)
    for d, f in ((y_dim, fy), (x_dim, fx)):
        if da_c.sizes[d] % f:
            raise NotImplementedError("padding with cval for non-divisible blocks is not implemented on the device")
    method, policy = _resolve_reduction(reduction_function)
    t, outer = horizontal_last(da_c, y_dim, x_dim)
    res = ops.block_reduce(t, (fy, fx), op=method, nan_policy=policy)
    out = DataArray(like_input(res, da_c.data), dims=tuple(outer) + (y_dim, x_dim), name=da_c.name, attrs=da_c.attrs,
                    coords=_coarsened_coords(da_c, {d: block_sizes[d] for d in reduction_dims}, coord_func))
    out = out.transpose(*da_c.dims)
    if "__unit__" in out.dims:
        out = out.isel({"__unit__": 0})
    return from_compat(out, da)


def xarray_block_reduce(obj, block_sizes, reduction_function, cval=np.nan, coord_func: CoordFunc = coarsen_coords_coord_func):
    """coarsen.py:463-517."""
    o = to_compat(obj)
    result = _map_dataset(
        o, lambda da: _xarray_block_reduce_dataarray(da, block_sizes, reduction_function, cval=cval, coord_func=coord_func)
    )
    return from_compat(_propagate_attrs(o, result), obj)


def block_median(obj, coarsening_factor: int, x_dim="xaxis_1", y_dim="yaxis_1",
                 coord_func: CoordFunc = coarsen_coords_coord_func):
    """coarsen.py:557-588."""
    return horizontal_block_reduce(obj, coarsening_factor, np.median, x_dim=x_dim, y_dim=y_dim, coord_func=coord_func)


def _block_mode(obj, coarsening_factor: int, x_dim="xaxis_1", y_dim="yaxis_1",
                coord_func: CoordFunc = coarsen_coords_coord_func, nan_policy: str = "propagate"):
    """coarsen.py:750-786 (scipy.stats.mode 1.7.3 semantics: most frequent value, smallest on ties)."""
    import functools

    return horizontal_block_reduce(
        obj, coarsening_factor, functools.partial(_mode_reduce, nan_policy=nan_policy), x_dim=x_dim, y_dim=y_dim,
        coord_func=coord_func,
    )


def block_edge_coarsen(obj, coarsening_factor: int, x_dim="xaxis_1", y_dim="yaxis_1", edge: str = "x",
                       coord_func: CoordFunc = coarsen_coords_coord_func, method="sum"):
    """Coarsen by an operation along a block edge (coarsen.py:629-683)."""
    coarsen_dim, downsample_dim = _edge_dims(edge, x_dim, y_dim)
    if method not in ("sum", "mean", "min", "max"):
        raise AttributeError(f"coarsen objects have no method {method!r}")
    o = to_compat(obj)
    f = coarsening_factor
    fy, fx = (1, f) if edge == "x" else (f, 1)

    def one(da):
        if coarsen_dim not in da.dims or downsample_dim not in da.dims:
            return da
        if da.sizes[coarsen_dim] % f:
            raise ValueError(f"Could not coarsen a dimension of size {da.sizes[coarsen_dim]} with window {f}")
        down = _coarsen_downsample_coordinate(da, downsample_dim, f, coord_func)
        bare = da._replace(coords={k: v for k, v in da.coords.items() if k != downsample_dim})
        out = _block_reduce_da(bare, fy, fx, f, f, method, x_dim, y_dim, {coarsen_dim: f}, coord_func, require_exact=False)
        if down is not None:
            out = out.assign_coords({downsample_dim: down})
        return out

    result = _map_dataset(o, one)
    return from_compat(_propagate_attrs(o, result), obj)


def block_edge_sum(obj, coarsening_factor: int, x_dim="xaxis_1", y_dim="yaxis_1", edge: str = "x",
                   coord_func: CoordFunc = coarsen_coords_coord_func):
    """coarsen.py:591-626."""
    return block_edge_coarsen(obj, coarsening_factor, x_dim=x_dim, y_dim=y_dim, edge=edge, coord_func=coord_func, method="sum")


# ---------------------------------------------------------------------------------------------
# upsampling
# ---------------------------------------------------------------------------------------------
def _upsample_da(da: DataArray, factor: int, dims: List[Hashable]) -> DataArray:
    present = [d for d in dims if d in da.dims]
    if not present:
        return da
    if len(present) > 2:
        raise NotImplementedError("block_upsample over more than two dimensions is not implemented")
    work = da
    if len(present) == 1:  # the kernel repeats along two dims: add a unit axis (size 1 is 'staggered': not repeated)
        work = DataArray(da.data[..., None], dims=da.dims + ("__unit__",), coords=da.coords, name=da.name, attrs=da.attrs)
        present = present + ["__unit__"]
    y_dim, x_dim = sorted(present, key=work.dims.index)
    t, outer = horizontal_last(work, y_dim, x_dim)
    res = ops.block_upsample(t, factor)
    coords = {k: v for k, v in da.coords.items() if k not in dims}  # upsampled dims lose their coordinates
    out = DataArray(like_input(res, da.data), dims=tuple(outer) + (y_dim, x_dim), name=da.name, attrs=da.attrs, coords=coords)
    out = out.transpose(*work.dims)
    if "__unit__" in out.dims:
        out = out.isel({"__unit__": 0})
    return out


def block_upsample(obj, upsampling_factor: int, dims: List[Hashable]):
    """Repeat values n times along each of ``dims``; a dim of odd size is a staggered one whose
    last point is not repeated (coarsen.py:843-897)."""
    o = to_compat(obj)
    result = _map_dataset(o, lambda da: _upsample_da(da, upsampling_factor, list(dims)))
    return from_compat(result, obj)


def block_upsample_like(da, reference_da, x_dim: Hashable = "xaxis_1", y_dim: Hashable = "yaxis_1"):
    """Upsample back to the resolution (and horizontal coordinates) of ``reference_da``
    (coarsen.py:900-938)."""
    d, ref = to_compat(da), to_compat(reference_da)
    x_is_staggered_dim = d.sizes[x_dim] % 2 == 1
    if x_is_staggered_dim:
        factor = (ref.sizes[x_dim] - 1) // (d.sizes[x_dim] - 1)
    else:
        factor = ref.sizes[x_dim] // d.sizes[x_dim]
    result = _upsample_da(d, factor, [x_dim, y_dim])
    coords = {k: ref.coords[k] for k in (x_dim, y_dim) if k in ref.coords}
    return from_compat(result.assign_coords(coords), da)
