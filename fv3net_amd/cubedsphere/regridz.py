"""Vertical regridding to coarse pressure levels with the interface of
``vcm.cubedsphere.regridz`` (external/vcm/vcm/cubedsphere/regridz.py), on the device.

The reference moves the vertical dim last, flattens to [column, level], lets f2py copy to
Fortran order and calls ``mappm`` per chunk.  Here the arrays stay in whatever layout they have
(the native [tile, z, y, x] included) and the remap kernel walks the level axis in place.
"""
import numpy as np

from .. import ops
from ..xr_compat import DataArray, Dataset, from_compat, to_compat
from ._device import like_input, on_device
from .coarsen import block_upsample_like, edge_weighted_block_average, weighted_block_average
from .constants import (
    FV_CORE_X_CENTER,
    FV_CORE_X_OUTER,
    FV_CORE_Y_CENTER,
    FV_CORE_Y_OUTER,
    RESTART_Z_CENTER,
    RESTART_Z_OUTER,
)
from .grid import interp_center_to_outer

SURFACE_LEVEL = -1


def pressure_at_interface(*args, **kwargs):
    from ..thermo import pressure_at_interface as f  # (thermo imports this package's device helpers)

    return f(*args, **kwargs)


def pressure_at_midpoint_log(*args, **kwargs):
    from ..thermo import pressure_at_midpoint_log as f

    return f(*args, **kwargs)


def regrid_vertical(p_in, f_in, p_out, iv: int = 1, kord: int = 1, z_dim_center: str = RESTART_Z_CENTER,
                    z_dim_outer: str = RESTART_Z_OUTER):
    """Vertical regridding with the PPM remap (regridz.py:223-301).  ``p_in``/``p_out`` hold
    interface pressures along ``z_dim_outer``, ``f_in`` layer means along ``z_dim_center``; the
    result has ``f_in``'s dim order and attrs, float32, with ``p_out``'s layers."""
    return _regrid_vertical_many(p_in, [f_in], p_out, iv, kord, z_dim_center, z_dim_outer)[0]


def _regrid_vertical_many(p_in, fields, p_out, iv: int = 1, kord: int = 1, z_dim_center: str = RESTART_Z_CENTER,
                          z_dim_outer: str = RESTART_Z_OUTER, p_out_factor: int = None, hor_dims=None):
    """``regrid_vertical`` of several fields between the same two pressure grids (what regridz.py:180-185
    loops over): fields with the same dims share one multi-field remap sweep.  ``p_out_factor``: ``p_out`` is still on the
    grid coarsened by that factor along ``hor_dims`` = (y_dim, x_dim) and stands for its block-upsampled copy."""
    if z_dim_center == z_dim_outer:
        raise ValueError("'z_dim_center' and 'z_dim_outer' must not be equal.")
    pi, po = to_compat(p_in), to_compat(p_out)
    compat = [to_compat(f) for f in fields]
    groups = {}
    for n, fi in enumerate(compat):
        groups.setdefault((fi.dims, tuple(fi.shape)), []).append(n)
    results = [None] * len(compat)
    for members in groups.values():
        outs = _regrid_group(pi, [compat[n] for n in members], po, iv, kord, z_dim_center, z_dim_outer, p_out_factor, hor_dims)
        for n, out in zip(members, outs):
            results[n] = from_compat(out, fields[n])
    return results


def _regrid_group(pi, group, po, iv, kord, z_dim_center, z_dim_outer, po_factor=None, hor_dims=None):
    fi = group[0]
    if po_factor is not None and po_factor > 1:
        # the fused path needs [..., z, y, x]; any other order goes through the upsampled copy
        if tuple(fi.dims[-3:]) == (z_dim_center,) + tuple(hor_dims):
            order_p = [z_dim_outer if d == z_dim_center else d for d in fi.dims]
            if set(pi.dims) != set(order_p) or set(po.dims) != set(order_p):
                raise ValueError("All dimensions except vertical must be same size for p_in, f_in and p_out")
            pi_t, po_t = pi.transpose(*order_p), po.transpose(*order_p)
            if fi.sizes[z_dim_center] != pi_t.sizes[z_dim_outer] - 1:
                raise ValueError("f_in must have a vertical dimension one shorter than p_in")
            res = ops.mappm_multi_coarse_target(on_device(pi_t.data), [on_device(f.data) for f in group], on_device(po_t.data), po_factor,
                                                iv=iv, kord=kord, z_axis=-3)
            return [DataArray(like_input(r, f.data), dims=f.dims, coords={k: v for k, v in f.coords.items() if k != z_dim_center},
                              name=f.name, attrs=f.attrs) for r, f in zip(res, group)]
        from .coarsen import block_upsample

        po = to_compat(block_upsample(po, po_factor, list(hor_dims)))
    dims_except_z = [d for d in fi.dims if d != z_dim_center]
    # same dim order for all three, the vertical dim where f_in has it
    order_f = list(fi.dims)
    order_p = [z_dim_outer if d == z_dim_center else d for d in order_f]
    for name, arr in (("p_in", pi), ("p_out", po)):
        if set(arr.dims) != set(order_p):
            raise ValueError("All dimensions except vertical must be same size for p_in, f_in and p_out")
    pi_t, po_t = pi.transpose(*order_p), po.transpose(*order_p)
    n_columns = 1
    for d in dims_except_z:
        n_columns *= fi.sizes[d]
        if pi_t.sizes[d] != fi.sizes[d] or po_t.sizes[d] != fi.sizes[d]:
            raise ValueError("All dimensions except vertical must be same size for p_in, f_in and p_out")
    if fi.sizes[z_dim_center] != pi_t.sizes[z_dim_outer] - 1:
        raise ValueError("f_in must have a vertical dimension one shorter than p_in")
    axis = fi.get_axis_num(z_dim_center)
    res = ops.mappm_multi(on_device(pi_t.data), [on_device(f.data) for f in group], on_device(po_t.data), iv=iv, kord=kord,
                          z_axis=axis)
    return [DataArray(like_input(r, f.data), dims=f.dims, coords={k: v for k, v in f.coords.items() if k != z_dim_center},
                      name=f.name, attrs=f.attrs) for r, f in zip(res, group)]


def _mask_weights(weights, pfull_coarse_on_fine, phalf_coarse_on_fine, phalf_fine, dim_center: str = RESTART_Z_CENTER,
                  dim_outer: str = RESTART_Z_OUTER, extrapolate: bool = False, coarse_factor: int = None, hor_dims=None):
    """regridz.py:200-220: weights where the coarse level lies above the fine surface, else 0.  ``coarse_factor``: the
    coarse pressures are passed on their own grid (coarsened by that factor along ``hor_dims``), not upsampled."""
    w, pf = to_compat(weights), to_compat(phalf_fine)
    pc = to_compat(pfull_coarse_on_fine if extrapolate else phalf_coarse_on_fine)
    if coarse_factor is not None and coarse_factor > 1 and tuple(pf.dims[-3:]) != (dim_outer,) + tuple(hor_dims):
        from .coarsen import block_upsample

        pc, coarse_factor = to_compat(block_upsample(pc, coarse_factor, list(hor_dims))), None
    zc = dim_center if extrapolate else dim_outer
    order = list(pf.dims)
    axis = order.index(dim_outer)
    pc_t = pc.transpose(*[zc if d == dim_outer else d for d in order])
    w_order = [d for d in order if d != dim_outer]
    if set(w.dims) != set(w_order):
        # broadcast the weights over the missing non-vertical dims (e.g. area [tile, y, x] vs time)
        missing = [d for d in w_order if d not in w.dims]
        data = on_device(w.data)
        for _ in missing:
            data = data.unsqueeze(0)
        data = data.expand(*[pf.sizes[d] for d in missing], *w.shape).contiguous()
        w = DataArray(data, dims=tuple(missing) + w.dims, coords=w.coords, attrs=w.attrs, name=w.name)
    w_t = w.transpose(*w_order)
    res = ops.mask_weights(on_device(w_t.data), on_device(pc_t.data), on_device(pf.data), axis, extrapolate=extrapolate,
                           coarse_factor=coarse_factor)
    dims = tuple(dim_center if d == dim_outer else d for d in order)
    coords = {k: v for k, v in w.coords.items()}
    out = DataArray(like_input(res, to_compat(weights).data), dims=dims, coords=coords, name=w.name, attrs=w.attrs)
    return from_compat(out, weights)


def _regrid_given_delp(ds, delp_fine, delp_coarse, weights, toa_pressure, x_dim: str = FV_CORE_X_CENTER,
                       y_dim: str = FV_CORE_Y_CENTER, z_dim: str = RESTART_Z_CENTER, extrapolate: bool = False):
    """regridz.py:149-197."""
    # The reference upsamples the coarse thicknesses and integrates them on the fine grid; the column sums of a block are
    # those of its coarse column, bit for bit, so the interfaces are integrated on the coarse grid (1/f^2 of the columns)
    # and upsampled -- one fine-size pass instead of two.
    # ... and never upsampled at all where the remap and the mask can read them through (y // f, x // f).
    phalf_fine = pressure_at_interface(delp_fine, dim_center=z_dim, dim_outer=RESTART_Z_OUTER, toa_pressure=toa_pressure)
    phalf_coarse = pressure_at_interface(delp_coarse, dim_center=z_dim, dim_outer=RESTART_Z_OUTER, toa_pressure=toa_pressure)
    dc, df = to_compat(delp_coarse), to_compat(delp_fine)
    staggered = dc.sizes[x_dim] % 2 == 1
    factor = (df.sizes[x_dim] - 1) // (dc.sizes[x_dim] - 1) if staggered else df.sizes[x_dim] // dc.sizes[x_dim]
    hor = (y_dim, x_dim)
    d = to_compat(ds)
    if isinstance(d, Dataset):
        regridded = Dataset(attrs=d.attrs)
        names = list(d)
        for var, out in zip(names, _regrid_vertical_many(phalf_fine, [d[v] for v in names], phalf_coarse, z_dim_center=z_dim,
                                                         p_out_factor=factor, hor_dims=hor)):
            regridded[var] = out
    else:
        regridded = _regrid_vertical_many(phalf_fine, [d], phalf_coarse, z_dim_center=z_dim, p_out_factor=factor, hor_dims=hor)[0]
    pfull_coarse = pressure_at_midpoint_log(delp_coarse, dim=z_dim, toa_pressure=toa_pressure) if extrapolate else None
    masked_weights = _mask_weights(weights, pfull_coarse, phalf_coarse, phalf_fine, dim_center=z_dim, extrapolate=extrapolate,
                                   coarse_factor=factor, hor_dims=hor)
    return from_compat(regridded, ds), masked_weights


def regrid_to_area_weighted_pressure(ds, delp, area, toa_pressure: float, coarsening_factor: int,
                                     x_dim: str = FV_CORE_X_CENTER, y_dim: str = FV_CORE_Y_CENTER,
                                     z_dim: str = RESTART_Z_CENTER, extrapolate: bool = False):
    """Vertically regrid cell-centred quantities to coarsened pressure levels (regridz.py:31-78).
    Returns (regridded dataset, area masked wherever the coarse layer is below the fine surface)."""
    delp_coarse = weighted_block_average(delp, area, coarsening_factor, x_dim=x_dim, y_dim=y_dim)
    return _regrid_given_delp(ds, delp, delp_coarse, area, toa_pressure, x_dim=x_dim, y_dim=y_dim, z_dim=z_dim,
                              extrapolate=extrapolate)


def fused_block_mean_mode() -> str:
    """``FV3NET_AMD_FUSED_BLOCK_MEAN`` = ``0`` | ``1`` | ``auto`` (default): whether the pressure-level means of cell-centred
    fields go through the fused remap + block-mean kernel (``ops.mappm_block_mean``) -- never, always, or while it pays:
    the kernel is faster where the 64 columns of a coarse cell stay within a few target layers of each other (a real restart
    file; 12-14 % of a whole pressure-level pipeline call) and slower where they do not (BASELINE configs[2]'s iid
    thicknesses), the values are the same either way, and the kernel counts the blocks whose waves ran out of
    ring.  ``auto`` starts fused, reads that count -- asynchronously, when the next call of the same shape begins -- and
    takes the three launches while more than 40 % of the blocks gave up, looking again every 64th call."""
    import os

    v = os.environ.get("FV3NET_AMD_FUSED_BLOCK_MEAN", "auto").lower()
    return {"0": "0", "false": "0", "off": "0", "1": "1", "true": "1", "on": "1"}.get(v, "auto")


_FUSED_ROUTE = {}   # (device, shape, dtype, fields) -> the adaptive route's state


def _use_fused(key, n_blocks: int) -> bool:
    mode = fused_block_mean_mode()
    if mode != "auto":
        return mode == "1"
    import torch

    if torch.cuda.is_current_stream_capturing():   # (a captured call keeps the route it is captured with: the static default)
        return False
    st = _FUSED_ROUTE.setdefault(key, {"fused": True, "pending": None, "calls_since": 0})
    pending = st["pending"]
    if pending is not None and pending[1].query():
        st["fused"] = int(pending[0][2]) * 10 <= 4 * n_blocks   # blocks whose waves gave up summing, of the last sweep
        st["pending"], st["calls_since"] = None, 0
    st["calls_since"] += 1
    if not st["fused"] and st["calls_since"] > 64:
        st["fused"] = True   # the data may have changed: look again
    return st["fused"]


def _watch_fused(key, dev):
    """A pinned buffer for the call's counters and the event that says they have arrived (``auto`` mode only)."""
    if fused_block_mean_mode() != "auto":
        return None
    import torch

    st = _FUSED_ROUTE.get(key)
    if st is None or st["pending"] is not None:
        return None
    st["pending"] = (torch.zeros(4, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    return st["pending"]


def area_weighted_pressure_means(ds, delp, area, toa_pressure: float, coarsening_factor: int, x_dim: str = FV_CORE_X_CENTER,
                                 y_dim: str = FV_CORE_Y_CENTER, z_dim: str = RESTART_Z_CENTER, extrapolate: bool = False,
                                 side_stream=None, side_work=None):
    """``weighted_block_average(*regrid_to_area_weighted_pressure(ds, delp, area, ...), coarsening_factor)`` -- what the
    pressure-level restart pipelines do with every cell-centred field (coarsen_restarts.py:483-495, 940-961) -- as a
    two-stream pipeline over groups of four fields: the remap sweep of group g + 1 (latency-bound, it fills the chip's
    wave slots but not its memory system) runs on the calling stream while the masked block mean of group g (HBM-bound) and,
    first of all, the masked area run on a stream beside it.  Same kernels, same values and labels as the two calls.
    With factor 8 and a float32 area one fused kernel per group instead where that pays (``fused_block_mean_mode``).
    Arrays must come in [.., z, y, x] order (the restart files'); anything else takes the two calls.
    ``side_stream``: the stream to use beside the caller's (the HIP runtime multiplexes streams onto four hardware queues:
    a pipeline that spreads over more than two or three streams serialises on queue sharing); ``side_work``: a callable that
    is run -- under ``side_stream`` -- once the first sweep is enqueued, i.e. while the device is busy with it."""
    import torch

    def two_calls():
        if side_work is not None:
            with torch.cuda.stream(side_stream):
                side_work()
        regridded, masked = regrid_to_area_weighted_pressure(ds, delp, area, toa_pressure, coarsening_factor, x_dim=x_dim, y_dim=y_dim,
                                                             z_dim=z_dim, extrapolate=extrapolate)
        return weighted_block_average(regridded, masked, coarsening_factor, x_dim=x_dim, y_dim=y_dim)

    d, dl, ar = to_compat(ds), to_compat(delp), to_compat(area)
    das = [d[v] for v in d] if isinstance(d, Dataset) else [d]
    order = tuple(dl.dims)
    f = int(coarsening_factor)
    if (f < 2 or order[-3:] != (z_dim, y_dim, x_dim) or any(tuple(a.dims) != order or a.shape != dl.shape for a in das)
            or tuple(ar.dims[-2:]) != (y_dim, x_dim) or tuple(ar.dims[:-2]) != order[: len(ar.dims) - 2]
            or dl.sizes[y_dim] % f or dl.sizes[x_dim] % f):
        return two_calls()
    delp_t, area_t = on_device(dl.data), on_device(ar.data)
    fields = [on_device(a.data) for a in das]
    if any(q.dtype != delp_t.dtype for q in fields) or not delp_t.is_cuda:
        return two_calls()
    delp_coarse = ops.weighted_block_average(delp_t, area_t, f)
    phalf_fine = ops.pressure_at_interface(delp_t, toa_pressure, -3)
    phalf_coarse = ops.pressure_at_interface(delp_coarse, toa_pressure, -3)
    level = ops.pressure_at_midpoint_log(delp_coarse, toa_pressure, -3) if extrapolate else None
    means = None
    if f == 8 and area_t.dtype == torch.float32:
        key = (delp_t.device.index, tuple(delp_t.shape), str(delp_t.dtype), len(fields), bool(extrapolate), ops.MAPPM_ARITHMETIC)
        n_blocks = min(delp_t.numel() // int(delp_t.shape[-3]) // 64, (1 << 20) // 64)   # (the counters are those of the last launch: <= 2^20 columns)
        if _use_fused(key, n_blocks):
            watch = _watch_fused(key, delp_t.device)
            means = ops.mappm_block_mean(phalf_fine, fields, phalf_coarse, area_t, level_coarse=level, iv=1, kord=1,
                                         counters=None if watch is None else watch[0])
            if watch is not None:
                if means is None:
                    _FUSED_ROUTE[key]["pending"] = None
                else:
                    watch[1].record(torch.cuda.current_stream(delp_t.device))
    if means is None:
        dev = delp_t.device
        main = torch.cuda.current_stream(dev)
        if side_stream is None:
            from ._device import side_streams

            side_stream = side_streams(dev)[0]
        side = side_stream
        batch = tuple(phalf_fine.shape[:-3])
        w = area_t
        if tuple(w.shape[:-2]) != batch:  # the area [tile, y, x] shared by the time axis
            w = w.reshape(tuple(w.shape[:-2]) + (1,) * (len(batch) - (w.dim() - 2)) + tuple(w.shape[-2:])).expand(*batch, *w.shape[-2:]).contiguous()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            masked = ops.mask_weights(w, level if extrapolate else phalf_coarse, phalf_fine, -3, extrapolate=extrapolate, coarse_factor=f)
        means, remapped = [], []
        for g0 in range(0, len(fields), 4):
            q2 = ops.mappm_multi_coarse_target(phalf_fine, fields[g0:g0 + 4], phalf_coarse, f, iv=1, kord=1, z_axis=-3)
            if g0 == 0 and side_work is not None:
                with torch.cuda.stream(side):
                    side_work()
                side_work = None
            side.wait_stream(main)
            with torch.cuda.stream(side):
                means.extend(ops.weighted_block_average_multi(q2, masked, f) if len(q2) > 1 else [ops.weighted_block_average(q2[0], masked, f)])
            remapped.append(q2)   # (kept until the join: freed now, the next sweep would write into it while the mean still reads)
        main.wait_stream(side)
        # Memory crossed the two streams in both directions without `record_stream`: what the calling stream allocated and the
        # side stream read (remapped fields, pressures, weights) stays referenced until this join; what the side stream
        # allocated (the means) returns to ITS pool when the caller drops it, and that pool serves only work that begins by
        # waiting for the calling stream again.
        del remapped
    if side_work is not None:   # (the fused route, or no field at all)
        with torch.cuda.stream(side_stream):
            side_work()
    from .coarsen import _coarsened_coords, coarsen_coords_coord_func

    def label(t, a):
        coords = _coarsened_coords(a, {x_dim: f, y_dim: f}, coarsen_coords_coord_func)
        coords = {k: v for k, v in coords.items() if k != z_dim}   # (regrid_vertical drops the vertical coordinate)
        return DataArray(like_input(t, a.data), dims=a.dims, name=a.name, attrs=a.attrs, coords=coords)

    if not isinstance(d, Dataset):
        return from_compat(label(means[0], das[0]), ds)
    out = Dataset(attrs=d.attrs)
    for v, m, a in zip(list(d), means, das):
        out[v] = label(m, a)
    return from_compat(out, ds)


def compute_edge_delp(delp, edge: str, x_dim: str = FV_CORE_X_CENTER, y_dim: str = FV_CORE_Y_CENTER, step: int = 1):
    """Pressure thickness on grid cell edges (coarsen_restarts.py:825-853): ``delp`` interpolated
    across the cube's faces to the edges the ``edge``-directed wind component lives on; the new
    staggered dimension keeps the name passed for it, with coordinate 1..n+1 (float32).
    ``step`` > 1 (not in the reference): every step-th of those edge lines only -- all an edge-weighted block average keeps."""
    hor_dims = {"x": x_dim, "y": y_dim}
    interp_dim = "x" if edge == "y" else "y"
    outer_names = {"x": FV_CORE_X_OUTER, "y": FV_CORE_Y_OUTER}
    staggered = to_compat(interp_center_to_outer(delp, interp_dim, x_center=FV_CORE_X_CENTER, x_outer=FV_CORE_X_OUTER,
                                                 y_center=FV_CORE_Y_CENTER, y_outer=FV_CORE_Y_OUTER, step=step))
    d = to_compat(delp)
    new_dim = outer_names[interp_dim]
    wanted = hor_dims[interp_dim]
    if wanted == new_dim:  # (with other names the reference attaches a coordinate that indexes no dimension)
        staggered = staggered.assign_coords({wanted: np.arange(1, d.sizes[hor_dims[edge]] + 2, dtype=np.float32)[::int(step)]})
    return from_compat(staggered, delp)


def regrid_to_edge_weighted_pressure(ds, delp, length, toa_pressure: float, coarsening_factor: int,
                                     x_dim: str = FV_CORE_X_CENTER, y_dim: str = FV_CORE_Y_OUTER,
                                     z_dim: str = RESTART_Z_CENTER, edge: str = "x", extrapolate: bool = False):
    """Vertically regrid edge-valued quantities (D-grid winds) to coarsened pressure levels
    (regridz.py:81-146).  ``delp`` is interpolated to the cell edges across the cube's faces -- the one
    step of the coarse-graining path that needs data from neighbouring tiles -- coarsened along the
    edges with the edge lengths, and the fields are remapped column by column.  Returns (regridded
    dataset, edge lengths masked wherever the coarse layer is below the fine surface)."""
    if edge not in ("x", "y"):
        raise ValueError(f"'edge' most be either 'x' or 'y'; got {edge}.")
    delp_staggered = compute_edge_delp(delp, edge, x_dim=x_dim, y_dim=y_dim)
    delp_staggered_coarse = edge_weighted_block_average(delp_staggered, length, coarsening_factor, x_dim=x_dim,
                                                        y_dim=y_dim, edge=edge)
    return _regrid_given_delp(ds, delp_staggered, delp_staggered_coarse, length, toa_pressure, x_dim=x_dim,
                              y_dim=y_dim, z_dim=z_dim, extrapolate=extrapolate)


class EdgeLines:
    """The edge pressure thickness of one D-grid wind component on the lines its edge-weighted means keep
    (every f-th of the n + 1 edge lines): device tensors in [outer..., z, y, x] order, shared by the pressure-level remap
    and the blending weights of one pipeline call."""

    def __init__(self, delp, length, factor: int, edge: str, x_dim, y_dim, z_dim=RESTART_Z_CENTER):
        from .coarsen import _edge_dims

        self.factor, self.edge, self.x_dim, self.y_dim, self.z_dim = int(factor), edge, x_dim, y_dim, z_dim
        self.coarsen_dim, self.down_dim = _edge_dims(edge, x_dim, y_dim)
        f = self.factor
        d = to_compat(compute_edge_delp(delp, edge, x_dim=x_dim, y_dim=y_dim, step=f))
        self.outer = [dim for dim in d.dims if dim not in (z_dim, y_dim, x_dim)]
        self.order = self.outer + [z_dim, y_dim, x_dim]
        self.delp = on_device(d.transpose(*self.order).data)                      # [outer, z, lines..]
        ln = to_compat(length)
        w_outer = [dim for dim in ln.dims if dim not in (y_dim, x_dim)]
        if w_outer != self.outer[: len(w_outer)]:
            raise ValueError(f"the edge lengths' dims {ln.dims} do not lead the field's {tuple(self.order)}")
        self.axis = 1 if self.down_dim == y_dim else 0                             # ops convention: 0 = x, 1 = y
        self.length = ops.take_lines(on_device(ln.transpose(*w_outer, y_dim, x_dim).data), f, self.axis)
        self.window = (1, f) if edge == "x" else (f, 1)
        self.delp_coarse = ops.weighted_window_average(self.delp, self.length, self.window, self.window)

    def lines_of(self, da):
        return ops.take_lines(on_device(da.transpose(*self.order).data), self.factor, self.axis)

    def mean_along_edge(self, field, weights):
        return ops.weighted_window_average(field, weights, self.window, self.window)

    def coarse_like(self, tensor, da, drop_z=False):
        """The coarse result ``tensor`` labelled as ``edge_weighted_block_average`` labels the mean of ``da``."""
        from .coarsen import _coarsen_downsample_coordinate, _coarsened_coords, coarsen_coords_coord_func

        coords = _coarsened_coords(da, {self.coarsen_dim: self.factor}, coarsen_coords_coord_func)
        down = _coarsen_downsample_coordinate(da, self.down_dim, self.factor, coarsen_coords_coord_func)
        if down is not None:
            coords[self.down_dim] = down
        dims = [dim for dim in self.order if not (drop_z and dim == self.z_dim)]
        coords = {k: v for k, v in coords.items() if not (drop_z and k == self.z_dim)}
        out = DataArray(like_input(tensor, da.data), dims=tuple(dims), name=da.name, attrs=da.attrs, coords=coords)
        return out.transpose(*[dim for dim in da.dims if dim in dims])


def edge_weighted_pressure_means(ds, delp, length, toa_pressure: float, coarsening_factor: int, x_dim: str = FV_CORE_X_CENTER,
                                 y_dim: str = FV_CORE_Y_OUTER, z_dim: str = RESTART_Z_CENTER, edge: str = "x",
                                 extrapolate: bool = False, lines: EdgeLines = None):
    """``edge_weighted_block_average(*regrid_to_edge_weighted_pressure(ds, delp, length, ...), edge=edge)``
    (coarsen_restarts.py:497-556) without the seven eighths of the work the reference throws away: the edge-weighted mean
    keeps every f-th edge line only (coarsen.py:265-271), so the edge thicknesses are interpolated, integrated, remapped and
    masked on those lines alone -- column for column the same arithmetic, hence the same values."""
    if edge not in ("x", "y"):
        raise ValueError(f"'edge' most be either 'x' or 'y'; got {edge}.")
    L = lines or EdgeLines(delp, length, coarsening_factor, edge, x_dim, y_dim, z_dim)
    f = L.factor
    d = to_compat(ds)
    names = list(d) if isinstance(d, Dataset) else None
    das = [d[v] for v in names] if names is not None else [d]
    for a in das:
        if set(a.dims) != set(L.order):
            raise ValueError(f"field dims {a.dims} do not match the edge thickness' {tuple(L.order)}")
    fields = [L.lines_of(a) for a in das]
    rep = (1, f) if edge == "x" else (f, 1)
    phalf_fine = ops.pressure_at_interface(L.delp, toa_pressure, -3)
    phalf_coarse = ops.repeat(ops.pressure_at_interface(L.delp_coarse, toa_pressure, -3), *rep)
    regridded = ops.mappm_multi(phalf_fine, fields, phalf_coarse, iv=1, kord=1, z_axis=-3)
    w = L.length
    batch = tuple(phalf_fine.shape[:-3])
    if tuple(w.shape[:-2]) != batch:  # the lengths [tile, y, x] shared by the time axis
        w = w.reshape(tuple(w.shape[:-2]) + (1,) * (len(batch) - (w.dim() - 2)) + tuple(w.shape[-2:])).expand(*batch, *w.shape[-2:]).contiguous()
    if extrapolate:
        level = ops.repeat(ops.pressure_at_midpoint_log(L.delp_coarse, toa_pressure, -3), *rep)
    else:
        level = phalf_coarse
    masked = ops.mask_weights(w, level, phalf_fine, -3, extrapolate=extrapolate)
    means = [L.coarse_like(L.mean_along_edge(r, masked), a) for r, a in zip(regridded, das)]
    if names is None:
        return from_compat(means[0], ds)
    out = Dataset(attrs=d.attrs)
    for v, m in zip(names, means):
        out[v] = m
    return from_compat(out, ds)
