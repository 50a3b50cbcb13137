"""The restart coarse-graining pipelines of ``vcm.cubedsphere.coarsen_restarts``
(external/vcm/vcm/cubedsphere/coarsen_restarts.py:21-332 and the per-category functions they call,
:335-556, :559-822, :856-1017), composed from the device entry points of this package.

``restarts``: mapping with the keys "fv_core.res", "fv_srf_wnd.res", "fv_tracer.res", "sfc_data"
(Datasets with the restart files' dimension names); ``grid_spec``: Dataset with ``area``, ``dx``, ``dy``
on the ``grid_xt/grid_yt/grid_x/grid_y`` dims.  Returns the same mapping, ``coarsening_factor`` times
coarser.  The reference's regression fixtures for all seven configurations are reproduced through
these functions (tests/test_gpu_api.py).
"""
import os
from typing import Hashable, Mapping

from .. import ops
from ..xr_compat import DataArray, Dataset, from_compat, merge, to_compat
from ._device import like_input, on_device
from .coarsen import (block_coarsen, block_edge_coarsen, edge_weighted_block_average, mass_weighted_block_average,
                      weighted_block_average)
from .constants import (
    COORD_X_CENTER,
    COORD_X_OUTER,
    COORD_Y_CENTER,
    COORD_Y_OUTER,
    FV_CORE_X_CENTER,
    FV_CORE_X_OUTER,
    FV_CORE_Y_CENTER,
    FV_CORE_Y_OUTER,
    FV_SRF_WND_X_CENTER,
    FV_SRF_WND_Y_CENTER,
    FV_TRACER_X_CENTER,
    FV_TRACER_Y_CENTER,
    RESTART_Z_CENTER,
    SFC_DATA_X_CENTER,
    SFC_DATA_Y_CENTER,
)
from .regridz import (EdgeLines, area_weighted_pressure_means, compute_edge_delp, edge_weighted_pressure_means,
                      pressure_at_midpoint_log, regrid_to_area_weighted_pressure, regrid_to_edge_weighted_pressure)  # (pressure_at_midpoint_log: thermo imports this package's device helpers)
from .sfc_data import _coarse_grain_sfc_data_complex

CATEGORY_LIST = ["fv_core.res", "fv_srf_wnd.res", "fv_tracer.res", "sfc_data"]
SIGMA_BLEND = 0.9
FRACTION_TRACERS = ["cld_amt"]
NON_FRACTION_TRACERS = ["sphum", "liq_wat", "rainwat", "ice_wat", "snowwat", "graupel", "o3mr", "sgs_tke"]


def _mul(a: DataArray, b: DataArray) -> DataArray:
    """``a * b`` for a [.., z, y, x] field and a [.., y, x] weight (``delp * area``), on the device."""
    a, b = to_compat(a), to_compat(b)
    extra = [d for d in a.dims[:-2] if d not in b.dims]   # e.g. Time, z: the weight is repeated over these
    lead = [d for d in a.dims[:-2] if d in b.dims]
    tdims = lead + extra + list(a.dims[-2:])
    at = on_device(a.transpose(*tdims).data)
    bt = on_device(b.transpose(*(lead + list(a.dims[-2:]))).data)
    n_lead = len(lead)
    flat = at.reshape(tuple(at.shape[:n_lead]) + (-1,) + tuple(at.shape[-2:]))
    res = ops.ew("mul", flat, bt).reshape(at.shape)
    return DataArray(like_input(res, a.data), dims=tuple(tdims)).transpose(*a.dims)


# ------------------------------------------------------------------------------------------------
# model-level ("sigma") coarse-graining
# ------------------------------------------------------------------------------------------------
def _coarse_grain_fv_core(ds, delp, area, dx, dy, coarsening_factor, coarsen_agrid_winds=False, mass_weighted=True):
    """coarsen_restarts.py:335-427."""
    if mass_weighted:
        area_weighted_vars, mass_weighted_vars = ["phis", "delp", "DZ"], ["W", "T"]
    else:
        area_weighted_vars, mass_weighted_vars = ["phis", "delp", "DZ", "W", "T"], []
    if coarsen_agrid_winds:
        if not ("ua" in ds and "va" in ds):
            raise ValueError("If 'coarsen_agrid_winds' is active, 'ua' and 'va' must be present in the 'fv_core.res' restart files.")
        (mass_weighted_vars if mass_weighted else area_weighted_vars).extend(["ua", "va"])
    parts = [weighted_block_average(ds[area_weighted_vars], area, coarsening_factor, x_dim=FV_CORE_X_CENTER, y_dim=FV_CORE_Y_CENTER)]
    if mass_weighted_vars:
        parts.append(mass_weighted_block_average(ds[mass_weighted_vars], delp, area, coarsening_factor,
                                                 x_dim=FV_CORE_X_CENTER, y_dim=FV_CORE_Y_CENTER))
    parts.append(edge_weighted_block_average(ds[["u"]], dx, coarsening_factor, x_dim=FV_CORE_X_CENTER, y_dim=FV_CORE_Y_OUTER, edge="x"))
    parts.append(edge_weighted_block_average(ds[["v"]], dy, coarsening_factor, x_dim=FV_CORE_X_OUTER, y_dim=FV_CORE_Y_CENTER, edge="y"))
    return merge(parts)


def _coarse_grain_fv_tracer(ds, delp, area, coarsening_factor, mass_weighted=True):
    """coarsen_restarts.py:856-900."""
    if mass_weighted:
        area_weighted_vars, mass_weighted_vars = FRACTION_TRACERS, NON_FRACTION_TRACERS
    else:
        area_weighted_vars, mass_weighted_vars = FRACTION_TRACERS + NON_FRACTION_TRACERS, []
    parts = [weighted_block_average(ds[area_weighted_vars], area, coarsening_factor, x_dim=FV_TRACER_X_CENTER, y_dim=FV_TRACER_Y_CENTER)]
    if mass_weighted_vars:
        parts.append(mass_weighted_block_average(ds[mass_weighted_vars], delp, area, coarsening_factor,
                                                 x_dim=FV_TRACER_X_CENTER, y_dim=FV_TRACER_Y_CENTER))
    return merge(parts)


def _coarse_grain_fv_srf_wnd(ds, area, coarsening_factor):
    """coarsen_restarts.py:964-987."""
    return weighted_block_average(ds[["u_srf", "v_srf"]], area, coarsening_factor, x_dim="xaxis_1", y_dim="yaxis_1")


# ------------------------------------------------------------------------------------------------
# pressure-level coarse-graining
# ------------------------------------------------------------------------------------------------
def _masked_core_vars(ds, coarsen_agrid_winds):
    masked_area_weighted_vars = ["W", "T"]
    if coarsen_agrid_winds:
        if not ("ua" in ds and "va" in ds):
            raise ValueError("If 'coarsen_agrid_winds' is active, 'ua' and 'va' must be present in the 'fv_core.res' restart files.")
        masked_area_weighted_vars.extend(["ua", "va"])
    return masked_area_weighted_vars


def _area_weighted_pressure_means(core, tracer, delp, area, toa_pressure, coarsening_factor, coarsen_agrid_winds, extrapolate,
                                  side_stream=None, side_work=None):
    """The cell-centred fields of fv_core (W, T, ua, va) and all tracers are remapped between the same two pressure
    grids with the same masked area weights (coarsen_restarts.py:483-495 and :940-961 compute them twice): here
    once -- one pressure context and one multi-field remap for the 11-13 fields -- with identical results per field.
    Returns (coarse fv_core fields, coarse tracers)."""
    names_core = _masked_core_vars(core, coarsen_agrid_winds)
    names_tracer = FRACTION_TRACERS + NON_FRACTION_TRACERS
    t = to_compat(tracer)[names_tracer].rename({FV_TRACER_Y_CENTER: FV_CORE_Y_CENTER})
    both = merge([to_compat(core)[names_core], t])
    # (regrid_to_area_weighted_pressure + weighted_block_average as a two-stream pipeline over groups of four fields)
    means = to_compat(area_weighted_pressure_means(both, delp, area, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_CENTER,
                                                   y_dim=FV_CORE_Y_CENTER, extrapolate=extrapolate, side_stream=side_stream,
                                                   side_work=side_work))
    return means[names_core], means[names_tracer].rename({FV_CORE_Y_CENTER: FV_TRACER_Y_CENTER})


def _coarse_grain_fv_core_on_pressure(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, coarsen_agrid_winds=False,
                                      extrapolate=False, area_means=None, edge_lines=None):
    """coarsen_restarts.py:430-556: delp, DZ, phis on model surfaces, the rest on surfaces of constant pressure.
    ``area_means``: the coarse W, T, (ua, va) when the caller has them already (``_area_weighted_pressure_means``)."""
    masked_area_weighted_vars = _masked_core_vars(ds, coarsen_agrid_winds)
    if area_means is None:
        area_regridded, masked_area = regrid_to_area_weighted_pressure(
            ds[masked_area_weighted_vars], delp, area, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_CENTER,
            y_dim=FV_CORE_Y_CENTER, extrapolate=extrapolate)
        area_means = weighted_block_average(area_regridded, masked_area, coarsening_factor, x_dim=FV_CORE_X_CENTER,
                                            y_dim=FV_CORE_Y_CENTER)
    plain, u_mean, v_mean = _fv_core_on_pressure_beside(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, extrapolate, edge_lines)
    return merge([plain, area_means, u_mean, v_mean])


def _fv_core_on_pressure_beside(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, extrapolate=False, edge_lines=None):
    """What ``_coarse_grain_fv_core_on_pressure`` computes beside the remapped cell-centred fields, none of it depending on
    them: (phis, delp, DZ on model surfaces; the pressure-level mean of u; of v)."""
    # the D-grid winds: remapped and averaged on the edge lines the average keeps (regridz.edge_weighted_pressure_means ==
    # edge_weighted_block_average(*regrid_to_edge_weighted_pressure(...)), coarsen_restarts.py:497-540)
    edge_lines = edge_lines or {}
    u_mean = edge_weighted_pressure_means(ds[["u"]], delp, dx, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_CENTER,
                                          y_dim=FV_CORE_Y_OUTER, edge="x", extrapolate=extrapolate, lines=edge_lines.get("x"))
    v_mean = edge_weighted_pressure_means(ds[["v"]], delp, dy, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_OUTER,
                                          y_dim=FV_CORE_Y_CENTER, edge="y", extrapolate=extrapolate, lines=edge_lines.get("y"))
    plain = weighted_block_average(ds[["phis", "delp", "DZ"]], area, coarsening_factor, x_dim=FV_CORE_X_CENTER, y_dim=FV_CORE_Y_CENTER)
    return plain, u_mean, v_mean


def _coarse_grain_fv_tracer_on_pressure(ds, delp, area, toa_pressure, coarsening_factor, extrapolate=False):
    """coarsen_restarts.py:903-961."""
    ds_regridded, masked_area = regrid_to_area_weighted_pressure(
        ds, delp, area, toa_pressure, coarsening_factor, x_dim=FV_TRACER_X_CENTER, y_dim=FV_TRACER_Y_CENTER,
        extrapolate=extrapolate)
    return weighted_block_average(ds_regridded[FRACTION_TRACERS + NON_FRACTION_TRACERS], masked_area, coarsening_factor,
                                  x_dim=FV_TRACER_X_CENTER, y_dim=FV_TRACER_Y_CENTER)


# ------------------------------------------------------------------------------------------------
# blending and hydrostatic balance
# ------------------------------------------------------------------------------------------------
def surface_pressure_from_delp(delp, p_toa: float = 300.0, vertical_dim: Hashable = "z"):
    """``delp.sum(vertical_dim) + p_toa`` (vertically_dependent.py:189-208)."""
    d = to_compat(delp)
    axis = d.get_axis_num(vertical_dim)
    res = ops.column_sum(on_device(d.data), axis, addend=p_toa)
    dims = tuple(k for k in d.dims if k != vertical_dim)
    out = DataArray(like_input(res, d.data), dims=dims, coords={k: v for k, v in d.coords.items() if k != vertical_dim},
                    attrs={"long_name": "surface pressure", "units": "Pa"})
    return from_compat(out, delp)


def compute_blending_weights(blending_pressure, ps_coarse, pfull_coarse):
    """coarsen_restarts.py:559-576."""
    pb, ps, pf = to_compat(blending_pressure), to_compat(ps_coarse), to_compat(pfull_coarse)
    zdims = [d for d in pf.dims if d not in ps.dims]
    if len(zdims) != 1:
        raise ValueError("pfull_coarse must have exactly one (vertical) dimension more than ps_coarse")
    axis = pf.get_axis_num(zdims[0])
    order = [d for d in pf.dims if d != zdims[0]]
    res = ops.blend_weights(on_device(pb.transpose(*order).data), on_device(ps.transpose(*order).data), on_device(pf.data), axis)
    return from_compat(DataArray(like_input(res, pf.data), dims=pf.dims, coords=dict(pf.coords)), pfull_coarse)


def _scale(da, factor: float):
    d = to_compat(da)
    t = on_device(d.data)
    return d._replace(data=like_input(ops.ew("mul_s", t, scalar=factor), d.data))


def _compute_blending_weights_agrid(delp, area, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_CENTER, y_dim=FV_CORE_Y_CENTER):
    """coarsen_restarts.py:579-622."""
    delp_coarse = weighted_block_average(delp, area, coarsening_factor, x_dim=x_dim, y_dim=y_dim)
    pfull_coarse = pressure_at_midpoint_log(delp_coarse, toa_pressure=toa_pressure, dim=RESTART_Z_CENTER)
    ps = surface_pressure_from_delp(delp, p_toa=toa_pressure, vertical_dim=RESTART_Z_CENTER)
    ps_coarse = surface_pressure_from_delp(delp_coarse, p_toa=toa_pressure, vertical_dim=RESTART_Z_CENTER)
    blending_pressure = _scale(block_coarsen(ps, coarsening_factor, x_dim=x_dim, y_dim=y_dim, method="min"), SIGMA_BLEND)
    return compute_blending_weights(blending_pressure, ps_coarse, pfull_coarse)


def _compute_blending_weights_dgrid(delp, length, toa_pressure, coarsening_factor, edge, x_dim, y_dim, lines=None):
    """coarsen_restarts.py:625-661, on the edge lines the edge-weighted / edge-min reductions keep (``lines``: the
    pipeline's EdgeLines of this component, else built here)."""
    L = lines or EdgeLines(delp, length, coarsening_factor, edge, x_dim, y_dim)
    pfull_coarse = ops.pressure_at_midpoint_log(L.delp_coarse, toa_pressure, -3)
    ps = ops.column_sum(L.delp, -3, addend=toa_pressure)
    ps_coarse = ops.column_sum(L.delp_coarse, -3, addend=toa_pressure)
    blending_pressure = ops.ew("mul_s", ops.block_reduce(ps, L.window, L.window, op="min"), scalar=SIGMA_BLEND)
    res = ops.blend_weights(blending_pressure, ps_coarse, pfull_coarse, pfull_coarse.dim() - 3)
    return DataArray(res, dims=tuple(L.order))


def _blend_da(weights: DataArray, pressure_level: DataArray, model_level: DataArray) -> DataArray:
    w = weights.transpose(*pressure_level.dims)
    m = model_level.transpose(*pressure_level.dims)
    res = ops.ew("blend", on_device(w.data), on_device(pressure_level.data), on_device(m.data))
    return pressure_level._replace(data=like_input(res, pressure_level.data))


def blend(weights, pressure_level, model_level):
    """``weights * pressure_level + (1 - weights) * model_level`` (coarsen_restarts.py:664-676)."""
    w, p, m = to_compat(weights), to_compat(pressure_level), to_compat(model_level)
    if isinstance(p, Dataset):
        out = Dataset(attrs=p.attrs)
        for name in p:
            out[name] = _blend_da(w, p[name], m[name])
        return from_compat(out, pressure_level)
    return from_compat(_blend_da(w, p, m), pressure_level)


def _impose_hydrostatic_balance(ds_fv_core, ds_fv_tracer, toa_pressure, dim=RESTART_Z_CENTER):
    """Layer thicknesses from hydrostatic balance, surface geopotential adjusted to keep the model-top
    height (coarsen_restarts.py:990-1017)."""
    core, tracer = to_compat(ds_fv_core), to_compat(ds_fv_tracer)
    dz = core["DZ"]
    order = list(dz.dims)
    axis = order.index(dim)
    sphum = tracer["sphum"].rename({FV_TRACER_Y_CENTER: FV_CORE_Y_CENTER}).transpose(*order)
    phis_order = [d for d in order if d != dim]
    new_dz, new_phis = ops.hydrostatic_balance(
        on_device(dz.data), on_device(core["phis"].transpose(*phis_order).data), on_device(core["T"].transpose(*order).data),
        on_device(sphum.data), on_device(core["delp"].transpose(*order).data), toa_pressure, axis)
    out = Dataset(attrs=core.attrs)
    for name in core:
        out[name] = core[name]
    out["DZ"] = dz._replace(data=like_input(new_dz, dz.data))
    out["phis"] = DataArray(like_input(new_phis, dz.data), dims=tuple(phis_order), attrs=core["phis"].attrs).transpose(*core["phis"].dims)
    return from_compat(out, ds_fv_core)


def _names(ds, with_z: bool):
    return [v for v in ds if (RESTART_Z_CENTER in ds[v].dims) == with_z]


def _coarse_grain_fv_core_via_blended_method(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, coarsen_agrid_winds=False,
                                             mass_weighted=True, area_means=None):
    """coarsen_restarts.py:679-778."""
    beside = _fv_core_blended_beside(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, coarsen_agrid_winds, mass_weighted)
    return _fv_core_blended_finish(ds, beside, area_means, delp, area, toa_pressure, coarsening_factor, coarsen_agrid_winds)


def _fv_core_blended_beside(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, coarsen_agrid_winds=False, mass_weighted=True):
    """Everything of the blended fv_core result that does not depend on the remapped cell-centred fields: the pressure-level
    winds and model-surface fields, the whole model-level result, the three sets of blending weights."""
    # the edge thicknesses of u and v on the lines their means keep: one interpolation across the cube faces per component,
    # shared by the pressure-level remap and the blending weights
    edge_lines = {"x": EdgeLines(delp, dx, coarsening_factor, "x", FV_CORE_X_CENTER, FV_CORE_Y_OUTER),
                  "y": EdgeLines(delp, dy, coarsening_factor, "y", FV_CORE_X_OUTER, FV_CORE_Y_CENTER)}
    on_pressure = _fv_core_on_pressure_beside(ds, delp, area, dx, dy, toa_pressure, coarsening_factor, False, edge_lines)
    model_level = to_compat(_coarse_grain_fv_core(ds, delp, area, dx, dy, coarsening_factor, coarsen_agrid_winds, mass_weighted))
    weights_agrid = _compute_blending_weights_agrid(delp, area, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_CENTER,
                                                    y_dim=FV_CORE_Y_CENTER)
    weights_u = _compute_blending_weights_dgrid(delp, dx, toa_pressure, coarsening_factor, "x", x_dim=FV_CORE_X_CENTER,
                                                y_dim=FV_CORE_Y_OUTER, lines=edge_lines["x"])
    weights_v = _compute_blending_weights_dgrid(delp, dy, toa_pressure, coarsening_factor, "y", x_dim=FV_CORE_X_OUTER,
                                                y_dim=FV_CORE_Y_CENTER, lines=edge_lines["y"])
    return on_pressure, model_level, weights_agrid, weights_u, weights_v


def _fv_core_blended_finish(ds, beside, area_means, delp, area, toa_pressure, coarsening_factor, coarsen_agrid_winds):
    on_pressure, model_level, weights_agrid, weights_u, weights_v = beside
    if area_means is None:
        regridded, masked_area = regrid_to_area_weighted_pressure(
            ds[_masked_core_vars(ds, coarsen_agrid_winds)], delp, area, toa_pressure, coarsening_factor, x_dim=FV_CORE_X_CENTER,
            y_dim=FV_CORE_Y_CENTER)
        area_means = weighted_block_average(regridded, masked_area, coarsening_factor, x_dim=FV_CORE_X_CENTER, y_dim=FV_CORE_Y_CENTER)
    pressure_level = to_compat(merge([on_pressure[0], area_means, on_pressure[1], on_pressure[2]]))
    d = to_compat(ds)
    ignore = ["u", "v"] + ([] if coarsen_agrid_winds else ["ua", "va"])
    names_2d = _names(d, False)
    names_3d = [v for v in _names(d, True) if v not in ignore]
    return merge([
        model_level[names_2d],  # 2-D fields could come from either result
        blend(weights_agrid, pressure_level[names_3d], model_level[names_3d]),
        blend(weights_u, pressure_level["u"], model_level["u"]).rename("u"),
        blend(weights_v, pressure_level["v"], model_level["v"]).rename("v"),
    ])


def _coarse_grain_fv_tracer_via_blended_method(ds, delp, area, toa_pressure, coarsening_factor, mass_weighted=True,
                                               pressure_level=None):
    """coarsen_restarts.py:781-822.  ``pressure_level``: the pressure-level result when the caller has it already."""
    if pressure_level is None:
        pressure_level = _coarse_grain_fv_tracer_on_pressure(ds, delp, area, toa_pressure, coarsening_factor)
    weights, model_level = _fv_tracer_blended_beside(ds, delp, area, toa_pressure, coarsening_factor, mass_weighted)
    return blend(weights, pressure_level, model_level)


def _fv_tracer_blended_beside(ds, delp, area, toa_pressure, coarsening_factor, mass_weighted=True):
    """(blending weights, model-level tracers): what the blended tracers need beside their pressure-level means."""
    model_level = _coarse_grain_fv_tracer(ds, delp, area, coarsening_factor, mass_weighted)
    weights = _compute_blending_weights_agrid(delp, area, toa_pressure, coarsening_factor, x_dim=FV_TRACER_X_CENTER,
                                              y_dim=FV_TRACER_Y_CENTER)
    return weights, model_level


# ------------------------------------------------------------------------------------------------
# the three pipelines
# ------------------------------------------------------------------------------------------------
def _sync_dimension_order(a, b):
    a, b = to_compat(a), to_compat(b)
    out = Dataset(attrs=a.attrs)
    for var in a:
        out[var] = a[var].transpose(*b[var].dims)
    return out


def _grid(grid_spec, name: str, x_dim, y_dim):
    g = to_compat(grid_spec)[name]
    ren = {COORD_X_CENTER: x_dim, COORD_X_OUTER: x_dim, COORD_Y_CENTER: y_dim, COORD_Y_OUTER: y_dim}
    return g.rename({d: ren[d] for d in g.dims if d in ren})


def _common(coarsening_factor, grid_spec, restarts):
    out = {}
    out["fv_srf_wnd.res"] = _coarse_grain_fv_srf_wnd(
        restarts["fv_srf_wnd.res"], _grid(grid_spec, "area", FV_SRF_WND_X_CENTER, FV_SRF_WND_Y_CENTER), coarsening_factor)
    out["sfc_data"] = _coarse_grain_sfc_data_complex(
        restarts["sfc_data"], _grid(grid_spec, "area", SFC_DATA_X_CENTER, SFC_DATA_Y_CENTER), coarsening_factor)
    return out


def _entry_event():
    """An event on the calling stream at the moment a pipeline is entered: what the surface categories' side stream has to
    wait for (the caller's inputs) -- not the 3-D work the pipeline enqueues afterwards."""
    import torch

    from ._device import compute_device

    dev = compute_device()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return ev


def _side_stream(name, entry_event):
    """The side stream ``name`` of the device, made to wait for the pipeline's inputs (``entry_event``).  Two side streams
    per pipeline at most (this one and the surface categories'): the HIP runtime multiplexes streams onto four hardware
    queues, and kernels of independent streams that share a queue run one after the other."""
    import torch

    from ._device import compute_device

    dev = compute_device()
    if os.environ.get("FV3NET_AMD_PIPELINE_STREAMS", "1") == "0":   # (everything on the calling stream: for A/B timing)
        return torch.cuda.current_stream(dev)
    from ._device import side_streams

    side = side_streams(dev)[{"beside": 0, "surface": 1}[name]]
    side.wait_event(entry_event)
    return side


def _join(result, side):
    """The calling stream waits for ``side``; ``result`` (memory of the side stream's pool) is the caller's from here on.
    No ``record_stream``: a block that returns to the side stream's pool is handed out again only to work on that stream, and
    every use of a side stream begins by waiting for an event the calling stream records at the NEXT pipeline entry --
    after whatever the caller enqueued on these results before dropping them.  (Recording ~70 small arrays per call made
    the allocator track an event per array and cost the eager call more than the overlap gained.)"""
    import torch

    from ._device import compute_device

    torch.cuda.current_stream(compute_device()).wait_stream(side)
    return result


def _common_beside(coarsening_factor, grid_spec, restarts, entry_event):
    """``_common`` on a second HIP stream: the surface categories are some 150 launches on 2-D fields (a few microseconds of
    device time each) that depend on nothing the 3-D categories produce -- issued after them, on a stream of their own that
    only waits for the pipeline's inputs, they run in the shadow of the large kernels instead of behind them.  The calling
    stream waits for the side stream before the results are handed out."""
    import torch

    from ._device import compute_device

    dev = compute_device()
    main = torch.cuda.current_stream(dev)
    from ._device import side_streams

    side = side_streams(dev)[1]   # (also under the A/B switch, which moves the 3-D branches only)
    side.wait_event(entry_event)
    with torch.cuda.stream(side):
        out = _common(coarsening_factor, grid_spec, restarts)
    main.wait_stream(side)   # (memory of the side stream's pool, the caller's from here on: see _join)
    return out


def _finish(coarsened, restarts):
    return {category: from_compat(_sync_dimension_order(coarsened[category], restarts[category]), restarts[category])
            for category in CATEGORY_LIST}


def coarsen_restarts_on_sigma(coarsening_factor: int, grid_spec, restarts: Mapping[str, object], coarsen_agrid_winds: bool = False,
                              mass_weighted: bool = True):
    """Coarsen a complete set of restart files on model levels, 'complex' surface method (coarsen_restarts.py:21-95)."""
    core = to_compat(restarts["fv_core.res"])
    coarsened, entered = {}, _entry_event()
    coarsened["fv_core.res"] = _coarse_grain_fv_core(
        core, core["delp"], _grid(grid_spec, "area", FV_CORE_X_CENTER, FV_CORE_Y_CENTER),
        _grid(grid_spec, "dx", FV_CORE_X_CENTER, FV_CORE_Y_OUTER), _grid(grid_spec, "dy", FV_CORE_X_OUTER, FV_CORE_Y_CENTER),
        coarsening_factor, coarsen_agrid_winds, mass_weighted)
    coarsened["fv_tracer.res"] = _coarse_grain_fv_tracer(
        restarts["fv_tracer.res"], core["delp"].rename({FV_CORE_Y_CENTER: FV_TRACER_Y_CENTER}),
        _grid(grid_spec, "area", FV_TRACER_X_CENTER, FV_TRACER_Y_CENTER), coarsening_factor, mass_weighted)
    coarsened.update(_common_beside(coarsening_factor, grid_spec, restarts, entered))
    return _finish(coarsened, restarts)


def coarsen_restarts_on_pressure(coarsening_factor: int, grid_spec, toa_pressure: float, restarts: Mapping[str, object],
                                 coarsen_agrid_winds: bool = False, extrapolate: bool = False):
    """Coarsen a complete set of restart files on surfaces of constant pressure, then impose hydrostatic
    balance (coarsen_restarts.py:98-237)."""
    core = to_compat(restarts["fv_core.res"])
    coarsened, entered = {}, _entry_event()
    area = _grid(grid_spec, "area", FV_CORE_X_CENTER, FV_CORE_Y_CENTER)
    # Two branches that meet at the merge: the remap sweeps of the 11-13 cell-centred fields (latency-bound kernels that fill
    # the chip's wave slots but not its memory system) on the calling stream, and beside them -- on a stream of its own that
    # waits only for the inputs -- the D-grid winds on their edge lines and the model-surface fields (short launches, 1 ms).
    # (the sweeps are ENQUEUED first -- a few launches -- so that the device has its long kernels while the host is still
    # issuing the many short ones of the other branch: eager calls are bound by the ~20 us of Python per launch)
    side, beside = _side_stream("beside", entered), []
    core_means, coarsened["fv_tracer.res"] = _area_weighted_pressure_means(
        core, restarts["fv_tracer.res"], core["delp"], area, toa_pressure, coarsening_factor, coarsen_agrid_winds, extrapolate,
        side_stream=side, side_work=lambda: beside.append(_fv_core_on_pressure_beside(
            core, core["delp"], area, _grid(grid_spec, "dx", FV_CORE_X_CENTER, FV_CORE_Y_OUTER),
            _grid(grid_spec, "dy", FV_CORE_X_OUTER, FV_CORE_Y_CENTER), toa_pressure, coarsening_factor, extrapolate)))
    plain, u_mean, v_mean = _join(beside[0], side)
    coarsened["fv_core.res"] = merge([plain, core_means, u_mean, v_mean])
    coarsened["fv_core.res"] = _impose_hydrostatic_balance(coarsened["fv_core.res"], coarsened["fv_tracer.res"], toa_pressure)
    coarsened.update(_common_beside(coarsening_factor, grid_spec, restarts, entered))
    return _finish(coarsened, restarts)


def coarsen_restarts_via_blended_method(coarsening_factor: int, grid_spec, toa_pressure: float, restarts: Mapping[str, object],
                                        coarsen_agrid_winds: bool = False, mass_weighted: bool = True):
    """Blended pressure-level / model-level coarse-graining of the 3-D fields (coarsen_restarts.py:240-332)."""
    core = to_compat(restarts["fv_core.res"])
    coarsened, entered = {}, _entry_event()
    area = _grid(grid_spec, "area", FV_CORE_X_CENTER, FV_CORE_Y_CENTER)
    delp_t = core["delp"].rename({FV_CORE_Y_CENTER: FV_TRACER_Y_CENTER})
    area_t = _grid(grid_spec, "area", FV_TRACER_X_CENTER, FV_TRACER_Y_CENTER)
    # as in coarsen_restarts_on_pressure: the remap sweeps on the calling stream; beside them everything the blend needs
    # that does not come out of a sweep (pressure-level winds, the whole model-level result, the blending weights -- HBM-bound
    # passes that fit beside the latency-bound sweeps)
    side, beside = _side_stream("beside", entered), []
    core_means, tracer_means = _area_weighted_pressure_means(
        core, restarts["fv_tracer.res"], core["delp"], area, toa_pressure, coarsening_factor, coarsen_agrid_winds, False,
        side_stream=side, side_work=lambda: beside.append((
            _fv_core_blended_beside(core, core["delp"], area, _grid(grid_spec, "dx", FV_CORE_X_CENTER, FV_CORE_Y_OUTER),
                                    _grid(grid_spec, "dy", FV_CORE_X_OUTER, FV_CORE_Y_CENTER), toa_pressure, coarsening_factor,
                                    coarsen_agrid_winds, mass_weighted),
            _fv_tracer_blended_beside(restarts["fv_tracer.res"], delp_t, area_t, toa_pressure, coarsening_factor, mass_weighted))))
    core_beside, (tracer_weights, tracer_model_level) = _join(beside[0], side)
    coarsened["fv_core.res"] = _fv_core_blended_finish(core, core_beside, core_means, core["delp"], area, toa_pressure,
                                                      coarsening_factor, coarsen_agrid_winds)
    coarsened["fv_tracer.res"] = blend(tracer_weights, tracer_means, tracer_model_level)
    coarsened["fv_core.res"] = _impose_hydrostatic_balance(coarsened["fv_core.res"], coarsened["fv_tracer.res"], toa_pressure)
    coarsened.update(_common_beside(coarsening_factor, grid_spec, restarts, entered))
    return _finish(coarsened, restarts)
