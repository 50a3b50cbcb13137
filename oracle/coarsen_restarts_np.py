"""CPU restatement (numpy) of the restart coarse-graining pipelines -- TEST INFRASTRUCTURE ONLY.

Follows external/vcm/vcm/cubedsphere/coarsen_restarts.py: ``coarsen_restarts_on_sigma`` (:21-95),
``coarsen_restarts_on_pressure`` (:98-237), ``coarsen_restarts_via_blended_method`` (:240-332) and the
per-category functions they call (:335-556, :679-822, :856-1017), on plain arrays [tile, (z,) y, x]
with the Time axis squeezed.  Pinned by every array of the reference's regression fixtures
(``_coarsen_restarts_regression_tests/reference/*.json``: 7 configurations x 4 restart categories).
"""
import numpy as np

from . import coarsen_np as C
from . import sfc_data_np

FRACTION_TRACERS = ["cld_amt"]
NON_FRACTION_TRACERS = ["sphum", "liq_wat", "rainwat", "ice_wat", "snowwat", "graupel", "o3mr", "sgs_tke"]
_GRAVITY, _RDGAS, _RVGAS = 9.80665, 287.05, 461.5   # vcm/calc/thermo/constants.py:2-4


def _wavg(x, w, f):
    w = w if w.ndim == x.ndim else w[:, None]
    return C.weighted_block_average(x, w, f)


def impose_hydrostatic_balance(core, tracer, toa):
    """coarsen_restarts.py:990-1017 with vertically_dependent.py:69-99,182-186,211-235."""
    dz, phis = core["DZ"], core["phis"]
    bottom = phis[:, None] / _GRAVITY
    stack = np.concatenate([-dz, bottom], axis=1)
    height_top = np.cumsum(stack[:, ::-1], axis=1)[:, ::-1][:, 0]
    pi = C.pressure_at_interface(core["delp"], toa, 1)
    tv = core["T"] * (1 + (_RVGAS / _RDGAS - 1) * tracer["sphum"])
    dz_new = -np.diff(np.log(pi), axis=1) * _RDGAS * tv / _GRAVITY
    return {**core, "DZ": dz_new, "phis": _GRAVITY * (height_top + dz_new.sum(axis=1))}


def fv_core_on_sigma(core, delp, area, dx, dy, f, agrid, mass_weighted):
    area_vars = ["phis", "delp", "DZ"] + ([] if mass_weighted else ["W", "T"])
    mass_vars = ["W", "T"] if mass_weighted else []
    if agrid:
        (mass_vars if mass_weighted else area_vars).extend(["ua", "va"])
    out = {v: _wavg(core[v], area, f) for v in area_vars}
    out.update({v: _wavg(core[v], delp * area[:, None], f) for v in mass_vars})
    out["u"] = C.edge_weighted_block_average(core["u"], dx[:, None], f, "x")
    out["v"] = C.edge_weighted_block_average(core["v"], dy[:, None], f, "y")
    return out


def fv_core_on_pressure(core, delp, area, dx, dy, toa, f, agrid, extrapolate, mappm_fn):
    masked = ["W", "T"] + (["ua", "va"] if agrid else [])
    reg, m_area = C.regrid_to_area_weighted_pressure({v: core[v] for v in masked}, delp, area, toa, f, mappm_fn, extrapolate)
    out = {v: _wavg(core[v], area, f) for v in ["phis", "delp", "DZ"]}
    out.update({v: C.weighted_block_average(reg[v], m_area, f) for v in masked})
    for var, length, edge in (("u", dx, "x"), ("v", dy, "y")):
        r, m_len = C.regrid_to_edge_weighted_pressure({var: core[var]}, delp, length, toa, f, mappm_fn, edge, extrapolate)
        out[var] = C.edge_weighted_block_average(r[var], m_len, f, edge)
    return out


def fv_tracer_on_sigma(tracer, delp, area, f, mass_weighted):
    out = {}
    for v in FRACTION_TRACERS + NON_FRACTION_TRACERS:
        w = delp * area[:, None] if (mass_weighted and v in NON_FRACTION_TRACERS) else area
        out[v] = _wavg(tracer[v], w, f)
    return out


def fv_tracer_on_pressure(tracer, delp, area, toa, f, extrapolate, mappm_fn):
    reg, m_area = C.regrid_to_area_weighted_pressure({v: tracer[v] for v in FRACTION_TRACERS + NON_FRACTION_TRACERS},
                                                     delp, area, toa, f, mappm_fn, extrapolate)
    return {v: C.weighted_block_average(reg[v], m_area, f) for v in reg}


def coarsen_restarts(method, restarts, grid, f, toa, mappm_fn, coarsen_agrid_winds=False, mass_weighted=True,
                     extrapolate=False):
    """``method`` in {'sigma', 'pressure', 'blended'}; ``restarts``: category -> {name: array}; ``grid``:
    {'area', 'dx', 'dy'}.  Returns category -> {name: array}."""
    core, tracer = restarts["fv_core.res"], restarts["fv_tracer.res"]
    delp, area, dx, dy = core["delp"], grid["area"], grid["dx"], grid["dy"]
    out = {"fv_srf_wnd.res": {v: _wavg(restarts["fv_srf_wnd.res"][v], area, f) for v in ("u_srf", "v_srf")},
           "sfc_data": sfc_data_np.coarse_grain_sfc_data_complex(restarts["sfc_data"], area, f)}
    if method == "sigma":
        out["fv_core.res"] = fv_core_on_sigma(core, delp, area, dx, dy, f, coarsen_agrid_winds, mass_weighted)
        out["fv_tracer.res"] = fv_tracer_on_sigma(tracer, delp, area, f, mass_weighted)
        return out
    if method == "pressure":
        out["fv_core.res"] = fv_core_on_pressure(core, delp, area, dx, dy, toa, f, coarsen_agrid_winds, extrapolate, mappm_fn)
        out["fv_tracer.res"] = fv_tracer_on_pressure(tracer, delp, area, toa, f, extrapolate, mappm_fn)
    elif method == "blended":
        p_core = fv_core_on_pressure(core, delp, area, dx, dy, toa, f, coarsen_agrid_winds, False, mappm_fn)
        m_core = fv_core_on_sigma(core, delp, area, dx, dy, f, coarsen_agrid_winds, mass_weighted)
        wa = C.blending_weights_agrid(delp, area, toa, f)
        wu = C.blending_weights_dgrid(delp, dx, toa, f, "x")
        wv = C.blending_weights_dgrid(delp, dy, toa, f, "y")
        b_core = {}
        for v, ml in m_core.items():
            if v in ("u", "v"):
                b_core[v] = C.blend(wu if v == "u" else wv, p_core[v], ml)
            elif ml.ndim == 3:  # 2-D fields come from the model-level result
                b_core[v] = ml
            else:
                b_core[v] = C.blend(wa, p_core[v], ml)
        out["fv_core.res"] = b_core
        p_tr = fv_tracer_on_pressure(tracer, delp, area, toa, f, False, mappm_fn)
        m_tr = fv_tracer_on_sigma(tracer, delp, area, f, mass_weighted)
        out["fv_tracer.res"] = {v: C.blend(wa, p_tr[v], m_tr[v]) for v in m_tr}
    else:
        raise ValueError(method)
    out["fv_core.res"] = impose_hydrostatic_balance(out["fv_core.res"], out["fv_tracer.res"], toa)
    return out
