"""Shared by the oracle test and the GPU test: regenerates the synthetic restart inputs of the
reference's coarsen-restarts regression test and lists the expected outputs held by its fixtures."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "coarsen_restarts_reference.npz")


def load():
    z = np.load(GOLDEN)
    meta = json.loads(bytes(z["meta_json"]).decode())
    expected = {e["key"]: (e, z[e["key"]]) for e in meta["expected"]}
    return meta, expected


def synth_array(shape, dtype, lo, hi):
    """external/synth/synth/core.py:62-67"""
    np.random.seed(0)
    return np.random.uniform(low=lo, high=hi, size=shape).astype(dtype)


def inputs(meta):
    """category -> variable -> (dims, array), with the reference test's ranges."""
    out = {}
    for category, variables in meta["inputs"].items():
        out[category] = {}
        for name, info in variables.items():
            lo, hi = meta["ranges"].get(name, meta["default_range"])
            out[category][name] = (info["dims"], synth_array(info["shape"], np.dtype(info["dtype"]), lo, hi))
    return out


def plan(tag):
    """Which weighting each (category, variable) gets for a tag
    (external/vcm/vcm/cubedsphere/coarsen_restarts.py:335-427, 430-556, 856-987)."""
    tracers = ["sphum", "liq_wat", "rainwat", "ice_wat", "snowwat", "graupel", "o3mr", "sgs_tke"]
    if tag.startswith("area-weighted-model-level"):
        return {"area": {"fv_core.res": ["phis", "delp", "DZ", "W", "T"], "fv_tracer.res": ["cld_amt"] + tracers,
                         "fv_srf_wnd.res": ["u_srf", "v_srf"]},
                "mass": {}, "edge_x": {"fv_core.res": ["u"]}, "edge_y": {"fv_core.res": ["v"]},
                "sfc_data": True}  # the 'complex' surface-data method; the same for every tag
    if tag.startswith("mass-weighted-model-level"):
        return {"area": {"fv_core.res": ["phis", "delp", "DZ"], "fv_tracer.res": ["cld_amt"],
                         "fv_srf_wnd.res": ["u_srf", "v_srf"]},
                "mass": {"fv_core.res": ["W", "T", "ua", "va"], "fv_tracer.res": tracers},
                "edge_x": {"fv_core.res": ["u"]}, "edge_y": {"fv_core.res": ["v"]}}
    if tag.startswith("pressure-level"):
        return {"area": {"fv_core.res": ["delp"], "fv_srf_wnd.res": ["u_srf", "v_srf"]},
                "pressure": {"fv_core.res": ["W", "T", "ua", "va"], "fv_tracer.res": ["cld_amt"] + tracers},
                # D-grid winds: delp interpolated to the cell edges across the cube's faces, then remapped
                "pressure_edge_x": {"fv_core.res": ["u"]}, "pressure_edge_y": {"fv_core.res": ["v"]},
                "extrapolate": "extrapolate" in tag}
    raise KeyError(tag)


def medium_inputs(meta, n, nz, seed):
    """The fixture schema's variables, dims and value ranges on C{n} tiles with ``nz`` levels (float64 restarts):
    category -> variable -> (dims, array)."""
    rng = np.random.default_rng(seed)
    inp = {}
    for category, variables in meta["inputs"].items():
        inp[category] = {}
        for name, info in variables.items():
            lo, hi = meta["ranges"].get(name, meta["default_range"])
            shape = []
            for d, s0 in zip(info["dims"], info["shape"]):
                if d in ("tile", "Time"):
                    shape.append(s0)
                elif d.startswith("zaxis"):
                    shape.append(nz if category in ("fv_core.res", "fv_tracer.res") else s0)
                else:  # horizontal: staggered dims are one longer than the centred ones in the fixture
                    shape.append(n + (s0 - 4))
            inp[category][name] = (info["dims"], rng.uniform(lo, hi, shape).astype(info["dtype"]))
    return inp
