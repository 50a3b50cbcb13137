"""CPU restatement (numpy) of the 'complex' sfc_data coarse-graining -- TEST INFRASTRUCTURE ONLY.

Follows external/vcm/vcm/cubedsphere/coarsen_restarts.py:1111-1470 (``_coarse_grain_sfc_data_complex``,
``_compute_arguments_for_complex_sfc_coarsening``, the per-variable methods of
``SFC_DATA_COARSENING_METHOD``, ``_apply_surface_chgres_corrections``, ``_doubles_to_floats``) on plain
arrays whose last two axes are (y, x); ``area`` is [tile, y, x] and broadcasts over leading axes.
Pinned by the reference's regression fixtures ``*-sfc_data.json`` (tests/test_oracle_coarsen.py).
"""
import numpy as np

from . import coarsen_np as C

FREEZING_TEMPERATURE = 273.16   # coarsen_restarts.py:48-51
SHDMIN_THRESHOLD = 0.011
STYPE_LAND_ICE = 16.0
VTYPE_LAND_ICE = 15.0

AREA_WEIGHTED = ["tsea", "alvsf", "alvwf", "alnsf", "alnwf", "facsf", "facwf", "f10m", "t2m", "q2m", "uustar", "ffmm",
                 "ffhh", "tprcp", "snwdph"]
OVER_DOMINANT_SFC = ["tg3", "vfrac", "fice", "sncovr", "shdmin", "shdmax", "snoalb"]
VFRAC_OVER_SFC_AND_VTYPE = ["canopy", "zorl"]
OVER_SFC_AND_STYPE = ["smc", "slc", "stc"]


def _bcast(w, x):
    """[tile, y, x] weights against [tile, ..., y, x] data."""
    w = np.asarray(w)
    while w.ndim < x.ndim:
        w = w[:, None]
    return np.broadcast_to(w, x.shape)


def _where(x, mask):
    return np.where(_bcast(mask, x), x, np.nan)


def _wavg(x, w, f):
    return C.weighted_block_average(x, _bcast(w, x), f)


def coarse_grain_sfc_data_complex(ds, area, f):
    """``ds``: dict of float arrays [tile, (level,) y, x]; returns dict of float32 arrays."""
    slmsk_c = C.block_mode(ds["slmsk"], f, "omit")
    dom_sfc = np.isclose(ds["slmsk"], C.block_upsample(slmsk_c, f))
    vtype_c = C.block_mode(_where(ds["vtype"], dom_sfc), f, "omit")
    stype_c = C.block_mode(_where(ds["stype"], dom_sfc), f, "omit")
    dom_v = np.isclose(ds["vtype"], C.block_upsample(vtype_c, f))
    dom_s = np.isclose(ds["stype"], C.block_upsample(stype_c, f))
    out = {"slmsk": slmsk_c, "vtype": vtype_c, "stype": stype_c}
    for name, x in ds.items():
        if name in out:
            continue
        if name in AREA_WEIGHTED:
            out[name] = _wavg(x, area, f)
        elif name in OVER_DOMINANT_SFC:
            out[name] = _wavg(_where(x, dom_sfc), _where(_bcast(area, x), dom_sfc), f)
        elif name in VFRAC_OVER_SFC_AND_VTYPE:
            mask = dom_sfc & dom_v
            av = area * ds["vfrac"]
            a_mean = _wavg(_where(x, mask), _where(_bcast(area, x), mask), f)
            av_mean = _wavg(_where(x, mask), _where(_bcast(av, x), mask), f)
            av_sum = C.block_coarsen(_where(_bcast(av, x), mask), f, "sum")
            out[name] = np.where(av_sum > 0.0, av_mean, a_mean)
        elif name in OVER_SFC_AND_STYPE:
            mask = dom_sfc & dom_s
            out[name] = _wavg(_where(x, mask), _where(_bcast(area, x), mask), f)
        elif name == "srflag":
            out[name] = C.block_mode(x, f, "omit")
        elif name == "slope":
            out[name] = C.block_mode(_where(x, dom_sfc), f, "omit")
        elif name == "sheleg":
            r = _wavg(x, area * ds["sncovr"], f)
            out[name] = np.where(np.isnan(r), 0.0, r)
        elif name == "hice":
            r = _wavg(x, area * ds["fice"], f)
            out[name] = np.where(np.isnan(r), 0.0, r)
        elif name == "tisfc":
            sea_ice = _wavg(_where(x, dom_sfc), _where(_bcast(area * ds["fice"], x), dom_sfc), f)
            other = _wavg(_where(x, dom_sfc), _where(_bcast(area, x), dom_sfc), f)
            out[name] = np.where(np.isclose(slmsk_c, 2.0), sea_ice, other)
        else:
            raise KeyError(f"no coarsening method for {name!r}")
    # surface_chgres corrections (coarsen_restarts.py:1403-1470)
    land_ice = np.isclose(out["vtype"], VTYPE_LAND_ICE)
    clip = lambda a: np.where(a < FREEZING_TEMPERATURE, a, FREEZING_TEMPERATURE)
    out["tsea"] = np.where(land_ice, clip(out["tsea"]), out["tsea"])
    out["tg3"] = np.where(land_ice, clip(out["tg3"]), out["tg3"])
    out["stype"] = np.where(land_ice, STYPE_LAND_ICE, out["stype"])
    out["canopy"] = np.where(out["shdmin"] < SHDMIN_THRESHOLD, 0.0, out["canopy"])
    out["shdmin"] = np.where(np.isclose(out["vtype"], VTYPE_LAND_ICE), 0.0, out["shdmin"])
    return {k: np.asarray(v).astype(np.float32) for k, v in out.items()}
