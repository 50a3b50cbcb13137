"""'Complex' coarse-graining of the ``sfc_data`` restart category on the device
(external/vcm/vcm/cubedsphere/coarsen_restarts.py:1111-1470: ``_coarse_grain_sfc_data_complex``,
``_compute_arguments_for_complex_sfc_coarsening``, the per-variable methods of
``SFC_DATA_COARSENING_METHOD`` and ``_apply_surface_chgres_corrections``).

The categorical fields (``slmsk``, ``vtype``, ``stype``, ``srflag``, ``slope``) are block MODES
(``scipy.stats.mode`` semantics, NaNs omitted -- bit-exact index maps); everything else is an area-weighted
mean restricted to the cells of the dominant surface / vegetation / soil type, with the
``surface_chgres`` corrections applied to the coarse result.  Every step is a HIP launch: block mode,
block upsample, NaN-skipping weighted block average / block sum, and the elementwise mask vocabulary
``fv3hip_ew``; fields of a call stay on the device until the end.
"""
from typing import Dict, Mapping

import torch

from .. import ops
from ..xr_compat import DataArray, Dataset, from_compat, to_compat
from ._device import like_input, on_device

X_DIM = "xaxis_1"
Y_DIM = "yaxis_1"
FREEZING_TEMPERATURE = 273.16   # coarsen_restarts.py:48-51
SHDMIN_THRESHOLD = 0.011
STYPE_LAND_ICE = 16.0
VTYPE_LAND_ICE = 15.0

_AREA_WEIGHTED = ("tsea", "alvsf", "alvwf", "alnsf", "alnwf", "facsf", "facwf", "f10m", "t2m", "q2m", "uustar", "ffmm",
                  "ffhh", "tprcp", "snwdph")
_OVER_DOMINANT_SFC = ("tg3", "vfrac", "fice", "sncovr", "shdmin", "shdmax", "snoalb")
_VFRAC_OVER_SFC_AND_VTYPE = ("canopy", "zorl")
_OVER_SFC_AND_STYPE = ("smc", "slc", "stc")
SFC_DATA_VARIABLES = (("slmsk", "vtype", "stype", "srflag", "slope", "sheleg", "hice", "tisfc") + _AREA_WEIGHTED
                      + _OVER_DOMINANT_SFC + _VFRAC_OVER_SFC_AND_VTYPE + _OVER_SFC_AND_STYPE)


def _mode(x: torch.Tensor, f: int) -> torch.Tensor:
    return ops.block_reduce(x, (f, f), op="mode", nan_policy="omit")


def coarse_grain_sfc_data_tensors(fields: Mapping[str, torch.Tensor], area: torch.Tensor, f: int) -> Dict[str, torch.Tensor]:
    """The method on device tensors whose last two dims are (y, x): 2-D fields share their leading
    dims with ``area``; 3-D (soil level) fields have one extra axis before (y, x)."""
    dt = torch.float64 if any(t.dtype == torch.float64 for t in fields.values()) or area.dtype == torch.float64 else torch.float32
    names = list(fields)
    fields = dict(zip(names, ops.cast_many([fields[n] for n in names], dt)))  # (one launch for all that need it)
    area = ops.cast(area, dt)
    slmsk_c = _mode(fields["slmsk"], f)
    dom_sfc = ops.ew("isclose", fields["slmsk"], ops.block_upsample(slmsk_c, f))
    vtype_c = _mode(ops.ew("where_nan", fields["vtype"], dom_sfc), f)
    stype_c = _mode(ops.ew("where_nan", fields["stype"], dom_sfc), f)
    dom_v = ops.ew("isclose", fields["vtype"], ops.block_upsample(vtype_c, f))
    dom_s = ops.ew("isclose", fields["stype"], ops.block_upsample(stype_c, f))
    out = {"slmsk": slmsk_c, "vtype": vtype_c, "stype": stype_c}
    sfc_and_v = ops.ew("and", dom_sfc, dom_v)
    sfc_and_s = ops.ew("and", dom_sfc, dom_s)
    area_sfc = ops.ew("where_nan", area, dom_sfc)
    area_sv = area_ss = None

    # Masked means: the reference averages `x.where(mask)` with the weights `w.where(mask)`.  A NaN weight makes the product
    # NaN whatever x holds, and NaN products are skipped, so masking the WEIGHTS is enough -- no masked copy of every field.
    # Fields that share their weights (and shape) go four to a launch (`weighted_block_average_multi`).
    def group(field_names, weights):
        by_shape = {}
        for n in field_names:
            by_shape.setdefault(tuple(fields[n].shape), []).append(n)
        for ns in by_shape.values():
            for n, res in zip(ns, ops.weighted_block_average_multi([fields[n] for n in ns], weights, f)):
                out[n] = res

    present = lambda seq: [n for n in seq if n in fields]
    group(present(_AREA_WEIGHTED), area)
    group(present(_OVER_DOMINANT_SFC), area_sfc)
    if present(_OVER_SFC_AND_STYPE):
        area_ss = ops.ew("where_nan", area, sfc_and_s)
        group(present(_OVER_SFC_AND_STYPE), area_ss)
    if present(_VFRAC_OVER_SFC_AND_VTYPE):
        area_sv = ops.ew("where_nan", area, sfc_and_v)
        av = ops.ew("where_nan", ops.ew("mul", area, fields["vfrac"]), sfc_and_v)
        av_sum = ops.block_reduce(av, (f, f), op="sum")
        for name in present(_VFRAC_OVER_SFC_AND_VTYPE):
            x = fields[name]
            a_mean = ops.weighted_block_average(x, area_sv, f)
            av_mean = ops.weighted_block_average(x, av, f)
            s_ = av_sum
            if s_.shape != av_mean.shape:  # (a field with a level axis: the 2-D sum applies to every level)
                s_ = torch.broadcast_to(s_.unsqueeze(-3), av_mean.shape).contiguous()
            out[name] = ops.ew("select", av_mean, a_mean, ops.ew("gt_s", s_, scalar=0.0))
    for name, x in fields.items():
        if name in out:
            continue
        if name == "srflag":
            out[name] = _mode(x, f)
        elif name == "slope":
            out[name] = _mode(ops.ew("where_nan", x, dom_sfc), f)
        elif name == "sheleg":
            out[name] = ops.ew("fillna_s", ops.weighted_block_average(x, ops.ew("mul", area, fields["sncovr"]), f), scalar=0.0)
        elif name == "hice":
            out[name] = ops.ew("fillna_s", ops.weighted_block_average(x, ops.ew("mul", area, fields["fice"]), f), scalar=0.0)
        elif name == "tisfc":
            sea_ice = ops.weighted_block_average(x, ops.ew("where_nan", ops.ew("mul", area, fields["fice"]), dom_sfc), f)
            other = ops.weighted_block_average(x, area_sfc, f)
            out[name] = ops.ew("select", sea_ice, other, ops.ew("isclose_s", slmsk_c, scalar=2.0))
        else:
            raise KeyError(f"no coarsening method for sfc_data variable {name!r}")
    out = {n: out[n] for n in names}  # (the inputs' order)
    # surface_chgres corrections (coarsen_restarts.py:1403-1470), in the reference's order
    land_ice = ops.ew("isclose_s", out["vtype"], scalar=VTYPE_LAND_ICE)
    if "tsea" in out:
        out["tsea"] = ops.ew("select", ops.ew("min_s", out["tsea"], scalar=FREEZING_TEMPERATURE), out["tsea"], land_ice)
    if "tg3" in out:
        out["tg3"] = ops.ew("select", ops.ew("min_s", out["tg3"], scalar=FREEZING_TEMPERATURE), out["tg3"], land_ice)
    out["stype"] = ops.ew("select_s", out["stype"], land_ice, scalar=STYPE_LAND_ICE)
    if "canopy" in out and "shdmin" in out:
        out["canopy"] = ops.ew("select_s", out["canopy"], ops.ew("lt_s", out["shdmin"], scalar=SHDMIN_THRESHOLD), scalar=0.0)
    if "shdmin" in out:
        out["shdmin"] = ops.ew("select_s", out["shdmin"], land_ice, scalar=0.0)
    keys = list(out)
    return dict(zip(keys, ops.cast_many([out[k] for k in keys], torch.float32)))  # _doubles_to_floats, one launch


def _coarse_grain_sfc_data_complex(ds, area, coarsening_factor: int):
    """Coarse grain a set of sfc_data restart files using the 'complicated' method
    (coarsen_restarts.py:1111-1161).  ``ds``: Dataset with horizontal dims 'yaxis_1', 'xaxis_1';
    ``area``: DataArray with the same horizontal dims.  Returns a Dataset of float32 fields on the
    coarse grid, dims in the inputs' order."""
    d, a = to_compat(ds), to_compat(area)
    fields, orders = {}, {}
    for name in d:
        da = d[name]
        lead = [k for k in da.dims if k not in (Y_DIM, X_DIM)]
        a_lead = [k for k in lead if k in a.dims or k == "Time"]
        extra = [k for k in lead if k not in a_lead]
        if len(extra) > 1:
            raise ValueError(f"{name}: at most one non-horizontal axis besides tile/Time is supported, got {extra}")
        order = a_lead + extra + [Y_DIM, X_DIM]
        orders[name] = (order, da)
        fields[name] = on_device(da.transpose(*order).data)
    some2d = next(v for k, v in orders.items() if len(v[0]) == min(len(o[0]) for o in orders.values()))
    lead2d = some2d[0][:-2]
    at = on_device(a.transpose(*[k for k in lead2d if k in a.dims], Y_DIM, X_DIM).data)
    ref2d = fields[next(k for k, v in orders.items() if v is some2d)]
    if at.dim() != ref2d.dim():  # area has no Time axis: insert the unit axes the fields carry
        shape = [ref2d.shape[i] if lead2d[i] in a.dims else 1 for i in range(len(lead2d))] + list(at.shape[-2:])
        at = torch.broadcast_to(at.reshape(shape), ref2d.shape).contiguous()
    res = coarse_grain_sfc_data_tensors(fields, at, int(coarsening_factor))
    out = Dataset(attrs=d.attrs)
    for name, (order, da) in orders.items():
        arr = DataArray(like_input(res[name], da.data), dims=tuple(order), name=name, attrs=da.attrs)
        out[name] = arr.transpose(*da.dims)
    return from_compat(out, ds)


def coarse_grain_sfc_data(ds, area, coarsening_factor: int, version: str = "complex"):
    """coarsen_restarts.py:1020-1048 (``_coarse_grain_sfc_data``); the pipelines use 'complex'."""
    if version == "complex":
        return _coarse_grain_sfc_data_complex(ds, area, coarsening_factor)
    if version == "simple":
        from .coarsen import block_median

        return block_median(ds, coarsening_factor, x_dim=X_DIM, y_dim=Y_DIM)
    raise ValueError(f"Currently the only supported versions are 'simple' and 'complex'. Got {version}.")
