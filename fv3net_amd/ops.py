"""Array-level entry points: torch device tensors in, torch device tensors out.

Every function here is a thin shape/dtype check around one C-ABI call into ``libfv3hip.so``
(``include/fv3hip.h``), enqueued on torch's current HIP stream.  PyTorch only supplies device
memory and the stream; no torch kernel does any of the arithmetic.  There is no CPU path:
tensors must live on a ``cuda`` (ROCm) device.
"""
import ctypes
import os
import math
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

_DTYPE_CODE = {
    torch.float32: _lib.F32,
    torch.float64: _lib.F64,
    torch.int32: _lib.I32,
    torch.int64: _lib.I64,
}

_OPS = {
    "sum": _lib.OP_SUM,
    "mean": _lib.OP_MEAN,
    "min": _lib.OP_MIN,
    "max": _lib.OP_MAX,
    "median": _lib.OP_MEDIAN,
    "mode": _lib.OP_MODE,
}

_initialised_devices = set()


def _require_device(*tensors):
    dev = None
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t)}")
        if not t.is_cuda:
            raise RuntimeError(
                "fv3net_amd.ops works on device tensors only (there is no CPU fallback); "
                "move the input to a 'cuda' (ROCm) device first"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors live on different devices: {dev} and {t.device}")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _initialised_devices:
        _lib.call("fv3hip_init", idx)
        _initialised_devices.add(idx)
    return dev


def _code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}") from None


def _float_code(t: torch.Tensor) -> int:
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"expected float32 or float64, got {t.dtype}")
    return _DTYPE_CODE[t.dtype]


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


try:  # the raw handle of torch's current stream without building a Stream object (several microseconds per launch)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None


def _stream(dev) -> ctypes.c_void_p:
    if _raw_stream is not None:
        idx = None if dev is None else torch.device(dev).index
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _prod(xs) -> int:
    return int(math.prod(int(x) for x in xs))


def _weights_repeat(obj: torch.Tensor, weights: torch.Tensor):
    """Return (weights_contiguous, w_repeat) such that weight slice ``o // w_repeat`` applies to
    outer slice ``o`` of ``obj`` (both viewed as [n_outer, ny, nx])."""
    if weights.shape[-2:] != obj.shape[-2:]:
        raise ValueError(
            f"horizontal shape of weights {tuple(weights.shape[-2:])} does not match "
            f"the field's {tuple(obj.shape[-2:])}"
        )
    o_outer, w_outer = tuple(obj.shape[:-2]), tuple(weights.shape[:-2])
    if w_outer == o_outer:
        return weights.contiguous(), 1
    # weights' outer dims are a prefix of the field's: shared by the trailing outer dims
    if len(w_outer) <= len(o_outer) and o_outer[: len(w_outer)] == w_outer:
        return weights.contiguous(), max(_prod(o_outer[len(w_outer):]), 1)
    # anything else: materialise the broadcast (numpy broadcasting rules)
    expanded = torch.broadcast_to(weights, obj.shape).contiguous()
    return expanded, 1


def _promoted(a: torch.Tensor, b: torch.Tensor) -> torch.dtype:
    return torch.float64 if torch.float64 in (a.dtype, b.dtype) else torch.float32


def weighted_block_average(obj: torch.Tensor, weights: torch.Tensor, factor: int) -> torch.Tensor:
    """``nansum(obj*w)/nansum(w)`` over factor x factor blocks of the last two dims
    (vcm.cubedsphere.weighted_block_average, coarsen.py:183-218)."""
    dev = _require_device(obj, weights)
    factor = int(factor)
    if obj.dim() < 2:
        raise ValueError("field must have at least two (horizontal) dimensions")
    obj = obj.contiguous()
    weights, w_repeat = _weights_repeat(obj, weights)
    ny, nx = int(obj.shape[-2]), int(obj.shape[-1])
    if factor < 1 or ny % factor or nx % factor:
        raise ValueError(
            f"horizontal extents ({ny}, {nx}) are not multiples of the coarsening factor {factor}"
        )
    n_outer = _prod(obj.shape[:-2])
    out = torch.empty(
        tuple(obj.shape[:-2]) + (ny // factor, nx // factor), dtype=_promoted(obj, weights), device=dev
    )
    _lib.call_on(dev,
        "fv3hip_weighted_block_average", _ptr(obj), _float_code(obj), _ptr(weights), _float_code(weights),
        n_outer, ny, nx, w_repeat, factor, _ptr(out), _stream(dev),
    )
    return out


def weighted_block_average_multi(fields: Sequence[torch.Tensor], weights: torch.Tensor, factor: int) -> list:
    """``weighted_block_average(field, weights, factor)`` for several fields of one shape and dtype that share their weights
    (2-D, or the fields' shape): four fields per launch read the weights once.  Same results as the single-field call when
    fields and weights have one dtype (the surface-data arithmetic does)."""
    fields = [f.contiguous() for f in fields]
    if not fields:
        return []
    f0 = fields[0]
    if len(fields) == 1 or any(f.shape != f0.shape or f.dtype != f0.dtype for f in fields) or weights.dtype != f0.dtype:
        return [weighted_block_average(f, weights, factor) for f in fields]
    dev = _require_device(weights, *fields)
    factor = int(factor)
    w, w_repeat = _weights_repeat(f0, weights)
    ny, nx = int(f0.shape[-2]), int(f0.shape[-1])
    if factor < 1 or ny % factor or nx % factor:
        raise ValueError(f"horizontal extents ({ny}, {nx}) are not multiples of the coarsening factor {factor}")
    outs = [torch.empty(tuple(f0.shape[:-2]) + (ny // factor, nx // factor), dtype=f0.dtype, device=dev) for _ in fields]
    n = len(fields)
    try:
        _lib.call_on(dev, "fv3hip_mass_weighted_block_average", (ctypes.c_void_p * n)(*[f.data_ptr() for f in fields]), n,
                     _float_code(f0), None, _ptr(w), _float_code(w), _prod(f0.shape[:-2]), ny, nx, w_repeat, factor,
                     (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), _stream(dev))
    except _lib.Fv3HipError as err:
        if err.code != _lib.EUNSUPPORTED:
            raise
        return [weighted_block_average(f, weights, factor) for f in fields]
    return outs


def mass_weighted_block_average(fields: Sequence[torch.Tensor], delp: torch.Tensor, area: torch.Tensor, factor: int) -> list:
    """``weighted_block_average(field, delp * area, factor)`` for several fields that share ``delp`` [.., ny, nx] (the
    fields' shape and dtype) and ``area`` (2-D weights shared by trailing outer dims, as in ``weighted_block_average``): the
    product is formed in registers and read once per four fields (coarsen_restarts.py:335-427, 856-900).  Falls back to the
    product + ``weighted_block_average`` kernels for shapes the fused kernel does not take."""
    fields = [f.contiguous() for f in fields]
    if not fields:
        return []
    dev = _require_device(delp, area, *fields)
    factor = int(factor)
    f0 = fields[0]
    if any(f.shape != f0.shape or f.dtype != f0.dtype for f in fields) or delp.shape != f0.shape:
        raise ValueError("fields and delp must share one shape and dtype")
    delp = cast(delp, f0.dtype).contiguous()
    area_b, a_repeat = _weights_repeat(f0, area)
    ny, nx = int(f0.shape[-2]), int(f0.shape[-1])
    if factor < 1 or ny % factor or nx % factor:
        raise ValueError(f"horizontal extents ({ny}, {nx}) are not multiples of the coarsening factor {factor}")
    n_outer = _prod(f0.shape[:-2])
    out_dtype = _promoted(f0, area_b)
    outs = [torch.empty(tuple(f0.shape[:-2]) + (ny // factor, nx // factor), dtype=out_dtype, device=dev) for _ in fields]
    n = len(fields)
    f_ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fields])
    o_ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    try:
        _lib.call_on(dev, "fv3hip_mass_weighted_block_average", f_ptrs, n, _float_code(f0), _ptr(delp), _ptr(area_b),
                     _float_code(area_b), n_outer, ny, nx, a_repeat, factor, o_ptrs, _stream(dev))
    except _lib.Fv3HipError as err:
        if err.code != _lib.EUNSUPPORTED:
            raise
        # the same result from the product kernel and the single-field average
        lead = area_b.dim() - 2
        if a_repeat > 1:
            w = ew("mul", delp.reshape(tuple(delp.shape[:lead]) + (-1, ny, nx)), area_b).reshape(delp.shape)
        else:
            w = ew("mul", delp, area_b)
        if w.dtype != out_dtype:
            w = w.to(out_dtype)
        return [weighted_block_average(f, w, factor) for f in fields]
    return outs


def edge_weighted_block_average(
    obj: torch.Tensor, spacing: torch.Tensor, factor: int, edge: str = "x"
) -> torch.Tensor:
    """Weighted mean over ``factor`` cells along one horizontal dim, every factor-th line kept
    along the other (vcm.cubedsphere.edge_weighted_block_average, coarsen.py:221-273).
    The last two dims are (y, x)."""
    if edge not in ("x", "y"):
        raise ValueError(f"'edge' most be either 'x' or 'y'; got {edge}.")
    dev = _require_device(obj, spacing)
    factor = int(factor)
    obj = obj.contiguous()
    spacing, w_repeat = _weights_repeat(obj, spacing)
    ny, nx = int(obj.shape[-2]), int(obj.shape[-1])
    if edge == "x":
        if nx % factor:
            raise ValueError(f"x extent {nx} is not a multiple of the coarsening factor {factor}")
        oshape = (-(-ny // factor), nx // factor)
    else:
        if ny % factor:
            raise ValueError(f"y extent {ny} is not a multiple of the coarsening factor {factor}")
        oshape = (ny // factor, -(-nx // factor))
    out = torch.empty(tuple(obj.shape[:-2]) + oshape, dtype=_promoted(obj, spacing), device=dev)
    _lib.call_on(dev,
        "fv3hip_edge_weighted_block_average", _ptr(obj), _float_code(obj), _ptr(spacing),
        _float_code(spacing), _prod(obj.shape[:-2]), ny, nx, w_repeat, factor, 0 if edge == "x" else 1,
        _ptr(out), _stream(dev),
    )
    return out


def weighted_window_average(obj: torch.Tensor, weights: torch.Tensor, window: Sequence[int], stride: Sequence[int]) -> torch.Tensor:
    """``sum(obj * weights) / sum(weights)`` over windows (by, bx) every (sy, sx) cells of the last two dims -- the general
    form of ``weighted_block_average`` (f, f / f, f) and ``edge_weighted_block_average`` ('x': (1, f) / (f, f)); on fields
    already reduced to the lines the edge-weighted mean keeps: (1, f) / (1, f)."""
    dev = _require_device(obj, weights)
    by, bx = (int(v) for v in window)
    sy, sx = (int(v) for v in stride)
    obj = obj.contiguous()
    weights, w_repeat = _weights_repeat(obj, weights)
    ny, nx = int(obj.shape[-2]), int(obj.shape[-1])
    if ny < by or nx < bx:
        raise ValueError(f"window ({by}, {bx}) is larger than the field ({ny}, {nx})")
    out = torch.empty(tuple(obj.shape[:-2]) + ((ny - by) // sy + 1, (nx - bx) // sx + 1), dtype=_promoted(obj, weights), device=dev)
    _lib.call_on(dev, "fv3hip_weighted_window_average", _ptr(obj), _float_code(obj), _ptr(weights), _float_code(weights),
                 _prod(obj.shape[:-2]), ny, nx, w_repeat, by, bx, sy, sx, _ptr(out), _stream(dev))
    return out


def take_lines(x: torch.Tensor, step: int, axis: int) -> torch.Tensor:
    """Every ``step``-th line of the last (``axis`` = 0, x) or second-to-last (``axis`` = 1, y) dim, starting with the
    first: ``x.isel(dim=slice(None, None, step))`` as a device gather (coarsen.py:265-271 keeps these lines)."""
    window, stride = (1, 1), ((1, int(step)) if axis == 0 else (int(step), 1))
    return block_reduce(x, window, stride, op="max", nan_policy="propagate")


def repeat(x: torch.Tensor, fy: int, fx: int) -> torch.Tensor:
    """Repeat each value ``fy`` times along the second-to-last and ``fx`` times along the last dim
    (``xarray_utils.repeat``, vcm/xarray_utils.py:37-82, a count per horizontal dim)."""
    dev = _require_device(x)
    x = x.contiguous()
    if x.element_size() not in (4, 8):
        raise TypeError(f"unsupported dtype {x.dtype}")
    ny, nx = int(x.shape[-2]), int(x.shape[-1])
    out = torch.empty(tuple(x.shape[:-2]) + (ny * int(fy), nx * int(fx)), dtype=x.dtype, device=dev)
    _lib.call_on(dev, "fv3hip_repeat", _ptr(x), x.element_size(), _prod(x.shape[:-2]), ny, nx, int(fy), int(fx), _ptr(out), _stream(dev))
    return out


def block_reduce(
    x: torch.Tensor,
    window: Sequence[int],
    stride: Optional[Sequence[int]] = None,
    op: str = "sum",
    nan_policy: str = "skip",
) -> torch.Tensor:
    """Windowed reduction over the last two dims: window (by, bx), stride (sy, sx) (default =
    window).  ops: sum, mean, min, max (NaN-skipping like xarray's coarsen), median (numpy.median),
    mode (scipy.stats.mode 1.7.3, ``nan_policy`` 'propagate' or 'omit')."""
    dev = _require_device(x)
    by, bx = (int(v) for v in window)
    sy, sx = (by, bx) if stride is None else (int(v) for v in stride)
    if op not in _OPS:
        raise ValueError(f"unknown block reduction {op!r}")
    policy = {"skip": _lib.NAN_SKIP, "propagate": _lib.NAN_PROPAGATE, "omit": _lib.NAN_OMIT}[nan_policy]
    x = x.contiguous()
    ny, nx = int(x.shape[-2]), int(x.shape[-1])
    if ny < by or nx < bx:
        raise ValueError(f"window ({by}, {bx}) is larger than the field ({ny}, {nx})")
    nyo, nxo = (ny - by) // sy + 1, (nx - bx) // sx + 1
    out = torch.empty(tuple(x.shape[:-2]) + (nyo, nxo), dtype=x.dtype, device=dev)
    _lib.call_on(dev,
        "fv3hip_block_reduce", _ptr(x), _code(x), _prod(x.shape[:-2]), ny, nx, by, bx, sy, sx, _OPS[op],
        policy, _ptr(out), _stream(dev),
    )
    return out


def block_upsample(x: torch.Tensor, factor: int) -> torch.Tensor:
    """Repeat each value ``factor`` times along the last two dims; a dim of odd size is treated
    as staggered and its last point is not repeated (coarsen.py:843-897)."""
    dev = _require_device(x)
    factor = int(factor)
    x = x.contiguous()
    if x.element_size() not in (4, 8):
        raise TypeError(f"unsupported dtype {x.dtype}")
    ny, nx = int(x.shape[-2]), int(x.shape[-1])
    nyo = (ny - 1) * factor + 1 if ny % 2 == 1 else ny * factor
    nxo = (nx - 1) * factor + 1 if nx % 2 == 1 else nx * factor
    out = torch.empty(tuple(x.shape[:-2]) + (nyo, nxo), dtype=x.dtype, device=dev)
    _lib.call_on(dev,
        "fv3hip_block_upsample", _ptr(x), x.element_size(), _prod(x.shape[:-2]), ny, nx, factor, _ptr(out),
        _stream(dev),
    )
    return out


def _column_view(x: torch.Tensor, z_axis: int):
    z_axis = z_axis % x.dim()
    return z_axis, _prod(x.shape[:z_axis]), int(x.shape[z_axis]), _prod(x.shape[z_axis + 1:])


def column_sum(x: torch.Tensor, z_axis: int, addend: float = 0.0) -> torch.Tensor:
    """``x.sum(z_axis) + addend`` (surface_pressure_from_delp, vertically_dependent.py:189-208)."""
    dev = _require_device(x)
    code = _float_code(x)
    x = x.contiguous()
    z_axis, nb, nz, ni = _column_view(x, z_axis)
    out = torch.empty(tuple(x.shape[:z_axis]) + tuple(x.shape[z_axis + 1:]), dtype=x.dtype, device=dev)
    _lib.call_on(dev, "fv3hip_column_sum", _ptr(x), code, nb, nz, ni, float(addend), _ptr(out), _stream(dev))
    return out


def blend_weights(blending_pressure: torch.Tensor, ps_coarse: torch.Tensor, pfull_coarse: torch.Tensor, z_axis: int) -> torch.Tensor:
    """``(ps - p) / (ps - pb)`` where ``p > pb`` else 1 (coarsen_restarts.py:559-576); ``pfull_coarse`` has
    the z axis, the other two the same shape without it."""
    dev = _require_device(blending_pressure, ps_coarse, pfull_coarse)
    code = _float_code(pfull_coarse)
    p = pfull_coarse.contiguous()
    z_axis, nb, nz, ni = _column_view(p, z_axis)
    pb, ps = blending_pressure.to(p.dtype).contiguous(), ps_coarse.to(p.dtype).contiguous()
    want = tuple(p.shape[:z_axis]) + tuple(p.shape[z_axis + 1:])
    if tuple(pb.shape) != want or tuple(ps.shape) != want:
        raise ValueError(f"blending and surface pressures must have shape {want}")
    out = torch.empty_like(p)
    _lib.call_on(dev, "fv3hip_blend_weights", _ptr(pb), _ptr(ps), _ptr(p), code, nb, nz, ni, _ptr(out), _stream(dev))
    return out


def hydrostatic_balance(dz: torch.Tensor, phis: torch.Tensor, t: torch.Tensor, q: torch.Tensor, delp: torch.Tensor,
                        toa_pressure: float, z_axis: int):
    """Hydrostatic layer thicknesses and the surface geopotential that keeps the model-top height
    (coarsen_restarts.py:990-1017).  Returns (dz, phis)."""
    dev = _require_device(dz, phis, t, q, delp)
    dt = torch.float64 if any(a.dtype == torch.float64 for a in (dz, phis, t, q, delp)) else torch.float32
    dz, phis, t, q, delp = (a.to(dt).contiguous() for a in (dz, phis, t, q, delp))
    z_axis, nb, nz, ni = _column_view(dz, z_axis)
    for a in (t, q, delp):
        if tuple(a.shape) != tuple(dz.shape):
            raise ValueError("DZ, T, sphum and delp must have the same shape")
    if tuple(phis.shape) != tuple(dz.shape[:z_axis]) + tuple(dz.shape[z_axis + 1:]):
        raise ValueError("phis must have DZ's shape without the vertical axis")
    dz_out, phis_out = torch.empty_like(dz), torch.empty_like(phis)
    _lib.call_on(dev, "fv3hip_hydrostatic_balance", _ptr(dz), _ptr(phis), _ptr(t), _ptr(q), _ptr(delp), _DTYPE_CODE[dt], nb, nz, ni,
              float(toa_pressure), _ptr(dz_out), _ptr(phis_out), _stream(dev))
    return dz_out, phis_out


def level_scale(x: torch.Tensor, scale: torch.Tensor, z_axis: int) -> torch.Tensor:
    """``scale[z] * x`` along ``z_axis`` in float64 (TaperConfig.apply, _shared/config.py:11-24)."""
    dev = _require_device(x, scale)
    code = _float_code(x)
    x = x.contiguous()
    z_axis, nb, nz, ni = _column_view(x, z_axis)
    scale = scale.to(torch.float64).contiguous()
    if tuple(scale.shape) != (nz,):
        raise ValueError(f"scale must have shape ({nz},), got {tuple(scale.shape)}")
    out = torch.empty(x.shape, dtype=torch.float64, device=dev)
    _lib.call_on(dev, "fv3hip_level_scale", _ptr(x), code, _ptr(scale), nb, nz, ni, _ptr(out), _stream(dev))
    return out


def _flux_operands(arrays, surface, z_axis: int):
    """Column arrays and [columns-without-z] surface arrays in one float dtype (numpy's promotion), contiguous."""
    dt = torch.float64 if any(t is not None and t.dtype == torch.float64 for t in list(arrays) + list(surface)) else torch.float32
    arrays = [cast(t, dt).contiguous() for t in arrays]
    shape = tuple(arrays[0].shape)
    for t in arrays[1:]:
        if tuple(t.shape) != shape:
            raise ValueError(f"column arrays differ in shape: {tuple(t.shape)} and {shape}")
    z_axis, nb, nz, ni = _column_view(arrays[0], z_axis)
    flat = shape[:z_axis] + shape[z_axis + 1:]
    out_surface = []
    for t in surface:
        if t is not None:
            t = cast(t, dt).contiguous()
            if tuple(t.shape) != flat:
                raise ValueError(f"surface array has shape {tuple(t.shape)}, the columns have {flat}")
        out_surface.append(t)
    return dt, arrays, out_surface, (nb, nz, ni), flat


def tendency_to_flux(tendency: torch.Tensor, delp: torch.Tensor, toa_net_flux: Optional[torch.Tensor],
                     surface_upward_flux: torch.Tensor, z_axis: int, rectify: bool = True, closure_only: bool = False):
    """``vcm.calc.flux_form._tendency_to_flux`` (flux_form.py:7-46): (net flux at the interface above each cell, surface
    downward flux); ``closure_only``: ``_tendency_to_implied_surface_downward_flux`` (:49-75), (None, downward flux)."""
    dev = _require_device(tendency, delp, surface_upward_flux)
    dt, (tend, dp), (toa, up), (nb, nz, ni), flat = _flux_operands([tendency, delp], [toa_net_flux, surface_upward_flux], z_axis)
    flux = None if closure_only else torch.empty_like(tend)
    down = torch.empty(flat, dtype=dt, device=dev)
    _lib.call_on(dev, "fv3hip_tendency_to_flux", _ptr(tend), _ptr(dp), _ptr(toa), _ptr(up), _float_code(tend), nb, nz, ni,
                 int(bool(rectify)), int(bool(closure_only)), _ptr(flux), _ptr(down), _stream(dev))
    return flux, down


def flux_to_tendency(net_flux: torch.Tensor, surface_downward_flux: torch.Tensor, surface_upward_flux: torch.Tensor,
                     delp: torch.Tensor, z_axis: int) -> torch.Tensor:
    """``vcm.calc.flux_form._flux_to_tendency`` (flux_form.py:78-104)."""
    dev = _require_device(net_flux, delp)
    dt, (flux, dp), (down, up), (nb, nz, ni), _ = _flux_operands([net_flux, delp], [surface_downward_flux, surface_upward_flux], z_axis)
    out = torch.empty_like(flux)
    _lib.call_on(dev, "fv3hip_flux_to_tendency", _ptr(flux), _ptr(down), _ptr(up), _ptr(dp), _float_code(flux), nb, nz, ni, _ptr(out),
                 _stream(dev))
    return out


def minmax_score(variables: Sequence[torch.Tensor], scales: Sequence[torch.Tensor], offsets: Sequence[torch.Tensor]) -> torch.Tensor:
    """MinMaxNoveltyDetector's score (fv3fit/sklearn/_min_max_novelty_detector.py:94-121) of ``[feature, sample]`` arrays
    (any strides) scaled as ``MinMaxScaler.transform`` does: ``max(max_f - 1, 0) + max(-min_f, 0)``, float64 ``[sample]``."""
    dev = _require_device(*variables)
    n = int(variables[0].shape[1])
    run_max = torch.empty(n, dtype=torch.float64, device=dev)
    run_min = torch.empty(n, dtype=torch.float64, device=dev)
    score = torch.empty(n, dtype=torch.float64, device=dev)
    for k, (t, sc, off) in enumerate(zip(variables, scales, offsets)):
        if t.dim() != 2 or int(t.shape[1]) != n or tuple(sc.shape) != (t.shape[0],) or tuple(off.shape) != (t.shape[0],):
            raise ValueError("variables must be [feature, sample] arrays over the same samples with [feature] scales and offsets")
        _lib.call_on(dev, "fv3hip_minmax_score", _ptr(t), _float_code(t), int(t.stride(0)), int(t.stride(1)), int(t.shape[0]),
                     _ptr(sc.to(torch.float64).contiguous()), _ptr(off.to(torch.float64).contiguous()), n, int(k == 0),
                     int(k == len(variables) - 1), _ptr(run_max), _ptr(run_min), _ptr(score), _stream(dev))
    return score


def ocsvm_score(x: torch.Tensor, mean: torch.Tensor, scale: torch.Tensor, support_vectors: torch.Tensor, dual_coef: torch.Tensor,
                gamma: float) -> torch.Tensor:
    """``-Pipeline(StandardScaler, OneClassSVM(rbf)).score_samples`` of the packed float64 ``x`` [feature, sample]
    (fv3fit/sklearn/_ocsvm_novelty_detector.py:124-160)."""
    dev = _require_device(x, mean, scale, support_vectors, dual_coef)
    x = x.to(torch.float64).contiguous()
    nf, n = int(x.shape[0]), int(x.shape[1])
    sv = support_vectors.to(torch.float64).contiguous()
    if tuple(sv.shape[1:]) != (nf,) or tuple(dual_coef.shape) != (sv.shape[0],) or tuple(mean.shape) != (nf,) or tuple(scale.shape) != (nf,):
        raise ValueError("support vectors [n_sv, feature], dual_coef [n_sv], mean / scale [feature] do not fit x [feature, sample]")
    out = torch.empty(n, dtype=torch.float64, device=dev)
    _lib.call_on(dev, "fv3hip_ocsvm_score", _ptr(x), nf, n, _ptr(mean.to(torch.float64).contiguous()), _ptr(scale.to(torch.float64).contiguous()),
                 _ptr(sv), _ptr(dual_coef.to(torch.float64).contiguous()), int(sv.shape[0]), float(gamma), _ptr(out), _stream(dev))
    return out


def member_reduce(members: Sequence[torch.Tensor], op: str) -> torch.Tensor:
    """NaN-skipping ``mean`` / ``median`` over same-shaped member arrays (EnsembleModel.predict, models.py:253-260)."""
    dev = _require_device(*members)
    dt = torch.float64 if any(m.dtype == torch.float64 for m in members) else torch.float32
    ms = [m.to(dt).contiguous() for m in members]
    if any(tuple(m.shape) != tuple(ms[0].shape) for m in ms):
        raise ValueError("ensemble members differ in shape")
    if op not in ("mean", "median"):
        raise NotImplementedError(f"Got reduction {op}: only mean, median supported")
    out = torch.empty_like(ms[0])
    ptrs = (ctypes.c_void_p * len(ms))(*[m.data_ptr() for m in ms])
    _lib.call_on(dev, "fv3hip_member_reduce", ptrs, len(ms), _DTYPE_CODE[dt], _OPS[op], out.numel(), _ptr(out), _stream(dev))
    return out


def interpolate_2d(xp: torch.Tensor, x: torch.Tensor, y: torch.Tensor, fill_value: float = float("nan"),
                   z_axis: int = -1) -> torch.Tensor:
    """``mappm.interpolate_2d`` (interpolate_2d.f90:1-28): linear interpolation of ``y(x)`` onto ``xp``
    along ``z_axis`` (x, y: n_in points there, xp: n_out), ``fill_value`` outside each column's range.
    Arrays share all other dims; computed and returned in float64."""
    dev = _require_device(xp, x, y)
    z_axis = z_axis % x.dim()
    if tuple(x.shape) != tuple(y.shape):
        raise ValueError("x and y must have the same shape")
    rest = lambda t: tuple(t.shape[:z_axis]) + tuple(t.shape[z_axis + 1:])
    if rest(xp) != rest(x):
        raise ValueError("xp must match x in every dimension but the interpolated one")
    xp, x, y = (t.to(torch.float64).contiguous() for t in (xp, x, y))
    n_in, n_out = int(x.shape[z_axis]), int(xp.shape[z_axis])
    n_batch, n_inner = _prod(x.shape[:z_axis]), _prod(x.shape[z_axis + 1:])
    layout = _lib.LAYOUT_LEVEL_COL if n_inner > 1 else _lib.LAYOUT_COL_LEVEL
    out = torch.empty_like(xp)
    _lib.call_on(dev, "fv3hip_interpolate_2d", _ptr(xp), _ptr(x), _ptr(y), n_batch, n_inner, n_in, n_out, float(fill_value), layout,
              _ptr(out), _stream(dev))
    return out


EW_OPS = {"mul": 0, "isclose": 1, "isclose_s": 2, "where_nan": 3, "select": 4, "select_s": 5, "gt_s": 6, "lt_s": 7,
          "fillna_s": 8, "and": 9, "min_s": 10, "blend": 11, "mul_s": 12, "where_s": 13, "add": 14, "add_s": 15,
          "sub": 16, "log_floor_s": 17, "exp": 18, "relu_threshold_s": 19, "below_s": 20, "div_s": 21, "incloud_to_gridcell": 22,
          "clip01": 23, "pow_base_s": 24, "minimum_s": 25, "div": 26, "where_pos_s": 27, "sign": 28, "abs": 29, "rsub_s": 30,
          "rdiv_s": 31, "where_gt_s": 32, "le_s": 33, "sin": 34, "cos": 35}


def ew(op: str, a: torch.Tensor, b: Optional[torch.Tensor] = None, c: Optional[torch.Tensor] = None,
       scalar: float = 0.0) -> torch.Tensor:
    """One step of the surface-data mask arithmetic (coarsen_restarts.py:1140-1470), see
    ``fv3hip_ew``: ``b`` / ``c`` have ``a``'s shape, or are [.., y, x] fields shared by the extra
    (level) axis of ``a`` [.., level, y, x]."""
    dev = _require_device(a)
    code = _float_code(a)
    a = a.contiguous()
    inner = int(a.shape[-1] * a.shape[-2]) if a.dim() >= 2 else max(int(a.numel()), 1)

    def operand(t):
        if t is None:
            return None, 1
        t = cast(t, a.dtype).contiguous()
        if tuple(t.shape) == tuple(a.shape):
            return t, 1
        if a.dim() >= 3 and tuple(t.shape) == tuple(a.shape[:-3]) + tuple(a.shape[-2:]):
            return t, int(a.shape[-3])
        raise ValueError(f"operand shape {tuple(t.shape)} does not match {tuple(a.shape)}")

    b, b_rep = operand(b)
    c, c_rep = operand(c)
    out = torch.empty_like(a)
    _lib.call_on(dev, "fv3hip_ew", EW_OPS[op], _ptr(a), _ptr(b), _ptr(c), float(scalar), code, a.numel(), inner, b_rep, c_rep,
              _ptr(out), _stream(dev))
    return out


_CAST_IN = {torch.float32: _lib.F32, torch.float64: _lib.F64, torch.int32: _lib.I32, torch.int64: _lib.I64}


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``x`` in ``dtype`` (float32 / float64): the tensor itself when it already is, else one pass of ``cast_kernel``."""
    if x.dtype == dtype:
        return x
    if x.dtype not in _CAST_IN or dtype not in (torch.float32, torch.float64):
        return x.to(dtype)  # (bool / half etc.: not on the restart path)
    dev = _require_device(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=dev)
    _lib.call_on(dev, "fv3hip_cast", _ptr(x), _CAST_IN[x.dtype], _ptr(out), _CAST_IN[dtype], x.numel(), _stream(dev))
    return out


def cast_many(tensors: Sequence[torch.Tensor], dtype: torch.dtype) -> list:
    """``cast`` of several tensors with one launch for all that need converting (``fv3hip_cast_many``)."""
    tensors = list(tensors)
    todo = [i for i, t in enumerate(tensors) if t.dtype != dtype and t.dtype in _CAST_IN]
    if dtype not in (torch.float32, torch.float64) or len(todo) < 2:
        return [cast(t, dtype) for t in tensors]
    dev = _require_device(*[tensors[i] for i in todo])
    src = [tensors[i].contiguous() for i in todo]
    dst = [torch.empty(t.shape, dtype=dtype, device=dev) for t in src]
    n = len(src)
    _lib.call_on(dev, "fv3hip_cast_many", (ctypes.c_void_p * n)(*[t.data_ptr() for t in src]),
                 (ctypes.c_int * n)(*[_CAST_IN[t.dtype] for t in src]), (ctypes.c_void_p * n)(*[t.data_ptr() for t in dst]),
                 _CAST_IN[dtype], (ctypes.c_int64 * n)(*[t.numel() for t in src]), n, _stream(dev))
    out = list(tensors)
    for i, d in zip(todo, dst):
        out[i] = d
    return [t if t.dtype == dtype else cast(t, dtype) for t in out]  # (anything left: exotic dtypes, one by one)


def halo_pick(rows: torch.Tensor, nbr, row, flip) -> torch.Tensor:
    """(lo, hi) stacked as [2, n_local, ..., n] from the boundary-vector table ``rows`` [6, 4, ..., n] (``cube_edge_rows``):
    entry ``side * n_local + i`` of the host lists names the neighbour tile, which of its four vectors, and whether it is
    read reversed (cubedsphere/grid.py holds the connectivity)."""
    dev = _require_device(rows)
    rows = rows.contiguous()
    n_local = len(nbr) // 2
    n = int(rows.shape[-1])
    mid = tuple(rows.shape[2:-1])
    out = torch.empty((2, n_local) + mid + (n,), dtype=rows.dtype, device=dev)
    arr = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])
    _lib.call_on(dev, "fv3hip_halo_pick", _ptr(rows), rows.element_size(), n_local, _prod(mid), n, arr(nbr), arr(row), arr(flip),
                 _ptr(out), _stream(dev))
    return out


def cube_edge_rows(x: torch.Tensor) -> torch.Tensor:
    """The four boundary vectors of every square tile of ``x`` [tile, ..., n, n] ->
    [tile, 4, ..., n]: 0: x = 0, 1: x = n-1 (indexed by y), 2: y = 0, 3: y = n-1 (indexed by x).
    The one-cell halo the neighbouring cube faces need (xgcm.py:7-34)."""
    dev = _require_device(x)
    x = x.contiguous()
    if x.element_size() not in (4, 8):
        raise TypeError(f"unsupported dtype {x.dtype}")
    if x.dim() < 3 or x.shape[-1] != x.shape[-2]:
        raise ValueError(f"cube faces must be square [tile, ..., n, n], got {tuple(x.shape)}")
    n_tiles, n = int(x.shape[0]), int(x.shape[-1])
    mid = tuple(x.shape[1:-2])
    rows = torch.empty((n_tiles, 4) + mid + (n,), dtype=x.dtype, device=dev)
    _lib.call_on(dev, "fv3hip_cube_edge_rows", _ptr(x), x.element_size(), n_tiles, _prod(mid), n, _ptr(rows), _stream(dev))
    return rows


def interp_center_to_outer(x: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor, axis: int, step: int = 1) -> torch.Tensor:
    """``0.5 * (left + right)`` from cell centres to the n+1 cell edges along the last (``axis`` = 0,
    x) or second-to-last (``axis`` = 1, y) dim; ``lo`` / ``hi`` [..., n_edge] hold the neighbours
    beyond the two ends (regridz.py:123-135).  ``step`` > 1: only every step-th edge (n / step + 1 of them)."""
    dev = _require_device(x)
    code = _float_code(x)
    x = x.contiguous()
    lo = cast(lo, x.dtype).contiguous()
    hi = cast(hi, x.dtype).contiguous()
    ny, nx = int(x.shape[-2]), int(x.shape[-1])
    want = tuple(x.shape[:-2]) + ((ny,) if axis == 0 else (nx,))
    if tuple(lo.shape) != want or tuple(hi.shape) != want:
        raise ValueError(f"halo shape must be {want}, got {tuple(lo.shape)} and {tuple(hi.shape)}")
    step = int(step)
    if step < 1 or (nx if axis == 0 else ny) % step:
        raise ValueError(f"the extent along the axis must be a multiple of step={step}")
    oshape = (ny // step + 1 if axis == 1 else ny, nx // step + 1 if axis == 0 else nx)
    out = torch.empty(tuple(x.shape[:-2]) + oshape, dtype=x.dtype, device=dev)
    _lib.call_on(dev, "fv3hip_interp_center_to_outer_lines", _ptr(x), code, _prod(x.shape[:-2]), ny, nx, int(axis), step,
              _ptr(lo), _ptr(hi), _ptr(out), _stream(dev))
    return out


def pressure_at_interface(delp: torch.Tensor, toa_pressure: float, z_axis: int) -> torch.Tensor:
    """``p[0] = toa; p[k+1] = p[k] + delp[k]`` along ``z_axis`` (size nz -> nz + 1), accumulated
    sequentially in delp's dtype (vertically_dependent.py:41-66)."""
    dev = _require_device(delp)
    delp = delp.contiguous()
    z_axis = z_axis % delp.dim()
    nz = int(delp.shape[z_axis])
    n_batch, n_inner = _prod(delp.shape[:z_axis]), _prod(delp.shape[z_axis + 1:])
    shape = list(delp.shape)
    shape[z_axis] = nz + 1
    out = torch.empty(shape, dtype=delp.dtype, device=dev)
    _lib.call_on(dev,
        "fv3hip_pressure_at_interface", _ptr(delp), _float_code(delp), n_batch, nz, n_inner,
        float(toa_pressure), _ptr(out), _stream(dev),
    )
    return out


def pressure_at_midpoint_log(delp: torch.Tensor, toa_pressure: float, z_axis: int) -> torch.Tensor:
    """``delp / diff(log(p_interface))`` along ``z_axis`` (vertically_dependent.py:153-179)."""
    dev = _require_device(delp)
    delp = delp.contiguous()
    z_axis = z_axis % delp.dim()
    nz = int(delp.shape[z_axis])
    n_batch, n_inner = _prod(delp.shape[:z_axis]), _prod(delp.shape[z_axis + 1:])
    out = torch.empty_like(delp)
    _lib.call_on(dev,
        "fv3hip_pressure_at_midpoint_log", _ptr(delp), _float_code(delp), n_batch, nz, n_inner,
        float(toa_pressure), _ptr(out), _stream(dev),
    )
    return out


def mask_weights(
    weights: torch.Tensor, p_coarse: torch.Tensor, p_fine: torch.Tensor, z_axis: int, extrapolate: bool = False,
    coarse_factor: Optional[int] = None,
) -> torch.Tensor:
    """``weights where p_level < p_fine[surface] else 0`` (regridz.py:200-220).
    ``extrapolate=False``: ``p_coarse`` holds the nz+1 coarse interface pressures and level k is
    compared through its bottom interface; ``extrapolate=True``: ``p_coarse`` holds the nz coarse
    midpoint pressures.  ``p_fine`` has nz+1 levels along ``z_axis``; ``weights`` has the pressure
    shape without the z axis.  ``coarse_factor``: ``p_coarse`` is still on its horizontally coarser grid (last two dims;
    the z axis third from last) and is read through (y // factor, x // factor) instead of an upsampled copy."""
    if coarse_factor is not None and int(coarse_factor) > 1:
        if z_axis % p_fine.dim() != p_fine.dim() - 3:
            return mask_weights(weights, block_upsample(p_coarse, int(coarse_factor)), p_fine, z_axis, extrapolate)
        dev = _require_device(weights, p_coarse, p_fine)
        if p_coarse.dtype != p_fine.dtype:
            raise ValueError("p_coarse and p_fine must have the same dtype")
        p_coarse, p_fine, weights = p_coarse.contiguous(), p_fine.contiguous(), weights.contiguous()
        nz, ny, nx = int(p_fine.shape[-3]) - 1, int(p_fine.shape[-2]), int(p_fine.shape[-1])
        cmp_levels, cmp_offset = (nz, 0) if extrapolate else (nz + 1, 1)
        batch_shape = tuple(p_fine.shape[:-3])
        if tuple(weights.shape) != batch_shape + (ny, nx) or int(p_coarse.shape[-3]) != cmp_levels:
            raise ValueError("weights / p_coarse do not match the fine pressures' shape")
        # the C entry point indexes p_coarse through (y // f) * nxc + x // f and cannot see its size
        f_ = int(coarse_factor)
        coarse_extent = lambda n: (n - 1) // f_ + 1 if n % 2 else n // f_  # (an odd, staggered extent keeps its last point)
        want_c = batch_shape + (cmp_levels, coarse_extent(ny), coarse_extent(nx))
        if tuple(p_coarse.shape) != want_c:
            raise ValueError(f"p_coarse has shape {tuple(p_coarse.shape)}, expected {want_c} for fine pressures "
                             f"{tuple(p_fine.shape)} coarsened by {f_}")
        out = torch.empty(batch_shape + (nz, ny, nx), dtype=weights.dtype, device=dev)
        try:
            _lib.call_on(dev, "fv3hip_mask_weights_coarse", _ptr(weights), _float_code(weights), _ptr(p_coarse), cmp_levels, cmp_offset,
                         _ptr(p_fine), _float_code(p_fine), _prod(batch_shape), nz, ny, nx, int(coarse_factor), 1, _ptr(out), _stream(dev))
        except _lib.Fv3HipError as err:
            if err.code != _lib.EUNSUPPORTED:
                raise
            return mask_weights(weights, block_upsample(p_coarse, int(coarse_factor)), p_fine, z_axis, extrapolate)
        return out
    dev = _require_device(weights, p_coarse, p_fine)
    if p_coarse.dtype != p_fine.dtype:
        raise ValueError("p_coarse and p_fine must have the same dtype")
    p_coarse, p_fine, weights = p_coarse.contiguous(), p_fine.contiguous(), weights.contiguous()
    z_axis = z_axis % p_fine.dim()
    nz = int(p_fine.shape[z_axis]) - 1
    cmp_levels, cmp_offset = (nz, 0) if extrapolate else (nz + 1, 1)
    expect = list(p_fine.shape)
    expect[z_axis] = cmp_levels
    if list(p_coarse.shape) != expect:
        raise ValueError(f"p_coarse has shape {tuple(p_coarse.shape)}, expected {tuple(expect)}")
    batch_shape, inner_shape = tuple(p_fine.shape[:z_axis]), tuple(p_fine.shape[z_axis + 1:])
    n_batch, n_inner = _prod(batch_shape), _prod(inner_shape)
    if tuple(weights.shape) != batch_shape + inner_shape:
        raise ValueError(
            f"weights shape {tuple(weights.shape)} must be the pressure shape without its z axis "
            f"{batch_shape + inner_shape}"
        )
    out = torch.empty(batch_shape + (nz,) + inner_shape, dtype=weights.dtype, device=dev)
    _lib.call_on(dev,
        "fv3hip_mask_weights", _ptr(weights), _float_code(weights), _ptr(p_coarse), cmp_levels, cmp_offset,
        _ptr(p_fine), _float_code(p_fine), n_batch, nz, n_inner, 1, _ptr(out), _stream(dev),
    )
    return out


_workspaces = {}


def _workspace(dev, nbytes: int) -> torch.Tensor:
    """Scratch of the remap, one per (device, HIP stream): it holds a call's list of columns to redo and the fallback
    planes, so calls on different streams must not share it.  A grown workspace replaces the old tensor only for its own
    stream, whose earlier launches are ordered before the new ones."""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws


# Arithmetic of the remap when a call does not say: "exact" -- bit-identical to the compiled reference Fortran, the
# library's contract.  "fast" (reciprocal-multiply, fused multiply-adds and hardware minima in the sweep kernel: a few ulp
# from the reference on every level, except that the reference's limiter is discontinuous where a slope cancels to exactly
# zero -- such a level may take the other, equally monotone profile) is an opt-in: per call (``arith="fast"``), per process
# (``ops.MAPPM_ARITHMETIC = "fast"``) or per environment (``FV3NET_AMD_MAPPM_ARITH=fast``).  bench.py times both.
MAPPM_ARITHMETIC = os.environ.get("FV3NET_AMD_MAPPM_ARITH", "exact")


def _arith_code(arith: Optional[str]) -> int:
    mode = MAPPM_ARITHMETIC if arith is None else arith
    if mode not in ("exact", "fast"):
        raise ValueError(f"arith must be 'exact' or 'fast', got {mode!r}")
    return _lib.ARITH_FAST if mode == "fast" else _lib.ARITH_EXACT


def mappm(
    pe1: torch.Tensor,
    q1: torch.Tensor,
    pe2: torch.Tensor,
    iv: int = 1,
    kord: int = 1,
    z_axis: int = -1,
    arith: Optional[str] = None,
) -> torch.Tensor:
    """PPM vertical remap of ``q1`` from interface pressures ``pe1`` to ``pe2`` (mappm.f90).
    ``arith``: "exact" | "fast" (default: ``ops.MAPPM_ARITHMETIC``), see ``FV3HIP_ARITH_*`` in include/fv3hip.h.
    ``z_axis`` is the level axis of all three arrays (km+1, km, kn+1 levels); the other dims
    must match.  ``z_axis=-1`` is the [column, level] layout f2py callers use; any other
    position is handled in place as the native [.., level, .. columns ..] layout.
    Returns float32 with kn levels along ``z_axis``."""
    dev = _require_device(pe1, q1, pe2)
    if not (pe1.dtype == q1.dtype == pe2.dtype):
        common = torch.float64 if torch.float64 in (pe1.dtype, q1.dtype, pe2.dtype) else torch.float32
        pe1, q1, pe2 = pe1.to(common), q1.to(common), pe2.to(common)
    pe1, q1, pe2 = pe1.contiguous(), q1.contiguous(), pe2.contiguous()
    nd = q1.dim()
    z_axis = z_axis % nd
    km, kn = int(q1.shape[z_axis]), int(pe2.shape[z_axis]) - 1
    if int(pe1.shape[z_axis]) != km + 1:
        raise ValueError("f_in must have a vertical dimension one shorter than p_in")

    def others(t):
        return tuple(t.shape[:z_axis]) + tuple(t.shape[z_axis + 1:])

    if not (others(pe1) == others(q1) == others(pe2)):
        raise ValueError("All dimensions except vertical must be same size for p_in, f_in and p_out")
    n_batch, n_inner = _prod(q1.shape[:z_axis]), _prod(q1.shape[z_axis + 1:])
    if z_axis == nd - 1:
        layout, nb, ni = _lib.LAYOUT_COL_LEVEL, n_batch, 1
    else:
        layout, nb, ni = _lib.LAYOUT_LEVEL_COL, n_batch, n_inner
    shape = list(q1.shape)
    shape[z_axis] = kn
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    ncol = nb * ni
    nbytes = int(_lib.load().fv3hip_mappm_workspace_bytes(ncol, km))
    ws = _workspace(dev, nbytes)
    _lib.call_on(dev,
        "fv3hip_mappm", _ptr(pe1), _ptr(q1), _ptr(pe2), _float_code(q1), _ptr(out), nb, ni, km, kn,
        int(iv), int(kord), layout, _arith_code(arith), _ptr(ws), ws.numel(), _stream(dev),
    )
    return out


def mappm_multi(pe1: torch.Tensor, fields: Sequence[torch.Tensor], pe2: torch.Tensor, iv: int = 1, kord: int = 1,
                z_axis: int = -1, arith: Optional[str] = None) -> list:
    """``mappm`` of several fields that share ``pe1`` and ``pe2`` (every variable of a dataset in
    regridz.py:163-185): one sweep per four fields computes the control flow and the pressure-only terms
    once.  Each result is bit-identical to ``mappm`` on that field in the same arithmetic mode (``arith``: see ``mappm``;
    the default "exact" is bit-identical to the compiled reference)."""
    fields = list(fields)
    if not fields:
        return []
    dev = _require_device(pe1, pe2, *fields)
    dtypes = {t.dtype for t in (pe1, pe2, *fields)}
    if len(dtypes) > 1:
        common = torch.float64 if torch.float64 in dtypes else torch.float32
        pe1, pe2, fields = pe1.to(common), pe2.to(common), [q.to(common) for q in fields]
    pe1, pe2, fields = pe1.contiguous(), pe2.contiguous(), [q.contiguous() for q in fields]
    q1 = fields[0]
    nd = q1.dim()
    z_axis = z_axis % nd
    km, kn = int(q1.shape[z_axis]), int(pe2.shape[z_axis]) - 1
    if int(pe1.shape[z_axis]) != km + 1:
        raise ValueError("f_in must have a vertical dimension one shorter than p_in")

    def others(t):
        return tuple(t.shape[:z_axis]) + tuple(t.shape[z_axis + 1:])

    if any(tuple(q.shape) != tuple(q1.shape) for q in fields) or not (others(pe1) == others(q1) == others(pe2)):
        raise ValueError("All dimensions except vertical must be same size for p_in, f_in and p_out")
    n_batch, n_inner = _prod(q1.shape[:z_axis]), _prod(q1.shape[z_axis + 1:])
    if z_axis == nd - 1:
        layout, nb, ni = _lib.LAYOUT_COL_LEVEL, n_batch, 1
    else:
        layout, nb, ni = _lib.LAYOUT_LEVEL_COL, n_batch, n_inner
    shape = list(q1.shape)
    shape[z_axis] = kn
    outs = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in fields]
    ws = _workspace(dev, int(_lib.load().fv3hip_mappm_workspace_bytes(nb * ni, km)))
    n = len(fields)
    q_ptrs = (ctypes.c_void_p * n)(*[q.data_ptr() for q in fields])
    o_ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    _lib.call_on(dev, "fv3hip_mappm_multi", _ptr(pe1), q_ptrs, _ptr(pe2), _float_code(q1), o_ptrs, n, nb, ni, km, kn, int(iv), int(kord),
              layout, _arith_code(arith), _ptr(ws), ws.numel(), _stream(dev))
    return outs


def mappm_multi_coarse_target(pe1: torch.Tensor, fields: Sequence[torch.Tensor], pe2_coarse: torch.Tensor, factor: int, iv: int = 1,
                              kord: int = 1, z_axis: int = -3, arith: Optional[str] = None) -> list:
    """``mappm_multi(pe1, fields, block_upsample(pe2_coarse, factor), ...)`` without the upsampled copy: the target interfaces
    stay on their coarse grid ``[..., z, ny_c, nx_c]`` and fine column (y, x) reads coarse column (y // factor, x // factor)
    (regridz.py:119-185; a staggered dim's last point maps to the last coarse point, as ``block_upsample`` repeats it).  The
    vertical axis must be third from last; shapes the fused kernel does not take go through the upsampled copy.  Same results."""
    fields = list(fields)
    if not fields:
        return []
    factor = int(factor)
    q1 = fields[0]
    nd = q1.dim()

    def fallback():
        return mappm_multi(pe1, fields, block_upsample(pe2_coarse, factor), iv=iv, kord=kord, z_axis=z_axis, arith=arith)

    if nd < 3 or z_axis % nd != nd - 3 or factor < 2:
        return fallback()
    dev = _require_device(pe1, pe2_coarse, *fields)
    dtypes = {t.dtype for t in (pe1, pe2_coarse, *fields)}
    if len(dtypes) > 1:
        common = torch.float64 if torch.float64 in dtypes else torch.float32
        pe1, pe2_coarse, fields = pe1.to(common), pe2_coarse.to(common), [q.to(common) for q in fields]
    pe1, pe2_coarse, fields = pe1.contiguous(), pe2_coarse.contiguous(), [q.contiguous() for q in fields]
    q1 = fields[0]
    km, kn = int(q1.shape[-3]), int(pe2_coarse.shape[-3]) - 1
    ny, nx = int(q1.shape[-2]), int(q1.shape[-1])
    if int(pe1.shape[-3]) != km + 1:
        raise ValueError("f_in must have a vertical dimension one shorter than p_in")
    coarse = lambda n: (n - 1) // factor + 1 if n % 2 == 1 else n // factor
    if (any(tuple(q.shape) != tuple(q1.shape) for q in fields) or tuple(pe1.shape[:-3]) != tuple(q1.shape[:-3])
            or tuple(pe1.shape[-2:]) != (ny, nx) or tuple(pe2_coarse.shape[:-3]) != tuple(q1.shape[:-3])
            or tuple(pe2_coarse.shape[-2:]) != (coarse(ny), coarse(nx))):
        raise ValueError("All dimensions except vertical must be same size for p_in, f_in and (the upsampled) p_out")
    nb = _prod(q1.shape[:-3])
    outs = [torch.empty(tuple(q1.shape[:-3]) + (kn, ny, nx), dtype=torch.float32, device=dev) for _ in fields]
    ws = _workspace(dev, int(_lib.load().fv3hip_mappm_workspace_bytes(nb * ny * nx, km)))
    n = len(fields)
    q_ptrs = (ctypes.c_void_p * n)(*[q.data_ptr() for q in fields])
    o_ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    try:
        _lib.call_on(dev, "fv3hip_mappm_multi_coarse_target", _ptr(pe1), q_ptrs, _ptr(pe2_coarse), _float_code(q1), o_ptrs, n, nb, ny, nx,
                     factor, km, kn, int(iv), int(kord), _arith_code(arith), _ptr(ws), ws.numel(), _stream(dev))
    except _lib.Fv3HipError as err:
        if err.code != _lib.EUNSUPPORTED:
            raise
        return fallback()
    return outs


def mappm_block_mean(pe1: torch.Tensor, fields: Sequence[torch.Tensor], pe2_coarse: torch.Tensor, area: torch.Tensor,
                     level_coarse: Optional[torch.Tensor] = None, factor: int = 8, iv: int = 1, kord: int = 1,
                     arith: Optional[str] = None, counters: Optional[torch.Tensor] = None) -> Optional[list]:
    """``weighted_block_average(mappm_multi_coarse_target(pe1, fields, pe2_coarse, factor), mask_weights(area, level, pe1, ...,
    coarse_factor=factor), factor)`` in one kernel (regridz.py:149-220 followed by coarsen.py:183-218, as the pressure-level
    restart pipelines call them): a wavefront owns one 8 x 8 block, sums its remapped, masked values in LDS and writes the
    coarse means -- the fine remapped fields and the masked weights never exist.  ``level_coarse``: the coarse midpoint
    pressures [.., kn, ny/8, nx/8] (``extrapolate=True``); None compares each layer's bottom interface (``pe2_coarse``).
    ``area`` [.., ny, nx] float32 with leading dims that lead the fields' (shared by the rest).  Arrays in [.., z, y, x] order.
    Bit-identical to the three calls above in the same ``arith``.  Returns None where the kernel does not apply (factor != 8,
    extents not multiples of 8, a non-float32 area, kord > 3, ...): the caller takes the three calls.
    ``counters``: a pinned int32 host tensor of 4 elements that receives, asynchronously on the current stream, the call's
    last [columns listed for the sequential routine, blocks redone, blocks whose waves gave up summing, 0] -- what an adaptive
    caller looks at before it chooses the route of its next call (``regridz.area_weighted_pressure_means``)."""
    fields = list(fields)
    if not fields:
        return []
    q1 = fields[0]
    nd = q1.dim()
    if int(factor) != 8 or nd < 3 or area.dtype != torch.float32:
        return None
    dev = _require_device(pe1, pe2_coarse, area, *fields)
    lvl, cmp_offset = (pe2_coarse, 1) if level_coarse is None else (level_coarse, 0)
    dtypes = {t.dtype for t in (pe1, pe2_coarse, lvl, *fields)}
    if len(dtypes) > 1:
        return None
    km, kn = int(q1.shape[-3]), int(pe2_coarse.shape[-3]) - 1
    ny, nx = int(q1.shape[-2]), int(q1.shape[-1])
    batch = tuple(q1.shape[:-3])
    if ny % 8 or nx % 8 or int(kord) > 3 or km < 8 or kn < 1 or kn + 1 > 128:
        return None
    if int(pe1.shape[-3]) != km + 1:
        raise ValueError("f_in must have a vertical dimension one shorter than p_in")
    coarse_hw = (ny // 8, nx // 8)
    if (any(tuple(q.shape) != tuple(q1.shape) for q in fields) or tuple(pe1.shape) != batch + (km + 1, ny, nx)
            or tuple(pe2_coarse.shape) != batch + (kn + 1,) + coarse_hw
            or tuple(lvl.shape) != batch + (kn + cmp_offset,) + coarse_hw):
        raise ValueError("All dimensions except vertical must be same size for p_in, f_in and (the upsampled) p_out")
    lead = tuple(area.shape[:-2])
    if tuple(area.shape[-2:]) != (ny, nx) or lead != batch[: len(lead)]:
        raise ValueError(f"area of shape {tuple(area.shape)} does not lead the fields' {tuple(q1.shape)}")
    nb = _prod(batch)
    if min(len(fields), 4) * nb * kn * ny * nx * 4 > _BLOCK_MEAN_SCRATCH_LIMIT:
        return None   # (the scratch rows are fine-size: 71 GB for a whole C3072 cube -- such calls take the three launches)
    area_repeat = nb // max(_prod(lead), 1)
    pe1, pe2_coarse, lvl, area = pe1.contiguous(), pe2_coarse.contiguous(), lvl.contiguous(), area.contiguous()
    fields = [q.contiguous() for q in fields]
    n = len(fields)
    lib = _lib.load()
    ws = _workspace(dev, int(lib.fv3hip_mappm_block_mean_workspace_bytes(nb * ny * nx, km)))
    # fine-size rows that only evicted values and redone blocks pass through: one allocation, kept per (device, stream, size)
    scratch = _scratch_rows(dev, n, nb * kn * ny * nx)
    outs = [torch.empty(batch + (kn,) + coarse_hw, dtype=torch.float32, device=dev) for _ in fields]
    q_ptrs = (ctypes.c_void_p * n)(*[q.data_ptr() for q in fields])
    s_ptrs = (ctypes.c_void_p * n)(*[scratch[i % len(scratch)].data_ptr() for i in range(n)])  # (a sweep takes four fields)
    o_ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    try:
        _lib.call_on(dev, "fv3hip_mappm_block_mean", _ptr(pe1), q_ptrs, _ptr(pe2_coarse), _ptr(lvl), kn + cmp_offset, cmp_offset,
                     _float_code(q1), _ptr(area), area_repeat, s_ptrs, o_ptrs, n, nb, ny, nx, 8, km, kn, int(iv), int(kord),
                     _arith_code(arith), _ptr(ws), ws.numel(), _stream(dev))
    except _lib.Fv3HipError as err:
        if err.code != _lib.EUNSUPPORTED:
            raise
        return None
    if counters is not None:
        counters.copy_(ws[:16].view(torch.int32), non_blocking=True)
    return outs


_scratch = {}
_BLOCK_MEAN_SCRATCH_LIMIT = 24 << 30   # bytes of scratch the fused remap + block mean may hold per (device, stream)


def _scratch_rows(dev, n: int, numel: int) -> list:
    """``n`` float32 scratch arrays of ``numel`` elements for the fused remap + block mean, one set per (device, HIP stream)
    like the remap's workspace: a sweep's four fields use the first four, the next sweep of the same call reuses them in
    stream order."""
    n = min(n, 4)
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    have = _scratch.get(key)
    if have is None or have.numel() < n * numel:
        have = _scratch[key] = torch.empty(n * numel, dtype=torch.float32, device=dev)
    return [have[i * numel:(i + 1) * numel] for i in range(n)]


class HipTimer:
    """HIP events recorded on torch's current stream (used by bench.py)."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        _lib.call("fv3hip_timer_create", ctypes.byref(self._h))

    def start(self, dev=None):
        _lib.call_on(dev, "fv3hip_timer_start", self._h, _stream(dev))

    def stop(self, dev=None):
        _lib.call_on(dev, "fv3hip_timer_stop", self._h, _stream(dev))

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        _lib.call("fv3hip_timer_elapsed_ms", self._h, ctypes.byref(ms))
        return float(ms.value)

    def __del__(self):
        try:
            _lib.load().fv3hip_timer_destroy(self._h)
        except Exception:
            pass


def device_info() -> dict:
    info = _lib.DeviceInfo()
    _lib.call("fv3hip_device_info", ctypes.byref(info))
    return {
        "name": info.name.decode(),
        "arch": info.arch.decode(),
        "compute_units": info.compute_units,
        "wavefront_size": info.wavefront_size,
        "lds_bytes_per_cu": info.lds_bytes_per_cu,
        "clock_mhz": info.clock_mhz,
        "hbm_bytes": int(info.hbm_bytes),
    }


def as_numpy(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()
