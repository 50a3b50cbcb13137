"""Zhao-Carr specific post-processing of the emulator's outputs, on the device
(external/emulation/emulation/zhao_carr.py:60-344, same function names and state/emulator
dictionaries of [feature(z), sample] arrays).  Each function is one HIP launch
(``csrc/emulation.hip``); numpy arrays are uploaded and the results come back as numpy, device
tensors stay on the device.  Arithmetic follows numpy's promotion: float64 as soon as one operand
(the Fortran state) is float64.
"""

import numpy as np
import torch

from .. import _lib
from ..cubedsphere._device import like_input, on_device
from ..ops import _ptr, _stream

# zhao_carr.py:33-37 (physcons.f)
GRAVITY = 9.80665
CP = 1.0046e3
LV = 2.5e6
RHO_WATER = 1000.0

# sorted(fv3fit.emulation.transforms.zhao_carr.CLASS_NAMES) (transforms/zhao_carr.py:24-36)
CLASS_NAMES = ["negative_tendency", "positive_tendency", "zero_cloud", "zero_tendency"]
ZERO_CLOUD, ZERO_TENDENCY = "zero_cloud", "zero_tendency"
POSITIVE_TENDENCY, NEGATIVE_TENDENCY = "positive_tendency", "negative_tendency"

__all__ = [
    "infer_gscond_cloud_from_conservation", "squash_gscond", "squash_precpd", "mask_where_fortran_cloud_identical",
    "mask_where_fortran_cloud_vanishes_gscond", "mask_zero_tend_classifier", "mask_zero_cloud_classifier",
    "mask_zero_cloud_classifier_precpd", "enforce_conservative_gscond", "enforce_conservative_phase_dependent",
    "enforce_conservative_precpd", "conservative_precip_simple",
]


class Input:
    cloud_water = "cloud_water_mixing_ratio_input"
    humidity = "specific_humidity_input"
    temperature = "air_temperature_input"
    delp = "pressure_thickness_of_atmospheric_layer"


class GscondOutput:
    cloud_water = "cloud_water_mixing_ratio_after_gscond"
    humidity = "specific_humidity_after_gscond"
    temperature = "air_temperature_after_gscond"


class PrecpdOutput:
    cloud_water = "cloud_water_mixing_ratio_after_precpd"
    humidity = "specific_humidity_after_precpd"
    temperature = "air_temperature_after_precpd"
    precip = "total_precipitation"


_CODE = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _dev(a) -> torch.Tensor:
    t = on_device(a)
    if t.dtype not in _CODE:
        t = t.to(torch.float64)  # numpy promotes integer / bool operands to float64
    return t.contiguous()


def _same_dtype(*ts):
    """State arrays travel as one dtype (float64 if any is)."""
    dt = torch.float64 if any(t.dtype == torch.float64 for t in ts) else torch.float32
    return [t if t.dtype == dt else t.to(dt) for t in ts], dt


def _out_dtype(*dts):
    return torch.float64 if any(d == torch.float64 for d in dts) else torch.float32


def _n01(t: torch.Tensor):
    if t.dim() == 1:
        return 1, int(t.shape[0])
    return int(np.prod(t.shape[:-1])), int(t.shape[-1])


def squash_water_water_conserving(cloud, humidity, bound: float):
    c, h = _dev(cloud), _dev(humidity)
    odt = _out_dtype(c.dtype, h.dtype)
    cloud_out = torch.empty_like(c)
    qv_out = torch.empty(c.shape, dtype=odt, device=c.device)
    _lib.call_on(c.device, "fv3hip_zc_squash", _ptr(c), _CODE[c.dtype], _ptr(h), _CODE[h.dtype], c.numel(), float(bound), _CODE[odt],
              _ptr(cloud_out), _ptr(qv_out), _stream(c.device))
    return like_input(cloud_out, cloud), like_input(qv_out, cloud)


def _apply_squash(struct, output_state, cloud_squash: float):
    out = {**output_state}
    if struct.cloud_water in output_state:
        cloud, humidity = squash_water_water_conserving(output_state[struct.cloud_water], output_state[struct.humidity],
                                                        cloud_squash)
        out[struct.cloud_water] = cloud
        out[struct.humidity] = humidity
    return out


def squash_gscond(state, emulator, cloud_squash):
    return _apply_squash(GscondOutput, emulator, cloud_squash)


def squash_precpd(state, emulator, cloud_squash):
    return _apply_squash(PrecpdOutput, emulator, cloud_squash)


def infer_gscond_cloud_from_conservation(state, emulator):
    (c_in, qv_in), sdt = _same_dtype(_dev(state[Input.cloud_water]), _dev(state[Input.humidity]))
    qv_e = _dev(emulator[GscondOutput.humidity])
    odt = _out_dtype(sdt, qv_e.dtype)
    out = torch.empty(c_in.shape, dtype=odt, device=c_in.device)
    _lib.call_on(c_in.device, "fv3hip_zc_infer_cloud", _ptr(c_in), _ptr(qv_in), _CODE[sdt], _ptr(qv_e), _CODE[qv_e.dtype], c_in.numel(),
              _CODE[odt], _ptr(out), _stream(c_in.device))
    return {**emulator, GscondOutput.cloud_water: like_input(out, emulator[GscondOutput.humidity])}


_MODES = {"none": 0, "fortran_vanishes": 1, "fortran_identical": 2, "class_zero_cloud": 3, "class_zero_tend": 4}


def _gscond_conserve(state, emulator, mode: str, phase_dependent: bool):
    """The gscond mask ``mode`` followed by ``_update_with_net_condensation`` (zhao_carr.py:97-246)."""
    (c_in, qv_in, t_in), sdt = _same_dtype(_dev(state[Input.cloud_water]), _dev(state[Input.humidity]),
                                           _dev(state[Input.temperature]))
    c_e = _dev(emulator[GscondOutput.cloud_water])
    aux, n_class, cls = None, 0, 0
    dts = [sdt, c_e.dtype]
    if mode in ("fortran_vanishes", "fortran_identical"):
        aux = _dev(state[GscondOutput.cloud_water])
        dts.append(aux.dtype)
    elif mode in ("class_zero_cloud", "class_zero_tend"):
        aux = _dev(emulator["gscond_classes"])
        n_class = int(aux.shape[0])
        if n_class != len(CLASS_NAMES):
            raise ValueError(f"gscond_classes must hold {len(CLASS_NAMES)} classes along its first axis, got {n_class}")
        cls = CLASS_NAMES.index(ZERO_CLOUD if mode == "class_zero_cloud" else ZERO_TENDENCY)
    odt = _out_dtype(*dts)
    n0, n1 = _n01(c_in)
    outs = [torch.empty(c_in.shape, dtype=odt, device=c_in.device) for _ in range(3)]
    _lib.call_on(c_in.device, "fv3hip_zc_gscond_conserve", _ptr(c_in), _ptr(qv_in), _ptr(t_in), _CODE[sdt], _ptr(c_e), _CODE[c_e.dtype],
              _MODES[mode], _ptr(aux), _CODE[aux.dtype] if aux is not None else 0, n_class, cls, n0, n1,
              1 if phase_dependent else 0, _CODE[odt], _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _stream(c_in.device))
    ref = emulator[GscondOutput.cloud_water]
    return {**emulator, GscondOutput.cloud_water: like_input(outs[0], ref), GscondOutput.humidity: like_input(outs[1], ref),
            GscondOutput.temperature: like_input(outs[2], ref)}


def mask_where_fortran_cloud_vanishes_gscond(state, emulator):
    return _gscond_conserve(state, emulator, "fortran_vanishes", False)


def mask_where_fortran_cloud_identical(state, emulator):
    return _gscond_conserve(state, emulator, "fortran_identical", False)


def mask_zero_cloud_classifier(state, emulator):
    return _gscond_conserve(state, emulator, "class_zero_cloud", False)


def mask_zero_tend_classifier(state, emulator):
    return _gscond_conserve(state, emulator, "class_zero_tend", False)


def enforce_conservative_gscond(state, emulator):
    return _gscond_conserve(state, emulator, "none", False)


def enforce_conservative_phase_dependent(state, emulator):
    return _gscond_conserve(state, emulator, "none", True)


def mask_zero_cloud_classifier_precpd(state, emulator):
    x = _dev(emulator[PrecpdOutput.cloud_water])
    logits = _dev(emulator["precpd_classes"])
    out = torch.empty_like(x)
    _lib.call_on(x.device, "fv3hip_zc_class_zero", _ptr(x), _CODE[x.dtype], _ptr(logits), _CODE[logits.dtype], int(logits.shape[0]),
              CLASS_NAMES.index(ZERO_CLOUD), x.numel(), _ptr(out), _stream(x.device))
    return {**emulator, PrecpdOutput.cloud_water: like_input(out, emulator[PrecpdOutput.cloud_water])}


def enforce_conservative_precpd(state, emulator):
    (c_g, qv_g, t_g, delp), sdt = _same_dtype(_dev(state[GscondOutput.cloud_water]), _dev(state[GscondOutput.humidity]),
                                              _dev(state[GscondOutput.temperature]), _dev(state[Input.delp]))
    (c_p, qv_p), edt = _same_dtype(_dev(emulator[PrecpdOutput.cloud_water]), _dev(emulator[PrecpdOutput.humidity]))
    if c_g.dim() != 2:
        raise ValueError("Expected 2D inputs to the strict conservative precip function")
    odt = _out_dtype(sdt, edt)
    n0, n1 = int(c_g.shape[0]), int(c_g.shape[1])
    outs = [torch.empty(c_g.shape, dtype=odt, device=c_g.device) for _ in range(3)]
    precip = torch.empty((n1,), dtype=odt, device=c_g.device)
    _lib.call_on(c_g.device, "fv3hip_zc_precpd_conserve", _ptr(c_g), _ptr(qv_g), _ptr(t_g), _ptr(delp), _CODE[sdt], _ptr(c_p), _ptr(qv_p),
              _CODE[edt], n0, n1, _CODE[odt], _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(precip), _stream(c_g.device))
    ref = emulator[PrecpdOutput.cloud_water]
    return {**emulator, PrecpdOutput.cloud_water: like_input(outs[0], ref), PrecpdOutput.humidity: like_input(outs[1], ref),
            PrecpdOutput.temperature: like_input(outs[2], ref), PrecpdOutput.precip: like_input(precip, ref)}


def conservative_precip_simple(state, emulator, sum_axis=0):
    if sum_axis != 0:
        raise NotImplementedError("conservative_precip_simple sums over the first (level) axis")
    (c_g, qv_g, delp), sdt = _same_dtype(_dev(state[GscondOutput.cloud_water]), _dev(state[GscondOutput.humidity]),
                                         _dev(state[Input.delp]))
    (c_p, qv_p), edt = _same_dtype(_dev(emulator[PrecpdOutput.cloud_water]), _dev(emulator[PrecpdOutput.humidity]))
    odt = _out_dtype(sdt, edt)
    n0, n1 = int(c_g.shape[0]), int(np.prod(c_g.shape[1:]))
    precip = torch.empty(tuple(c_g.shape[1:]), dtype=odt, device=c_g.device)
    _lib.call_on(c_g.device, "fv3hip_zc_precip_simple", _ptr(c_g), _ptr(qv_g), _ptr(delp), _CODE[sdt], _ptr(c_p), _ptr(qv_p), _CODE[edt],
              n0, n1, _CODE[odt], _ptr(precip), _stream(c_g.device))
    return {**emulator, PrecpdOutput.precip: like_input(precip, emulator[PrecpdOutput.cloud_water])}


def mixing_ratio_to_mass(x, delp):
    """kg/kg -> kg/m2 (host arithmetic on scalars / small arrays; zhao_carr.py:281-283)."""
    return x * delp / GRAVITY


def mass_to_mixing_ratio(x, delp):
    return x / delp * GRAVITY


def liquid_water_equivalent(x):
    return x / RHO_WATER


def latent_heat_phase_dependent(iw):
    return 2.5e6 + iw * 3.3358e5
