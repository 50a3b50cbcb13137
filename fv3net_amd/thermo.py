"""The pieces of ``vcm.calc.thermo`` that sit on the hot path, on the device: the pressure helpers of
the coarse-graining path (external/vcm/vcm/calc/thermo/vertically_dependent.py:41-66, 153-179, 189-)
and the humidity limiters applied to the ML tendencies every timestep
(external/vcm/vcm/calc/thermo/non_negative_sphum.py:6-45)."""
from typing import Hashable, Optional, Tuple

import torch

from . import _lib, ops
from .cubedsphere._device import like_input, on_device
from .cubedsphere.constants import COORD_Z_CENTER, COORD_Z_OUTER, TOA_PRESSURE
from .xr_compat import DataArray, from_compat, to_compat


def pressure_at_interface(delp, toa_pressure: float = TOA_PRESSURE, dim_center: Hashable = COORD_Z_CENTER,
                          dim_outer: Hashable = COORD_Z_OUTER):
    """Pressure at layer interfaces: TOA pressure followed by the running sum of ``delp`` along
    ``dim_center``; the vertical dim is renamed to ``dim_outer`` and loses its coordinate."""
    d = to_compat(delp)
    axis = d.get_axis_num(dim_center)
    res = ops.pressure_at_interface(on_device(d.data), toa_pressure, axis)
    dims = tuple(dim_outer if x == dim_center else x for x in d.dims)
    coords = {k: v for k, v in d.coords.items() if k != dim_center}
    out = DataArray(like_input(res, d.data), dims=dims, coords=coords, name=None)
    return from_compat(out, delp)


def pressure_at_midpoint_log(delp, toa_pressure: float = TOA_PRESSURE, dim: Hashable = COORD_Z_CENTER):
    """Layer-midpoint pressure, Simmons and Burridge (1981) eq. 3.17: ``delp / diff(log(p_interface))``."""
    d = to_compat(delp)
    axis = d.get_axis_num(dim)
    res = ops.pressure_at_midpoint_log(on_device(d.data), toa_pressure, axis)
    return from_compat(d._replace(data=like_input(res, d.data), name=None, attrs={}), delp)


# ---------------------------------------------------------------------------------------------
# humidity limiters (non_negative_sphum.py:6-45; constants.py:3-12, local.py:25-28,317-360)
# ---------------------------------------------------------------------------------------------
_RDGAS = 287.05
_SPECIFIC_HEAT_CONST_PRESSURE = 1004
_LATENT_HEAT_VAPORIZATION_0_C = 2.5e6
_HEAT_CAPACITY = _SPECIFIC_HEAT_CONST_PRESSURE - _RDGAS


def moist_static_energy_tendency(temperature_tendency, specific_humidity_tendency):
    """(cp - Rd) dT/dt + Lv(273.15 K) dq/dt  [W/kg] (local.py:317-337); host arithmetic on whatever is passed."""
    return _HEAT_CAPACITY * temperature_tendency + _LATENT_HEAT_VAPORIZATION_0_C * specific_humidity_tendency


def temperature_tendency(moist_static_energy_tendency, specific_humidity_tendency):
    """Inverse of :func:`moist_static_energy_tendency` for the temperature tendency (local.py:340-365)."""
    return (moist_static_energy_tendency - _LATENT_HEAT_VAPORIZATION_0_C * specific_humidity_tendency) / _HEAT_CAPACITY


def _limiter(sphum, q1, q2, dt: float, mse_conserving: bool):
    s, b = to_compat(sphum), to_compat(q2)
    a = to_compat(q1) if q1 is not None else None
    dims = s.dims
    ts = on_device(s.data)
    tb = on_device(b.transpose(*dims).data)
    ta = on_device(a.transpose(*dims).data) if a is not None else None
    dt_ = torch.float32 if all(t is None or t.dtype == torch.float32 for t in (ts, ta, tb)) else torch.float64
    ts, tb = ts.to(dt_).contiguous(), tb.to(dt_).contiguous()
    ta = ta.to(dt_).contiguous() if ta is not None else None
    if tuple(tb.shape) != tuple(ts.shape) or (ta is not None and tuple(ta.shape) != tuple(ts.shape)):
        raise ValueError("sphum and the tendencies must have the same dimensions")
    out2 = torch.empty_like(tb)
    out1 = torch.empty_like(ta) if ta is not None else None
    _lib.call_on(ts.device, "fv3hip_non_negative_sphum", ops._ptr(ts), ops._ptr(ta), ops._ptr(tb), _lib.F64 if dt_ == torch.float64 else _lib.F32,
              ts.numel(), float(dt), int(mse_conserving), ops._ptr(out1), ops._ptr(out2), ops._stream(ts.device))
    wrap = lambda t, ref, orig: from_compat(DataArray(like_input(t, s.data), dims=dims, coords=dict(s.coords)), orig)
    return (wrap(out1, a, q1) if out1 is not None else None), wrap(out2, b, q2)


def non_negative_sphum(sphum, dQ1, dQ2, dt: float) -> Tuple[object, object]:
    """Scale dQ1 and dQ2 by ``-sphum / (dt dQ2)`` wherever ``sphum + dQ2 dt`` would be negative
    (non_negative_sphum.py:6-13).  Returns (dQ1_updated, dQ2_updated)."""
    q1, q2 = _limiter(sphum, dQ1, dQ2, dt, False)
    return q1, q2


def update_moisture_tendency_to_ensure_non_negative_humidity(sphum, q2, dt: float):
    """``q2`` where ``sphum + q2 dt >= 0``, else ``-sphum / dt`` (non_negative_sphum.py:16-19)."""
    return _limiter(sphum, None, q2, dt, True)[1]


def update_temperature_tendency_to_conserve_mse(q1, q2_old, q2_new):
    """The heating that keeps the moist static energy tendency when q2 is changed (non_negative_sphum.py:22-27)."""
    return temperature_tendency(moist_static_energy_tendency(q1, q2_old), q2_new)


def non_negative_sphum_mse_conserving(sphum, q2, dt: float, q1: Optional[object] = None):
    """(q2_new, q1_new or None): limited moistening and the heating that conserves MSE
    (non_negative_sphum.py:30-45), one launch."""
    q1_new, q2_new = _limiter(sphum, q1, q2, dt, True)
    return q2_new, q1_new
