"""The pressure helpers of ``vcm.calc.thermo`` that sit on the coarse-graining path
(external/vcm/vcm/calc/thermo/vertically_dependent.py:41-66, 153-179, 189-), on the device."""
from typing import Hashable

from . import ops
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
