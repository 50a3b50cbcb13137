"""The rest of ``vcm.DerivedMapping`` (external/vcm/vcm/derived_mapping.py:115-577): everything a ``DerivedModel`` may be asked
for as a derived output (fv3fit/_shared/models.py:210-220 accepts any registered name) -- the rotated winds and wind
tendencies, the EAMXX shortwave splits, the surface-type masks, column integrals, in-cloud condensate, midpoint pressure,
relative humidity, cos zenith angle.  Registered names, required inputs and attributes are the reference's; the arithmetic
runs on the device through ``fv3hip_ew`` / ``fv3hip_column_sum`` in the reference's operation order (so float64 results agree
with numpy to the last place wherever no transcendental function is involved)."""
import datetime
import math

import numpy as np

from .. import ops
from ..cubedsphere._device import like_input, on_device
from ..xr_compat import DataArray, Dataset, to_compat
from .derived import DerivedMapping, _binary, _scalar

# external/vcm/vcm/calc/thermo/constants.py:2-21
_GRAVITY = 9.80665
_RDGAS = 287.05
_RVGAS = 461.5
_LATENT_HEAT_VAPORIZATION_0_C = 2.5e6
_SPECIFIC_ENTHALPY_LIQUID = 4185.5
_SPECIFIC_ENTHALPY_VAP0R = 1846
_SPECIFIC_HEAT_CONST_PRESSURE = 1004
_FREEZING_TEMPERATURE = 273.15
_DEFAULT_SURFACE_TEMPERATURE = _FREEZING_TEMPERATURE + 15
_KG_M2S_TO_MM_DAY = (1e3 * 86400) / 997.0
CLIMIT1, CLIMIT2 = 0.001, 0.05   # calc/clouds.py (GFS radiation_clouds.f)


def _broadcast(op: str, a: DataArray, b: DataArray) -> DataArray:
    """``a op b`` where ``b`` may lack ONE dimension of ``a`` (a [.., y, x] coefficient against a [.., z, y, x] field): the
    kernel shares ``b`` over that axis.  The result carries xarray's dimension order for ``b * a`` (``b``'s dims first)."""
    if set(a.dims) == set(b.dims):
        return _binary(op, a, b)
    missing = [d for d in a.dims if d not in b.dims]
    if set(b.dims) - set(a.dims) or len(missing) != 1 or len(b.dims) < 2:
        raise ValueError(f"cannot combine arrays over {a.dims} and {b.dims}")
    order = tuple(b.dims[:-2]) + (missing[0],) + tuple(b.dims[-2:])
    at = a.transpose(*order)
    ta, tb = on_device(at.data).contiguous(), on_device(b.data).contiguous()
    if ta.dtype != tb.dtype:
        ta, tb = ta.double(), tb.double()
    out = at._replace(data=like_input(ops.ew(op, ta, tb), a.data), name=None)
    return out.transpose(*(tuple(b.dims) + (missing[0],)))


def _sum_z(a: DataArray, dim: str = "z") -> DataArray:
    axis = a.get_axis_num(dim)
    res = ops.column_sum(on_device(a.data).contiguous(), axis)
    coords = {k: v for k, v in a.coords.items() if k != dim}
    return DataArray(like_input(res, a.data), dims=tuple(d for d in a.dims if d != dim), coords=coords, attrs={})


def mass_integrate(da: DataArray, delp: DataArray, dim: str = "z") -> DataArray:
    """``(da * delp / g).sum(dim)`` (calc/thermo/vertically_dependent.py:18-22)."""
    return _sum_z(_scalar("div_s", _binary("mul", da, delp), _GRAVITY), dim)


def _limit_sw_positive(da: DataArray, downward_toa_shortwave_flux: DataArray) -> DataArray:
    # xr.where(toa > 0, da, 0.0)  (derived_mapping.py:243-244)
    b = downward_toa_shortwave_flux.transpose(*da.dims)
    return da._replace(data=like_input(ops.ew("where_pos_s", on_device(da.data).contiguous(), on_device(b.data).contiguous(), scalar=0.0),
                                       da.data), name=None)


# ---- evaporation, winds ------------------------------------------------------------------------------------------------------
@DerivedMapping.register("evaporation", required_inputs=["latent_heat_flux"])
def evaporation(self):
    # lhf / Lv(288.15 K)  (calc/thermo/local.py:25-28, 69-82)
    lv = _LATENT_HEAT_VAPORIZATION_0_C + (_SPECIFIC_ENTHALPY_LIQUID - _SPECIFIC_ENTHALPY_VAP0R) * (
        _DEFAULT_SURFACE_TEMPERATURE - _FREEZING_TEMPERATURE)
    return _scalar("div_s", self["latent_heat_flux"], lv)


EDGE_TO_CENTER_DIMS = {"x_interface": "x", "y_interface": "y"}


def shift_edge_var_to_center(edge: DataArray, edge_to_center_dims=None) -> DataArray:
    """Mean of the two edge values of a cell along the (first) staggered dimension (cubedsphere/coarsen.py:54-80)."""
    mapping = edge_to_center_dims or {"grid_x": "grid_xt", "grid_y": "grid_yt"}
    for dim in (d for d in mapping if d in edge.dims):
        n = edge.sizes[dim]
        lo, hi = edge.isel({dim: slice(0, n - 1)}), edge.isel({dim: slice(1, n)})
        t = ops.ew("mul_s", ops.ew("add", on_device(hi.data).contiguous(), on_device(lo.data).contiguous()), scalar=0.5)
        dims = tuple(mapping.get(d, d) for d in edge.dims)
        coords = {k: v for k, v in edge.coords.items() if k != dim and k in dims}
        return DataArray(like_input(t, edge.data), dims=dims, coords=coords, attrs=edge.attrs, name=edge.name)
    raise ValueError("Variable to shift to center must be centered on one horizontal axis and edge-valued on the other.")


def center_and_rotate_xy_winds(wind_rotation_matrix: Dataset, x_component: DataArray, y_component: DataArray):
    """D-grid x / y winds -> A-grid eastward / northward winds (cubedsphere/rotate.py:9-55)."""
    m = to_compat(wind_rotation_matrix)
    xc = shift_edge_var_to_center(to_compat(x_component), EDGE_TO_CENTER_DIMS)
    yc = shift_edge_var_to_center(to_compat(y_component), EDGE_TO_CENTER_DIMS)
    common = {k: m.coords[k] for k in ("x", "y") if k in m.coords}
    xc, yc = xc.assign_coords(common), yc.assign_coords(common)
    yc = yc.transpose(*xc.dims)
    east = _binary("add", _broadcast("mul", xc, m["eastward_wind_u_coeff"]), _broadcast("mul", yc, m["eastward_wind_v_coeff"]))
    north = _binary("add", _broadcast("mul", xc, m["northward_wind_u_coeff"]), _broadcast("mul", yc, m["northward_wind_v_coeff"]))
    return east, north


def _rotate(self, x, y):
    matrix = self.dataset(["eastward_wind_u_coeff", "eastward_wind_v_coeff", "northward_wind_u_coeff", "northward_wind_v_coeff"])
    return center_and_rotate_xy_winds(matrix, self[x], self[y])


@DerivedMapping.register("dQu", required_inputs=["dQxwind", "dQywind"], use_nonderived_if_exists=True)
def dQu(self):
    return _rotate(self, "dQxwind", "dQywind")[0]


@DerivedMapping.register("dQv", required_inputs=["dQxwind", "dQywind"], use_nonderived_if_exists=True)
def dQv(self):
    return _rotate(self, "dQxwind", "dQywind")[1]


@DerivedMapping.register("eastward_wind", use_nonderived_if_exists=True)
def eastward_wind(self):
    return _rotate(self, "x_wind", "y_wind")[0]


@DerivedMapping.register("northward_wind", use_nonderived_if_exists=True)
def northward_wind(self):
    return _rotate(self, "x_wind", "y_wind")[1]


def _parallel(wind: DataArray, tendency: DataArray) -> DataArray:
    # sign(wind / tendency) * |tendency|  (derived_mapping.py:164-176)
    sign = _unary("sign", _binary("div", wind, tendency))
    return _binary("mul", sign, _unary("abs", tendency))


def _unary(op: str, a: DataArray) -> DataArray:
    return a._replace(data=like_input(ops.ew(op, on_device(a.data).contiguous()), a.data), name=None)


@DerivedMapping.register("dQu_parallel_to_eastward_wind", required_inputs=["eastward_wind", "dQu"])
def dQu_parallel_to_eastward_wind_direction(self):
    return _parallel(self["eastward_wind"], self["dQu"])


@DerivedMapping.register("dQv_parallel_to_northward_wind", required_inputs=["northward_wind", "dQv"])
def dQv_parallel_to_northward_wind_direction(self):
    return _parallel(self["northward_wind"], self["dQv"])


@DerivedMapping.register("horizontal_wind_tendency_parallel_to_horizontal_wind",
                         required_inputs=["eastward_wind", "dQu", "northward_wind", "dQv"])
def horizontal_wind_tendency_parallel_to_horizontal_wind(self):
    # (e dQu + n dQv) / np.linalg.norm((e, n)) -- the norm of the STACKED pair, i.e. one number for the whole array, as the
    # reference computes it (derived_mapping.py:183-188)
    e, n = self["eastward_wind"], self["northward_wind"]
    te, tn = on_device(e.data), on_device(n.data)
    norm = float(((te * te).sum() + (tn * tn).sum()).sqrt())
    dot = _binary("add", _binary("mul", e, self["dQu"]), _binary("mul", n, self["dQv"]))
    return _scalar("div_s", dot, norm)


# ---- shortwave splits (EAMXX radiation) -----------------------------------------------------------------------------------------
_TOA = "total_sky_downward_shortwave_flux_at_top_of_atmosphere"
_SFC = "total_sky_downward_shortwave_flux_at_surface"


@DerivedMapping.register("shortwave_transmissivity_of_atmospheric_column", required_inputs=[_SFC, _TOA], use_nonderived_if_exists=True)
def shortwave_transmissivity_of_atmospheric_column(self):
    toa = self[_TOA]
    return _limit_sw_positive(_binary("div", self[_SFC], toa), toa)


@DerivedMapping.register("downward_shortwave_total_nir_at_surface", required_inputs=["sfc_flux_dir_nir", "sfc_flux_dif_nir"])
def downward_shortwave_total_nir_at_surface(self):
    return _binary("add", self["sfc_flux_dir_nir"], self["sfc_flux_dif_nir"])


@DerivedMapping.register("downward_shortwave_total_vis_at_surface", required_inputs=["sfc_flux_dir_vis", "sfc_flux_dif_vis"])
def downward_shortwave_total_vis_at_surface(self):
    return _binary("add", self["sfc_flux_dir_vis"], self["sfc_flux_dif_vis"])


@DerivedMapping.register("downward_vis_fraction_at_surface", required_inputs=[_SFC, "downward_shortwave_total_nir_at_surface", _TOA],
                         use_nonderived_if_exists=True)
def downward_vis_fraction_at_surface(self):
    return _limit_sw_positive(_binary("div", self["downward_shortwave_total_vis_at_surface"], self[_SFC]), self[_TOA])


def _one_minus(a: DataArray) -> DataArray:
    return _scalar("rsub_s", a, 1.0)


@DerivedMapping.register("downward_nir_fraction_at_surface", required_inputs=["downward_vis_fraction_at_surface", _TOA])
def downward_nir_fraction_at_surface(self):
    return _limit_sw_positive(_one_minus(self["downward_vis_fraction_at_surface"]), self[_TOA])


@DerivedMapping.register("downward_vis_diffuse_fraction_at_surface",
                         required_inputs=["downward_shortwave_total_vis_at_surface", "sfc_flux_dif_vis", _TOA], use_nonderived_if_exists=True)
def downward_vis_diffuse_fraction_at_surface(self):
    return _limit_sw_positive(_binary("div", self["sfc_flux_dif_vis"], self["downward_shortwave_total_vis_at_surface"]), self[_TOA])


@DerivedMapping.register("downward_vis_direct_fraction_at_surface", required_inputs=["downward_vis_diffuse_fraction_at_surface", _TOA],
                         use_nonderived_if_exists=True)
def downward_vis_direct_fraction_at_surface(self):
    return _limit_sw_positive(_one_minus(self["downward_vis_diffuse_fraction_at_surface"]), self[_TOA])


@DerivedMapping.register("downward_nir_diffuse_fraction_at_surface",
                         required_inputs=["downward_shortwave_total_nir_at_surface", "sfc_flux_dif_nir", _TOA], use_nonderived_if_exists=True)
def downward_nir_diffuse_fraction_at_surface(self):
    return _limit_sw_positive(_binary("div", self["sfc_flux_dif_nir"], self["downward_shortwave_total_nir_at_surface"]), self[_TOA])


@DerivedMapping.register("downward_nir_direct_fraction_at_surface", required_inputs=["downward_nir_diffuse_fraction_at_surface", _TOA],
                         use_nonderived_if_exists=True)
def downward_nir_direct_fraction(self):
    return _limit_sw_positive(_one_minus(self["downward_nir_diffuse_fraction_at_surface"]), self[_TOA])


# ---- surface type one-hots --------------------------------------------------------------------------------------------------------
def _is_type(mask: DataArray, value: float) -> DataArray:
    # xr.where(isclose(mask, value), 1.0, 0.0): a float64 one-hot (derived_mapping.py:392-413)
    t = on_device(mask.data).double().contiguous()
    return mask._replace(data=like_input(ops.ew("isclose_s", t, scalar=value), mask.data), name=None)


@DerivedMapping.register("is_land", required_inputs=["land_sea_mask"])
def is_land(self):
    return _is_type(self["land_sea_mask"], 1.0)


@DerivedMapping.register("is_sea", required_inputs=["land_sea_mask"])
def is_sea(self):
    return _is_type(self["land_sea_mask"], 0.0)


@DerivedMapping.register("is_sea_ice", required_inputs=["land_sea_mask"])
def is_sea_ice(self):
    return _is_type(self["land_sea_mask"], 2.0)


# ---- energetics and column integrals ------------------------------------------------------------------------------------------------
_DELP = "pressure_thickness_of_atmospheric_layer"


@DerivedMapping.register("internal_energy", required_inputs=["air_temperature"])
def internal_energy(self):
    res = _scalar("mul_s", self._mapper["air_temperature"], _SPECIFIC_HEAT_CONST_PRESSURE - _RDGAS)
    return res.assign_attrs({"long_name": "internal energy", "units": "J/kg"})


def _column_heating(t: DataArray, delp: DataArray) -> DataArray:
    res = _scalar("mul_s", mass_integrate(t, delp), _SPECIFIC_HEAT_CONST_PRESSURE - _RDGAS)
    return res.assign_attrs({"long_name": "column integrated heating", "units": "W/m**2"})


def _column_moistening(q: DataArray, delp: DataArray) -> DataArray:
    # -(KG_M2S_TO_MM_DAY * mass_integrate(q * -1, delp))  (vertically_dependent.py:310-332, derived_mapping.py:467-473)
    minus = _scalar("mul_s", mass_integrate(_scalar("mul_s", q, -1.0), delp), _KG_M2S_TO_MM_DAY)
    return _scalar("mul_s", minus, -1.0).assign_attrs({"long_name": "column integrated moistening", "units": "mm/day"})


@DerivedMapping.register("column_integrated_dQ1", required_inputs=["dQ1", _DELP])
def column_integrated_dQ1(self):
    return _column_heating(self._mapper["dQ1"], self._mapper[_DELP])


@DerivedMapping.register("column_integrated_dQ2", required_inputs=["dQ2", _DELP])
def column_integrated_dQ2(self):
    return _column_moistening(self._mapper["dQ2"], self._mapper[_DELP])


@DerivedMapping.register("column_integrated_Q1", required_inputs=["Q1", _DELP])
def column_integrated_Q1(self):
    return _column_heating(self._mapper["Q1"], self._mapper[_DELP])


@DerivedMapping.register("column_integrated_Q2", required_inputs=["Q2", _DELP])
def column_integrated_Q2(self):
    return _column_moistening(self._mapper["Q2"], self._mapper[_DELP])


@DerivedMapping.register("water_vapor_path", required_inputs=["specific_humidity", _DELP], use_nonderived_if_exists=True)
def water_vapor_path(self):
    res = mass_integrate(self._mapper["specific_humidity"], self._mapper[_DELP], dim="z")
    return res.assign_attrs({"long_name": "column integrated water vapor", "units": "mm"})


@DerivedMapping.register("upward_heat_flux_at_surface", required_inputs=[
    "total_sky_upward_shortwave_flux_at_surface", "total_sky_upward_longwave_flux_at_surface", "sensible_heat_flux"])
def upward_heat_flux_at_surface(self):
    res = _binary("add", _binary("add", self["total_sky_upward_shortwave_flux_at_surface"], self["total_sky_upward_longwave_flux_at_surface"]),
                  self["sensible_heat_flux"])
    return res.assign_attrs(long_name="Upward heat (sensible+radiative) flux at surface", units="W/m**2")


# ---- clouds, pressure, humidity ----------------------------------------------------------------------------------------------------------
def gridcell_to_incloud_condensate(cloud_fraction: DataArray, gridcell_cloud_condensate: DataArray, climit1: float = CLIMIT1,
                                   climit2: float = CLIMIT2) -> DataArray:
    """Gridcell-mean -> in-cloud condensate by the cloud fraction (calc/clouds.py:7-37)."""
    cf = on_device(cloud_fraction.data).contiguous()
    g = on_device(gridcell_cloud_condensate.transpose(*cloud_fraction.dims).data).contiguous()
    if cf.dtype != g.dtype:
        cf, g = cf.double(), g.double()
    ratio = ops.ew("rdiv_s", ops.ew("where_gt_s", cf, scalar=climit2), scalar=1.0)
    res = ops.ew("select", g, ops.ew("mul", g, ratio), ops.ew("le_s", cf, scalar=climit1))   # g where cf <= climit1 else g * ratio
    out = cloud_fraction._replace(data=like_input(res, gridcell_cloud_condensate.data), name=gridcell_cloud_condensate.name,
                                  attrs=gridcell_cloud_condensate.attrs)
    return out.transpose(*gridcell_cloud_condensate.dims)


@DerivedMapping.register("incloud_water_mixing_ratio", required_inputs=["cloud_amount", "cloud_water_mixing_ratio"])
def incloud_water_mixing_ratio(self):
    res = gridcell_to_incloud_condensate(self["cloud_amount"], self["cloud_water_mixing_ratio"])
    return res.assign_attrs(long_name="in-cloud water mixing ratio", units="kg/kg")


@DerivedMapping.register("incloud_ice_mixing_ratio", required_inputs=["cloud_amount", "cloud_ice_mixing_ratio"])
def incloud_ice_mixing_ratio(self):
    res = gridcell_to_incloud_condensate(self["cloud_amount"], self["cloud_ice_mixing_ratio"])
    return res.assign_attrs(long_name="in-cloud ice mixing ratio", units="kg/kg")


@DerivedMapping.register("pressure", required_inputs=[_DELP])
def pressure(self):
    from ..thermo import pressure_at_midpoint_log

    res = to_compat(pressure_at_midpoint_log(self[_DELP], dim="z"))
    return res.assign_attrs(long_name="pressure at layer midpoint", units="Pa")


def relative_humidity_from_pressure(temperature: DataArray, specific_humidity: DataArray, pressure: DataArray) -> DataArray:
    """Wallace and Hobbs (2006) eq. 3.59 over the August-Roche-Magnus saturation pressure (calc/thermo/local.py:211-263)."""
    q, p = specific_humidity.transpose(*temperature.dims), pressure.transpose(*temperature.dims)
    mixing_ratio = _binary("div", q, _one_minus(q))
    partial = _binary("div", _binary("mul", p, mixing_ratio), _scalar("add_s", mixing_ratio, _RDGAS / _RVGAS))
    tc = _scalar("add_s", temperature, -273.15)
    saturation = _scalar("mul_s", _unary("exp", _binary("div", _scalar("mul_s", tc, 17.625), _scalar("add_s", tc, 243.04))), 610.94)
    return _binary("div", partial, saturation)


@DerivedMapping.register("relative_humidity", required_inputs=["air_temperature", "specific_humidity", "pressure"])
def relative_humidity(self):
    res = relative_humidity_from_pressure(self["air_temperature"], self["specific_humidity"], self["pressure"])
    return res.assign_attrs(long_name="relative humidity", units="-")


# ---- cos zenith angle (calc/_zenith_angle.py) ----------------------------------------------------------------------------------------------
def _days_from_2000(model_time) -> float:
    """Days since 2000-01-01 12:00 of a ``datetime.datetime`` or a cftime ``DatetimeJulian`` (anything else is refused, as in
    the reference: the formulas below assume a calendar with real leap years)."""
    if isinstance(model_time, datetime.datetime):
        return (model_time - datetime.datetime(2000, 1, 1, 12, 0)) / datetime.timedelta(days=1)
    if type(model_time).__name__ == "DatetimeJulian":   # (cftime is not a dependency: its Julian-calendar dates by their fields)
        def julian_day_number(y, m, d):   # Julian calendar: a leap year every fourth year
            a = (14 - m) // 12
            yy, mm = y + 4800 - a, m + 12 * a - 3
            return d + (153 * mm + 2) // 5 + 365 * yy + yy // 4 - 32083
        t = model_time
        days = julian_day_number(t.year, t.month, t.day) - julian_day_number(2000, 1, 1)
        seconds = t.hour * 3600 + t.minute * 60 + t.second + getattr(t, "microsecond", 0) * 1e-6 - 12 * 3600
        return days + seconds / 86400.0
    raise ValueError("model_time has an invalid date type. It must be either datetime.datetime or cftime.DatetimeJulian. "
                     f"Got {type(model_time)}.")


def _sun_position(model_time):
    """(Greenwich mean sidereal time, right ascension, declination) in radians, float64 host scalars
    (calc/_zenith_angle.py:108-211: the AIAA 2006 sidereal time, Meeus' low-accuracy solar coordinates)."""
    t = _days_from_2000(model_time) / 36525.0
    theta = 67310.54841 + t * (876600 * 3600 + 8640184.812866 + t * (0.093104 - t * 6.2 * 10e-6))
    gmst = math.radians(theta / 240.0) % (2 * math.pi)
    mean_anomaly = math.radians(357.52910 + 35999.05030 * t - 0.0001559 * t * t - 0.00000048 * t * t * t)
    mean_longitude = math.radians(280.46645 + 36000.76983 * t + 0.0003032 * (t ** 2))
    d_l = math.radians((1.914600 - 0.004817 * t - 0.000014 * (t ** 2)) * math.sin(mean_anomaly)
                       + (0.019993 - 0.000101 * t) * math.sin(2 * mean_anomaly) + 0.000290 * math.sin(3 * mean_anomaly))
    eclon = mean_longitude + d_l
    eps = math.radians(23.0 + 26.0 / 60.0 + 21.406 / 3600.0 - (46.836769 * t - 0.0001831 * (t ** 2) + 0.00200340 * (t ** 3)
                                                             - 0.576e-6 * (t ** 4) - 4.34e-8 * (t ** 5)) / 3600.0)
    x, y, z = math.cos(eclon), math.cos(eps) * math.sin(eclon), math.sin(eps) * math.sin(eclon)
    r = math.sqrt(1.0 - z * z)
    return gmst, 2 * math.atan2(y, x + r), math.atan2(z, r)


def _degrees(da: DataArray) -> DataArray:
    units = str(da.attrs.get("units", "")).lower()
    return _scalar("mul_s", da, 180.0 / math.pi).assign_attrs(units="degrees") if "rad" in units else da


def cos_zenith_angle(time, lon, lat):
    """Cosine of the solar zenith angle at ``time`` (UTC; one time or an array of times) for ``lon`` / ``lat`` in degrees (or
    radians when their ``units`` attribute says so), float64 (calc/_zenith_angle.py:59-98, 226-242)."""
    if isinstance(lon, (int, float, np.ndarray, np.generic)):   # plain numbers / numpy arrays: host arithmetic
        lon_r, lat_r = np.asarray(lon, dtype=np.float64) * (np.pi / 180.0), np.asarray(lat, dtype=np.float64) * (np.pi / 180.0)
        gmst, ra, dec = _sun_position(time)
        return np.sin(lat_r) * np.sin(dec) + np.cos(lat_r) * np.cos(dec) * np.cos(gmst + lon_r - ra)
    lon_c, lat_c = _degrees(to_compat(lon)), _degrees(to_compat(lat))
    lat_c = lat_c.transpose(*lon_c.dims)
    lon_r = ops.ew("mul_s", on_device(lon_c.data).double().contiguous(), scalar=np.pi / 180.0)
    lat_r = ops.ew("mul_s", on_device(lat_c.data).double().contiguous(), scalar=np.pi / 180.0)
    sin_lat, cos_lat = ops.ew("sin", lat_r), ops.ew("cos", lat_r)
    t = to_compat(time)
    times = np.asarray(t.values if isinstance(t, DataArray) else t, dtype=object)
    slabs = []
    for one in times.ravel():
        gmst, ra, dec = _sun_position(one)
        hour = ops.ew("cos", ops.ew("add_s", lon_r, scalar=gmst - ra))
        slabs.append(ops.ew("add", ops.ew("mul_s", sin_lat, scalar=math.sin(dec)),
                            ops.ew("mul", ops.ew("mul_s", cos_lat, scalar=math.cos(dec)), hour)))
    if times.ndim == 0:
        return DataArray(like_input(slabs[0], lon_c.data), dims=lon_c.dims, coords=lon_c.coords, name="cos_zenith_angle", attrs={"units": ""})
    import torch

    tdims = tuple(t.dims) if isinstance(t, DataArray) else ("time",)
    data = torch.stack(slabs).reshape(tuple(times.shape) + tuple(slabs[0].shape))
    coords = dict(lon_c.coords)
    if isinstance(t, DataArray):
        coords.update(t.coords)
    return DataArray(like_input(data, lon_c.data), dims=tdims + tuple(lon_c.dims), coords=coords, name="cos_zenith_angle", attrs={"units": ""})


@DerivedMapping.register("cos_zenith_angle", required_inputs=["time", "lon", "lat"])
def cos_zenith_angle_variable(self):
    return cos_zenith_angle(self["time"], self["lon"], self["lat"])
