"""The derived variables of ``vcm.DerivedMapping`` on the device (fv3net_amd/fit/derived_more.py) against the numpy oracle
(oracle/derived_np.py, pinned by the reference's known answers in tests/test_oracle_derived.py); float64 results equal numpy's
to the last place wherever no transcendental function is involved."""
from datetime import datetime

import numpy as np
import pytest

from fv3net_amd.fit.derived import DerivedMapping, DerivedModel
from fv3net_amd.fit.testing import ConstantOutputPredictor
from fv3net_amd.xr_compat import DataArray, Dataset
from oracle import derived_np as o

pytestmark = pytest.mark.gpu

TOA, SFC = "total_sky_downward_shortwave_flux_at_top_of_atmosphere", "total_sky_downward_shortwave_flux_at_surface"
DELP = "pressure_thickness_of_atmospheric_layer"


def _np(da):
    return np.asarray(da.values)


def _state(rng, dtype=np.float64):
    nt, nz, ny, nx = 2, 5, 3, 4
    d3, d2 = ["tile", "z", "y", "x"], ["tile", "y", "x"]
    u = lambda lo, hi, *shape: rng.uniform(lo, hi, shape).astype(dtype)
    toa = u(0, 400, nt, ny, nx)
    toa[0, 0, :2] = 0.0   # night
    ds = Dataset({
        "latent_heat_flux": DataArray(u(-50, 300, nt, ny, nx), dims=d2),
        TOA: DataArray(toa, dims=d2), SFC: DataArray(u(0, 300, nt, ny, nx), dims=d2),
        **{k: DataArray(u(0, 100, nt, ny, nx), dims=d2) for k in ("sfc_flux_dir_nir", "sfc_flux_dif_nir", "sfc_flux_dir_vis", "sfc_flux_dif_vis")},
        "land_sea_mask": DataArray(rng.integers(0, 3, (nt, ny, nx)).astype(np.float32), dims=d2),
        "air_temperature": DataArray(u(200, 310, nt, nz, ny, nx), dims=d3),
        "specific_humidity": DataArray(u(0, 0.02, nt, nz, ny, nx), dims=d3),
        DELP: DataArray(u(300, 1500, nt, nz, ny, nx), dims=d3),
        "dQ1": DataArray(u(-1e-4, 1e-4, nt, nz, ny, nx), dims=d3), "dQ2": DataArray(u(-1e-8, 1e-8, nt, nz, ny, nx), dims=d3),
        "Q1": DataArray(u(-1e-4, 1e-4, nt, nz, ny, nx), dims=d3), "Q2": DataArray(u(-1e-8, 1e-8, nt, nz, ny, nx), dims=d3),
        "cloud_amount": DataArray(np.where(rng.uniform(size=(nt, nz, ny, nx)) < 0.3, 0.0005, rng.uniform(0, 1, (nt, nz, ny, nx))).astype(dtype), dims=d3),
        "cloud_water_mixing_ratio": DataArray(u(0, 1e-3, nt, nz, ny, nx), dims=d3),
        "cloud_ice_mixing_ratio": DataArray(u(0, 1e-4, nt, nz, ny, nx), dims=d3),
        "total_sky_upward_shortwave_flux_at_surface": DataArray(u(0, 100, nt, ny, nx), dims=d2),
        "total_sky_upward_longwave_flux_at_surface": DataArray(u(200, 500, nt, ny, nx), dims=d2),
        "sensible_heat_flux": DataArray(u(-50, 200, nt, ny, nx), dims=d2),
        "x_wind": DataArray(u(-30, 30, nt, nz, ny + 1, nx), dims=["tile", "z", "y_interface", "x"]),
        "y_wind": DataArray(u(-30, 30, nt, nz, ny, nx + 1), dims=["tile", "z", "y", "x_interface"]),
        "dQxwind": DataArray(u(-1e-3, 1e-3, nt, nz, ny + 1, nx), dims=["tile", "z", "y_interface", "x"]),
        "dQywind": DataArray(u(-1e-3, 1e-3, nt, nz, ny, nx + 1), dims=["tile", "z", "y", "x_interface"]),
        **{k: DataArray(u(-1, 1, nt, ny, nx), dims=d2) for k in
           ("eastward_wind_u_coeff", "eastward_wind_v_coeff", "northward_wind_u_coeff", "northward_wind_v_coeff")},
        "lon": DataArray(u(0, 360, nt, ny, nx), dims=d2), "lat": DataArray(u(-90, 90, nt, ny, nx), dims=d2),
    })
    return ds


def test_every_reference_variable_is_registered():
    names = {"cos_zenith_angle", "evaporation", "dQu", "dQv", "eastward_wind", "northward_wind", "dQu_parallel_to_eastward_wind",
             "dQv_parallel_to_northward_wind", "horizontal_wind_tendency_parallel_to_horizontal_wind", "net_shortwave_sfc_flux_derived",
             "downward_shortwave_sfc_flux_via_transmissivity", "net_shortwave_sfc_flux_via_transmissivity",
             "shortwave_transmissivity_of_atmospheric_column", "downward_shortwave_total_nir_at_surface",
             "downward_shortwave_total_vis_at_surface", "downward_vis_fraction_at_surface", "downward_nir_fraction_at_surface",
             "downward_vis_diffuse_fraction_at_surface", "downward_vis_direct_fraction_at_surface",
             "downward_nir_diffuse_fraction_at_surface", "downward_nir_direct_fraction_at_surface", "is_land", "is_sea", "is_sea_ice",
             "Q1", "Q2", "pQ1", "pQ2", "internal_energy", "column_integrated_dQ1", "column_integrated_dQ2", "column_integrated_Q1",
             "column_integrated_Q2", "water_vapor_path", "upward_heat_flux_at_surface", "incloud_water_mixing_ratio",
             "incloud_ice_mixing_ratio", "pressure", "relative_humidity"}   # external/vcm/vcm/derived_mapping.py, every @register
    assert names <= set(DerivedMapping.VARIABLES)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_elementwise_variables_equal_numpy(dtype):
    rng = np.random.default_rng(0)
    ds = _state(rng, dtype)
    m = DerivedMapping(ds)
    g = lambda k: _np(ds[k])
    toa = g(TOA)
    exact = lambda name, want: np.testing.assert_array_equal(_np(m[name]), want, err_msg=name)
    exact("evaporation", o.evaporation(g("latent_heat_flux")).astype(dtype))
    exact("shortwave_transmissivity_of_atmospheric_column", o.transmissivity(g(SFC), toa))
    nir, vis = g("sfc_flux_dir_nir") + g("sfc_flux_dif_nir"), g("sfc_flux_dir_vis") + g("sfc_flux_dif_vis")
    exact("downward_shortwave_total_nir_at_surface", nir)
    exact("downward_shortwave_total_vis_at_surface", vis)
    vis_frac = o.fraction(vis, g(SFC), toa).astype(dtype)
    exact("downward_vis_fraction_at_surface", vis_frac)
    exact("downward_nir_fraction_at_surface", o.complement(vis_frac, toa).astype(dtype))
    vdif = o.fraction(g("sfc_flux_dif_vis"), vis, toa).astype(dtype)
    exact("downward_vis_diffuse_fraction_at_surface", vdif)
    exact("downward_vis_direct_fraction_at_surface", o.complement(vdif, toa).astype(dtype))
    ndif = o.fraction(g("sfc_flux_dif_nir"), nir, toa).astype(dtype)
    exact("downward_nir_diffuse_fraction_at_surface", ndif)
    exact("downward_nir_direct_fraction_at_surface", o.complement(ndif, toa).astype(dtype))
    for name, value in (("is_sea", 0), ("is_land", 1), ("is_sea_ice", 2)):
        got = _np(m[name])
        assert got.dtype == np.float64
        np.testing.assert_array_equal(got, o.one_hot(g("land_sea_mask"), value))
    exact("internal_energy", ((o.CP - o.RDGAS) * g("air_temperature")).astype(dtype))
    exact("upward_heat_flux_at_surface", g("total_sky_upward_shortwave_flux_at_surface") + g("total_sky_upward_longwave_flux_at_surface")
          + g("sensible_heat_flux"))
    assert m["upward_heat_flux_at_surface"].attrs["units"] == "W/m**2"
    for name, cond in (("incloud_water_mixing_ratio", "cloud_water_mixing_ratio"), ("incloud_ice_mixing_ratio", "cloud_ice_mixing_ratio")):
        np.testing.assert_array_equal(_np(m[name]), o.incloud(g("cloud_amount"), g(cond)).astype(dtype), err_msg=name)


def test_column_integrals_pressure_and_humidity():
    rng = np.random.default_rng(1)
    ds = _state(rng)
    m = DerivedMapping(ds)
    g = lambda k: _np(ds[k])
    tol = dict(rtol=1e-13, atol=0)   # (the level sum runs in another order than numpy's pairwise one)
    np.testing.assert_allclose(_np(m["column_integrated_dQ1"]), o.column_heating(g("dQ1"), g(DELP), 1), **tol)
    np.testing.assert_allclose(_np(m["column_integrated_Q1"]), o.column_heating(g("Q1"), g(DELP), 1), **tol)
    np.testing.assert_allclose(_np(m["column_integrated_dQ2"]), o.column_moistening(g("dQ2"), g(DELP), 1), **tol)
    np.testing.assert_allclose(_np(m["column_integrated_Q2"]), o.column_moistening(g("Q2"), g(DELP), 1), **tol)
    np.testing.assert_allclose(_np(m["water_vapor_path"]), o.mass_integrate(g("specific_humidity"), g(DELP), 1), **tol)
    assert m["column_integrated_dQ2"].attrs == {"long_name": "column integrated moistening", "units": "mm/day"}
    assert m["water_vapor_path"].dims == ("tile", "y", "x")
    # midpoint pressure (Simmons & Burridge) and the relative humidity built on it
    pe = np.concatenate([np.full((2, 1, 3, 4), 300.0), 300.0 + np.cumsum(g(DELP), axis=1)], axis=1)
    p = g(DELP) / np.diff(np.log(pe), axis=1)
    np.testing.assert_allclose(_np(m["pressure"]), p, rtol=1e-12)
    np.testing.assert_allclose(_np(m["relative_humidity"]), o.relative_humidity(g("air_temperature"), g("specific_humidity"), p), rtol=1e-12)
    assert m["relative_humidity"].attrs["long_name"] == "relative humidity"


def test_winds_and_wind_tendencies():
    rng = np.random.default_rng(2)
    ds = _state(rng)
    m = DerivedMapping(ds)
    g = lambda k: _np(ds[k])
    coeffs = [g(k)[:, None] for k in ("eastward_wind_u_coeff", "eastward_wind_v_coeff", "northward_wind_u_coeff", "northward_wind_v_coeff")]
    for xk, yk, ek, nk in (("x_wind", "y_wind", "eastward_wind", "northward_wind"), ("dQxwind", "dQywind", "dQu", "dQv")):
        east, north = o.rotate(coeffs, o.shift_to_center(g(xk), 2), o.shift_to_center(g(yk), 3))
        for name, want in ((ek, east), (nk, north)):
            got = m[name]
            assert set(got.dims) == {"tile", "z", "y", "x"}
            np.testing.assert_array_equal(_np(got.transpose("tile", "z", "y", "x")), want, err_msg=name)
    e, n, dqu, dqv = (_np(m[k].transpose("tile", "z", "y", "x")) for k in ("eastward_wind", "northward_wind", "dQu", "dQv"))
    np.testing.assert_array_equal(_np(m["dQu_parallel_to_eastward_wind"].transpose("tile", "z", "y", "x")), o.parallel(e, dqu))
    np.testing.assert_array_equal(_np(m["dQv_parallel_to_northward_wind"].transpose("tile", "z", "y", "x")), o.parallel(n, dqv))
    np.testing.assert_allclose(_np(m["horizontal_wind_tendency_parallel_to_horizontal_wind"].transpose("tile", "z", "y", "x")),
                               o.tendency_projection(e, dqu, n, dqv), rtol=1e-12)
    # a wind that is in the data is taken from there (use_nonderived_if_exists)
    ds2 = Dataset({"dQu": DataArray(np.ones((2, 3)), dims=["y", "x"])})
    np.testing.assert_array_equal(_np(DerivedMapping(ds2)["dQu"]), 1.0)


class _Julian:   # a stand-in with cftime.DatetimeJulian's name and fields (cftime is not installed)
    def __init__(self, *a):
        self.year, self.month, self.day, self.hour, self.minute, self.second = a
        self.microsecond = 0


_Julian.__name__ = "DatetimeJulian"


class _NoLeap(_Julian):
    pass


_NoLeap.__name__ = "DatetimeNoLeap"


def test_cos_zenith_angle():
    from fv3net_amd.fit.derived_more import cos_zenith_angle

    for make in (datetime, _Julian):   # the reference's table, external/vcm/tests/test__zenith_angle.py:10-28
        for args, lon, lat, expected in (((2020, 3, 21, 12, 0, 0), 0.0, 0.0, 1.0), ((2020, 3, 21, 18, 0, 0), -90.0, 0.0, 1.0),
                                         ((2020, 3, 21, 18, 0, 0), 270.0, 0.0, 1.0), ((2020, 7, 6, 12, 0, 0), -90.0, 0.0, -0.0196310),
                                         ((2020, 7, 6, 9, 0, 0), 40.0, 40.0, 0.9501915), ((2020, 7, 6, 12, 0, 0), 0.0, 90.0, 0.3843733)):
            assert float(cos_zenith_angle(make(*args), lon, lat)) == pytest.approx(expected, abs=1e-3)
    with pytest.raises(ValueError, match="model_time has an invalid date type"):
        cos_zenith_angle(_NoLeap(2000, 1, 1, 0, 0, 0), 0.0, 0.0)
    rng = np.random.default_rng(3)
    ds = _state(rng)
    times = [datetime(2016, 8, 1, 0, 15), datetime(2016, 8, 1, 6, 0), datetime(2017, 1, 15, 18, 30)]
    got = cos_zenith_angle(DataArray(np.array(times, dtype=object), dims=["time"]), ds["lon"], ds["lat"])
    assert got.dims == ("time", "tile", "y", "x") and got.name == "cos_zenith_angle" and _np(got).dtype == np.float64
    for i, t in enumerate(times):
        np.testing.assert_allclose(_np(got)[i], o.cos_zenith_angle(t, _np(ds["lon"]), _np(ds["lat"])), rtol=0, atol=1e-14)
    # radians by their units attribute
    lon_r = DataArray(np.deg2rad(_np(ds["lon"])), dims=ds["lon"].dims, attrs={"units": "radians"})
    lat_r = DataArray(np.deg2rad(_np(ds["lat"])), dims=ds["lat"].dims, attrs={"units": "radians"})
    one = cos_zenith_angle(DataArray(np.array(times[1], dtype=object), dims=[]), lon_r, lat_r)
    np.testing.assert_allclose(_np(one), o.cos_zenith_angle(times[1], _np(ds["lon"]), _np(ds["lat"])), rtol=0, atol=1e-13)


def test_derived_model_with_the_new_outputs():
    rng = np.random.default_rng(4)
    ds = _state(rng)
    base = ConstantOutputPredictor(input_variables=["air_temperature"], output_variables=["dQ1", SFC])
    base.set_outputs(dQ1=np.full(5, 2.0e-5), **{SFC: 120.0})
    model = DerivedModel(base, ["column_integrated_dQ1", "shortwave_transmissivity_of_atmospheric_column", "is_land"])
    assert set(model.input_variables) == {"air_temperature", DELP, TOA, "land_sea_mask"}
    out = model.predict(ds)
    assert set(out) == {"dQ1", SFC, "column_integrated_dQ1", "shortwave_transmissivity_of_atmospheric_column", "is_land"}
    np.testing.assert_allclose(_np(out["column_integrated_dQ1"]), o.column_heating(np.full((2, 5, 3, 4), 2.0e-5), _np(ds[DELP]), 1), rtol=1e-13)
    np.testing.assert_array_equal(_np(out["shortwave_transmissivity_of_atmospheric_column"]),
                                  o.transmissivity(np.full((2, 3, 4), 120.0), _np(ds[TOA])))
    with pytest.raises(ValueError, match="Invalid variables"):
        DerivedModel(base, ["not_a_derived_variable"])
