"""oracle/derived_np.py against the reference's own known answers for the derived variables
(external/vcm/tests/test__zenith_angle.py:10-28, test_derived_mapping.py:33-59, 85-89, 147-211)."""
from datetime import datetime

import numpy as np
import pytest

from oracle import derived_np as o


@pytest.mark.parametrize("time, lon, lat, expected", [
    (datetime(2020, 3, 21, 12, 0, 0), 0.0, 0.0, 1.0),
    (datetime(2020, 3, 21, 18, 0, 0), -90.0, 0.0, 1.0),
    (datetime(2020, 3, 21, 18, 0, 0), 270.0, 0.0, 1.0),
    (datetime(2020, 7, 6, 12, 0, 0), -90.0, 0.0, -0.0196310),
    (datetime(2020, 7, 6, 9, 0, 0), 40.0, 40.0, 0.9501915),
    (datetime(2020, 7, 6, 12, 0, 0), 0.0, 90.0, 0.3843733),
])
def test_cos_zenith_angle_known_answers(time, lon, lat, expected):
    assert float(o.cos_zenith_angle(time, lon, lat)) == pytest.approx(expected, abs=1e-3)


@pytest.mark.parametrize("dqu, dqv, e, n, projection", [
    (1.0, 0.0, 1.0, 0.0, 1.0), (1.0, 0.0, -1.0, 0.0, -1.0), (1.0, 1.0, 1.0, 1.0, np.sqrt(2)), (1.0, 0.0, 1.0, 1.0, 1 / np.sqrt(2)),
    (-1.0, 0.0, 1.0, 1.0, -1 / np.sqrt(2))])
def test_wind_tendency_projection(dqu, dqv, e, n, projection):
    """The projection of (dQu, dQv) onto the unit wind vector, as the formula gives it.  (The reference's own test,
    test_derived_mapping.py:33-59, passes `projection` as the tolerance argument of pytest.approx and so asserts nothing;
    its parametrised values for the diagonal winds are off by a factor of two.)"""
    got = o.tendency_projection(np.array([e]), np.array([dqu]), np.array([n]), np.array([dqv]))
    np.testing.assert_allclose(got, [projection], rtol=1e-12)


def test_rotated_winds_with_zero_coefficients_vanish():
    ny, nx = 1, 2
    xw, yw = np.ones((ny + 1, nx)), np.ones((ny, nx + 1))
    xc, yc = o.shift_to_center(xw, 0), o.shift_to_center(yw, 1)
    assert xc.shape == yc.shape == (ny, nx) and np.all(xc == 1.0)
    east, north = o.rotate([np.zeros((ny, nx))] * 4, xc, yc)
    np.testing.assert_array_almost_equal(east, 0.0)
    np.testing.assert_array_almost_equal(north, 0.0)


def test_shortwave_known_answers():
    albedo, down = np.array([0, 0.5, 1.0]), np.array([1.0, 1.0, 1.0])
    np.testing.assert_array_almost_equal((1 - albedo) * down, [1.0, 0.5, 0.0])
    toa, trans = np.array([2.0, 1.0, 3.0]), np.array([0.5, 0.75, 1.0])
    np.testing.assert_array_almost_equal(trans * toa, [1.0, 0.75, 3.0])
    np.testing.assert_array_almost_equal((1 - albedo) * (trans * toa), [1.0, 0.375, 0.0])
    # night-time columns: every fraction is 0 where no shortwave arrives at the top (derived_mapping.py:243-244)
    toa = np.array([0.0, 100.0])
    np.testing.assert_array_equal(o.transmissivity(np.array([0.0, 50.0]), toa), [0.0, 0.5])
    np.testing.assert_array_equal(o.complement(np.array([0.3, 0.3]), toa), [0.0, 0.7])


def test_surface_type_one_hots():
    mask = np.array([0, 1, 2])
    np.testing.assert_array_equal(o.one_hot(mask, 0), [1.0, 0.0, 0.0])
    np.testing.assert_array_equal(o.one_hot(mask, 1), [0.0, 1.0, 0.0])
    np.testing.assert_array_equal(o.one_hot(mask, 2), [0.0, 0.0, 1.0])


def test_thermodynamic_sanity():
    assert o.saturation_pressure(np.array(273.15)) == pytest.approx(610.94)
    # saturated air at 1000 hPa and 20 C holds about 14.7 g/kg
    assert float(o.relative_humidity(np.array(293.15), np.array(0.0147), np.array(1.0e5))) == pytest.approx(1.0, abs=0.02)
    np.testing.assert_allclose(o.incloud(np.array([0.0005, 0.02, 0.5]), np.array([1.0, 1.0, 1.0])), [1.0, 20.0, 2.0])
    delp = np.full((3, 2), 1000.0)
    np.testing.assert_allclose(o.mass_integrate(np.ones((3, 2)), delp, 0), 3000.0 / 9.80665)
