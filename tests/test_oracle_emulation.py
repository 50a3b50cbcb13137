"""The post-processing oracle against the reference's own known answers
(external/emulation/tests/test_zhao_carr.py:15-147, test_mask.py:7-47, _regtest_outputs/*.out)."""
import numpy as np
import pytest

from oracle import emulation_np as E


def test_limit_net_condensation_known_answer():
    qv = np.array([[1, 1, 1], [0, 0, 0]], dtype=np.float64)
    qc = np.array([[0, 0, 0], [1, 1, 0]], dtype=np.float64)
    net = np.array([[1.5, 0.5, 0], [-1.5, -0.5, 0]], dtype=np.float64)
    res = E.limit_net_condensation({E.CLOUD_IN: qc, E.QV_IN: qv}, net)
    np.testing.assert_array_equal(res, np.array([[1, 0.5, 0], [-1, -0.5, 0]]))


def test_ice_water_flag_known_answers():
    iw = E.ice_water_flag(np.array([[10, 0, -10, -15, -16]]), np.array([[0, 0, 0, 1, 0]]))
    np.testing.assert_array_equal(iw, np.array([[0, 0, 0.0, 1.0, 1.0]]))
    iw = E.ice_water_flag(np.array([[-14, -16]]), np.array([[0, 0]]))
    np.testing.assert_array_equal(iw, np.array([[0, 1.0]]))


def test_regtest_scalars():
    assert E.latent_heat_phase_dependent(0.5) == 2666790.0
    assert np.array(10.0) / E.RHO_WATER == 0.01
    assert np.array(2.0) * np.array(1.0) / E.GRAVITY == 0.20394324259558566


def test_strict_precip_scan_known_answer():
    c, v, total = E.strict_precip_scan(np.array([[1.0], [-2.0], [3.0]]), np.array([[4.0], [-1.0], [2.0]]))
    np.testing.assert_equal(c, [[1.0], [0.0], [3.0]])
    np.testing.assert_equal(v, [[2.0], [0.0], [2.0]])
    np.testing.assert_equal(total, np.zeros_like(total))


def _states():
    shp = (5, 10)
    state = {E.CLOUD_G: np.ones(shp) * 4, E.QV_G: np.ones(shp), E.T_G: np.ones(shp) * 10, E.DELP: np.ones(shp)}
    emulator = {E.CLOUD_P: np.ones(shp) * 2, E.QV_P: np.ones(shp) * 2}
    return state, emulator


def test_enforce_conservative_precpd_properties():
    state, emulator = _states()
    res = E.enforce_conservative_precpd(state, emulator)
    assert E.PRECIP in res and E.T_P in res
    dummy = -1 * np.ones_like(state[E.QV_G])
    res = E.enforce_conservative_precpd(state, {E.CLOUD_P: dummy * -10, E.QV_P: dummy, E.T_P: dummy, E.PRECIP: dummy})
    for v in res.values():
        assert not np.any(v == -1)
    assert not np.any(res[E.CLOUD_P] == 10)


def test_simple_conservative_overwrites_precip():
    state, emulator = _states()
    res = E.conservative_precip_simple(state, emulator)
    res[E.PRECIP] = -1 * np.ones_like(res[E.PRECIP])
    assert not np.any(E.conservative_precip_simple(state, res)[E.PRECIP] == -1)


def test_range_mask_known_answers():
    assert E.range_mask({"foo": 0.5}, "foo", 0, 1) == {"foo": 0.5}
    assert E.range_mask({"foo": 1.5}, "foo", 0, 1) == {"foo": 1.0}
    assert E.range_mask({"foo": -1.5}, "foo", 0, 1) == {"foo": 0}


@pytest.mark.parametrize("start, stop", [(2, 3), (2, 5), (None, 2), (None, None)])
def test_level_mask_known_answers(start, stop):
    ones = np.ones((4, 2))
    zeros = ones * 0
    res = E.level_mask({"foo": zeros}, {"foo": ones}, "foo", start, stop)
    sl = slice(start, stop)
    np.testing.assert_array_equal(res["foo"][sl], zeros[sl])
    assert np.sum(res["foo"]) == res["foo"].size - zeros[sl].size
    res = E.level_mask({}, {"foo": ones}, "foo", 0, 2, fill_value=0.5)
    np.testing.assert_array_equal(0.5, res["foo"][:2])
    a = ones * 1.1
    res = E.level_mask({"a": a}, {"foo": ones}, "foo", 0, 2, fill_value="a")
    np.testing.assert_array_equal(a[:2], res["foo"][:2])
