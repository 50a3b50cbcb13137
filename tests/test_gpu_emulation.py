"""Zhao-Carr post-processing on the device against the numpy oracle (oracle/emulation_np.py, itself
pinned by the reference's known answers in tests/test_oracle_emulation.py) and against those known
answers directly (external/emulation/tests/test_zhao_carr.py, test_mask.py).  float64 state (what
the Fortran hook passes) with float32 emulator outputs: results must agree to the last bit or two
(the same IEEE operations in the same order; tolerances 1e-12 relative for float64, 1e-6 for float32)."""
import numpy as np
import pytest
import torch

from oracle import emulation_np as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _state(rng, n0=79, n1=257, dt=np.float64):
    t = rng.uniform(230, 300, (n0, n1))
    qv = 10 ** rng.uniform(-6, -2, (n0, n1))
    qc = np.where(rng.random((n0, n1)) < 0.5, 0.0, 10 ** rng.uniform(-9, -3, (n0, n1)))
    state = {E.T_IN: t, E.QV_IN: qv, E.CLOUD_IN: qc, E.DELP: rng.uniform(300, 1500, (n0, n1))}
    dq = rng.normal(0, 2e-4, (n0, n1))
    state[E.CLOUD_G] = np.where(rng.random((n0, n1)) < 0.3, qc, np.maximum(qc + dq, 0))
    state[E.QV_G] = qv - (state[E.CLOUD_G] - qc)
    state[E.T_G] = t + 2.5e6 / 1004.6 * (state[E.CLOUD_G] - qc)
    return {k: v.astype(dt) for k, v in state.items()}


def _emulator(rng, state, dt=np.float32):
    sh = state[E.T_IN].shape
    em = {
        E.CLOUD_G: state[E.CLOUD_IN] + rng.normal(0, 3e-4, sh),
        E.QV_G: state[E.QV_IN] + rng.normal(0, 3e-4, sh),
        E.T_G: state[E.T_IN] + rng.normal(0, 1, sh),
        E.CLOUD_P: state[E.CLOUD_G] + rng.normal(0, 3e-4, sh),
        E.QV_P: state[E.QV_G] + rng.normal(0, 3e-4, sh),
        E.T_P: state[E.T_G] + rng.normal(0, 1, sh),
        E.PRECIP: rng.uniform(0, 1e-3, sh[1]),
        "gscond_classes": rng.normal(0, 1, (4,) + sh),
        "precpd_classes": rng.normal(0, 1, (4,) + sh),
    }
    return {k: v.astype(dt) for k, v in em.items()}


def _close(res, ref, name):
    assert res.shape == ref.shape and res.dtype == ref.dtype, name
    rtol = 1e-12 if ref.dtype == np.float64 else 2e-6
    np.testing.assert_allclose(res, ref, rtol=rtol, atol=0, err_msg=name)


@pytest.mark.parametrize("sdt, edt", [(np.float64, np.float32), (np.float64, np.float64), (np.float32, np.float32)])
def test_gscond_conservation_and_masks(device, sdt, edt):
    from fv3net_amd.emulation import zhao_carr as zc

    rng = np.random.default_rng(3)
    state = _state(rng, dt=sdt)
    em = _emulator(rng, state, dt=edt)
    cases = [
        (zc.enforce_conservative_gscond, "none", False),
        (zc.enforce_conservative_phase_dependent, "none", True),
        (zc.mask_where_fortran_cloud_vanishes_gscond, "fortran_vanishes", False),
        (zc.mask_where_fortran_cloud_identical, "fortran_identical", False),
        (zc.mask_zero_cloud_classifier, "class_zero_cloud", False),
        (zc.mask_zero_tend_classifier, "class_zero_tend", False),
    ]
    for fn, mode, phase in cases:
        res = fn(state, em)
        ref = E.update_with_net_condensation(E.gscond_cloud_choice(state, em, mode), state, em, phase_dependent=phase)
        for key in (E.CLOUD_G, E.QV_G, E.T_G):
            _close(res[key], ref[key], f"{fn.__name__}:{key}")
        assert res[E.CLOUD_P] is em[E.CLOUD_P]  # untouched entries pass through


def test_ice_water_flag_scan_known_answers_through_the_kernel(device):
    """test_zhao_carr.py:29-42: the flag itself is internal here, so read it back from the latent
    heat it selects: T_out - T_in = lv * net / cp with net = 1e-3 everywhere."""
    from fv3net_amd.emulation import zhao_carr as zc

    for t_c, cloud, expected in (([10, 0, -10, -15, -16], [0, 0, 0, 1, 0], [0, 0, 0.0, 1.0, 1.0]),
                                 ([-14, -16], [0, 0], [0, 1.0])):
        t = np.array([t_c], dtype=np.float64) + 273.16
        c = np.array([cloud], dtype=np.float64)
        state = {E.T_IN: t, E.CLOUD_IN: c, E.QV_IN: np.ones_like(t)}
        em = {E.CLOUD_G: c + 1e-3}
        res = zc.enforce_conservative_phase_dependent(state, em)
        lv = (res[E.T_G] - t) * E.CP / 1e-3
        iw = (lv - E.LV) / E.HFUS
        np.testing.assert_allclose(iw, np.array([expected]), atol=1e-6)


def test_ice_water_flag_long_rows(device):
    """Rows longer than the 256 scan segments, with runs that carry the flag across segment borders."""
    from fv3net_amd.emulation import zhao_carr as zc

    rng = np.random.default_rng(4)
    n0, n1 = 5, 4099
    t = np.where(rng.random((n0, n1)) < 0.02, 250.0, rng.uniform(258.2, 273.1, (n0, n1)))   # mostly the carry range
    t[:, rng.integers(0, n1, 20)] = 280.0
    c = np.where(rng.random((n0, n1)) < 0.97, 1e-5, 0.0)
    state = {E.T_IN: t, E.CLOUD_IN: c, E.QV_IN: np.full_like(t, 1e-2)}
    em = {E.CLOUD_G: (c + 1e-4).astype(np.float32)}
    res = zc.enforce_conservative_phase_dependent(state, em)
    ref = E.update_with_net_condensation(em[E.CLOUD_G], state, em, phase_dependent=True)
    _close(res[E.T_G], ref[E.T_G], "T after gscond")


@pytest.mark.parametrize("sdt, edt", [(np.float64, np.float32), (np.float32, np.float32)])
def test_precpd_conservation(device, sdt, edt):
    from fv3net_amd.emulation import zhao_carr as zc

    rng = np.random.default_rng(5)
    state = _state(rng, dt=sdt)
    em = _emulator(rng, state, dt=edt)
    res, ref = zc.enforce_conservative_precpd(state, em), E.enforce_conservative_precpd(state, em)
    for key in (E.CLOUD_P, E.QV_P, E.T_P, E.PRECIP):
        assert res[key].shape == ref[key].shape
        np.testing.assert_allclose(res[key], ref[key].astype(res[key].dtype), rtol=1e-12 if sdt == np.float64 else 5e-5,
                                   atol=0 if sdt == np.float64 else 1e-9, err_msg=key)
    assert np.all(res[E.PRECIP] >= 0)
    res, ref = zc.conservative_precip_simple(state, em), E.conservative_precip_simple(state, em)
    np.testing.assert_allclose(res[E.PRECIP], ref[E.PRECIP], rtol=1e-10 if sdt == np.float64 else 2e-3, atol=1e-12 if sdt == np.float64 else 1e-6)


def test_strict_scan_known_answer(device):
    """test_zhao_carr.py:49-70 through enforce_conservative_precpd: delp = g so that mass = mixing ratio."""
    from fv3net_amd.emulation import zhao_carr as zc

    c_to_p = np.array([[1.0], [-2.0], [3.0]])
    p_to_v = np.array([[4.0], [-1.0], [2.0]])
    z = np.zeros_like(c_to_p)
    state = {E.CLOUD_G: z, E.QV_G: z, E.T_G: z, E.DELP: np.full_like(z, E.GRAVITY)}
    res = zc.enforce_conservative_precpd(state, {E.CLOUD_P: -c_to_p, E.QV_P: p_to_v})
    np.testing.assert_allclose(-res[E.CLOUD_P], [[1.0], [0.0], [3.0]], atol=1e-15)
    np.testing.assert_allclose(res[E.QV_P], [[2.0], [0.0], [2.0]], atol=1e-15)
    np.testing.assert_array_equal(res[E.PRECIP], [0.0])


def test_squash_infer_and_class_zero(device):
    from fv3net_amd.emulation import zhao_carr as zc

    rng = np.random.default_rng(6)
    state = _state(rng)
    em = _emulator(rng, state)
    for fn, ckey, qkey in ((zc.squash_gscond, E.CLOUD_G, E.QV_G), (zc.squash_precpd, E.CLOUD_P, E.QV_P)):
        res = fn(state, em, 1e-4)
        c_ref, q_ref = E.squash(em[ckey], em[qkey], 1e-4)
        _close(res[ckey], c_ref.astype(np.float32), ckey)
        _close(res[qkey], q_ref, qkey)
    res, ref = zc.infer_gscond_cloud_from_conservation(state, em), E.infer_gscond_cloud_from_conservation(state, em)
    _close(res[E.CLOUD_G], ref[E.CLOUD_G], "inferred cloud")
    res, ref = zc.mask_zero_cloud_classifier_precpd(state, em), E.mask_zero_cloud_classifier_precpd(state, em)
    _close(res[E.CLOUD_P], ref[E.CLOUD_P].astype(np.float32), "class-zero precpd cloud")


def test_limit_net_condensation_known_answer(device):
    """test_zhao_carr.py:15-27 through enforce_conservative_gscond: cloud_out - cloud_in is the limited net."""
    from fv3net_amd.emulation import zhao_carr as zc

    qv = np.array([[1, 1, 1], [0, 0, 0]], dtype=np.float64)
    qc = np.array([[0, 0, 0], [1, 1, 0]], dtype=np.float64)
    net = np.array([[1.5, 0.5, 0], [-1.5, -0.5, 0]], dtype=np.float64)
    state = {E.CLOUD_IN: qc, E.QV_IN: qv, E.T_IN: np.full_like(qv, 280.0)}
    res = zc.enforce_conservative_gscond(state, {E.CLOUD_G: qc + net})
    np.testing.assert_array_equal(res[E.CLOUD_G] - qc, np.array([[1, 0.5, 0], [-1, -0.5, 0]]))


def test_range_and_level_masks_known_answers(device):
    """test_mask.py:7-47."""
    from fv3net_amd.emulation.masks import LevelMask, RangeMask, compose_masks

    mask = RangeMask("foo", min=0, max=1)
    assert mask({}, {"foo": 0.5}) == {"foo": 0.5} and mask({}, {"foo": 1.5}) == {"foo": 1.0} and mask({}, {"foo": -1.5}) == {"foo": 0}
    x = np.array([[-1.5, 0.5], [np.nan, 1.5]], dtype=np.float32)
    res = mask({}, {"foo": x})["foo"]
    np.testing.assert_array_equal(res, np.array([[0, 0.5], [np.nan, 1.0]], dtype=np.float32))
    assert res.dtype == np.float32
    assert compose_masks([])({}, {"a": 1}) == {"a": 1}
    ones = np.ones((4, 2))
    zeros = ones * 0
    for start, stop in [(2, 3), (2, 5), (None, 2), (None, None), (-1, None)]:
        res = LevelMask("foo", start, stop)(state={"foo": zeros}, emulator={"foo": ones.astype(np.float32)})
        ref = E.level_mask({"foo": zeros}, {"foo": ones}, "foo", start, stop)
        np.testing.assert_array_equal(res["foo"], ref["foo"])
        assert res["foo"].dtype == np.float64
    res = LevelMask("foo", 0, 2, fill_value=0.5)(state={}, emulator={"foo": ones})
    np.testing.assert_array_equal(0.5, res["foo"][:2])
    a = ones * 1.1
    res = LevelMask("foo", 0, 2, fill_value="a")(state={"a": a}, emulator={"foo": ones})
    np.testing.assert_array_equal(a[:2], res["foo"][:2])


def test_hook_applies_configured_masks(device, tmp_path):
    """ModelConfig composes the masks in the reference's order and the hook applies them after the
    network (here the identity 'model' of a config without a path, as the reference tests do)."""
    from fv3net_amd.emulation.config import ModelConfig

    rng = np.random.default_rng(8)
    state = _state(rng, n0=12, n1=40)
    em = _emulator(rng, state, dt=np.float64)
    full = {**state, **{k: v for k, v in em.items() if k not in state}}
    full[E.CLOUD_P] = em[E.CLOUD_P]
    cfg = ModelConfig.from_dict({"cloud_squash": 1e-4, "enforce_strict_precpd_conservative": True,
                                 "ranges": {E.PRECIP: {"min": 0.0}},
                                 "mask_emulator_levels": {E.T_P: {"start": 8, "stop": None}}})
    mask = cfg._build_mask()
    fortran = {**state, E.T_P: state[E.T_G] + 0.5}  # the Fortran scheme's own answer sits in the state
    emul = {k: full[k] for k in (E.CLOUD_P, E.QV_P, E.T_P, E.PRECIP)}
    res = mask(fortran, emul)
    ref = E.range_mask(emul, E.PRECIP, 0.0, None)
    c, q = E.squash(ref[E.CLOUD_P], ref[E.QV_P], 1e-4)
    ref = E.enforce_conservative_precpd(fortran, {**ref, E.CLOUD_P: c, E.QV_P: q})
    ref = E.level_mask(fortran, ref, E.T_P, 8, None)
    for key in (E.CLOUD_P, E.QV_P, E.T_P, E.PRECIP):
        np.testing.assert_allclose(res[key], ref[key], rtol=1e-12, err_msg=key)
    np.testing.assert_array_equal(res[E.T_P][8:], fortran[E.T_P][8:])
