"""The mappm oracle (oracle/mappm_oracle.c) pinned to the reference: its own known answers,
golden vectors produced by the reference's Fortran, and -- where oracle/_ref is present -- the
compiled reference itself, bit for bit."""
import os

import numpy as np
import pytest

from oracle import mappm_c

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mappm_reference.npz")


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def test_reference_known_answers():
    # external/vcm/tests/test_mappm.py:5-44
    p_in = np.asarray([0.0, 1.0, 2.0, 3.0, 4.0, 5.0])[None, :]
    f_in = np.asarray([0.0, 1.0, 2.0, 3.0, 4.0])[None, :]
    p_out = np.asarray([0.5, 1.2, 2.4, 2.8, 3.2, 4.5])[None, :]
    result = mappm_c.mappm(p_in, f_in, p_out, 1, 1)
    assert result.dtype == np.float32
    np.testing.assert_almost_equal(result, np.asarray([[0.35, 1.3, 2.1, 2.5, 3.35]], np.float32), decimal=5)

    p_in = np.asarray([1.0, 2.0, 3.0, 4.0, 5.0])[None, :]
    f_in = np.asarray([1.5, 2.5, 3.5, 4.5])[None, :]
    p_out = np.asarray([0.0, 2.5, 3.5, 4.5, 50.0])[None, :]
    result = mappm_c.mappm(p_in, f_in, p_out, 1, 1)
    np.testing.assert_almost_equal(result, np.asarray([[1.5, 3.0, 4.0, 4.502747]], np.float32), decimal=5)

    p_in = np.asarray([1.0, 2.0, 3.0, 2.0, 5.0])[None, :]
    f_in = np.full((1, 4), np.nan)
    result = mappm_c.mappm(p_in, f_in, p_out, 1, 1)
    assert np.all(np.isnan(result))


def test_golden_vectors_from_reference_fortran_bit_exact():
    z = np.load(GOLDEN)
    n = int(z["n_cases"])
    assert n >= 100 and {int(z[f"case{i}_ivkord"][1]) for i in range(n)} >= set(range(1, 18))
    for i in range(n):
        iv, kord = (int(v) for v in z[f"case{i}_ivkord"])
        got = mappm_c.mappm(z[f"case{i}_pe1"], z[f"case{i}_q1"], z[f"case{i}_pe2"], iv, kord)
        assert bits_equal(got, z[f"case{i}_q2"]), (i, iv, kord)


@pytest.mark.skipif(not mappm_c.have_reference(), reason="oracle/_ref is only built where /root/reference is mounted")
@pytest.mark.parametrize("iv", [-2, -1, 0, 1, 2])
def test_against_compiled_reference_bit_exact(iv):
    rng = np.random.default_rng(iv + 10)
    for kord in range(1, 8):
        ncol, km, kn = 512, 79, 79
        pe1 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, km)), 1)], 1)
        pe2 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, kn)), 1)], 1)
        q = rng.uniform(-1000, 1000, (ncol, km))
        q = np.abs(q) if iv == 0 else q
        assert bits_equal(mappm_c.mappm(pe1, q, pe2, iv, kord), mappm_c.reference_mappm(pe1, q, pe2, iv, kord)), kord


@pytest.mark.skipif(not mappm_c.have_reference(), reason="oracle/_ref is only built where /root/reference is mounted")
@pytest.mark.parametrize("iv", [-1, 0, 1, 2])
def test_cs_profile_against_compiled_reference_bit_exact(iv):
    """kord > 7 (cs_profile / cs_limiters, mappm.f90:132-611): schemes 8 .. 16 and the linear one above, on noise, smooth
    profiles, ties and NaNs, short and long columns."""
    rng = np.random.default_rng(iv + 50)
    for kord in range(8, 19):
        for km, kn, ncol in ((79, 79, 256), (4, 9, 64), (5, 5, 64), (7, 12, 64), (30, 41, 128)):
            pe1 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, km)), 1)], 1)
            pe2 = np.concatenate([np.full((ncol, 1), rng.choice([100.0, 300.0, 500.0])), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, kn)), 1)], 1)
            z = np.linspace(0, 1, km)[None, :]
            for kind in ("noise", "smooth", "ties", "nans"):
                if kind == "noise":
                    q = rng.uniform(-1000, 1000, (ncol, km))
                elif kind == "smooth":
                    q = 300 * np.sin(2 * np.pi * (z * rng.uniform(0.5, 3, (ncol, 1)) + rng.uniform(0, 1, (ncol, 1))))
                    q = q + rng.normal(0, 1, (ncol, km)) * (rng.random((ncol, km)) < 0.1)
                else:
                    q = np.round(rng.uniform(-3, 3, (ncol, km)))
                    if kind == "nans":
                        q[rng.random((ncol, km)) < 0.03] = np.nan
                if iv == 0:
                    q = np.abs(q) if kind != "smooth" else q + 250
                assert bits_equal(mappm_c.mappm(pe1, q, pe2, iv, kord), mappm_c.reference_mappm(pe1, q, pe2, iv, kord)), (kord, km, kind)


def test_unsupported_kord_and_short_columns():
    z = np.zeros((2, 8))
    with pytest.raises(ValueError):  # cs_profile with iv = -2 reads an array mappm never sets
        mappm_c.mappm(z, np.zeros((2, 7)), z, -2, 9)
    with pytest.raises(ValueError):
        mappm_c.mappm(np.zeros((2, 4)), np.zeros((2, 3)), np.zeros((2, 4)), 1, 1)


def test_interpolate_2d_oracle_matches_reference_goldens():
    """oracle/mappm_oracle.c:fv3_oracle_interpolate_2d against outputs of the reference's own
    interpolate_2d.f90 (tests/golden/interpolate_2d_reference.npz, generated by make_golden.py from the
    compiled Fortran) -- bit for bit -- and the reference's known answer (test_interpolate.py:120-133)."""
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "interpolate_2d_reference.npz"))
    for n in range(int(z["n_cases"])):
        got = mappm_c.interpolate_2d(z[f"case{n}_xp"], z[f"case{n}_x"], z[f"case{n}_y"])
        want = z[f"case{n}_out"]
        np.testing.assert_array_equal(got, want)
    x = np.arange(10.0).reshape(1, 10)
    got = mappm_c.interpolate_2d(np.arange(12.0).reshape(1, 12), x, x ** 2)
    np.testing.assert_array_equal(got[:, :10], x ** 2)
    assert np.isnan(got[:, -2:]).all()
    if mappm_c.have_reference():
        rng = np.random.default_rng(1)
        xx = np.cumsum(rng.uniform(0.1, 1, (30, 12)), axis=1)
        yy = rng.normal(0, 1, (30, 12))
        xp = rng.uniform(0, xx.max(), (30, 17))
        np.testing.assert_array_equal(mappm_c.interpolate_2d(xp, xx, yy), mappm_c.reference_interpolate_2d(xp, xx, yy))
