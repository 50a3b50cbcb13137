"""Parity of the HIP vertical kernels with the oracle.

mappm is single precision with the reference's exact operation order and no FMA contraction,
so the HIP result must be BIT-IDENTICAL to the C restatement (which tests/test_oracle_mappm.py
pins bit-for-bit to the reference's own Fortran).  pressure_at_interface is a sequential
cumulative sum in the array dtype: bit-identical to numpy.cumsum.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import coarsen_np as onp
from oracle import mappm_c

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(autouse=True)
def _exact_remap_by_default(monkeypatch):
    """The tests of this module that do not say otherwise check bit-exactness: FV3HIP_ARITH_EXACT."""
    from fv3net_amd import ops

    monkeypatch.setattr(ops, "MAPPM_ARITHMETIC", "exact")


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _bits_equal(a, b):
    """Bit-for-bit equality of float32 arrays; NaNs must sit in the same places but may differ
    in sign/payload (x86 SSE and gfx950 generate different default NaNs)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def _columns(rng, ncol, km, kn, ptop2=None):
    dp1 = rng.uniform(300, 1500, (ncol, km))
    pe1 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(dp1, 1)], 1)
    dp2 = rng.uniform(300, 1500, (ncol, kn))
    top = 300.0 if ptop2 is None else ptop2
    pe2 = np.concatenate([np.full((ncol, 1), top), 300 + np.cumsum(dp2, 1)], 1)
    q = rng.uniform(-1000, 1000, (ncol, km))
    return pe1.astype(np.float32), q.astype(np.float32), pe2.astype(np.float32)


def test_mappm_reference_known_answers(device):
    # external/vcm/tests/test_mappm.py:5-44
    from fv3net_amd import ops

    p_in = np.asarray([0.0, 1.0, 2.0, 3.0, 4.0, 5.0])[None, :]
    f_in = np.asarray([0.0, 1.0, 2.0, 3.0, 4.0])[None, :]
    p_out = np.asarray([0.5, 1.2, 2.4, 2.8, 3.2, 4.5])[None, :]
    res = ops.as_numpy(ops.mappm(_dev(p_in, device), _dev(f_in, device), _dev(p_out, device)))
    assert res.dtype == np.float32
    np.testing.assert_almost_equal(res, np.asarray([[0.35, 1.3, 2.1, 2.5, 3.35]], np.float32), decimal=5)

    p_in = np.asarray([1.0, 2.0, 3.0, 4.0, 5.0])[None, :]
    f_in = np.asarray([1.5, 2.5, 3.5, 4.5])[None, :]
    p_out = np.asarray([0.0, 2.5, 3.5, 4.5, 50.0])[None, :]
    res = ops.as_numpy(ops.mappm(_dev(p_in, device), _dev(f_in, device), _dev(p_out, device)))
    np.testing.assert_almost_equal(res, np.asarray([[1.5, 3.0, 4.0, 4.502747]], np.float32), decimal=5)

    p_in = np.asarray([1.0, 2.0, 3.0, 2.0, 5.0])[None, :]
    f_in = np.full((1, 4), np.nan)
    res = ops.as_numpy(ops.mappm(_dev(p_in, device), _dev(f_in, device), _dev(p_out, device)))
    assert np.all(np.isnan(res))


@pytest.mark.parametrize("km,kn,ncol", [(79, 79, 5000), (63, 63, 1000), (7, 7, 300), (4, 9, 100), (79, 40, 500), (20, 90, 333)])
@pytest.mark.parametrize("iv,kord", [(1, 1), (0, 1), (-1, 4), (2, 6), (0, 7), (-2, 7), (1, 3)])
def test_mappm_bit_exact_col_level(device, km, kn, ncol, iv, kord):
    from fv3net_amd import ops

    rng = np.random.default_rng(km * 100 + kn)
    pe1, q, pe2 = _columns(rng, ncol, km, kn, ptop2=rng.choice([100.0, 300.0, 500.0]))
    if iv == 0:
        q = np.abs(q)
    ref = mappm_c.mappm(pe1, q, pe2, iv, kord)
    res = ops.as_numpy(ops.mappm(_dev(pe1, device), _dev(q, device), _dev(pe2, device), iv=iv, kord=kord))
    assert _bits_equal(res, ref), np.nanmax(np.abs(res - ref))


def _cs_columns(rng, ncol, km, kn, kind):
    """Columns for the cs_profile schemes: iid noise (every level an extremum), smooth profiles with a few kinks (the
    monotonic branches), small integers (ties and exact zeros), the same with NaNs."""
    pe1, q, pe2 = _columns(rng, ncol, km, kn, ptop2=rng.choice([100.0, 300.0, 500.0]))
    if kind == "smooth":
        z = np.linspace(0, 1, km)[None, :]
        q = 300 * np.sin(2 * np.pi * (z * rng.uniform(0.5, 3, (ncol, 1)) + rng.uniform(0, 1, (ncol, 1)))) + rng.normal(0, 1, (ncol, km)) * (rng.random((ncol, km)) < 0.1)
    elif kind in ("ties", "nans"):
        q = np.round(rng.uniform(-3, 3, (ncol, km)))
        if kind == "nans":
            q[rng.random((ncol, km)) < 0.03] = np.nan
    return pe1, q, pe2


@pytest.mark.parametrize("kord", list(range(8, 18)))
@pytest.mark.parametrize("iv", [-1, 0, 1, 2])
def test_mappm_cs_profile_schemes_bit_exact(device, iv, kord):
    """kord > 7: cs_profile / cs_limiters (mappm.f90:132-611; VERDICT r02 missing #4), every scheme (8 .. 16, and the
    perfectly linear one above 16) for every iv mappm can pass, bit for bit against the C restatement that
    tests/test_oracle_mappm.py pins to the compiled Fortran; both layouts, several fields per call."""
    from fv3net_amd import ops

    rng = np.random.default_rng(100 * kord + iv)
    for kind, (km, kn, ncol) in (("noise", (79, 79, 1500)), ("smooth", (79, 60, 1500)), ("ties", (30, 30, 800)), ("nans", (30, 41, 800)),
                                 ("smooth", (4, 9, 100)), ("noise", (5, 5, 100)), ("smooth", (7, 12, 100))):
        pe1, q, pe2 = _cs_columns(rng, ncol, km, kn, kind)
        if iv == 0:
            q = np.abs(q) if kind != "smooth" else q + 250  # (mostly positive, some columns dip below zero)
        ref = mappm_c.mappm(pe1, q, pe2, iv, kord)
        res = ops.as_numpy(ops.mappm(_dev(pe1, device), _dev(q, device), _dev(pe2, device), iv=iv, kord=kord))
        assert _bits_equal(res, ref), (kind, km, kn, np.nanmax(np.abs(res - ref)))
    nt, km, ny, nx = 2, 20, 8, 16
    pe1, q, pe2 = _cs_columns(rng, nt * ny * nx, km, km, "smooth")
    q2 = q[::-1].copy()

    def native(a):
        return np.ascontiguousarray(np.moveaxis(a.reshape(nt, ny, nx, -1), -1, 1))

    res = ops.mappm_multi(_dev(native(pe1).astype(np.float64), device), [_dev(native(f).astype(np.float64), device) for f in (q, q2)],
                          _dev(native(pe2).astype(np.float64), device), iv=iv, kord=kord, z_axis=1)
    for f, r in zip((q, q2), res):
        assert _bits_equal(np.moveaxis(ops.as_numpy(r), 1, -1).reshape(-1, km), mappm_c.mappm(pe1, f, pe2, iv, kord))


def test_mappm_level_col_layout_and_f64_inputs(device):
    from fv3net_amd import ops

    rng = np.random.default_rng(7)
    nt, km, ny, nx = 2, 79, 12, 16
    pe1, q, pe2 = _columns(rng, nt * ny * nx, km, km)
    ref = mappm_c.mappm(pe1, q, pe2)

    def native(a):  # [ncol, lev] -> [tile, lev, y, x]
        return np.ascontiguousarray(np.moveaxis(a.reshape(nt, ny, nx, -1), -1, 1))

    res = ops.as_numpy(ops.mappm(_dev(native(pe1), device), _dev(native(q), device), _dev(native(pe2), device), z_axis=1))
    assert res.shape == (nt, km, ny, nx)
    assert _bits_equal(np.moveaxis(res, 1, -1).reshape(-1, km), ref)
    # float64 inputs are rounded to float32 on load, like f2py's argument conversion
    res64 = ops.as_numpy(ops.mappm(_dev(native(pe1).astype(np.float64), device), _dev(native(q).astype(np.float64), device),
                                   _dev(native(pe2).astype(np.float64), device), z_axis=1))
    assert _bits_equal(res64, res)


def test_mappm_edge_cases_bit_exact(device):
    """Ties between interfaces, zero-thickness target layers, targets outside the source
    column, NaNs in the field, flat fields (dm == 0 branch)."""
    from fv3net_amd import ops

    rng = np.random.default_rng(11)
    ncol, km, kn = 2000, 30, 30
    for trial in range(4):
        dp1 = rng.integers(1, 4, (ncol, km)).astype(float)
        pe1 = np.concatenate([np.full((ncol, 1), 3.0), 3 + np.cumsum(dp1, 1)], 1)
        dp2 = rng.integers(0, 4, (ncol, kn)).astype(float)
        pe2 = np.concatenate([np.full((ncol, 1), float(rng.integers(0, 6))), 3 + np.cumsum(dp2, 1)], 1)
        q = rng.uniform(-10, 10, (ncol, km))
        if trial >= 1:
            q[rng.random((ncol, km)) < 0.05] = np.nan
        if trial >= 2:
            q = np.round(q)
        for iv, kord in [(1, 1), (0, 7), (-1, 4)]:
            ref = mappm_c.mappm(pe1, q, pe2, iv, kord)
            res = ops.as_numpy(ops.mappm(_dev(pe1, device), _dev(q, device), _dev(pe2, device), iv=iv, kord=kord))
            assert np.array_equal(np.isnan(res), np.isnan(ref))
            assert _bits_equal(res, ref)


def test_mappm_errors(device):
    from fv3net_amd import ops
    from fv3net_amd._lib import Fv3HipError

    z = torch.zeros(4, 6, device=device)
    with pytest.raises(ValueError, match="one shorter"):
        ops.mappm(z, z, z)
    with pytest.raises(ValueError, match="All dimensions except vertical"):
        ops.mappm(torch.zeros(4, 7, device=device), torch.zeros(5, 6, device=device), torch.zeros(4, 7, device=device))
    with pytest.raises(Fv3HipError, match="cs_profile"):  # (mappm never sets the qs that iv = -2 makes cs_profile read)
        ops.mappm(torch.zeros(4, 7, device=device), z, torch.zeros(4, 7, device=device), iv=-2, kord=9)


def test_mappm_empty(device):
    from fv3net_amd import ops

    res = ops.mappm(torch.zeros(0, 7, device=device), torch.zeros(0, 6, device=device), torch.zeros(0, 7, device=device))
    assert tuple(res.shape) == (0, 6)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("z_axis", [0, 1, -1])
def test_pressure_at_interface(device, dt, z_axis):
    from fv3net_amd import ops

    rng = np.random.default_rng(3)
    delp = rng.uniform(300, 1500, (4, 9, 6)).astype(dt)
    res = ops.as_numpy(ops.pressure_at_interface(_dev(delp, device), 300.0, z_axis))
    ref = onp.pressure_at_interface(delp, 300.0, z_axis)
    assert res.dtype == ref.dtype
    np.testing.assert_array_equal(res, ref)


def test_mask_weights(device):
    from fv3net_amd import ops

    # external/vcm/tests/test_regridz.py:113-147 (extrapolate=False case)
    weights = np.array([[1.0, 1.0]])                                    # [y=1, x=2]
    phalf_c = np.array([[[0.0, 0.0]], [[1.0, 1.0]], [[2.0, 2.0]], [[3.0, 3.0]]])  # [z+1, y, x]
    phalf_f = np.array([[[0.0, 0.0]], [[1.0, 1.0]], [[2.0, 2.5]], [[2.75, 3.5]]])
    res = ops.as_numpy(ops.mask_weights(_dev(weights, device), _dev(phalf_c, device), _dev(phalf_f, device), z_axis=0))
    ref = onp.mask_weights(weights, phalf_c, phalf_f, 0)
    np.testing.assert_array_equal(res, ref)
    np.testing.assert_array_equal(res[:, 0, :], np.array([[1.0, 1.0], [1.0, 1.0], [0.0, 1.0]]))


def test_full_size_c384_mappm(device):
    """C384 x 79 at full size (884 736 columns, native [tile, z, y, x] layout): remapping onto the
    same interfaces returns the field, a constant field stays constant, and every 16th column
    is bit-identical to the oracle.  (Column-integral conservation is NOT a property of the
    reference: a target layer whose top edge is at or above the source top is assigned
    q1(1) outright, mappm.f90:62-64.)"""
    from fv3net_amd import ops

    g = torch.Generator(device=device).manual_seed(0)
    nt, nz, n = 6, 79, 384
    delp = torch.rand((nt, nz, n, n), device=device, generator=g) * 1200 + 300
    delp2 = torch.rand((nt, nz, n, n), device=device, generator=g) * 1200 + 300
    q = torch.rand((nt, nz, n, n), device=device, generator=g) * 2000 - 1000
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2 = ops.pressure_at_interface(delp2, 300.0, 1)
    same = ops.mappm(pe1, q, pe1, z_axis=1)
    assert torch.allclose(same, q, rtol=1e-5, atol=1e-3)
    const = ops.mappm(pe1, torch.full_like(q, 7.5), pe2, z_axis=1)
    assert torch.allclose(const, torch.full_like(const, 7.5), rtol=1e-6, atol=0)
    r = ops.mappm(pe1, q, pe2, z_axis=1)

    def cols(t):  # [tile, lev, y, x] -> every 16th column as [ncol, lev]
        a = t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])[::16]
        return a.cpu().numpy()

    ref = mappm_c.mappm(cols(pe1), cols(q), cols(pe2))
    assert _bits_equal(cols(r), ref)


def test_mappm_ill_formed_columns_take_the_sequential_path(device):
    """Columns with non-monotone or NaN pressures leave the merge sweep and are redone by the
    sequential routine; the rest of the batch is unaffected.  Everything stays bit-identical to
    the oracle (whose semantics for such columns are the Fortran's, with per-column state)."""
    from fv3net_amd import ops

    rng = np.random.default_rng(21)
    ncol, km, kn = 3000, 40, 40
    pe1, q, pe2 = _columns(rng, ncol, km, kn)
    bad = rng.choice(ncol, 300, replace=False)
    for i, c in enumerate(bad):
        kind = i % 5
        if kind == 0:      # a source interface out of order
            j = rng.integers(1, km)
            pe1[c, j], pe1[c, j + 1] = pe1[c, j + 1], pe1[c, j]
        elif kind == 1:    # a target interface out of order
            j = rng.integers(1, kn)
            pe2[c, j], pe2[c, j + 1] = pe2[c, j + 1], pe2[c, j]
        elif kind == 2:    # NaN source pressure
            pe1[c, rng.integers(0, km + 1)] = np.nan
        elif kind == 3:    # NaN target pressure
            pe2[c, rng.integers(0, kn + 1)] = np.nan
        else:              # decreasing target column
            pe2[c] = pe2[c, ::-1].copy()
    for iv, kord in [(1, 1), (0, 4), (-1, 6)]:
        qq = np.abs(q) if iv == 0 else q
        ref = mappm_c.mappm(pe1, qq, pe2, iv, kord)
        res = ops.as_numpy(ops.mappm(_dev(pe1, device), _dev(qq, device), _dev(pe2, device), iv=iv, kord=kord))
        good = np.setdiff1d(np.arange(ncol), bad)
        assert _bits_equal(res[good], ref[good])
        assert _bits_equal(res[bad], ref[bad])
    # every column ill-formed: more work-list entries than fallback threads in a launch is fine
    pe1_all = pe1[:, ::-1].copy()
    ref = mappm_c.mappm(pe1_all, q, pe2, 1, 1)
    res = ops.as_numpy(ops.mappm(_dev(pe1_all, device), _dev(q, device), _dev(pe2, device)))
    assert _bits_equal(res, ref)


@pytest.mark.parametrize("n_fields", [1, 2, 3, 4, 5, 9])
@pytest.mark.parametrize("iv,kord", [(1, 1), (0, 4), (-1, 6), (2, 3), (0, 7)])
def test_mappm_multi_field_bit_exact(device, n_fields, iv, kord):
    """Fields that share their pressures go through one sweep (four at a time): every field is
    bit-identical to the oracle, i.e. to a single-field call -- ill-formed columns, NaNs in one
    field only, ties and zero-thickness layers included."""
    from fv3net_amd import ops

    rng = np.random.default_rng(100 * n_fields + kord)
    ncol, km, kn = 1500, 33, 29
    pe1, _, pe2 = _columns(rng, ncol, km, kn, ptop2=rng.choice([100.0, 300.0, 500.0]))
    # integer-valued pressures in part of the batch: ties between interfaces, zero-thickness targets
    t = slice(0, 300)
    pe1[t] = np.concatenate([np.full((300, 1), 3.0), 3 + np.cumsum(rng.integers(1, 4, (300, km)), 1)], 1)
    pe2[t] = np.concatenate([np.full((300, 1), 2.0), 3 + np.cumsum(rng.integers(0, 4, (300, kn)), 1)], 1)
    bad = rng.choice(np.arange(300, ncol), 60, replace=False)
    for i, c in enumerate(bad):
        if i % 3 == 0:
            pe1[c, 5], pe1[c, 6] = pe1[c, 6], pe1[c, 5]
        elif i % 3 == 1:
            pe2[c, rng.integers(0, kn + 1)] = np.nan
        else:
            pe2[c] = pe2[c, ::-1].copy()
    fields = []
    for f in range(n_fields):
        q = rng.uniform(-1000, 1000, (ncol, km)).astype(np.float32) * np.float32(10.0 ** (f - 2))
        if f % 2:
            q[rng.random((ncol, km)) < 0.02] = np.nan
        if f == 2:
            q = np.round(q)  # flat stretches: the dm == 0 branch of the limiter
        fields.append(np.abs(q) if iv == 0 else q)
    res = ops.mappm_multi(_dev(pe1, device), [_dev(q, device) for q in fields], _dev(pe2, device), iv=iv, kord=kord)
    assert len(res) == n_fields
    for f, q in enumerate(fields):
        ref = mappm_c.mappm(pe1, q, pe2, iv, kord)
        assert _bits_equal(ops.as_numpy(res[f]), ref), (f, np.nanmax(np.abs(ops.as_numpy(res[f]) - ref)))


def test_mappm_multi_level_col_layout_f64_and_errors(device):
    from fv3net_amd import ops

    rng = np.random.default_rng(8)
    nt, km, ny, nx = 2, 79, 12, 16
    pe1, _, pe2 = _columns(rng, nt * ny * nx, km, km)
    qs = [rng.uniform(-1, 1, (nt * ny * nx, km)).astype(np.float32) for _ in range(6)]

    def native(a):  # [ncol, lev] -> [tile, lev, y, x]
        return np.ascontiguousarray(np.moveaxis(a.reshape(nt, ny, nx, -1), -1, 1))

    for cast in (np.float32, np.float64):
        res = ops.mappm_multi(_dev(native(pe1).astype(cast), device), [_dev(native(q).astype(cast), device) for q in qs],
                              _dev(native(pe2).astype(cast), device), z_axis=1)
        for q, r in zip(qs, res):
            assert r.shape == (nt, km, ny, nx)
            assert _bits_equal(np.moveaxis(ops.as_numpy(r), 1, -1).reshape(-1, km), mappm_c.mappm(pe1, q, pe2))
    assert ops.mappm_multi(_dev(pe1, device), [], _dev(pe2, device)) == []
    with pytest.raises(ValueError, match="same size"):
        ops.mappm_multi(_dev(pe1, device), [_dev(qs[0], device), _dev(qs[1][:10], device)], _dev(pe2, device))
    from fv3net_amd._lib import Fv3HipError

    with pytest.raises(Fv3HipError, match="cs_profile"):
        ops.mappm_multi(_dev(pe1, device), [_dev(qs[0], device), _dev(qs[1], device)], _dev(pe2, device), iv=-2, kord=9)


@pytest.mark.parametrize("layout", ["col_level", "level_col"])
def test_mappm_more_columns_than_one_launch_chunk(device, layout):
    """The launcher cuts the columns into chunks of 2^20 (the work list and the fallback workspace are per
    chunk): 1.3 M columns -- ill-formed ones on both sides of the cut -- single-field and multi-field, against
    the oracle on the columns around the cut and a tiled copy property elsewhere."""
    from fv3net_amd import ops

    rng = np.random.default_rng(31)
    base, km, kn = 4096, 12, 10
    reps = 320                                # 1 310 720 columns
    pe1, q, pe2 = _columns(rng, base, km, kn)
    pe1[7, 3], pe1[7, 4] = pe1[7, 4], pe1[7, 3]   # an ill-formed column in every copy of the base block
    pe2[100, 5] = np.nan
    q2 = rng.uniform(-5, 5, (base, km)).astype(np.float32)
    ref, ref2 = mappm_c.mappm(pe1, q, pe2), mappm_c.mappm(pe1, q2, pe2)

    def tiled(a):
        t = _dev(np.tile(a, (reps, 1)), device)
        return t.t().contiguous() if layout == "level_col" else t   # [level, column]: z_axis 0

    z_axis = 0 if layout == "level_col" else -1
    P1, Q1, Q2, P2 = tiled(pe1), tiled(q), tiled(q2), tiled(pe2)
    single = ops.mappm(P1, Q1, P2, z_axis=z_axis)
    multi = ops.mappm_multi(P1, [Q1, Q2], P2, z_axis=z_axis)
    for got, want in ((single, ref), (multi[0], ref), (multi[1], ref2)):
        g = ops.as_numpy(got.t().contiguous() if layout == "level_col" else got).reshape(reps, base, kn)
        for r in (0, 255, 256, reps - 1):      # 256 * 4096 = 2^20: the copies on either side of the chunk cut, and the ends
            assert _bits_equal(g[r], want), r
        assert _bits_equal(g, np.broadcast_to(want, g.shape).copy())


@pytest.mark.parametrize("dt_np", [np.float32, np.float64])
def test_humidity_limiters(device, dt_np):
    """vcm.non_negative_sphum(_mse_conserving) on the device (non_negative_sphum.py:6-45) against the
    oracle (pinned by test_non_negative_sphum.py's known answers) and those answers directly."""
    from fv3net_amd import thermo
    from fv3net_amd.xr_compat import DataArray

    rng = np.random.default_rng(12)
    shape = (79, 24, 24)
    sphum = DataArray((10 ** rng.uniform(-7, -2, shape)).astype(dt_np), dims=["z", "y", "x"])
    dq2 = DataArray(rng.normal(0, 2e-6, shape).astype(dt_np), dims=["z", "y", "x"])
    dq1 = DataArray(rng.normal(0, 1e-4, shape).astype(dt_np), dims=["z", "y", "x"])
    dq2.values[0, 0, 0] = 0.0  # 0 / 0 in the ratio, unused because the humidity stays non-negative
    rtol = 1e-6 if dt_np == np.float32 else 1e-13
    q1, q2 = thermo.non_negative_sphum(sphum, dq1, dq2, 900.0)
    r1, r2 = onp.non_negative_sphum(sphum.values, dq1.values, dq2.values, dt_np(900.0))
    assert q1.dims == ("z", "y", "x") and q1.values.dtype == dt_np
    np.testing.assert_allclose(q1.values, r1, rtol=rtol)
    np.testing.assert_allclose(q2.values, r2, rtol=rtol)
    # the limited humidity is zero up to the rounding of the ratio (a few ulp of sphum)
    assert np.all(sphum.values + q2.values * 900.0 >= -(4e-7 if dt_np == np.float32 else 1e-15) * sphum.values)
    q2m, q1m = thermo.non_negative_sphum_mse_conserving(sphum, dq2, 900.0, q1=dq1)
    r2m, r1m = onp.non_negative_sphum_mse_conserving(sphum.values, dq2.values, dt_np(900.0), dq1.values)
    np.testing.assert_allclose(q2m.values, r2m, rtol=rtol)
    np.testing.assert_allclose(q1m.values, r1m, rtol=1e-4 if dt_np == np.float32 else 1e-12, atol=1e-9)
    q2_only, none = thermo.non_negative_sphum_mse_conserving(sphum, dq2, 900.0)
    assert none is None
    np.testing.assert_array_equal(q2_only.values, q2m.values)
    # test_non_negative_sphum.py:49-58
    lim = thermo.update_moisture_tendency_to_ensure_non_negative_humidity(DataArray(np.array([1.0, 2.0]), dims=["x"]),
                                                                          DataArray(np.array([-3.0, -1.0]), dims=["x"]), 1.0)
    np.testing.assert_array_equal(lim.values, [-1.0, -1.0])


@pytest.mark.parametrize("layout", ["col_level", "level_col"])
def test_interpolate_2d_bit_exact(device, layout):
    """The HIP interpolate_2d against the goldens produced by the reference's compiled Fortran and against
    the C oracle, bit for bit (NaN fill included), in both memory layouts."""
    import os

    from fv3net_amd import ops
    from oracle import mappm_c

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "interpolate_2d_reference.npz"))
    for n in range(int(z["n_cases"])):
        xp, x, y, want = (z[f"case{n}_{k}"] for k in ("xp", "x", "y", "out"))
        if layout == "col_level":
            got = ops.as_numpy(ops.interpolate_2d(_dev(xp, device), _dev(x, device), _dev(y, device), z_axis=-1))
        else:  # [level, column]: the interpolated axis first
            got = ops.as_numpy(ops.interpolate_2d(_dev(xp.T.copy(), device), _dev(x.T.copy(), device),
                                                  _dev(y.T.copy(), device), z_axis=0)).T
        np.testing.assert_array_equal(got, want)
    rng = np.random.default_rng(3)
    x = np.cumsum(rng.uniform(0.1, 1, (6, 79, 24, 24)), axis=1)
    y = rng.normal(0, 1, x.shape)
    xp = rng.uniform(0, x.max(), (6, 31, 24, 24))
    got = ops.as_numpy(ops.interpolate_2d(_dev(xp, device), _dev(x, device), _dev(y, device), z_axis=1))
    cols = lambda a: np.moveaxis(a, 1, -1).reshape(-1, a.shape[1])
    want = mappm_c.interpolate_2d(cols(xp), cols(x), cols(y)).reshape(6, 24, 24, 31)
    np.testing.assert_array_equal(got, np.moveaxis(want, -1, 1))


def test_interpolate_1d_reference_known_answer(device):
    """external/vcm/tests/test_interpolate.py:99-106 through the drop-in API."""
    from fv3net_amd import mappm
    from fv3net_amd.interpolate import interpolate_1d
    from fv3net_amd.xr_compat import DataArray

    xp = DataArray(np.array([[0.25, 0.5, 1.0], [0.25, 0.5, 1.0]]), dims=["x", "y_new"])
    inp = DataArray(np.array([[0, 1], [2, 3]]), dims=["x", "y"])
    x = DataArray(np.array([[0, 1], [0, 1]]), dims=["x", "y"])
    ans = interpolate_1d(xp, x, inp)
    assert ans.dims == ("x", "y_new")
    np.testing.assert_allclose(ans.values, [[0.25, 0.5, 1.0], [2.25, 2.50, 3.0]])
    xs = np.arange(10).reshape(1, 10)
    res = mappm.interpolate_2d(np.arange(12).reshape(1, 12), xs, xs ** 2, fill_value=np.nan)  # test_interpolate.py:120-133
    np.testing.assert_array_equal(res[:, :10], xs ** 2)
    assert np.isnan(res[:, -2:]).all() and res.dtype == np.float64


def test_interpolate_1d_constant_levels_and_pressure_levels_known_answers(device):
    """The 1-D-levels branch (metpy in the reference): external/vcm/tests/test_interpolate.py:60-96 and :135-147."""
    from fv3net_amd.interpolate import PRESSURE_GRID, interpolate_1d, interpolate_to_pressure_levels
    from fv3net_amd.xr_compat import DataArray, Dataset

    var = DataArray(np.array([[1.0, 2.0, 3.0], [-1.0, -2.0, -3.0]]), dims=["x", "pfull"])
    pressure = DataArray(np.array([[0.0, 1, 2], [0, 2, 4]]), dims=["x", "pfull"])
    out_p = DataArray(np.array([0.5, 2]), dims=["pressure_uniform"])
    got = interpolate_1d(out_p, pressure, var, "pfull")
    assert got.dims == ("x", "pressure_uniform")
    np.testing.assert_allclose(got.values, [[1.5, 3.0], [-1.25, -2.0]])
    np.testing.assert_array_equal(got.coords["pressure_uniform"], [0.5, 2])
    ds = interpolate_1d(out_p, pressure, Dataset({"interp_var": var, "pressure": pressure}), dim="pfull")
    np.testing.assert_allclose(ds["interp_var"].values, got.values)
    with pytest.raises(ValueError, match="dim argument"):
        interpolate_1d(out_p, pressure, var)
    # model top at 300 Pa, two 100 Pa layers: 350 Pa lies between the two midpoints -> no NaN
    out = interpolate_to_pressure_levels(DataArray(np.array([2.0, 1.0]), dims=["z"]), DataArray(np.array([100.0, 100.0]), dims=["z"]),
                                         levels=DataArray(np.array([350.0]), dims=["pressure"]), dim="z")
    assert out.dims == ("pressure",) and not np.isnan(out.values).any() and 1.0 < float(out.values[0]) < 2.0
    # the default grid on a [z, y, x] field: level axis in place, NaN below the surface and above the top layer's midpoint
    rng = np.random.default_rng(0)
    delp = DataArray(rng.uniform(800, 1400, (79, 6, 5)), dims=["pfull", "y", "x"])
    t = DataArray(rng.uniform(200, 300, (79, 6, 5)), dims=["pfull", "y", "x"])
    out = interpolate_to_pressure_levels(t, delp)
    assert out.dims == ("pressure", "y", "x") and out.shape == (35, 6, 5)
    mid = onp.pressure_at_midpoint_log(delp.values, 300.0, 0)
    want = np.full((35, 6, 5), np.nan)
    for j in range(6):
        for i in range(5):
            want[:, j, i] = np.interp(PRESSURE_GRID.values, mid[:, j, i], t.values[:, j, i], left=np.nan, right=np.nan)
    np.testing.assert_allclose(out.values, want, rtol=1e-12, equal_nan=True)
    # a field with a leading dim the pressure thickness does not have (time on the field, not on delp): x is broadcast,
    # as xr.apply_ufunc does in the reference (interpolate.py:165-179) -- in a Dataset too
    t2 = DataArray(np.stack([t.values, 2 * t.values]), dims=["time", "pfull", "y", "x"])
    out2 = interpolate_to_pressure_levels(t2, delp)
    assert out2.dims == ("time", "pressure", "y", "x")
    np.testing.assert_allclose(out2.values[0], want, rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(out2.values[1], 2 * want, rtol=1e-12, equal_nan=True)
    ds2 = interpolate_to_pressure_levels(Dataset({"t2": t2, "t": t}), delp)
    np.testing.assert_allclose(ds2["t2"].values[1], 2 * want, rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ds2["t"].values, want, rtol=1e-12, equal_nan=True)


def _native(a, nt, ny, nx):  # [ncol, lev] -> [tile, lev, y, x]
    return np.ascontiguousarray(np.moveaxis(a.reshape(nt, ny, nx, -1), -1, 1))


def _sweep_case(rng, nt, ny, nx, km, kn, n_fields, iv, ties=True, ill_formed=True):
    """Columns for the sweep kernel (native layout, n_inner = ny * nx a multiple of 64): random pressures, a block of
    integer-valued ones (ties between interfaces, zero-thickness targets, targets above the old top and below the old
    surface), ill-formed columns, NaNs and flat stretches in some fields."""
    ncol = nt * ny * nx
    pe1, _, pe2 = _columns(rng, ncol, km, kn, ptop2=rng.choice([100.0, 300.0, 500.0]))
    if ties:
        n = ncol // 5
        pe1[:n] = np.concatenate([np.full((n, 1), 3.0), 3 + np.cumsum(rng.integers(1, 4, (n, km)), 1)], 1)
        pe2[:n] = np.concatenate([np.full((n, 1), 2.0), 3 + np.cumsum(rng.integers(0, 4, (n, kn)), 1)], 1)
    bad = np.zeros(0, int)
    if ill_formed:
        bad = rng.choice(np.arange(ncol // 5, ncol), ncol // 25, replace=False)
        for i, c in enumerate(bad):
            if i % 4 == 0:
                pe1[c, 5], pe1[c, 6] = pe1[c, 6], pe1[c, 5]
            elif i % 4 == 1:
                pe2[c, rng.integers(0, kn + 1)] = np.nan
            elif i % 4 == 2:
                pe1[c, rng.integers(0, km + 1)] = np.nan
            else:
                pe2[c] = pe2[c, ::-1].copy()
    fields = []
    for f in range(n_fields):
        q = rng.uniform(-1000, 1000, (ncol, km)).astype(np.float32) * np.float32(10.0 ** (f - 2))
        if f % 2:
            q[rng.random((ncol, km)) < 0.02] = np.nan
        if f == 2:
            q = np.round(q)  # flat stretches: the dm == 0 branch of the limiter
        fields.append(np.abs(q) if iv == 0 else q)
    return pe1, pe2, fields, bad


@pytest.mark.parametrize("km,kn", [(79, 79), (33, 29), (20, 45), (8, 3)])
@pytest.mark.parametrize("n_fields,iv,kord,cast", [(1, 1, 1, np.float32), (1, 0, 2, np.float64), (2, -1, 3, np.float32),
                                                   (3, 2, 1, np.float32), (4, 1, 1, np.float64), (5, -2, 2, np.float32),
                                                   (9, 1, 1, np.float32)])
def test_mappm_sweep_kernel_exact_mode_is_bit_identical(device, km, kn, n_fields, iv, kord, cast):
    """The sweep kernel (csrc/remap.hip: native [tile, level, y, x] layout, kord <= 3, whole waves per tile plane) in
    FV3HIP_ARITH_EXACT: every field bit-identical to the oracle -- hence to the compiled reference Fortran
    (mappm.f90:10-126, 614-931) -- ties, zero-thickness layers, NaNs and ill-formed columns (worklist) included."""
    from fv3net_amd import ops

    rng = np.random.default_rng(1000 * km + 10 * n_fields + kord)
    nt, ny, nx = 2, 8, 16  # n_inner = 128
    pe1, pe2, fields, _ = _sweep_case(rng, nt, ny, nx, km, kn, n_fields, iv)
    nat = lambda a: _dev(_native(a, nt, ny, nx).astype(cast), device)
    if n_fields == 1:
        res = [ops.mappm(nat(pe1), nat(fields[0]), nat(pe2), iv=iv, kord=kord, z_axis=1, arith="exact")]
    else:
        res = ops.mappm_multi(nat(pe1), [nat(q) for q in fields], nat(pe2), iv=iv, kord=kord, z_axis=1, arith="exact")
    for f, q in enumerate(fields):
        ref = mappm_c.mappm(pe1, q, pe2, iv, kord)
        got = np.moveaxis(ops.as_numpy(res[f]), 1, -1).reshape(-1, kn)
        assert _bits_equal(got, ref), (f, np.nanmax(np.abs(got - ref)))


@pytest.mark.parametrize("km,kn", [(79, 79), (33, 29), (20, 45)])
@pytest.mark.parametrize("n_fields,iv,kord,cast", [(1, 1, 1, np.float32), (2, 0, 2, np.float64), (4, 1, 1, np.float32),
                                                   (7, -1, 3, np.float32)])
def test_mappm_sweep_kernel_fast_mode_within_tolerance(device, km, kn, n_fields, iv, kord, cast):
    """FV3HIP_ARITH_FAST (reciprocal-multiply, shared reciprocals): |fast - reference| <= 1e-5 x the column's value range
    on every level (the north star's 1e-5 relative; a result near zero has no relative accuracy in the reference
    either), NaNs in the same places, ill-formed columns (redone by the sequential routine) bit-identical."""
    from fv3net_amd import ops

    rng = np.random.default_rng(77 * km + n_fields)
    nt, ny, nx = 3, 16, 16  # n_inner = 256
    pe1, pe2, fields, bad = _sweep_case(rng, nt, ny, nx, km, kn, n_fields, iv)
    nat = lambda a: _dev(_native(a, nt, ny, nx).astype(cast), device)
    res = ops.mappm_multi(nat(pe1), [nat(q) for q in fields], nat(pe2), iv=iv, kord=kord, z_axis=1, arith="fast")
    worst, outliers = 0.0, 0
    for f, q in enumerate(fields):
        ref = mappm_c.mappm(pe1, q, pe2, iv, kord)
        got = np.moveaxis(ops.as_numpy(res[f]), 1, -1).reshape(-1, kn)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f
        assert _bits_equal(got[bad], ref[bad]), f
        scale = np.nanmax(np.abs(np.where(np.isfinite(q), q, np.nan)), axis=1, keepdims=True)
        err = np.abs(got - ref) / scale
        worst = max(worst, float(np.nanmax(err)))
        # Every level within 1e-5 of the column's scale (measured: <= 1.2e-6).  The reference's limiter is discontinuous
        # where dm == 0 exactly (mappm.f90:876-880); round 2's reciprocal arithmetic flipped that branch on 1 in 5e3 of
        # the integer-valued levels here, round 3's formulation on none -- were one to appear, what must hold is
        # test_fast_mode_outliers_stay_within_the_source_layers_bounds, not this line.
        assert np.nanmax(err) <= 1e-5, (f, float(np.nanmax(err)))
        assert np.nanmedian(err) <= 1e-6, f
        outliers += int(np.nansum(err > 1e-5))
    assert worst > 0  # (this really was the other arithmetic)
    print(f"fast mode: worst |fast - reference| / column scale {worst:.2e}, levels beyond 1e-5: {outliers}")


def test_mappm_fast_mode_full_size_c384(device):
    """BASELINE configs[2] at full size: 884 736 columns, 79 levels, the pipeline's own target grid; fast against exact
    on the device (exact is pinned to the oracle above), every 16th column of exact against the oracle."""
    from fv3net_amd import ops

    g = torch.Generator(device=device).manual_seed(5)
    n, nz = 384, 79
    delp = torch.rand((6, nz, n, n), device=device, generator=g) * 1200 + 300
    area = torch.rand((6, n, n), device=device, generator=g) * 0.5 + 0.5
    qs = [torch.rand((6, nz, n, n), device=device, generator=g) * 2000 - 1000 for _ in range(4)]
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2 = ops.pressure_at_interface(ops.block_upsample(ops.weighted_block_average(delp, area, 8), 8), 300.0, 1)
    exact = ops.mappm_multi(pe1, qs, pe2, z_axis=1, arith="exact")
    fast = ops.mappm_multi(pe1, qs, pe2, z_axis=1, arith="fast")
    single = ops.mappm(pe1, qs[0], pe2, z_axis=1, arith="fast")
    assert torch.equal(single, fast[0])
    for e, f in zip(exact, fast):
        d = (e - f).abs()
        assert int((d > 1e-5 * 1000).sum()) <= 8  # (isolated dm == 0 flips of the limiter, see the test above)
        assert float(d.mean()) <= 1e-7 * 1000
    cols = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])[::16].cpu().numpy()
    ref = mappm_c.mappm(cols(pe1), cols(qs[1]), cols(pe2))
    assert _bits_equal(cols(exact[1]), ref)


def _assert_outliers_within_source_bounds(pe1, q, pe2, fast, exact, tol):
    """[ncol, lev] numpy arrays.  Every target layer whose fast value differs from exact by more than ``tol`` (per column)
    must lie inside [min, max] of the source layers it overlaps plus one neighbour either side: the monotonicity constraint
    (mappm.f90:854-931) bounds a layer's parabola by the means of its neighbours, whichever branch of the limiter a level
    took.  Returns the number of outliers checked."""
    diff = np.abs(fast - exact)
    cols, levs = np.nonzero(diff > tol)
    km = q.shape[1]
    for c, k in zip(cols, levs):
        lo, hi = pe2[c, k], pe2[c, k + 1]
        j0 = int(np.clip(np.searchsorted(pe1[c], lo, side="right") - 1, 0, km - 1))
        j1 = int(np.clip(np.searchsorted(pe1[c], hi, side="left") - 1, 0, km - 1))
        a, b = max(j0 - 1, 0), min(max(j1, j0) + 1, km - 1)
        qmin, qmax = q[c, a:b + 1].min(), q[c, a:b + 1].max()
        slack = 1e-6 * max(abs(qmin), abs(qmax), 1.0)
        assert qmin - slack <= fast[c, k] <= qmax + slack, (c, k, fast[c, k], exact[c, k], qmin, qmax)
    return len(cols)


def test_fast_mode_outliers_stay_within_the_source_layers_bounds(device):
    """VERDICT r02 #6b.  FAST differs from EXACT by a few ulp -- except where the reference's limiter is discontinuous (a
    slope `dm` that cancels to exactly zero flattens the profile, mappm.f90:876-880): there the two arithmetics may take
    different branches.  What must hold for such a level is not closeness but monotonicity: the value stays inside the
    bounds of the source layers the target layer overlaps.  Checked (i) at BASELINE configs[2]'s full size on the
    pipeline's own target grid, (ii) on integer-valued fields over tied pressures, which provoke the flips by the hundred."""
    from fv3net_amd import ops

    # (i) C384, 4 fields
    g = torch.Generator(device=device).manual_seed(11)
    n, nz = 384, 79
    delp = torch.rand((6, nz, n, n), device=device, generator=g) * 1200 + 300
    area = torch.rand((6, n, n), device=device, generator=g) * 0.5 + 0.5
    qs = [torch.rand((6, nz, n, n), device=device, generator=g) * 2000 - 1000 for _ in range(4)]
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, 8), 300.0, 1)
    exact = ops.mappm_multi_coarse_target(pe1, qs, pe2c, 8, z_axis=1, arith="exact")
    fast = ops.mappm_multi_coarse_target(pe1, qs, pe2c, 8, z_axis=1, arith="fast")
    pe2 = ops.block_upsample(pe2c, 8)
    cols = lambda t, idx: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])[idx].cpu().numpy()
    n_out = 0
    for e, f, q in zip(exact, fast, qs):
        bad = ((e - f).abs() > 1e-5 * 2000).any(dim=1).reshape(-1).nonzero().reshape(-1)  # columns with an outlier
        assert float((e - f).abs().mean()) <= 1e-7 * 2000
        if bad.numel():
            n_out += _assert_outliers_within_source_bounds(cols(pe1, bad), cols(q, bad), cols(pe2, bad), cols(f, bad), cols(e, bad),
                                                           1e-5 * 2000)
    # (ii) integer-valued data on tied pressures
    rng = np.random.default_rng(3)
    nt, ny, nx, km, kn = 2, 16, 16, 40, 37
    ncol = nt * ny * nx
    pe1 = np.concatenate([np.full((ncol, 1), 3.0), 3 + np.cumsum(rng.integers(1, 4, (ncol, km)), 1)], 1).astype(np.float32)
    pe2 = np.concatenate([np.full((ncol, 1), 3.0), 3 + np.cumsum(rng.integers(1, 4, (ncol, kn)), 1)], 1).astype(np.float32)
    fields = [np.round(rng.uniform(-20, 20, (ncol, km))).astype(np.float32) for _ in range(4)]
    nat = lambda a: _dev(_native(a, nt, ny, nx), device)
    back = lambda t: np.moveaxis(ops.as_numpy(t), 1, -1).reshape(ncol, -1)
    e = ops.mappm_multi(nat(pe1), [nat(q) for q in fields], nat(pe2), z_axis=1, arith="exact")
    f = ops.mappm_multi(nat(pe1), [nat(q) for q in fields], nat(pe2), z_axis=1, arith="fast")
    flips = 0
    for ee, ff, q in zip(e, f, fields):
        flips += _assert_outliers_within_source_bounds(pe1, q, pe2, back(ff), back(ee), 1e-5 * 40)
    # (round 2's reciprocal arithmetic flipped a branch on ~1 level in 6e8 at C384 and on 1 in 5e3 of the integer-valued
    # levels; with round 3's formulation neither set shows one -- the bound is what would have to hold if it did)
    print(f"limiter flips beyond 1e-5 of the range: C384 {n_out}, integer-valued {flips}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("ny,nx,nfields", [(64, 64, 1), (64, 128, 4), (65, 64, 3), (64, 65, 2), (65, 16, 2)])
def test_coarse_target_remap_equals_the_upsampled_one(device, dtype, ny, nx, nfields):
    """``mappm_multi_coarse_target`` (target interfaces read through (y // f, x // f)) gives bit for bit what the reference's
    route gives -- upsample the coarse interfaces, then remap (regridz.py:119-185) -- for centred and staggered (odd) dims,
    1..4 fields, both input dtypes, shapes the sweep kernel takes and one it does not (65 x 16 columns are not whole waves: the fallback inside the op);
    so does ``mask_weights`` with the coarse pressures."""
    from fv3net_amd import ops

    f, nb, km = 8, 3, 19
    g = torch.Generator(device=device).manual_seed(ny * 1000 + nx)
    coarse = lambda n: (n - 1) // f + 1 if n % 2 else n // f
    delp = torch.rand((nb, km, ny, nx), device=device, generator=g, dtype=dtype) * 1200 + 300
    delp_c = torch.rand((nb, km, coarse(ny), coarse(nx)), device=device, generator=g, dtype=dtype) * 1200 + 300
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2_c = ops.pressure_at_interface(delp_c, 300.0, 1)
    qs = [torch.rand((nb, km, ny, nx), device=device, generator=g, dtype=dtype) * 200 - 100 for _ in range(nfields)]
    want = ops.mappm_multi(pe1, qs, ops.block_upsample(pe2_c, f), z_axis=1)
    got = ops.mappm_multi_coarse_target(pe1, qs, pe2_c, f, z_axis=1)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    w = torch.rand((nb, ny, nx), device=device, generator=g, dtype=torch.float32)
    for extrapolate, pc in ((False, pe2_c), (True, 0.5 * (pe2_c[:, 1:] + pe2_c[:, :-1]))):
        want = ops.mask_weights(w, ops.block_upsample(pc, f), pe1, 1, extrapolate=extrapolate)
        got = ops.mask_weights(w, pc, pe1, 1, extrapolate=extrapolate, coarse_factor=f)
        assert torch.equal(got, want) and 0 < float((got == 0).float().mean()) < 1


def test_coarse_target_sweep_on_a_plane_of_more_than_4_gib_of_levels(device):
    """A C3072 tile of float64 restarts is 75 MB per level and 6 GB per field: the byte offsets of its levels exceed 32 bits
    although nothing the sweep kernel keeps in 32 bits does (a level's stride, a result column, the coarse target array).
    2688 x 2688 float64 columns x 79 levels (4.6 GB of interfaces) in the FAST arithmetic -- which only the sweep kernel has -- must
    equal the same columns remapped in two halves, each below the old all-in-32-bits rule."""
    from fv3net_amd import ops

    n, km, f = 2688, 79, 8
    g = torch.Generator(device=device).manual_seed(5)
    delp = torch.rand((1, km, n, n), device=device, generator=g, dtype=torch.float64) * 1200 + 300
    delp_c = ops.weighted_block_average(delp, torch.ones((1, n, n), device=device, dtype=torch.float32), f)
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2c = ops.pressure_at_interface(delp_c, 300.0, 1)
    del delp
    qs = [torch.rand((1, km, n, n), device=device, generator=g, dtype=torch.float64) * 200 - 100 for _ in range(2)]
    assert pe1.numel() * 8 > 2 ** 32
    whole = ops.mappm_multi_coarse_target(pe1, qs, pe2c, f, z_axis=1, arith="fast")
    h = n // 2
    for rows, crow in ((slice(0, h), slice(0, h // f)), (slice(h, n), slice(h // f, n // f))):
        part = ops.mappm_multi_coarse_target(pe1[:, :, rows].contiguous(), [q[:, :, rows].contiguous() for q in qs],
                                             pe2c[:, :, crow].contiguous(), f, z_axis=1, arith="fast")
        for a, b in zip(whole, part):
            assert torch.equal(a[:, :, rows], b)
    exact = ops.mappm_multi_coarse_target(pe1, qs[:1], pe2c, f, z_axis=1, arith="exact")[0]
    assert not torch.equal(exact, whole[0])   # (the fast arithmetic really ran: the merge kernels the old rule fell back to have none)
    del whole, exact
    torch.cuda.empty_cache()
