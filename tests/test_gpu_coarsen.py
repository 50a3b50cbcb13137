"""Parity of the HIP coarsening kernels (through the C ABI) with the numpy oracle.

Tolerances: block sums are accumulated in a different order than numpy's, so values agree to
rounding of the accumulation: |gpu - oracle| <= 1e-5 * sum|obj*w| / sum(w) for float32
(1e-13 for float64).  NaN patterns and integer results must match exactly.
"""
import numpy as np
import pytest
import torch

from oracle import coarsen_np as onp

pytestmark = pytest.mark.gpu


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _check_wavg(res, obj, w, factor, tol):
    ref = onp.weighted_block_average(obj, w, factor)
    scale = onp._nansum_blocks(np.abs(obj * w), factor, factor) / np.abs(onp._nansum_blocks(np.broadcast_to(w, obj.shape) * 1.0, factor, factor))
    assert res.shape == ref.shape
    assert res.dtype == ref.dtype
    assert np.array_equal(np.isnan(res), np.isnan(ref))
    ok = ~np.isnan(ref)
    err = np.abs(res[ok] - ref[ok])
    assert np.all(err <= tol * np.maximum(scale[ok], 1e-30)), err.max()


@pytest.mark.parametrize("factor", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("odt,wdt", [(np.float32, np.float32), (np.float64, np.float64), (np.float64, np.float32), (np.float32, np.float64)])
def test_weighted_block_average_2d_weights(device, factor, odt, wdt):
    from fv3net_amd import ops

    rng = np.random.default_rng(factor)
    n = 64 if factor <= 16 else 128
    obj = rng.uniform(-1000, 1000, (3, 5, n, n)).astype(odt)
    w = rng.uniform(0.5, 1, (3, n, n)).astype(wdt)
    res = ops.as_numpy(ops.weighted_block_average(_dev(obj, device), _dev(w, device), factor))
    # a float32 operand caps the accuracy at float32 rounding of its own sums
    tol = 1e-13 if (odt == np.float64 and wdt == np.float64) else 1e-5
    _check_wavg(res, obj, w[:, None], factor, tol)


@pytest.mark.parametrize("shape,factor", [((2, 7, 6, 10), 2), ((1, 3, 9, 6), 3), ((4, 12, 20), 4), ((2, 2, 48, 48), 8), ((24, 40), 8)])
def test_weighted_block_average_3d_weights_and_odd_shapes(device, shape, factor):
    from fv3net_amd import ops

    rng = np.random.default_rng(1)
    obj = rng.uniform(-1000, 1000, shape).astype(np.float32)
    w = rng.uniform(3, 5, shape).astype(np.float32)
    res = ops.as_numpy(ops.weighted_block_average(_dev(obj, device), _dev(w, device), factor))
    _check_wavg(res, obj, w, factor, 1e-5)


def test_weighted_block_average_nan_and_zero_weights(device):
    from fv3net_amd import ops

    rng = np.random.default_rng(2)
    obj = rng.uniform(-10, 10, (2, 3, 16, 16)).astype(np.float32)
    w = rng.uniform(0.5, 1, (2, 3, 16, 16)).astype(np.float32)
    obj[0, 0, :2, :2] = np.nan          # fully-NaN block: numerator 0
    obj[0, 1, 3, 5] = np.nan            # partially NaN block
    w[1, 0, 4:8, 4:8] = 0.0             # zero denominator: 0/0 = NaN
    w[1, 1, 8, 8] = np.nan              # NaN weight is skipped in both sums
    for f in (2, 4):
        res = ops.as_numpy(ops.weighted_block_average(_dev(obj, device), _dev(w, device), f))
        ref = onp.weighted_block_average(obj, w, f)
        assert np.array_equal(np.isnan(res), np.isnan(ref))
        np.testing.assert_allclose(res, ref, rtol=2e-6, atol=1e-5)


def test_weighted_block_average_reference_known_answer(device):
    # external/vcm/tests/test_cubedsphere.py:206-237
    from fv3net_amd import ops

    data = np.array([[2.0, 6.0], [6.0, 2.0]])
    weights = np.array([[6.0, 2.0], [2.0, 6.0]])
    res = ops.as_numpy(ops.weighted_block_average(_dev(data, device), _dev(weights, device), 2))
    assert res.dtype == np.float64
    np.testing.assert_array_equal(res, np.array([[3.0]]))


def test_weighted_block_average_rejects_bad_factor(device):
    from fv3net_amd import ops

    x = torch.zeros(2, 6, 6, device=device)
    with pytest.raises(ValueError):
        ops.weighted_block_average(x, x, 4)


@pytest.mark.parametrize("edge", ["x", "y"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_edge_weighted_block_average(device, edge, dt):
    from fv3net_amd import ops

    rng = np.random.default_rng(3)
    f = 4
    shape = (2, 3, 4 * f + 1, 5 * f) if edge == "x" else (2, 3, 4 * f, 5 * f + 1)
    obj = rng.uniform(-50, 50, shape).astype(dt)
    sp = rng.uniform(0.5, 1, (2,) + shape[-2:]).astype(dt)
    res = ops.as_numpy(ops.edge_weighted_block_average(_dev(obj, device), _dev(sp, device), f, edge))
    ref = onp.edge_weighted_block_average(obj, sp[:, None], f, edge)
    assert res.shape == ref.shape and res.dtype == ref.dtype
    np.testing.assert_allclose(res, ref, rtol=1e-5 if dt == np.float32 else 1e-13)


def test_edge_weighted_reference_known_answers(device):
    # external/vcm/tests/test_cubedsphere.py:240-259, arrays given as [x_dim, y_dim]
    from fv3net_amd import ops

    data = np.array([[2, 6, 2], [6, 2, 6]], dtype=np.float64)  # dims (x, y)
    spacing = np.array([[6, 2, 6], [2, 6, 2]], dtype=np.float64)
    res = ops.as_numpy(ops.edge_weighted_block_average(_dev(data.T, device), _dev(spacing.T, device), 2, "x"))
    np.testing.assert_array_equal(res.T, np.array([[3.0, 3.0]]))
    data = np.array([[2, 6], [6, 2], [2, 6]], dtype=np.float64)
    spacing = np.array([[6, 2], [2, 6], [6, 2]], dtype=np.float64)
    res = ops.as_numpy(ops.edge_weighted_block_average(_dev(data.T, device), _dev(spacing.T, device), 2, "y"))
    np.testing.assert_array_equal(res.T, np.array([[3.0], [3.0]]))


@pytest.mark.parametrize("method", ["sum", "mean", "min", "max", "median"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_block_coarsen_float(device, method, dt):
    from fv3net_amd import ops

    rng = np.random.default_rng(4)
    a = rng.integers(-20, 20, (3, 12, 8)).astype(dt)  # small integers: sums are exact
    if method != "median":
        a[0, 0, 0] = np.nan
        a[1, :4, :4] = np.nan
    else:
        a[0, 0, 0] = np.nan
    res = ops.as_numpy(ops.block_reduce(_dev(a, device), (4, 4), op=method))
    ref = onp.block_coarsen(a, 4, method)
    np.testing.assert_array_equal(res, ref)


def test_block_reduce_matches_reference_regtest_hash_inputs(device):
    # _xarray_block_reduce_dataarray regtest (test_cubedsphere.py:262-290): arange(32) as
    # [x=4, y=4, z=2] float32, 2x2 blocks over (x, y), mean and median.
    from fv3net_amd import ops

    data = np.arange(32).reshape(4, 4, 2).astype(np.float32)
    zxy = np.moveaxis(data, -1, 0)  # [z, x, y]: horizontal dims last
    for method, fn in (("mean", np.mean), ("median", np.median)):
        res = ops.as_numpy(ops.block_reduce(_dev(zxy, device), (2, 2), op=method))
        ref = fn(data.reshape(2, 2, 2, 2, 2), axis=(1, 3))  # [X, Y, z]
        np.testing.assert_array_equal(np.moveaxis(res, 0, -1), ref)


@pytest.mark.parametrize("dt", [np.int32, np.int64])
def test_block_coarsen_int_and_edge(device, dt):
    from fv3net_amd import ops

    # external/vcm/tests/test_cubedsphere.py:386-419 (arrays given as [x_dim, y_dim])
    data = np.array([[2, 6, 2], [6, 2, 6]], dtype=dt).T.copy()  # -> [y, x]
    res = ops.as_numpy(ops.block_reduce(_dev(data, device), (1, 2), (2, 2), op="sum"))
    np.testing.assert_array_equal(res.T, np.array([[8, 8]]))
    res = ops.as_numpy(ops.block_reduce(_dev(data, device), (1, 2), (2, 2), op="min"))
    np.testing.assert_array_equal(res.T, np.array([[2, 2]]))
    data = np.array([[2, 6], [6, 2], [2, 6]], dtype=dt).T.copy()
    res = ops.as_numpy(ops.block_reduce(_dev(data, device), (2, 1), (2, 2), op="sum"))
    np.testing.assert_array_equal(res.T, np.array([[8], [8]]))


@pytest.mark.parametrize("policy", ["propagate", "omit"])
def test_block_mode(device, policy):
    from fv3net_amd import ops

    # external/vcm/tests/test_cubedsphere.py:731-769
    data = np.array(
        [[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0], [1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 0.0, np.nan]]
    )
    res = ops.as_numpy(ops.block_reduce(_dev(data, device), (2, 2), op="mode", nan_policy=policy))
    np.testing.assert_array_equal(res, np.array([[0.0, 1.0], [1.0, 0.0]]))
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4, (2, 16, 24)).astype(np.float32)  # categorical field, many ties
    a[rng.random(a.shape) < 0.1] = np.nan
    res = ops.as_numpy(ops.block_reduce(_dev(a, device), (4, 4), op="mode", nan_policy=policy))
    np.testing.assert_array_equal(res, onp.block_mode(a, 4, policy))


@pytest.mark.parametrize("f", [2, 3, 8, 16])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_block_mode_and_median_wave_and_serial_kernels(device, f, dt):
    """Blocks of up to 64 values go through the one-wavefront-per-block kernel, larger ones through the
    one-thread-per-block kernel: both against the oracle, with many ties, NaNs, all-NaN blocks and
    zeros of both signs (which compare equal)."""
    from fv3net_amd import ops

    rng = np.random.default_rng(17 + f)
    a = rng.integers(0, 5, (3, 4 * f, 6 * f)).astype(dt)
    a[rng.random(a.shape) < 0.15] = np.nan
    a[0, :f, :f] = np.nan                       # an all-NaN block
    a[1, :f, :f] = np.where(rng.random((f, f)) < 0.5, 0.0, -0.0)   # zeros of both signs only
    for policy in ("propagate", "omit"):
        res = ops.as_numpy(ops.block_reduce(_dev(a, device), (f, f), op="mode", nan_policy=policy))
        ref = onp.block_mode(a, f, policy)
        np.testing.assert_array_equal(res, ref)
    b = rng.normal(0, 1, (3, 4 * f, 6 * f)).astype(dt)
    b[2, :f, :f] = np.round(b[2, :f, :f])       # ties among the middle ranks
    b[0, 0, 0] = np.nan                          # numpy.median: NaN if any NaN
    res = ops.as_numpy(ops.block_reduce(_dev(b, device), (f, f), op="median"))
    np.testing.assert_array_equal(res, onp.block_coarsen(b, f, "median"))


@pytest.mark.parametrize("dt", [np.int32, np.int64])
def test_block_mode_integers(device, dt):
    from fv3net_amd import ops

    rng = np.random.default_rng(23)
    a = rng.integers(-3, 4, (2, 16, 24)).astype(dt)
    res = ops.as_numpy(ops.block_reduce(_dev(a, device), (4, 4), op="mode"))
    np.testing.assert_array_equal(res, onp.block_mode(a.astype(np.float64), 4, "propagate").astype(dt))


@pytest.mark.parametrize("shape", [(2, 2), (2, 3), (3, 3), (5, 4, 7)])
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.int32, np.int64])
def test_block_upsample(device, shape, dt):
    from fv3net_amd import ops

    a = np.arange(np.prod(shape)).reshape(shape).astype(dt)
    for f in (1, 2, 3):
        res = ops.as_numpy(ops.block_upsample(_dev(a, device), f))
        np.testing.assert_array_equal(res, onp.block_upsample(a, f))


def test_full_size_c384_properties(device):
    """C384 -> C48 at full size, checked through size-independent properties: a constant field
    averages to itself, the operator is linear in the field, and sum(w * mean) over the coarse
    grid equals sum(w * field) over the fine grid."""
    from fv3net_amd import ops

    g = torch.Generator(device=device).manual_seed(0)
    obj = (torch.rand((6, 79, 384, 384), device=device, generator=g) * 2000 - 1000)
    obj2 = (torch.rand((6, 79, 384, 384), device=device, generator=g) * 2000 - 1000)
    area = torch.rand((6, 384, 384), device=device, generator=g) * 0.5 + 0.5
    f = 8
    const = ops.weighted_block_average(torch.full_like(obj, 3.25), area, f)
    assert torch.allclose(const, torch.full_like(const, 3.25), rtol=1e-6, atol=0)
    a, b = ops.weighted_block_average(obj, area, f), ops.weighted_block_average(obj2, area, f)
    ab = ops.weighted_block_average(obj + 2 * obj2, area, f)
    assert torch.allclose(ab, a + 2 * b, rtol=0, atol=2e-3)
    wsum = ops.block_reduce(area, (f, f), op="sum")
    lhs = (a.double() * wsum[:, None].double()).sum(dim=(-1, -2))
    rhs = (obj.double() * area[:, None].double()).sum(dim=(-1, -2))
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1.0)


def test_full_size_c3072_properties(device):
    """BASELINE configs[4] at full size: one [6, 79, 3072, 3072] float32 field (4.47 G elements, 17.9 GB -- element
    offsets beyond 2^32) coarsened by 8 with 2-D area weights.  Size-independent properties: a constant field
    averages to itself; sum(w * mean) over the coarse grid equals sum(w * field) over the fine grid, per tile;
    and blocks sampled from all over the array -- the very last one included -- equal the float64 block
    average of the same values to float32 rounding."""
    from fv3net_amd import ops

    if torch.cuda.get_device_properties(device).total_memory < 64e9:
        pytest.skip("needs 64 GB of device memory")
    g = torch.Generator(device=device).manual_seed(1)
    n, nz, f = 3072, 79, 8
    area = torch.rand((6, n, n), device=device, generator=g) * 0.5 + 0.5
    obj = torch.empty((6, nz, n, n), device=device, dtype=torch.float32)
    for t in range(6):  # (filled per tile: the generator's temporaries stay small)
        obj[t] = torch.rand((nz, n, n), device=device, generator=g) * 2000 - 1000
    out = ops.weighted_block_average(obj, area, f)
    assert out.shape == (6, nz, n // f, n // f) and out.dtype == torch.float32
    wsum = ops.block_reduce(area, (f, f), op="sum")
    for t in range(6):
        lhs = (out[t].double() * wsum[t].double()).sum()
        rhs = sum((obj[t, k0:k0 + 8].double() * area[t].double()).sum() for k0 in range(0, nz, 8))
        assert abs(float(lhs - rhs)) <= 1e-6 * float((obj[t].abs().double().sum() * 0.75)), t
    rng = np.random.default_rng(2)
    samples = [(5, nz - 1, n // f - 1, n // f - 1), (0, 0, 0, 0)] + [
        (int(rng.integers(6)), int(rng.integers(nz)), int(rng.integers(n // f)), int(rng.integers(n // f))) for _ in range(200)]
    for t, k, Y, X in samples:
        blk = obj[t, k, Y * f:(Y + 1) * f, X * f:(X + 1) * f].double().cpu().numpy()
        w = area[t, Y * f:(Y + 1) * f, X * f:(X + 1) * f].double().cpu().numpy()
        want = (blk * w).sum() / w.sum()
        assert abs(float(out[t, k, Y, X]) - want) <= 1e-5 * np.abs(blk).max(), (t, k, Y, X)
    del obj
    torch.cuda.empty_cache()
    const = torch.full((6, nz, n, n), 3.25, device=device, dtype=torch.float32)
    res = ops.weighted_block_average(const, area, f)
    assert torch.allclose(res, torch.full_like(res, 3.25), rtol=1e-6, atol=0)


def test_full_size_c3072_column_and_upsample_kernels(device):
    """The other kernels of the pressure-level path on a 4.47 G-element field (offsets beyond 2^32): interface
    pressures (cumulative sums down 79 levels), their consistency with the column-sum kernel, the coarse-to-fine
    upsample and the weight mask, each checked on columns sampled from all over the array, the last one included."""
    from fv3net_amd import ops

    if torch.cuda.get_device_properties(device).total_memory < 128e9:
        pytest.skip("needs 128 GB of device memory")
    g = torch.Generator(device=device).manual_seed(3)
    n, nz, f = 3072, 79, 8
    delp = torch.empty((6, nz, n, n), device=device, dtype=torch.float32)
    for t in range(6):
        delp[t] = torch.rand((nz, n, n), device=device, generator=g) * 1200 + 300
    phalf = ops.pressure_at_interface(delp, 300.0, 1)
    assert phalf.shape == (6, nz + 1, n, n)
    ps = ops.column_sum(delp, 1, addend=300.0)
    rng = np.random.default_rng(4)
    cols = [(5, n - 1, n - 1), (0, 0, 0)] + [(int(rng.integers(6)), int(rng.integers(n)), int(rng.integers(n))) for _ in range(100)]
    for t, y, x in cols:
        d = delp[t, :, y, x].cpu().numpy()
        want = np.cumsum(np.concatenate([[np.float32(300.0)], d]), dtype=np.float32)  # p[0] = toa; p[k+1] = p[k] + delp[k]
        np.testing.assert_array_equal(phalf[t, :, y, x].cpu().numpy(), want)           # the same additions in the same order
        np.testing.assert_allclose(float(ps[t, y, x]), float(want[-1]), rtol=1e-6)     # (sum first, then + toa)
    # weight mask against the interface pressures of a second, coarser-looking grid
    area = torch.rand((6, n, n), device=device, generator=g) * 0.5 + 0.5
    coarse = ops.block_upsample(ops.weighted_block_average(delp, area, f), f)
    assert coarse.shape == delp.shape
    for t, y, x in cols[:20]:
        blk = coarse[t, :, (y // f) * f:(y // f + 1) * f, (x // f) * f:(x // f + 1) * f]
        assert bool((blk == blk[:, :1, :1]).all())  # every fine cell of a block holds the block's value
    del delp
    torch.cuda.empty_cache()
    phalf_c = ops.pressure_at_interface(coarse, 300.0, 1)
    del coarse
    masked = ops.mask_weights(area, phalf_c, phalf, z_axis=1)
    assert masked.shape == (6, nz, n, n)
    for t, y, x in cols:
        pc, pf, a = phalf_c[t, :, y, x].cpu().numpy(), phalf[t, :, y, x].cpu().numpy(), float(area[t, y, x])
        np.testing.assert_array_equal(masked[t, :, y, x].cpu().numpy(), np.where(pc[1:] < pf[-1], np.float32(a), np.float32(0)))


@pytest.mark.parametrize("axis", ["x", "y"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_cube_interp_center_to_outer_matches_oracle(device, axis, dt):
    """delp at the cell edges across the cube's faces (regridz.py:123-135, xgcm.py:7-34): the halo
    rows come from the neighbouring tiles through the connectivity table, reversed where the
    connection swaps axes.  The oracle is pinned by the reference's pressure-level u / v fixtures."""
    from fv3net_amd import ops
    from fv3net_amd.cubedsphere.grid import halos_from_rows, interp_tiles_to_edges

    rng = np.random.default_rng(11)
    a = rng.uniform(300, 1500, (6, 5, 12, 12)).astype(dt)
    res = ops.as_numpy(interp_tiles_to_edges(_dev(a, device), axis))
    ref = onp.interp_center_to_outer(a, axis)
    assert res.shape == ref.shape and res.dtype == ref.dtype
    np.testing.assert_array_equal(res, ref)  # 0.5 * (a + b): the same two roundings in the same order
    # the boundary vectors themselves (what a sharded run exchanges)
    rows = ops.as_numpy(ops.cube_edge_rows(_dev(a, device)))
    np.testing.assert_array_equal(rows[:, 0], a[..., :, 0])
    np.testing.assert_array_equal(rows[:, 1], a[..., :, -1])
    np.testing.assert_array_equal(rows[:, 2], a[..., 0, :])
    np.testing.assert_array_equal(rows[:, 3], a[..., -1, :])
    lo, hi = halos_from_rows(torch.from_numpy(rows), [2, 5], axis)
    assert lo.shape == (2, 5, 12) and hi.shape == (2, 5, 12)


def test_cube_interp_rejects_bad_shapes(device):
    from fv3net_amd import ops
    from fv3net_amd.cubedsphere.grid import interp_tiles_to_edges

    with pytest.raises(ValueError):
        interp_tiles_to_edges(torch.zeros(5, 3, 4, 4, device=device), "x")   # not six tiles
    with pytest.raises(ValueError):
        ops.cube_edge_rows(torch.zeros(6, 3, 4, 5, device=device))            # faces must be square
    with pytest.raises(ValueError):
        ops.interp_center_to_outer(torch.zeros(6, 3, 4, 4, device=device), torch.zeros(6, 3, 5, device=device),
                                   torch.zeros(6, 3, 4, device=device), 0)


def test_sfc_data_complex_with_categorical_fields(device):
    """The 'complex' surface-data method (coarsen_restarts.py:1111-1470) on fields that look like the
    real ones -- small-integer categoricals with ties, land / ocean / sea-ice patches, zero
    fractions -- against the oracle (pinned by the reference's sfc_data fixtures): categorical
    results bit-exact, means to float32 rounding of float64 sums."""
    from fv3net_amd.cubedsphere.sfc_data import SFC_DATA_VARIABLES, coarse_grain_sfc_data_tensors
    from oracle import sfc_data_np

    rng = np.random.default_rng(21)
    nt, n, f = 6, 24, 4
    ds = {}
    for name in SFC_DATA_VARIABLES:
        shape = (nt, 4, n, n) if name in ("smc", "slc", "stc") else (nt, n, n)
        ds[name] = rng.uniform(0, 1, shape)
    ds["slmsk"] = rng.integers(0, 3, (nt, n, n)).astype(np.float64)
    ds["vtype"] = rng.integers(13, 17, (nt, n, n)).astype(np.float64)   # includes land ice (15)
    ds["stype"] = rng.integers(1, 17, (nt, n, n)).astype(np.float64)
    ds["srflag"] = rng.integers(0, 2, (nt, n, n)).astype(np.float64)
    ds["slope"] = rng.integers(1, 10, (nt, n, n)).astype(np.float64)
    ds["tsea"] = rng.uniform(260, 290, (nt, n, n))
    ds["tg3"] = rng.uniform(260, 290, (nt, n, n))
    ds["vfrac"][rng.random((nt, n, n)) < 0.5] = 0.0
    ds["fice"][rng.random((nt, n, n)) < 0.7] = 0.0
    ds["sncovr"][rng.random((nt, n, n)) < 0.8] = 0.0
    ds["shdmin"] = rng.uniform(0, 0.03, (nt, n, n))
    area = rng.uniform(0.5, 1.0, (nt, n, n))
    ref = sfc_data_np.coarse_grain_sfc_data_complex(ds, area, f)
    res = coarse_grain_sfc_data_tensors({k: _dev(v, device) for k, v in ds.items()}, _dev(area, device), f)
    assert set(res) == set(ref)
    for name, want in ref.items():
        got = res[name].cpu().numpy()
        assert got.shape == want.shape and got.dtype == np.float32, name
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        if name in ("slmsk", "vtype", "stype", "srflag", "slope"):
            np.testing.assert_array_equal(got, want, err_msg=name)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-7, atol=0, err_msg=name)
