"""Property tests (hypothesis) of the host logic that every labelled entry point goes through: the slice of xarray
semantics in ``xr_compat`` (transpose / isel / rename / merge keep data and labels consistent), the partitions of
``parallel`` (every unit owned exactly once, blocks never split) and the oracle helpers the GPU tests lean on."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from fv3net_amd import parallel
from fv3net_amd.xr_compat import DataArray, Dataset, merge
from oracle import coarsen_np as onp
from oracle import mlp_np

DIMS = ["tile", "z", "y", "x"]


@st.composite
def arrays(draw):
    nd = draw(st.integers(1, 4))
    dims = draw(st.permutations(DIMS))[:nd]
    shape = [draw(st.integers(1, 4)) for _ in dims]
    data = np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape)
    coords = {d: np.arange(n) * 10 + i for i, (d, n) in enumerate(zip(dims, shape)) if draw(st.booleans())}
    return DataArray(data, dims=list(dims), coords=coords, name="a", attrs={"units": "K"})


@settings(max_examples=60, deadline=None)
@given(arrays(), st.data())
def test_transpose_isel_rename_keep_labels_and_values(da, data):
    perm = data.draw(st.permutations(list(da.dims)))
    t = da.transpose(*perm)
    assert t.dims == tuple(perm) and t.attrs == da.attrs and t.name == da.name
    np.testing.assert_array_equal(t.transpose(*da.dims).values, da.values)
    np.testing.assert_array_equal(t.values, np.transpose(da.values, [da.dims.index(d) for d in perm]))
    for d in da.dims:
        assert t.sizes[d] == da.sizes[d]
        if d in da.coords:
            np.testing.assert_array_equal(t.coords[d], da.coords[d])
    d0 = da.dims[0]
    i = data.draw(st.integers(0, da.sizes[d0] - 1))
    sel = da.isel({d0: i})
    assert sel.dims == da.dims[1:]
    np.testing.assert_array_equal(sel.values, da.values[i])
    ren = da.rename({d0: "renamed"})
    assert ren.dims == ("renamed",) + da.dims[1:] and ren.sizes["renamed"] == da.sizes[d0]
    np.testing.assert_array_equal(ren.values, da.values)
    if d0 in da.coords:
        np.testing.assert_array_equal(ren.coords["renamed"], da.coords[d0])


@settings(max_examples=40, deadline=None)
@given(arrays(), arrays())
def test_dataset_merge_and_selection(a, b):
    ds = merge([Dataset({"a": a}), Dataset({"b": b.rename("b")})]) if set(a.dims).isdisjoint(b.dims) or all(
        a.sizes[d] == b.sizes[d] for d in set(a.dims) & set(b.dims)) else None
    if ds is None:
        return
    assert list(ds) == ["a", "b"]
    np.testing.assert_array_equal(ds["a"].values, a.values)
    np.testing.assert_array_equal(ds["b"].values, b.values)
    sub = ds[["b"]]
    assert list(sub) == ["b"] and sub["b"].dims == b.dims
    for d, n in ds.dims.items():
        assert n == (a.sizes.get(d) or b.sizes.get(d))


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 5000), st.integers(1, 17))
def test_column_range_partitions_every_column_once(n, size):
    ranges = [parallel.column_range(n, size, r) for r in range(size)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 <= a1
    lengths = [b - a for a, b in ranges]
    assert max(lengths) - min(lengths) <= 1  # balanced


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 6), st.integers(1, 12), st.sampled_from([1, 2, 4, 8]), st.integers(1, 9))
def test_tile_bands_cover_every_row_once_and_never_split_a_block(n_tiles, blocks, factor, size):
    ny = blocks * factor
    per_rank = parallel.tile_bands(n_tiles, ny, factor, size)
    assert len(per_rank) == size
    covered = np.zeros((n_tiles, ny), int)
    for units in per_rank:
        for t, lo, hi in units:
            assert lo % factor == 0 and hi % factor == 0 and 0 <= lo < hi <= ny
            covered[t, lo:hi] += 1
    assert (covered == 1).all()


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 3), st.integers(1, 3), st.sampled_from([1, 2, 4]), st.integers(0, 2 ** 31 - 1))
def test_weighted_block_average_oracle_properties(ny_blocks, nx_blocks, f, seed):
    """The oracle the GPU coarsening tests lean on: constants are preserved, the weighted sum is conserved, and with a
    NaN inside a block the result is the average over the rest of it (xarray's skipna sums)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-5, 5, (2, ny_blocks * f, nx_blocks * f))
    w = rng.uniform(0.5, 1.0, x.shape)
    np.testing.assert_allclose(onp.weighted_block_average(np.full_like(x, 2.5), w, f), 2.5, rtol=1e-14)
    avg = onp.weighted_block_average(x, w, f)
    wsum = w.reshape(2, ny_blocks, f, nx_blocks, f).sum(axis=(2, 4))
    np.testing.assert_allclose((avg * wsum).sum(), (x * w).sum(), rtol=1e-12, atol=1e-12)
    if f > 1:
        x2 = x.copy()
        x2[0, 0, 0] = np.nan
        got = onp.weighted_block_average(x2, w, f)[0, 0, 0]
        blk_x, blk_w = x[0, :f, :f].ravel()[1:], w[0, :f, :f].ravel()[1:]
        np.testing.assert_allclose(got, (blk_x * blk_w).sum() / w[0, :f, :f].sum(), rtol=1e-12)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.floats(-50, 50, allow_nan=False), min_size=1, max_size=8), st.floats(-60, 60, allow_nan=False))
def test_piecewise_is_a_step_function_of_its_bins(edges, x):
    edges = np.unique(np.asarray(edges, np.float32))
    values = np.arange(len(edges), dtype=np.float32)
    got = float(mlp_np.piecewise(edges, values, np.asarray([x], np.float32))[0])
    want = max(int(np.sum(edges <= np.float32(x))) - 1, 0)
    assert got == want
