"""The fused remap + masked 8 x 8 block mean (fv3hip_mappm_block_mean, csrc/remap.hip MEAN) against the three calls it
replaces -- the coarse-target remap, the masked weights and the weighted block average (regridz.py:149-220 followed by
coarsen.py:183-218) -- BIT FOR BIT, in both arithmetic modes; the unfused route itself is pinned to the compiled reference
Fortran and the reference's fixtures elsewhere (tests/test_gpu_vertical.py, tests/test_gpu_api.py)."""
import numpy as np
import pytest
import torch

from fv3net_amd import ops

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _case(rng, batch, km, ny, nx, spread, dtype, n_fields, kn=None):
    """Fine thicknesses with `spread` of the iid part (1 = BASELINE configs[2]'s U(300, 1500) per cell), area, fields."""
    kn = km if kn is None else kn
    shape = batch + (km, ny, nx)
    base = rng.uniform(300, 1500, batch + (km, 1, 1))
    delp = np.maximum(base + spread * (rng.uniform(300, 1500, shape) - 900.0), 20.0)   # (thin layers, never negative ones)
    area = rng.uniform(0.5, 1.0, batch[:1] + (ny, nx)).astype(np.float32) if batch else rng.uniform(0.5, 1.0, (ny, nx)).astype(np.float32)
    fields = [rng.uniform(-1000, 1000, shape) for _ in range(n_fields)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(dtype))).to(_dev())
    delp_t, area_t = t(delp), torch.from_numpy(area).to(_dev())
    delp_c = ops.weighted_block_average(delp_t, area_t, 8)
    if kn != km:  # another number of coarse layers: the same column mass in kn layers
        tot = delp_c.sum(dim=-3, keepdim=True)
        frac = torch.from_numpy(rng.dirichlet(np.ones(kn) * 8).astype(dtype)).to(_dev()).reshape((1,) * len(batch) + (kn, 1, 1))
        delp_c = tot * frac
    pe1 = ops.pressure_at_interface(delp_t, 300.0, -3)
    pe2c = ops.pressure_at_interface(delp_c, 300.0, -3)
    pfull = ops.pressure_at_midpoint_log(delp_c, 300.0, -3)
    return pe1, [t(f) for f in fields], pe2c, pfull, area_t


def _unfused(pe1, fields, pe2c, pfull, area, extrapolate, arith, iv=1, kord=1):
    q2 = ops.mappm_multi_coarse_target(pe1, fields, pe2c, 8, iv=iv, kord=kord, arith=arith)
    batch = tuple(pe1.shape[:-3])
    w = area
    if tuple(w.shape[:-2]) != batch:
        w = w.reshape(tuple(w.shape[:-2]) + (1,) * (len(batch) - (w.dim() - 2)) + tuple(w.shape[-2:])).expand(*batch, *w.shape[-2:]).contiguous()
    if pe2c.shape[-3] == pe1.shape[-3]:
        mw = ops.mask_weights(w, pfull if extrapolate else pe2c, pe1, -3, extrapolate=extrapolate, coarse_factor=8)
    else:  # (mask_weights wants as many coarse layers as fine ones, which is all the reference ever has: spell it out)
        level = (pfull if extrapolate else pe2c[..., 1:, :, :]).repeat_interleave(8, dim=-2).repeat_interleave(8, dim=-1)
        mw = torch.where(level < pe1[..., -1:, :, :], w.unsqueeze(-3), torch.zeros((), dtype=w.dtype, device=w.device)).contiguous()
    return ops.weighted_block_average_multi(q2, mw, 8) if len(q2) > 1 else [ops.weighted_block_average(q2[0], mw, 8)]


def _same(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True), \
        f"{np.sum(a.view(np.uint32) != b.view(np.uint32))} of {a.size} values differ, max {np.nanmax(np.abs(a - b))}"


@pytest.mark.parametrize("arith", ["exact", "fast"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n_fields", [1, 2, 3, 4, 5])
def test_block_mean_is_the_unfused_route_bit_for_bit(arith, dtype, n_fields):
    rng = np.random.default_rng(n_fields)
    pe1, fields, pe2c, pfull, area = _case(rng, (2,), 20, 24, 32, 0.3, dtype, n_fields)
    for extrapolate in (False, True):
        got = ops.mappm_block_mean(pe1, fields, pe2c, area, level_coarse=pfull if extrapolate else None, arith=arith)
        assert got is not None and len(got) == n_fields
        for g, w in zip(got, _unfused(pe1, fields, pe2c, pfull, area, extrapolate, arith)):
            _same(g, w)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_block_mean_with_lanes_far_apart(arith):
    """BASELINE configs[2]'s iid thicknesses and stronger: the lanes of a block end up more rows apart than the LDS ring
    holds (16), so values take the detour through the scratch rows -- the means must not notice."""
    rng = np.random.default_rng(7)
    for spread, km in ((1.0, 79), (2.5, 79), (1.0, 127)):
        pe1, fields, pe2c, pfull, area = _case(rng, (1,), km, 16, 16, spread, np.float64, 4)
        got = ops.mappm_block_mean(pe1, fields, pe2c, area, arith=arith)
        for g, w in zip(got, _unfused(pe1, fields, pe2c, pfull, area, False, arith)):
            _same(g, w)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_block_mean_other_target_layer_counts_and_iv(arith):
    rng = np.random.default_rng(11)
    for kn, iv, kord in ((13, 1, 1), (40, 0, 3), (25, -1, 2)):
        pe1, fields, pe2c, pfull, area = _case(rng, (3,), 20, 16, 24, 0.5, np.float64, 2, kn=kn)
        got = ops.mappm_block_mean(pe1, fields, pe2c, area, iv=iv, kord=kord, arith=arith)
        for g, w in zip(got, _unfused(pe1, fields, pe2c, pfull, area, False, arith, iv=iv, kord=kord)):
            _same(g, w)


def test_block_mean_area_shared_by_a_time_axis_and_nan_fields():
    rng = np.random.default_rng(3)
    pe1, fields, pe2c, pfull, area = _case(rng, (2, 3), 16, 16, 16, 0.4, np.float64, 3)  # [time?, tile]: area leads with dim 0
    fields[1][0, 1, 4, 3, 5] = float("nan")
    fields[2][1, 2, :, 9, 9] = float("nan")
    area = area.clone()
    area[0, 2, 3] = float("nan")
    for arith in ("exact", "fast"):
        got = ops.mappm_block_mean(pe1, fields, pe2c, area, arith=arith)
        for g, w in zip(got, _unfused(pe1, fields, pe2c, pfull, area, False, arith)):
            _same(g, w)


def test_block_mean_redoes_blocks_with_ill_formed_columns():
    """NaN or non-monotone pressures in one column: the unfused route sends that column through the sequential routine;
    the fused kernel redoes the whole block that way (exact arithmetic), so compare in exact mode."""
    rng = np.random.default_rng(5)
    pe1, fields, pe2c, pfull, area = _case(rng, (2,), 24, 24, 24, 0.3, np.float64, 4)
    pe1 = pe1.clone()
    pe1[0, 7, 3, 4] = float("nan")            # a NaN interface
    pe1[1, 10, 17, 9] = pe1[1, 8, 17, 9]      # a non-monotone column
    pe1[1, -1, 20, 20] = float("nan")         # a NaN surface pressure (the mask compares against it)
    got = ops.mappm_block_mean(pe1, fields, pe2c, area, arith="exact")
    for g, w in zip(got, _unfused(pe1, fields, pe2c, pfull, area, False, "exact")):
        _same(g, w)
    # the fast mode differs from its unfused twin only inside the redone blocks (they come out in exact arithmetic)
    got = ops.mappm_block_mean(pe1, fields, pe2c, area, arith="fast")
    want = _unfused(pe1, fields, pe2c, pfull, area, False, "fast")
    exact = _unfused(pe1, fields, pe2c, pfull, area, False, "exact")
    touched = {(0, 0, 0), (1, 2, 1), (1, 2, 2)}
    for g, w, e in zip(got, want, exact):
        g, w, e = g.cpu().numpy(), w.cpu().numpy(), e.cpu().numpy()
        for b in range(2):
            for Y in range(3):
                for X in range(3):
                    refs = [e, w] if (b, Y, X) == (1, 2, 2) else [e] if (b, Y, X) in touched else [w]  # (a NaN surface alone: either)
                    assert any(np.array_equal(g[b, :, Y, X], r[b, :, Y, X], equal_nan=True) for r in refs), (b, Y, X)


def test_block_mean_declines_what_it_does_not_take():
    rng = np.random.default_rng(1)
    pe1, fields, pe2c, pfull, area = _case(rng, (1,), 12, 16, 16, 0.3, np.float64, 1)
    assert ops.mappm_block_mean(pe1, fields, pe2c, area.double()) is None          # float64 area
    assert ops.mappm_block_mean(pe1, fields, pe2c, area, factor=4) is None
    assert ops.mappm_block_mean(pe1, fields, pe2c, area, kord=7) is None
    with pytest.raises(ValueError):
        ops.mappm_block_mean(pe1, fields, pe2c[..., :1, :], area)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_block_mean_c384_tile(arith):
    """One C384 tile at full depth (147 456 columns, 2 304 blocks x 79 levels x 4 fields), configs[2] data."""
    rng = np.random.default_rng(2)
    pe1, fields, pe2c, pfull, area = _case(rng, (1,), 79, 384, 384, 1.0, np.float64, 4)
    got = ops.mappm_block_mean(pe1, fields, pe2c, area, arith=arith)
    for g, w in zip(got, _unfused(pe1, fields, pe2c, pfull, area, False, arith)):
        _same(g, w)


@pytest.mark.parametrize("method", ["pressure", "blended"])
def test_pipelines_with_the_fused_route_are_unchanged(method, monkeypatch):
    """FV3NET_AMD_FUSED_BLOCK_MEAN=1: the pressure-level and blended restart pipelines (C32 -> C4, factor 8, float64
    restarts of the fixture schema) with the cell-centred fields through the fused kernel -- every variable of every
    category identical to the default route, labels included."""
    from fv3net_amd.cubedsphere import coarsen_restarts_on_pressure, coarsen_restarts_via_blended_method
    from fv3net_amd.xr_compat import DataArray, Dataset
    import coarsen_restarts_cases as cases

    meta, _ = cases.load()
    inp = cases.medium_inputs(meta, 32, 16, seed=4)

    def dataset(category):
        return Dataset({v: DataArray(a, dims=d, name=v) for v, (d, a) in inp[category].items()})

    restarts = {c: dataset(c) for c in ("fv_core.res", "fv_tracer.res", "fv_srf_wnd.res", "sfc_data")}
    grid_spec = dataset("grid")
    fn = coarsen_restarts_on_pressure if method == "pressure" else coarsen_restarts_via_blended_method
    kwargs = {"coarsen_agrid_winds": True}
    calls = []
    real = ops.mappm_block_mean
    monkeypatch.setattr(ops, "mappm_block_mean", lambda *a, **k: calls.append(1) or real(*a, **k))
    monkeypatch.setenv("FV3NET_AMD_FUSED_BLOCK_MEAN", "0")
    want = fn(8, grid_spec, 300.0, restarts, **kwargs)
    assert not calls
    monkeypatch.setenv("FV3NET_AMD_FUSED_BLOCK_MEAN", "1")
    got = fn(8, grid_spec, 300.0, restarts, **kwargs)
    assert calls, "the fused kernel did not run"
    n = 0
    for category in want:
        assert list(got[category]) == list(want[category])
        for var in want[category]:
            g, w = got[category][var], want[category][var]
            assert g.dims == w.dims and g.dtype == w.dtype, (category, var)
            assert set(g.coords) == set(w.coords), (category, var)
            for c in w.coords:
                np.testing.assert_array_equal(np.asarray(g.coords[c]), np.asarray(w.coords[c]))
            assert np.array_equal(np.asarray(g.values), np.asarray(w.values), equal_nan=True), (category, var)
            n += 1
    assert n >= 55


def test_side_streams_and_the_one_stream_switch(monkeypatch):
    """The pipelines' two side streams come from a one-time probe of the hardware queues (cubedsphere/_device.py); with
    FV3NET_AMD_PIPELINE_STREAMS=0 the three-dimensional branches share the calling stream -- the same values either way."""
    from fv3net_amd.cubedsphere import coarsen_restarts_on_pressure
    from fv3net_amd.cubedsphere._device import side_streams
    from fv3net_amd.xr_compat import DataArray, Dataset
    import coarsen_restarts_cases as cases

    dev = _dev()
    a, b = side_streams(dev)
    main = torch.cuda.current_stream(dev)
    assert len({a.cuda_stream, b.cuda_stream, main.cuda_stream}) == 3
    assert [s.cuda_stream for s in side_streams(dev)] == [a.cuda_stream, b.cuda_stream]   # cached
    meta, _ = cases.load()
    inp = cases.medium_inputs(meta, 32, 16, seed=9)
    dataset = lambda c: Dataset({v: DataArray(x, dims=d, name=v) for v, (d, x) in inp[c].items()})
    restarts = {c: dataset(c) for c in ("fv_core.res", "fv_tracer.res", "fv_srf_wnd.res", "sfc_data")}
    grid_spec = dataset("grid")
    monkeypatch.setenv("FV3NET_AMD_PIPELINE_STREAMS", "1")
    two = coarsen_restarts_on_pressure(8, grid_spec, 300.0, restarts, coarsen_agrid_winds=True)
    monkeypatch.setenv("FV3NET_AMD_PIPELINE_STREAMS", "0")
    one = coarsen_restarts_on_pressure(8, grid_spec, 300.0, restarts, coarsen_agrid_winds=True)
    for category in two:
        for var in two[category]:
            assert np.array_equal(np.asarray(two[category][var].values), np.asarray(one[category][var].values), equal_nan=True), (category, var)


def test_the_adaptive_route_follows_the_data(monkeypatch):
    """FV3NET_AMD_FUSED_BLOCK_MEAN=auto (the default): the first call of a shape is fused; the kernel's count of blocks whose
    waves gave up summing decides the next ones -- the three launches on BASELINE configs[2]'s iid thicknesses, the fused
    kernel on smooth ones -- and the values never depend on the route."""
    from fv3net_amd.cubedsphere import regridz
    from fv3net_amd.xr_compat import DataArray, Dataset

    rng = np.random.default_rng(8)
    dims = ["tile", "zaxis_1", "yaxis_2", "xaxis_1"]
    calls = []
    real = ops.mappm_block_mean
    monkeypatch.setattr(ops, "mappm_block_mean", lambda *a, **k: calls.append(1) or real(*a, **k))
    for spread, expect_fused_later in ((1.0, False), (0.05, True)):
        nt, nz, n = 2, 79, 32
        delp = np.maximum(900.0 + spread * (rng.uniform(300, 1500, (nt, nz, n, n)) - 900.0), 20.0)
        ds = Dataset({k: DataArray(rng.uniform(-5, 5, (nt, nz, n, n)), dims=dims) for k in ("a", "b", "c")})
        delp_da, area = DataArray(delp, dims=dims), DataArray(rng.uniform(0.5, 1, (nt, n, n)).astype(np.float32), dims=["tile", "yaxis_2", "xaxis_1"])
        monkeypatch.setenv("FV3NET_AMD_FUSED_BLOCK_MEAN", "0")
        want = regridz.area_weighted_pressure_means(ds, delp_da, area, 300.0, 8)
        monkeypatch.setenv("FV3NET_AMD_FUSED_BLOCK_MEAN", "auto")
        regridz._FUSED_ROUTE.clear()
        seen = []
        for _ in range(4):
            del calls[:]
            got = regridz.area_weighted_pressure_means(ds, delp_da, area, 300.0, 8)
            torch.cuda.synchronize()
            seen.append(bool(calls))
            for v in want:
                assert np.array_equal(np.asarray(got[v].values), np.asarray(want[v].values), equal_nan=True), (spread, v)
        assert seen[0] is True and seen[-1] is expect_fused_later, (spread, seen)
