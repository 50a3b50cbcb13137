"""The N > 1 path on CPU: two gloo ranks partition the columns, each runs its own range with
no data-path exchange, and the optional gather reassembles the cube on rank 0."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    """Stands in for MlpModel (which needs a GPU): a fixed linear map per column."""

    def __init__(self):
        self.w = torch.arange(12, dtype=torch.float32).reshape(3, 4) / 7

    def predict(self, sources):
        return {"y": self.w @ sources["a"], "s": sources["a"].sum(dim=0, keepdim=True)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, size, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    from fv3net_amd import parallel

    assert parallel.world() == (rank, size)
    a = torch.arange(4 * n, dtype=torch.float32).reshape(4, n)
    local = parallel.predict_sharded(_FakeModel(), {"a": a})
    lo, hi = parallel.column_range(n, size, rank)
    assert local["y"].shape == (3, hi - lo)
    gathered = parallel.predict_sharded(_FakeModel(), {"a": a}, gather=True)
    if rank == 0:
        np.save(os.path.join(out_dir, "y.npy"), gathered["y"].numpy())
        np.save(os.path.join(out_dir, "s.npy"), gathered["s"].numpy())
    else:
        assert gathered["y"] is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_column_sharding(tmp_path):
    n = 1001  # not divisible by the world size
    mp.spawn(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    a = torch.arange(4 * n, dtype=torch.float32).reshape(4, n)
    ref = _FakeModel().predict({"a": a})
    np.testing.assert_array_equal(np.load(tmp_path / "y.npy"), ref["y"].numpy())
    np.testing.assert_array_equal(np.load(tmp_path / "s.npy"), ref["s"].numpy())


def _halo_worker(rank, size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    from fv3net_amd import parallel
    from fv3net_amd.cubedsphere.grid import halos_from_rows

    rng = np.random.default_rng(5)
    full = torch.from_numpy(rng.uniform(300, 1500, (6, 3, 8, 8)).astype(np.float32))  # same on every rank
    mine = parallel.tiles_of_rank(size, rank)
    local = full[mine]
    # the boundary vectors as ops.cube_edge_rows lays them out (that kernel needs a GPU; the exchange does not)
    rows = torch.stack([local[..., :, 0], local[..., :, -1], local[..., 0, :], local[..., -1, :]], dim=1)
    table = parallel.exchange_edge_rows(rows)
    assert table.shape == (6, 4, 3, 8)
    for axis in ("x", "y"):
        lo, hi = halos_from_rows(table, mine, axis)
        np.save(os.path.join(out_dir, f"lo_{axis}_{rank}.npy"), lo.numpy())
        np.save(os.path.join(out_dir, f"hi_{axis}_{rank}.npy"), hi.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_cube_halo_exchange(tmp_path):
    """Tiles 0-2 on rank 0, 3-5 on rank 1: after the one all-gather every rank pads its tiles
    exactly as the oracle (pinned by the reference's pressure-level u / v fixtures) does."""
    from fv3net_amd import parallel
    from oracle import coarsen_np as onp

    mp.spawn(_halo_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(5)
    full = rng.uniform(300, 1500, (6, 3, 8, 8)).astype(np.float32)
    for axis in ("x", "y"):
        ref = onp.interp_center_to_outer(full, axis)  # 0.5 * (neighbour + first/last cell) at the two ends
        for rank in range(2):
            mine = parallel.tiles_of_rank(2, rank)
            lo = np.load(tmp_path / f"lo_{axis}_{rank}.npy")
            hi = np.load(tmp_path / f"hi_{axis}_{rank}.npy")
            for i, t in enumerate(mine):
                first = full[t][:, :, 0] if axis == "x" else full[t][:, 0, :]
                last = full[t][:, :, -1] if axis == "x" else full[t][:, -1, :]
                edge0 = ref[t][:, :, 0] if axis == "x" else ref[t][:, 0, :]
                edge1 = ref[t][:, :, -1] if axis == "x" else ref[t][:, -1, :]
                np.testing.assert_array_equal(np.float32(0.5) * (lo[i] + first), edge0)
                np.testing.assert_array_equal(np.float32(0.5) * (last + hi[i]), edge1)


def _band_worker(rank, size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    from fv3net_amd import ops, parallel
    from oracle import coarsen_np as onp

    # (the device kernel needs a GPU; what is under test here is the partition, the gather and the sub-group plumbing)
    ops.weighted_block_average = lambda o, w, f: torch.from_numpy(onp.weighted_block_average(o.numpy(), w.numpy(), f))
    n, f = 16, 4
    rng = np.random.default_rng(9)
    field = rng.uniform(-1, 1, (6, 3, n, n)).astype(np.float32)  # the same cube on every rank; each keeps its bands
    area = rng.uniform(0.5, 1, (6, n, n)).astype(np.float32)
    units = parallel.units_of_rank(6, n, f, size, rank)
    objs = [torch.from_numpy(field[t, :, r0:r1]) for t, r0, r1 in units]
    wts = [torch.from_numpy(area[t, r0:r1]) for t, r0, r1 in units]
    bands = parallel.weighted_block_average_banded(objs, wts, f)
    assert [tuple(b.shape) for b in bands] == [(3, (r1 - r0) // f, n // f) for _, r0, r1 in units]
    cube = parallel.weighted_block_average_banded(objs, wts, f, ny=n, gather=True)
    if rank == 0:
        np.save(os.path.join(out_dir, "cube.npy"), cube.numpy())
        np.save(os.path.join(out_dir, "units.npy"), np.asarray(parallel.tile_bands(6, n, f, size), dtype=object), allow_pickle=True)
    else:
        assert cube is None
    # a sub-group (the six tile owners of an 8-rank job are one): ranks 0..size-2 exchange, the last rank stays out
    group = dist.new_group(list(range(size - 1)))
    if rank < size - 1:
        parallel.use_group(group)
        assert parallel.world() == (rank, size - 1)
        mine = parallel.tiles_of_rank(size - 1, rank)
        local = torch.from_numpy(field[mine])
        rows = torch.stack([local[..., :, 0], local[..., :, -1], local[..., 0, :], local[..., -1, :]], dim=1)
        assert parallel.exchange_edge_rows(rows).shape == (6, 4, 3, n)
        parallel.use_group(None)
    assert parallel.world() == (rank, size)
    dist.barrier()
    dist.destroy_process_group()


def test_band_sharded_cube_on_four_ranks(tmp_path):
    """BASELINE configs[4]'s partition on CPU ranks: 6 tiles x row bands over 4 ranks (2 bands per tile, 3 units per
    rank), every band coarsened where it lives with no exchange, the coarse cube gathered on rank 0 equals the
    whole-cube result; a process sub-group carries the tile-sharded halo exchange while the other ranks stay out."""
    from oracle import coarsen_np as onp

    size, n, f = 4, 16, 4
    mp.spawn(_band_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    rng = np.random.default_rng(9)
    field = rng.uniform(-1, 1, (6, 3, n, n)).astype(np.float32)
    area = rng.uniform(0.5, 1, (6, n, n)).astype(np.float32)
    want = onp.weighted_block_average(field, area[:, None], f)
    np.testing.assert_array_equal(np.load(tmp_path / "cube.npy"), want)
    from fv3net_amd import parallel

    plan = parallel.tile_bands(6, n, f, size)
    assert [len(u) for u in plan] == [3, 3, 3, 3] and sorted(sum(plan, [])) == [(t, b * 8, b * 8 + 8) for t in range(6) for b in range(2)]
    assert [len(u) for u in parallel.tile_bands(6, 3072, 8, 8)] == [3] * 8  # configs[4]: 4 bands of 768 rows per tile


def _idle_rank_worker(rank, size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    from fv3net_amd import ops, parallel
    from oracle import coarsen_np as onp

    ops.weighted_block_average = lambda o, w, f: torch.from_numpy(onp.weighted_block_average(o.numpy(), w.numpy(), f))
    n_tiles, n, f = 2, 4, 4  # one block per tile: whole tiles are dealt, ranks 2 and 3 get none
    units = parallel.units_of_rank(n_tiles, n, f, size, rank)
    objs = [torch.ones((3, r1 - r0, n)) for _, r0, r1 in units]
    wts = [torch.ones((r1 - r0, n)) for _, r0, r1 in units]
    try:
        parallel.weighted_block_average_banded(objs, wts, f, n_tiles=n_tiles, ny=n, gather=True)
        outcome = "returned"
    except ValueError as err:
        outcome = "raised" if "own no bands" in str(err) else f"other: {err}"
    with open(os.path.join(out_dir, f"outcome_{rank}.txt"), "w") as fh:
        fh.write(outcome)
    dist.barrier()  # every rank gets here: nobody is left waiting inside a gather
    dist.destroy_process_group()


def test_banded_gather_with_idle_ranks_raises_on_every_rank(tmp_path):
    """with more ranks than units (whole-tile fallback) a rank without bands used to raise alone while the
    others entered ``dist.gather`` and hung.  The partition is known to every rank, so all of them raise before any
    collective."""
    size = 4
    mp.spawn(_idle_rank_worker, args=(size, _free_port(), str(tmp_path)), nprocs=size, join=True)
    assert [open(tmp_path / f"outcome_{r}.txt").read() for r in range(size)] == ["raised"] * size
