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
