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
