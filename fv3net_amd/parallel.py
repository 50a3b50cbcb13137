"""One process per GPU: how the path shards and what (little) it exchanges.

Every unit of work on this path is independent -- a column for the MLP and the remap, an
f x f block for the horizontal coarsening -- so ranks never exchange data while computing
(the reference's MPI ranks are equally independent,
workflows/prognostic_c48_run/runtime/steppers/machine_learning.py:176-181), with ONE exception:
the pressure thickness interpolated to the cell edges for the D-grid winds needs the adjacent row
of the neighbouring cube faces (regridz.py:123-135) -- a one-cell halo, exchanged here as an
all-gather of every tile's four boundary vectors (C3072: 4 x 79 x 3072 floats = 3.9 MB per tile).
Besides that this module provides the partition (cube tiles / row bands / column ranges over
ranks) and the optional gather of results to one consumer, on ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in CPU tests).
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def column_range(n_columns: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of the columns owned by ``rank``."""
    base, extra = divmod(n_columns, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def tile_bands(n_tiles: int, ny: int, factor: int, world_size: int) -> List[List[Tuple[int, int, int]]]:
    """Partition ``n_tiles`` tiles of ``ny`` rows into (tile, row_start, row_stop) units, one list
    per rank.  Rows are cut in multiples of the coarsening factor so that no f x f block
    straddles two ranks (blocks never straddle tiles or sub-tiles in the reference either,
    external/vcm/vcm/cubedsphere/coarsen.py:83-132).  Bands per tile is the smallest count that
    makes tiles * bands a multiple of world_size (6 tiles on 8 GPUs -> 4 bands, 3 units each);
    if the rows cannot be cut that finely, whole tiles are dealt round-robin."""
    if ny % factor:
        raise ValueError(f"{ny} rows are not a multiple of the coarsening factor {factor}")
    blocks = ny // factor
    bands = next((b for b in range(1, blocks + 1) if (n_tiles * b) % world_size == 0 and blocks % b == 0), None)
    units: List[Tuple[int, int, int]] = []
    if bands is None:
        units = [(t, 0, ny) for t in range(n_tiles)]
    else:
        rows = (blocks // bands) * factor
        units = [(t, b * rows, (b + 1) * rows) for t in range(n_tiles) for b in range(bands)]
    out: List[List[Tuple[int, int, int]]] = [[] for _ in range(world_size)]
    per = -(-len(units) // world_size)
    for i, u in enumerate(units):
        out[min(i // per, world_size - 1)].append(u)
    return out


def gather_columns(local: torch.Tensor, n_columns: int, dst: int = 0):
    """Gather per-rank ``[features, n_local]`` blocks (split by :func:`column_range`) to rank
    ``dst`` as ``[features, n_columns]``; returns None on the other ranks.  The only collective
    on the path, and optional: needed only when a single consumer wants the whole cube."""
    rank, size = world()
    if size == 1:
        return local
    counts = [column_range(n_columns, size, r) for r in range(size)]
    width = max(b - a for a, b in counts)
    padded = torch.zeros((local.shape[0], width), dtype=local.dtype, device=local.device)
    padded[:, : local.shape[1]] = local
    if rank == dst:
        bufs = [torch.empty_like(padded) for _ in range(size)]
        dist.gather(padded, bufs, dst=dst)
        return torch.cat([b[:, : (hi - lo)] for b, (lo, hi) in zip(bufs, counts)], dim=1)
    dist.gather(padded, None, dst=dst)
    return None


def predict_sharded(model, sources: Dict[str, torch.Tensor], gather: bool = False, dst: int = 0):
    """Run ``model.predict`` (an :class:`fv3net_amd.mlp.MlpModel` or anything with the same call)
    on this rank's column range of ``[feature, sample]`` sources.  With ``gather`` the outputs are
    collected on rank ``dst``."""
    rank, size = world()
    n = next(iter(sources.values())).shape[-1]
    lo, hi = column_range(n, size, rank)
    local = {k: v[..., lo:hi] for k, v in sources.items()}
    outs = model.predict(local)
    if not gather:
        return outs
    return {k: gather_columns(v, n, dst) for k, v in outs.items()}


def tiles_of_rank(world_size: int, rank: int, n_tiles: int = 6) -> List[int]:
    """Whole cube tiles owned by ``rank`` (contiguous, balanced; ranks beyond the tile count own none)."""
    lo, hi = column_range(n_tiles, world_size, rank)
    return list(range(lo, hi))


def exchange_edge_rows(local_rows: torch.Tensor, n_tiles: int = 6) -> torch.Tensor:
    """All-gather the boundary vectors of the tiles each rank owns (``local_rows``
    [n_local, 4, ..., n], from ``ops.cube_edge_rows``; tiles dealt by :func:`tiles_of_rank`) into the
    full [n_tiles, 4, ..., n] table every rank needs to pad its tiles.  The one exchange step of the
    coarse-graining path; a no-op without ``torch.distributed``."""
    rank, size = world()
    if size == 1:
        return local_rows
    counts = [len(tiles_of_rank(size, r, n_tiles)) for r in range(size)]
    width = max(counts)
    if local_rows.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} owns {counts[rank]} tiles, got rows for {local_rows.shape[0]}")
    padded = torch.zeros((width,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    padded[: counts[rank]] = local_rows
    if padded.is_cuda and dist.get_backend() == "gloo":  # (CPU-backend rehearsals with device data: stage through the host)
        host = padded.cpu()
        bufs = [torch.empty_like(host) for _ in range(size)]
        dist.all_gather(bufs, host)
        bufs = [b.to(padded.device) for b in bufs]
    else:
        bufs = [torch.empty_like(padded) for _ in range(size)]
        dist.all_gather(bufs, padded.contiguous())
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def interp_tiles_to_edges_sharded(local: torch.Tensor, axis: str, n_tiles: int = 6) -> torch.Tensor:
    """``cubedsphere.grid.interp_tiles_to_edges`` for tile-sharded data: ``local`` [n_local, ..., n, n]
    holds this rank's tiles (:func:`tiles_of_rank`); the halo rows come from one all-gather."""
    from . import ops
    from .cubedsphere.grid import halos_from_rows

    rank, size = world()
    rows = exchange_edge_rows(ops.cube_edge_rows(local), n_tiles)
    lo, hi = halos_from_rows(rows, tiles_of_rank(size, rank, n_tiles), axis)
    return ops.interp_center_to_outer(local, lo, hi, 0 if axis == "x" else 1)
