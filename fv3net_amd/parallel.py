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


_GROUP = None  # process group the partition functions and the halo exchange run in (None = the default group)


def use_group(group) -> None:
    """Run the sharded entry points of this module inside ``group`` (a ``torch.distributed`` process group) instead of
    the default one -- e.g. the six tile owners of an 8-rank job for the tile-sharded restart pipelines
    (``dist.new_group(range(6))``; the other ranks do not call them).  ``None`` restores the default group."""
    global _GROUP
    _GROUP = group


def world() -> Tuple[int, int]:
    """(rank, world_size) in the active group; (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(_GROUP), dist.get_world_size(_GROUP)
    return 0, 1


def _global(group_rank: int) -> int:
    return group_rank if _GROUP is None else dist.get_global_rank(_GROUP, group_rank)


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


def units_of_rank(n_tiles: int, ny: int, factor: int, world_size: int, rank: int) -> List[Tuple[int, int, int]]:
    """The (tile, row_start, row_stop) row bands ``rank`` owns (:func:`tile_bands`)."""
    return tile_bands(n_tiles, ny, factor, world_size)[rank]


def weighted_block_average_banded(obj_bands: List[torch.Tensor], weight_bands: List[torch.Tensor], factor: int,
                                  n_tiles: int = 6, ny: int = None, gather: bool = False, dst: int = 0):
    """``weighted_block_average`` of a cube sharded by row bands (BASELINE configs[4]: 6 tiles x row bands over 8 GPUs; the
    reference's fine-resolution restarts arrive as such sub-tile files, external/vcm/vcm/cubedsphere/coarsen.py:27).
    ``obj_bands[i]`` [..., rows, nx] and ``weight_bands[i]`` ([rows, nx] or the field's shape) are this rank's units in
    the order of :func:`units_of_rank`; every f x f block lies inside one band, so there is no exchange step.  Returns the
    coarse bands [..., rows / f, nx / f]; with ``gather`` rank ``dst`` gets the assembled coarse cube
    [n_tiles, ..., ny / f, nx / f] (others None) -- the coarse field is 1 / f^2 of the data, one small collective."""
    from . import ops

    coarse = [ops.weighted_block_average(o, w, factor) for o, w in zip(obj_bands, weight_bands)]
    if not gather:
        return coarse
    rank, size = world()
    if ny is None:
        raise ValueError("gather needs the tile's row count ny")
    plan = tile_bands(n_tiles, ny, factor, size)
    if len(coarse) != len(plan[rank]):
        raise ValueError(f"rank {rank} owns {len(plan[rank])} bands, got {len(coarse)}")
    idle = [r for r, units in enumerate(plan) if not units]
    if idle and size > 1:
        # decided from the plan, which every rank computes alike: ALL ranks raise here, before the collective -- a rank
        # without bands has no buffer shape to contribute, and raising on it alone would leave the others in the gather
        raise ValueError(f"ranks {idle} own no bands of this partition ({n_tiles} tiles, {size} ranks): gather over the "
                         "ranks that do (parallel.use_group), or coarsen without gather")
    if size == 1:
        pieces = {0: coarse}
    else:
        # bands of one plan share their shape except when whole tiles were dealt unevenly: pad the count, not the shape
        width = max(len(units) for units in plan)
        ref = coarse[0]
        stacked = torch.zeros((width,) + tuple(ref.shape), dtype=ref.dtype, device=ref.device)
        for i, c in enumerate(coarse):
            stacked[i] = c
        staged = stacked.cpu() if (stacked.is_cuda and dist.get_backend(_GROUP) == "gloo") else stacked
        bufs = [torch.empty_like(staged) for _ in range(size)] if rank == dst else None
        dist.gather(staged, bufs, dst=_global(dst), group=_GROUP)
        if rank != dst:
            return None
        pieces = {r: [bufs[r][i].to(ref.device) for i in range(len(plan[r]))] for r in range(size)}
    first = pieces[0][0]
    out = torch.empty((n_tiles,) + tuple(first.shape[:-2]) + (ny // factor, first.shape[-1]), dtype=first.dtype, device=first.device)
    for r, units in enumerate(plan):
        for (t, r0, r1), c in zip(units, pieces[r]):
            out[t, ..., r0 // factor:r1 // factor, :] = c
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
        dist.gather(padded, bufs, dst=_global(dst), group=_GROUP)
        return torch.cat([b[:, : (hi - lo)] for b, (lo, hi) in zip(bufs, counts)], dim=1)
    dist.gather(padded, None, dst=_global(dst), group=_GROUP)
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
    if padded.is_cuda and dist.get_backend(_GROUP) == "gloo":  # (CPU-backend rehearsals with device data: stage through the host)
        host = padded.cpu()
        bufs = [torch.empty_like(host) for _ in range(size)]
        dist.all_gather(bufs, host, group=_GROUP)
        bufs = [b.to(padded.device) for b in bufs]
    else:
        bufs = [torch.empty_like(padded) for _ in range(size)]
        dist.all_gather(bufs, padded.contiguous(), group=_GROUP)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def interp_tiles_to_edges_sharded(local: torch.Tensor, axis: str, n_tiles: int = 6, step: int = 1) -> torch.Tensor:
    """``cubedsphere.grid.interp_tiles_to_edges`` for tile-sharded data: ``local`` [n_local, ..., n, n]
    holds this rank's tiles (:func:`tiles_of_rank`); the halo rows come from one all-gather."""
    from . import ops
    from .cubedsphere.grid import halos_from_rows

    rank, size = world()
    rows = exchange_edge_rows(ops.cube_edge_rows(local), n_tiles)
    lo, hi = halos_from_rows(rows, tiles_of_rank(size, rank, n_tiles), axis)
    return ops.interp_center_to_outer(local, lo, hi, 0 if axis == "x" else 1, step=step)
