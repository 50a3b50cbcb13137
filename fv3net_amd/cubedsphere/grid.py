"""Cube-face connectivity and the centre-to-edge interpolation the coarsening path takes from
xgcm (``vcm.cubedsphere.xgcm.create_fv3_grid(...).interp``; external/vcm/vcm/cubedsphere/xgcm.py:7-97,
used at regridz.py:123-135 and coarsen_restarts.py:825-853).

xgcm itself is not part of the reference tree (xgcm==0.6.1, constraints.txt:316).  What it
computes here: every tile is padded with one row of its two neighbours along the interpolated axis
-- the left neighbour's last line along its connecting axis, the right neighbour's first; a
neighbour connected through its *other* axis contributes that line reversed -- then
``0.5 * (left + right)`` on the n+1 cell edges.  The reference's pressure-level regression
fixtures (``u``, ``v``) pin this on all 12 cube edges (tests/test_oracle_coarsen.py).

The padding rows are the only data that ever crosses tiles on the coarse-graining path: with tiles
sharded over GPUs they are what ``fv3net_amd.parallel.exchange_edge_rows`` all-gathers.
"""
from typing import Hashable

import numpy as np
import torch

from .. import ops
from ..xr_compat import DataArray, from_compat, to_compat
from ._device import like_input, on_device
from .constants import COORD_X_CENTER, COORD_X_OUTER, COORD_Y_CENTER, COORD_Y_OUTER

# xgcm.py:7-34: tile -> axis -> ((left neighbour, its axis), (right neighbour, its axis)); none "reversed"
FV3_FACE_CONNECTIONS = {
    0: {"x": ((4, "y"), (1, "x")), "y": ((5, "y"), (2, "x"))},
    1: {"x": ((0, "x"), (3, "y")), "y": ((5, "x"), (2, "y"))},
    2: {"x": ((0, "y"), (3, "x")), "y": ((1, "y"), (4, "x"))},
    3: {"x": ((2, "x"), (5, "y")), "y": ((1, "x"), (4, "y"))},
    4: {"x": ((2, "y"), (5, "x")), "y": ((3, "y"), (0, "x"))},
    5: {"x": ((4, "x"), (1, "y")), "y": ((3, "x"), (0, "y"))},
}

# index of a tile's boundary vector in ops.cube_edge_rows: (axis, first/last line along it)
_ROW = {("x", "first"): 0, ("x", "last"): 1, ("y", "first"): 2, ("y", "last"): 3}


def halos_from_rows(rows: torch.Tensor, tiles, axis: str):
    """(lo, hi) [len(tiles), ..., n] for ``axis`` from the boundary vectors of ALL six tiles
    (``rows`` [6, 4, ..., n] as returned by ``ops.cube_edge_rows``): one pick-and-orient kernel, no copies."""
    nbr, row, flip = [[], []], [[], []], [[], []]
    for t in tiles:
        (ln, la), (rn, ra) = FV3_FACE_CONNECTIONS[int(t)][axis]
        for side, (n_, a_, which) in enumerate(((ln, la, "last"), (rn, ra, "first"))):
            nbr[side].append(n_)
            row[side].append(_ROW[(a_, which)])
            flip[side].append(0 if a_ == axis else 1)
    if not rows.is_cuda:  # (CPU-backend tests of the exchange: the same picks with array indexing)
        pick = lambda side: torch.stack([rows[n_, r_].flip(-1) if f_ else rows[n_, r_]
                                         for n_, r_, f_ in zip(nbr[side], row[side], flip[side])])
        return pick(0), pick(1)
    both = ops.halo_pick(rows, nbr[0] + nbr[1], row[0] + row[1], flip[0] + flip[1])
    return both[0], both[1]


def interp_tiles_to_edges(field: torch.Tensor, axis: str, step: int = 1) -> torch.Tensor:
    """[6, ..., n, n] cell-centred -> cell edges along ``axis`` ('x': [6, ..., n, n+1]; 'y':
    [6, ..., n+1, n]), all six tiles resident on one device.  ``step`` > 1: every step-th edge only."""
    if field.shape[0] != 6:
        raise ValueError("The leading dimension must hold the six tiles of the cube")
    rows = ops.cube_edge_rows(field)
    lo, hi = halos_from_rows(rows, range(6), axis)
    return ops.interp_center_to_outer(field, lo, hi, 0 if axis == "x" else 1, step=step)


def _validate_tile_coord(da: DataArray):
    """All six tiles -- or, with ``torch.distributed`` initialised, exactly the tiles this rank owns
    (``parallel.tiles_of_rank``): the cube is then sharded by tile and the halo rows are exchanged."""
    from ..parallel import tiles_of_rank, world

    if "tile" not in da.dims:
        raise ValueError("The input Dataset must have a `tile` coordinate.")
    rank, size = world()
    mine = list(range(6)) if size == 1 else tiles_of_rank(size, rank)
    if "tile" in da.coords and sorted(np.asarray(da.coords["tile"]).tolist()) != mine:
        raise ValueError(f"`tile` coordinate must contain each of {mine}")
    if da.sizes["tile"] != len(mine):
        raise ValueError(f"`tile` coordinate must contain each of {mine}")
    return mine


def interp_center_to_outer(da, axis: str, x_center: Hashable = COORD_X_CENTER, x_outer: Hashable = COORD_X_OUTER,
                           y_center: Hashable = COORD_Y_CENTER, y_outer: Hashable = COORD_Y_OUTER, step: int = 1):
    """``create_fv3_grid(ds, ...).interp(da, axis)`` for a cell-centred array: the result carries the
    outer dimension name in place of the centre one, with coordinate 0..n (xgcm.py:42-97).
    ``step`` > 1 (not in the reference): only every step-th of the n + 1 edges, ``.isel({outer: slice(None, None, step)})`` of the
    full result without computing the rest."""
    if axis not in ("x", "y"):
        raise ValueError(f"axis must be 'x' or 'y', got {axis!r}")
    d = to_compat(da)
    mine = _validate_tile_coord(d)
    for dim in (x_center, y_center):
        if dim not in d.dims:
            raise ValueError(f"{dim!r} is not a dimension of the array")
    outer = [dim for dim in d.dims if dim not in ("tile", y_center, x_center)]
    order = ["tile"] + outer + [y_center, x_center]
    t = on_device(d.transpose(*order).data)
    if "tile" in d.coords:  # tiles in coordinate order
        perm = np.argsort(np.asarray(d.coords["tile"]))
        if not np.array_equal(perm, np.arange(len(mine))):
            t = t[torch.as_tensor(perm, device=t.device)]
    if len(mine) == 6:
        res = interp_tiles_to_edges(t, axis, step=step)
    else:  # tile-sharded cube: the one exchange step of the path (parallel.exchange_edge_rows)
        from ..parallel import interp_tiles_to_edges_sharded

        res = interp_tiles_to_edges_sharded(t, axis, step=step)
    new_dim = x_outer if axis == "x" else y_outer
    old_dim = x_center if axis == "x" else y_center
    dims = tuple(new_dim if dim == old_dim else dim for dim in order)
    coords = {k: v for k, v in d.coords.items() if k != old_dim and k != "tile"}
    coords["tile"] = np.asarray(mine)
    coords[new_dim] = np.arange(res.shape[-1 if axis == "x" else -2]) * int(step)
    out = DataArray(like_input(res, d.data), dims=dims, coords=coords, name=d.name, attrs=d.attrs)
    out = out.transpose(*[new_dim if dim == old_dim else dim for dim in d.dims])
    return from_compat(out, da)
