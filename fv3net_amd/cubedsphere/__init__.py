"""``vcm.cubedsphere`` entry points on the hot path, computed on the device."""
from . import constants
from .coarsen import (
    _block_mode,
    add_coordinates,
    block_coarsen,
    block_edge_coarsen,
    block_edge_sum,
    block_median,
    block_upsample,
    block_upsample_like,
    coarsen_coords,
    coarsen_coords_coord_func,
    edge_weighted_block_average,
    horizontal_block_reduce,
    weighted_block_average,
    xarray_block_reduce,
)
from .regridz import regrid_to_area_weighted_pressure, regrid_vertical

__all__ = [
    "add_coordinates", "block_coarsen", "block_edge_coarsen", "block_edge_sum", "block_median", "block_upsample",
    "block_upsample_like", "coarsen_coords", "coarsen_coords_coord_func", "constants", "edge_weighted_block_average",
    "horizontal_block_reduce", "regrid_to_area_weighted_pressure", "regrid_vertical", "weighted_block_average",
    "xarray_block_reduce",
]
