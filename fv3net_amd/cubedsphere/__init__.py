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
from .coarsen_restarts import (
    coarsen_restarts_on_pressure,
    coarsen_restarts_on_sigma,
    coarsen_restarts_via_blended_method,
)
from .grid import FV3_FACE_CONNECTIONS, interp_center_to_outer
from .sfc_data import coarse_grain_sfc_data
from .regridz import (
    compute_edge_delp,
    regrid_to_area_weighted_pressure,
    regrid_to_edge_weighted_pressure,
    regrid_vertical,
)

__all__ = [
    "add_coordinates", "block_coarsen", "block_edge_coarsen", "block_edge_sum", "block_median", "block_upsample",
    "block_upsample_like", "coarsen_coords", "coarsen_coords_coord_func", "constants", "edge_weighted_block_average",
    "horizontal_block_reduce", "regrid_to_area_weighted_pressure", "regrid_to_edge_weighted_pressure", "regrid_vertical",
    "compute_edge_delp", "coarse_grain_sfc_data", "coarsen_restarts_on_pressure", "coarsen_restarts_on_sigma",
    "coarsen_restarts_via_blended_method", "interp_center_to_outer", "FV3_FACE_CONNECTIONS", "weighted_block_average",
    "xarray_block_reduce",
]
