"""Dimension and variable names of FV3 restart / diagnostic files used on the coarsening path
(the subset of vcm.cubedsphere.constants the path needs; external/vcm/vcm/cubedsphere/constants.py:1-36)."""
COORD_X_CENTER = "grid_xt"
COORD_X_OUTER = "grid_x"
COORD_Y_CENTER = "grid_yt"
COORD_Y_OUTER = "grid_y"
COORD_Z_CENTER = "pfull"
COORD_Z_OUTER = "phalf"
FV_CORE_X_CENTER = "xaxis_1"
FV_CORE_Y_CENTER = "yaxis_2"
FV_CORE_X_OUTER = "xaxis_2"
FV_CORE_Y_OUTER = "yaxis_1"
FV_SRF_WND_X_CENTER = "xaxis_1"
FV_SRF_WND_Y_CENTER = "yaxis_1"
FV_TRACER_X_CENTER = "xaxis_1"
FV_TRACER_Y_CENTER = "yaxis_1"
RESTART_Z_CENTER = "zaxis_1"
RESTART_Z_OUTER = "zaxis_2"
SFC_DATA_X_CENTER = "xaxis_1"
SFC_DATA_Y_CENTER = "yaxis_1"
TILE_COORDS = range(6)
NUM_TILES = 6
TOA_PRESSURE = 300.0  # Pa; vcm.calc.thermo.constants.TOA_PRESSURE
