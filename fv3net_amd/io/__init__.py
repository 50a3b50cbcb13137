"""Data formats either side of the coarse-graining path, dependency-free:
``zarr_v2``   writer / reader of uncompressed Zarr v2 stores (what ``Dataset.to_zarr`` of the reference's
              coarsen_c384_diagnostics.py:84 and pace's ZarrMonitor produce, minus compression);
``netcdf``    sub-tile restart / diagnostics files in the netCDF classic format through ``scipy.io.netcdf_file``
              (``"{prefix}.tile{tile}.nc.{subtile:04d}"``, external/vcm/vcm/cubedsphere/coarsen.py:27, io.py:6-39);
``pipeline``  host file -> pinned buffer -> device -> coarse zarr with the reads overlapped with the device work.
netCDF-4 / HDF5 files need libraries this image does not have and are out of reach."""
from . import netcdf, zarr_v2  # noqa: F401
from .pipeline import coarsen_subtile_files_to_zarr  # noqa: F401
