"""Sub-tile restart / diagnostics files in the netCDF classic format.

The fine-resolution model writes every tile as ``layout_x * layout_y`` files ``{prefix}.tile{tile}.nc.{subtile:04d}``
(external/vcm/vcm/cubedsphere/coarsen.py:27), each holding its rectangle of every variable together with the coordinate
values (1-based global indices along ``xaxis_1``, ``yaxis_2``, ...) that say where the rectangle sits; the reference
reassembles a tile with ``xr.combine_by_coords`` per variable (external/vcm/vcm/cubedsphere/io.py:6-28).  Here the same
placement is done directly: the union of the coordinate values along each dimension gives the tile's axes, each
rectangle is copied to its index range -- straight into a caller-supplied (pinned) host buffer when one is given, so that
the upload to the device can overlap the next read.

``scipy.io.netcdf_file`` reads and writes the classic format (CDF-1 / CDF-2, the "64-bit offset" variant included) with
memory mapping; netCDF-4 files are HDF5 containers and cannot be read without libraries this image lacks.
"""
import os
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

SUBTILE_FILE_PATTERN = "{prefix}.tile{tile:d}.nc.{subtile:04d}"
NUM_TILES = 6


def subtile_filenames(prefix: str, tile: int, num_subtiles: int = 16, pattern: str = SUBTILE_FILE_PATTERN) -> List[str]:
    return [pattern.format(prefix=prefix, tile=tile, subtile=s) for s in range(num_subtiles)]


def all_filenames(prefix: str, num_subtiles: int = 16, pattern: str = SUBTILE_FILE_PATTERN) -> List[str]:
    """io.py:35-39 of the reference: the files of all six tiles (1-based tile numbers)."""
    return [name for tile in range(1, NUM_TILES + 1) for name in subtile_filenames(prefix, tile, num_subtiles, pattern)]


def _open(path: str, mode: str = "r"):
    from scipy.io import netcdf_file

    return netcdf_file(path, mode, mmap=(mode == "r"), version=2)


_POOL = None


def _copy_pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        workers = int(os.environ.get("FV3NET_AMD_IO_THREADS", "0")) or min(4, os.cpu_count() or 1)
        _POOL = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="fv3net-amd-nc")
    return _POOL


class SubtileSet:
    """The sub-tile files of one tile, opened (memory-mapped) together."""

    def __init__(self, paths: Sequence[str]):
        missing = [p for p in paths if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(f"sub-tile files not found: {missing[:3]}{' ...' if len(missing) > 3 else ''}")
        self.files = [_open(p) for p in paths]
        first = self.files[0]
        self.dims: Dict[str, Tuple[str, ...]] = {}
        self.dtypes: Dict[str, np.dtype] = {}
        for name, var in first.variables.items():
            if name in first.dimensions and var.dimensions == (name,):
                continue  # a coordinate variable
            self.dims[name] = tuple(var.dimensions)
            self.dtypes[name] = np.dtype(var.data.dtype).newbyteorder("=")
        # the tile's axes: union of the files' coordinate values; a dimension without a coordinate variable is taken
        # whole from every file (it is not decomposed: levels, time)
        self.axes: Dict[str, Optional[np.ndarray]] = {}
        self.sizes: Dict[str, int] = {}
        for dim in {d for ds in self.dims.values() for d in ds}:
            if dim in first.variables and first.variables[dim].dimensions == (dim,):
                values = np.unique(np.concatenate([np.asarray(f.variables[dim].data, dtype=np.float64) for f in self.files]))
                self.axes[dim] = values
                self.sizes[dim] = int(values.size)
            else:
                self.axes[dim] = None
                length = first.dimensions[dim]
                if length is None:  # the record dimension: its length is that of any variable along it
                    owner = next(n for n, ds in self.dims.items() if dim in ds)
                    length = first.variables[owner].shape[self.dims[owner].index(dim)]
                self.sizes[dim] = int(length)

    def shape(self, name: str) -> Tuple[int, ...]:
        return tuple(self.sizes[d] for d in self.dims[name])

    def read(self, name: str, out: Optional[np.ndarray] = None) -> np.ndarray:
        """The whole tile of variable ``name``; ``out`` (native byte order, the tile's shape) is filled in place."""
        shape = self.shape(name)
        if out is None:
            out = np.empty(shape, dtype=self.dtypes[name])
        elif tuple(out.shape) != shape:
            raise ValueError(f"out has shape {out.shape}, variable {name!r} of this tile has {shape}")
        covered, pieces = 0, []
        for f in self.files:
            var = f.variables[name]
            sel = []
            for dim in var.dimensions:
                axis = self.axes[dim]
                if axis is None:
                    sel.append(slice(None))
                    continue
                c = np.asarray(f.variables[dim].data, dtype=np.float64)
                i0 = int(np.searchsorted(axis, c[0]))
                if not np.array_equal(axis[i0:i0 + c.size], c):
                    raise ValueError(f"coordinate {dim!r} of {f.filename} is not a contiguous run of the tile's axis")
                sel.append(slice(i0, i0 + c.size))
            pieces.append((tuple(sel), var.data))
            covered += int(np.prod(var.shape))
        if covered != int(np.prod(shape)):
            raise ValueError(f"the sub-tiles of {name!r} cover {covered} of {int(np.prod(shape))} points")
        # the copies byte-swap the files' big-endian values into place; numpy releases the GIL inside them, so the sub-tiles of a
        # tile are copied side by side (page-cache reads at 3 GB/s with one thread)
        list(_copy_pool().map(lambda piece: np.copyto(out[piece[0]], piece[1]), pieces))
        return out

    def coords(self) -> Dict[str, np.ndarray]:
        return {d: v for d, v in self.axes.items() if v is not None}

    def close(self):
        for f in self.files:
            try:
                f.close()
            except Exception:  # noqa: BLE001  (scipy warns when mmapped arrays are still referenced)
                pass
        self.files = []


def open_tile(prefix: str, tile: int, num_subtiles: int = 16, pattern: str = SUBTILE_FILE_PATTERN) -> SubtileSet:
    """The sub-tile files of 1-based ``tile``."""
    return SubtileSet(subtile_filenames(prefix, tile, num_subtiles, pattern))


def read_tile_dataset(prefix: str, tile: int, num_subtiles: int = 16, variables: Optional[Sequence[str]] = None):
    """One tile as an ``xr_compat.Dataset`` of numpy arrays (``combine_subtiles`` of the reference)."""
    from ..xr_compat import DataArray, Dataset

    tiles = open_tile(prefix, tile, num_subtiles)
    try:
        coords = tiles.coords()
        ds = Dataset()
        for name in (variables or tiles.dims):
            dims = tiles.dims[name]
            ds[name] = DataArray(tiles.read(name), dims=list(dims), name=name,
                                 coords={d: coords[d] for d in dims if d in coords})
        return ds
    finally:
        tiles.close()


def write_subtile_files(prefix: str, tile: int, variables: Mapping[str, Tuple[Sequence[str], np.ndarray]],
                        layout: Tuple[int, int] = (4, 4), x_dims: Sequence[str] = ("xaxis_1", "xaxis_2", "grid_xt", "grid_x"),
                        y_dims: Sequence[str] = ("yaxis_1", "yaxis_2", "grid_yt", "grid_y")) -> List[str]:
    """Split the tile's variables (name -> (dims, array)) into ``layout[0] x layout[1]`` rectangles and write them the way
    the model does (synthetic inputs for tests and benchmarks): classic 64-bit-offset files, 1-based index coordinates
    along the decomposed dimensions, staggered dimensions giving their extra point to the last rectangle."""
    lx, ly = layout
    paths = subtile_filenames(prefix, tile, lx * ly)
    os.makedirs(os.path.dirname(os.path.abspath(paths[0])), exist_ok=True)

    def cuts(n, parts):  # centred points split evenly; an odd (staggered) size leaves its last point to the last part
        base = (n - n % 2) // parts if n % parts else n // parts
        edges = [i * base for i in range(parts)] + [n]
        return [(edges[i], edges[i + 1]) for i in range(parts)]

    for j in range(ly):
        for i in range(lx):
            f = _open(paths[j * lx + i], "w")
            made = set()
            for name, (dims, data) in variables.items():
                data = np.asarray(data)
                sel = []
                for d, n in zip(dims, data.shape):
                    lo, hi = (cuts(n, lx)[i] if d in x_dims else cuts(n, ly)[j] if d in y_dims else (0, n))
                    sel.append(slice(lo, hi))
                    if d not in made:
                        f.createDimension(d, hi - lo)
                        if d in x_dims or d in y_dims:
                            c = f.createVariable(d, "f8", (d,))
                            c[:] = np.arange(lo + 1, hi + 1, dtype=np.float64)
                        made.add(d)
                var = f.createVariable(name, data.dtype.newbyteorder(">").char if data.dtype.kind != "f" else ("f4" if data.dtype.itemsize == 4 else "f8"), tuple(dims))
                var[:] = data[tuple(sel)]
            f.close()
    return paths
