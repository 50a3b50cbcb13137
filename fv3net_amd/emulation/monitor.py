"""The ``store`` hook: saves the microphysics state the Fortran model hands over every ``output_freq_sec``
(external/emulation/emulation/_monitor/monitor.py:28-305).

What is kept from the reference: ``StorageConfig``'s fields; the output times (``elapsed % output_freq_sec == 0`` after
``output_start_sec``, evaluated at ``model_time + dt``); the conversion of a state entry (``np.squeeze(x.astype(float32)).T``
with dims ``[]`` / ``[sample]`` / ``[sample, z]`` / ``[sample, z, category]`` by rank); the variable metadata lookup with the
``_input`` / ``_output`` suffix removed; the netCDF file names ``netcdf_output/state_{%Y%m%d.%H%M%S}_{rank}.nc`` with ``time``
and ``tile`` = rank coordinates; the zarr store ``state_output.zarr``.

What is different: the zarr side is written with this package's dependency-free writer instead of pace's ``ZarrMonitor``
over MPI -- arrays ``[time, rank, *dims]``, one chunk per (time, rank), every rank writing its own chunk files and (the same)
metadata, rank 0 extending the time axis; the netCDF files are classic-format (scipy), not netCDF-4; there is no TFRecord
output (TensorFlow is not part of this stack): ``save_tfrecord=True`` is refused.
"""
import dataclasses
import datetime
import json
import logging
import os
from typing import Any, Mapping, Optional

import numpy as np

from ..io import zarr_v2
from .schedule import julian_seconds, translate_time

logger = logging.getLogger(__name__)

TIME_FMT = "%Y%m%d.%H%M%S"
DIMS_MAP = {0: [], 1: ["sample"], 2: ["sample", "z"], 3: ["sample", "z", "category"]}


@dataclasses.dataclass
class StorageConfig:
    """monitor.py:28-53."""

    var_meta_path: str = ""
    output_freq_sec: int = 10_800
    output_start_sec: int = 0
    save_nc: bool = True
    save_zarr: bool = True
    save_tfrecord: bool = False


def _remove_io_suffix(key: str) -> str:
    for suffix in ("_input", "_output"):
        if key.endswith(suffix):
            return key[: -len(suffix)]
    return key


def _get_attrs(key: str, metadata: Mapping) -> dict:
    meta = metadata.get(_remove_io_suffix(key))
    return {k: json.dumps(v) for k, v in dict(meta).items()} if meta else {}


def _fields(state: Mapping[str, Any], metadata: Mapping):
    """name -> (dims, float32 array [sample, ...], attrs) for the entries the reference would store."""
    out = {}
    for key, data in state.items():
        arr = np.squeeze(np.asarray(data).astype(np.float32)).T
        dims = DIMS_MAP.get(arr.ndim)
        if dims is None:
            logger.info("Skipping %s ... unrecognized dimensions, ndim = %d", key, arr.ndim)
            continue
        attrs = _get_attrs(key, metadata)
        attrs["units"] = attrs.pop("units", "unknown")
        out[key] = (dims, np.ascontiguousarray(arr) if arr.ndim else np.asarray(arr), attrs)  # (ascontiguousarray makes 0-d 1-d)
    return out


def _as_datetime(t) -> datetime.datetime:
    return datetime.datetime(int(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[4]), int(t[5]))


class StorageHook:
    """monitor.py:194-305.  ``rank`` and ``n_ranks``: the MPI rank of this process and the number of ranks (the reference
    asks mpi4py; here they come from the state's ``rank`` entry and the constructor, default one rank)."""

    def __init__(self, output_freq_sec: int, output_start_sec: int = 0, dt_sec: int = 900, metadata: Any = None,
                 save_nc: bool = True, save_zarr: bool = True, save_tfrecord: bool = False, n_ranks: int = 1,
                 directory: Optional[str] = None):
        if save_tfrecord:
            raise ValueError("save_tfrecord needs TensorFlow, which this stack does not use: save_zarr / save_nc only")
        self.name = "emulation storage monitor"
        self.output_freq_sec = output_freq_sec
        self.output_start_sec = output_start_sec
        self.dt_sec = dt_sec
        self.metadata = dict(metadata or {})
        self.save_nc, self.save_zarr = save_nc, save_zarr
        self.n_ranks = int(n_ranks)
        self.directory = directory or os.getcwd()
        self.initial_time = None
        self._n_stored = 0

    def _store_data_at_time(self, seconds: int) -> bool:
        elapsed = seconds - self.initial_time
        return (elapsed % self.output_freq_sec == 0) and (elapsed >= self.output_start_sec)

    def store(self, state: Mapping[str, Any]) -> None:
        state = dict(**state)
        time = translate_time(state.pop("model_time"))
        # (the reference pops `model_time` only: a `rank` entry goes through the field conversion like any other key -- a
        # scalar has no recognised dims and is skipped there, an array of samples would be stored)
        rank = int(np.asarray(state.get("rank", 0)).reshape(-1)[0])
        seconds = julian_seconds(time)
        if self.initial_time is None:
            self.initial_time = seconds
        if not self._store_data_at_time(seconds + self.dt_sec):  # (we are in the middle of the time step)
            return
        # labelled with the UN-incremented model time, as the reference's _store_zarr / _store_netcdf are called
        # (monitor.py:273-281: `time + increment` only decides whether to store)
        when = _as_datetime(time)
        try:
            fields = _fields(state, self.metadata)
            if self.save_zarr:
                self._store_zarr(fields, when, rank)
            if self.save_nc:
                self._store_netcdf(fields, when, rank)
        except Exception:
            logger.critical("Failed to store state with shapes: %s", {k: np.shape(v) for k, v in state.items()})
            raise
        self._n_stored += 1

    # -- zarr: [time, rank, *dims], chunk (1, 1, *shape) ------------------------------------------------
    def _store_zarr(self, fields, when, rank):
        root = os.path.join(self.directory, "state_output.zarr")
        t = self._n_stored
        if t == 0:  # (every rank arrives here in its own process: whoever is first writes the metadata, the rest keep it)
            zarr_v2.create_group(root, exist_ok=True)
        for name, (dims, arr, attrs) in fields.items():
            path = os.path.join(root, name)
            shape = (t + 1, self.n_ranks) + tuple(arr.shape)
            if t == 0:
                zarr_v2.create_array(root, name, shape, (1, 1) + tuple(arr.shape), np.float32, ["time", "rank"] + list(dims), attrs,
                                     exist_ok=True)
            zarr_v2.write_chunk(path, (t, rank) + (0,) * arr.ndim, arr[None, None])
            if rank == 0 and t > 0:
                zarr_v2.set_shape(path, shape)
        if rank == 0:
            stamp = np.array([(when - datetime.datetime(1970, 1, 1)).total_seconds()], dtype=np.float64)
            if t == 0:
                zarr_v2.create_array(root, "time", (1,), (1,), np.float64, ["time"],
                                     {"units": "seconds since 1970-01-01 00:00:00", "calendar": "julian"})
            zarr_v2.write_chunk(os.path.join(root, "time"), (t,), stamp)
            zarr_v2.set_shape(os.path.join(root, "time"), (t + 1,))
            zarr_v2.consolidate(root)

    # -- netCDF classic: one file per (time, rank) ----------------------------------------------------------
    def _store_netcdf(self, fields, when, rank):
        from scipy.io import netcdf_file

        out_dir = os.path.join(self.directory, "netcdf_output")
        os.makedirs(out_dir, exist_ok=True)
        f = netcdf_file(os.path.join(out_dir, f"state_{when.strftime(TIME_FMT)}_{rank}.nc"), "w", version=2)
        made = {}
        for name, (dims, arr, attrs) in fields.items():
            for d, n in zip(dims, arr.shape):
                if made.setdefault(d, n) != n:  # (same-named dims of different length: the reference's xarray would fail too)
                    raise ValueError(f"dimension {d!r} has lengths {made[d]} and {n}")
                if d not in f.dimensions:
                    f.createDimension(d, n)
            var = f.createVariable(name, "f4", tuple(dims))
            if arr.ndim:
                var[:] = arr
            else:  # (scipy's assignValue indexes a scalar variable's 0-d buffer with [:])
                var.data[...] = arr
            for k, v in attrs.items():
                setattr(var, k, v)
        f.time = when.isoformat()
        f.tile = rank
        f.close()

    @classmethod
    def from_config(cls, config: StorageConfig, dt_sec: int = 900, n_ranks: int = 1, directory: Optional[str] = None):
        path = config.var_meta_path or os.environ.get("VAR_META_PATH", "")
        metadata = {}
        if path:
            with open(path) as f:
                metadata = json.load(f)
        return cls(config.output_freq_sec, config.output_start_sec, dt_sec, metadata, config.save_nc, config.save_zarr,
                   config.save_tfrecord, n_ranks, directory)
