"""Host files -> device -> coarse zarr, end to end (the shape of the reference's
workflows/coarsen_c384_diagnostics/coarsen_c384_diagnostics.py:64-89, whose inputs are zarr / sub-tile netCDF and whose
output is ``Dataset.to_zarr``).

Per tile: a reader thread assembles the tile's variables from their sub-tile files straight into PINNED host buffers
(two sets, used alternately) while the device works on the previous tile: upload on a copy stream, block average with
the tile's area weights, download of the coarse result.  At C3072 the kernel runs at ~6 TB/s and a tile's file data
arrives at disk / page-cache speed: the pipeline is I/O-bound by two to three orders of magnitude, which is why the reads
are what is overlapped.
"""
import threading
import time
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from .. import ops
from . import netcdf, zarr_v2


def coarsen_subtile_files_to_zarr(prefix: str, out_path: str, area: np.ndarray, coarsening_factor: int,
                                  variables: Optional[Sequence[str]] = None, num_subtiles: int = 16,
                                  x_dim: str = "xaxis_1", y_dim: str = "yaxis_1", attrs: Optional[Mapping] = None,
                                  device: Optional[torch.device] = None) -> Dict[str, float]:
    """``weighted_block_average`` of the variables stored in ``{prefix}.tile{1..6}.nc.{0000..}`` with the weights
    ``area`` [6, ny, nx], written to the zarr store ``out_path`` with a leading ``tile`` dimension.  Variables whose last
    two dims are not (``y_dim``, ``x_dim``) are skipped.  Returns the wall-clock seconds spent reading, on the device
    (uploads, kernels, downloads; overlapped with the reads) and writing, and the byte counts."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    f = int(coarsening_factor)
    area_t = torch.as_tensor(np.ascontiguousarray(area)).to(dev)
    copy_stream = torch.cuda.Stream(device=dev)
    t_read = t_dev = 0.0
    bytes_in = 0
    sets = [netcdf.open_tile(prefix, 1, num_subtiles)]
    names = [n for n in (variables or sets[0].dims) if sets[0].dims[n][-2:] == (y_dim, x_dim)]
    if not names:
        raise ValueError(f"no variable of {prefix} has trailing dims {(y_dim, x_dim)}")
    dims = {n: sets[0].dims[n] for n in names}
    # two sets of pinned staging buffers
    pinned = [{n: torch.empty(sets[0].shape(n), dtype=torch.from_numpy(np.empty(0, sets[0].dtypes[n])).dtype).pin_memory()
               for n in names} for _ in range(2)]
    coarse: Dict[str, list] = {n: [] for n in names}

    def read_tile(tile: int, slot: int, box: dict):
        t0 = time.perf_counter()
        ts = sets[0] if tile == 1 else netcdf.open_tile(prefix, tile, num_subtiles)
        for n in names:
            ts.read(n, out=pinned[slot][n].numpy())
        ts.close()
        box["seconds"] = time.perf_counter() - t0

    box = {}
    reader = threading.Thread(target=read_tile, args=(1, 0, box))
    reader.start()
    done_events = [None, None]  # the device has finished reading slot i's pinned buffers
    for tile in range(1, netcdf.NUM_TILES + 1):
        slot = (tile - 1) % 2
        reader.join()
        t_read += box["seconds"]
        if tile < netcdf.NUM_TILES:
            nxt = 1 - slot
            if done_events[nxt] is not None:
                done_events[nxt].synchronize()  # the previous upload from that slot must have finished
            box = {}
            reader = threading.Thread(target=read_tile, args=(tile + 1, nxt, box))
            reader.start()
        t0 = time.perf_counter()
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(copy_stream):
            dev_in = {n: pinned[slot][n].to(dev, non_blocking=True) for n in names}
            uploaded = torch.cuda.Event()
            uploaded.record(copy_stream)
        done_events[slot] = uploaded
        main.wait_event(uploaded)
        for n in names:
            x = dev_in[n]
            x.record_stream(main)
            bytes_in += x.numel() * x.element_size()
            coarse[n].append(ops.weighted_block_average(x, area_t[tile - 1], f).cpu().numpy())
        t_dev += time.perf_counter() - t0
    t0 = time.perf_counter()
    out_vars = {n: (("tile",) + tuple(dims[n]), np.stack(coarse[n]), {"coarsening_factor": f}) for n in names}
    zarr_v2.write_dataset(out_path, out_vars, coords={"tile": np.arange(netcdf.NUM_TILES)}, attrs=attrs)
    t_write = time.perf_counter() - t0
    return {"read_s": t_read, "device_s": t_dev, "write_s": t_write, "bytes_in": float(bytes_in),
            "bytes_out": float(sum(v[1].nbytes for v in out_vars.values()))}
