"""Zarr v2 directory stores, uncompressed, written and read with the standard library and numpy only.

Layout (zarr spec v2): a group is a directory with ``.zgroup`` (+ ``.zattrs``); an array is a directory with ``.zarray``
(shape, chunks, dtype, order "C", ``compressor: null``), ``.zattrs`` and one raw little-endian file per chunk named by the
chunk's grid indices joined with ".".  xarray reads such a store with ``open_zarr`` when every array's attributes carry
``_ARRAY_DIMENSIONS`` -- written here -- and ``.zmetadata`` holds the consolidated metadata (``consolidated=True`` in the
reference's ``to_zarr`` call, workflows/coarsen_c384_diagnostics/coarsen_c384_diagnostics.py:84).
"""
import itertools
import json
import math
import os
from typing import Dict, Mapping, Optional, Sequence

import numpy as np

DIMS_ATTR = "_ARRAY_DIMENSIONS"


def _dtype_str(dtype) -> str:
    dt = np.dtype(dtype)
    if dt.kind not in "fiub":
        raise TypeError(f"unsupported dtype {dt} (numeric and bool arrays only)")
    return dt.newbyteorder("<").str if dt.itemsize > 1 else dt.str


def _write_json(path: str, obj) -> None:
    """Publish a metadata document atomically.  Several processes (the ranks of a model run) may write the same
    document at the same time: each stages it in a file of its OWN (mkstemp in the target directory), so no writer can
    truncate or rename away another's staging file, and whichever rename lands last leaves a complete document."""
    import tempfile

    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=os.path.dirname(path) or ".")
    try:
        with os.fdopen(fd, "w") as f:
            json.dump(obj, f, indent=1, allow_nan=False)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise


def _fill_value(dtype):
    return "NaN" if np.dtype(dtype).kind == "f" else 0


def create_group(path: str, attrs: Optional[Mapping] = None, exist_ok: bool = False) -> None:
    """``exist_ok``: leave an existing group's metadata alone (a second process arriving at the same store)."""
    os.makedirs(path, exist_ok=True)
    if exist_ok and os.path.exists(os.path.join(path, ".zgroup")):
        return
    _write_json(os.path.join(path, ".zgroup"), {"zarr_format": 2})
    _write_json(os.path.join(path, ".zattrs"), dict(attrs or {}))


def create_array(group: str, name: str, shape: Sequence[int], chunks: Sequence[int], dtype, dims: Sequence[str],
                 attrs: Optional[Mapping] = None, exist_ok: bool = False) -> str:
    """Write the metadata of an array (no chunk yet); returns the array directory.  ``exist_ok``: an array some other
    process has created already keeps its metadata (its shape may have been extended since)."""
    if len(shape) != len(chunks) or len(shape) != len(dims):
        raise ValueError("shape, chunks and dims must have the same length")
    path = os.path.join(group, name)
    os.makedirs(path, exist_ok=True)
    if exist_ok and os.path.exists(os.path.join(path, ".zarray")):
        return path
    _write_json(os.path.join(path, ".zarray"), {
        "zarr_format": 2, "shape": [int(n) for n in shape], "chunks": [max(int(c), 1) for c in chunks],
        "dtype": _dtype_str(dtype), "compressor": None, "fill_value": _fill_value(dtype), "order": "C", "filters": None})
    _write_json(os.path.join(path, ".zattrs"), {DIMS_ATTR: list(dims), **dict(attrs or {})})
    return path


def write_chunk(array_dir: str, index: Sequence[int], data: np.ndarray) -> None:
    """One chunk, C order, little endian.  ``data`` must have the chunk's full shape (edge chunks are padded by the
    caller, as the spec stores them)."""
    meta = read_meta(array_dir)
    if tuple(data.shape) != tuple(meta["chunks"]):
        raise ValueError(f"chunk has shape {data.shape}, the array's chunks are {meta['chunks']}")
    a = np.ascontiguousarray(data, dtype=np.dtype(meta["dtype"]))
    with open(os.path.join(array_dir, ".".join(str(int(i)) for i in index) if len(index) else "0"), "wb") as f:
        f.write(a.tobytes())


def set_shape(array_dir: str, shape: Sequence[int]) -> None:
    meta = read_meta(array_dir)
    meta["shape"] = [int(n) for n in shape]
    _write_json(os.path.join(array_dir, ".zarray"), meta)


def read_meta(array_dir: str) -> dict:
    with open(os.path.join(array_dir, ".zarray")) as f:
        return json.load(f)


def write_array(group: str, name: str, data: np.ndarray, dims: Sequence[str], attrs: Optional[Mapping] = None,
                chunks: Optional[Sequence[int]] = None) -> None:
    """A whole array; default chunking: one chunk per index of the leading dimension (a tile, a time) for arrays of more
    than two dimensions, else a single chunk."""
    data = np.asarray(data)
    if chunks is None:
        chunks = ((1,) + tuple(data.shape[1:])) if data.ndim > 2 else tuple(data.shape)
    chunks = tuple(max(int(c), 1) for c in chunks)
    path = create_array(group, name, data.shape, chunks, data.dtype, dims, attrs)
    grid = [range(math.ceil(n / c)) for n, c in zip(data.shape, chunks)]
    for index in itertools.product(*grid):
        sel = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(index, chunks, data.shape))
        piece = data[sel]
        if piece.shape != chunks:  # edge chunk: stored at full size, padded with the fill value
            full = np.full(chunks, np.nan if data.dtype.kind == "f" else 0, dtype=data.dtype)
            full[tuple(slice(0, s) for s in piece.shape)] = piece
            piece = full
        write_chunk(path, index, piece)


def consolidate(group: str) -> None:
    """``.zmetadata``: every metadata document of the store in one file (zarr's consolidated metadata v1)."""
    docs = {}
    for root, _, files in os.walk(group):
        for name in files:
            if name in (".zgroup", ".zarray", ".zattrs"):
                rel = os.path.relpath(os.path.join(root, name), group).replace(os.sep, "/")
                with open(os.path.join(root, name)) as f:
                    docs[rel] = json.load(f)
    _write_json(os.path.join(group, ".zmetadata"), {"zarr_consolidated_format": 1, "metadata": dict(sorted(docs.items()))})


def write_dataset(path: str, variables: Mapping[str, tuple], coords: Optional[Mapping[str, np.ndarray]] = None,
                  attrs: Optional[Mapping] = None, consolidated: bool = True) -> None:
    """``variables``: name -> (dims, array[, attrs]); ``coords``: 1-D coordinate arrays by dimension name.  The store
    opens in xarray as the Dataset with those variables (``Dataset.to_zarr(path, mode="w", consolidated=True)``)."""
    create_group(path, attrs)
    for name, spec in variables.items():
        dims, data = spec[0], spec[1]
        write_array(path, name, np.asarray(data), dims, spec[2] if len(spec) > 2 else None)
    for dim, values in (coords or {}).items():
        write_array(path, dim, np.asarray(values), [dim])
    if consolidated:
        consolidate(path)


def read_array(group: str, name: str):
    """(array, dims, attrs) of a stored array; missing chunks read as the fill value."""
    path = os.path.join(group, name)
    meta = read_meta(path)
    if meta.get("compressor") is not None or meta.get("filters"):
        raise NotImplementedError("only uncompressed, unfiltered arrays are read here")
    with open(os.path.join(path, ".zattrs")) as f:
        attrs = json.load(f)
    dims = attrs.pop(DIMS_ATTR, None)
    shape, chunks, dtype = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    fill = meta.get("fill_value")
    fill = np.nan if fill == "NaN" else (0 if fill is None else fill)
    out = np.full(shape, fill, dtype=dtype)
    grid = [range(math.ceil(n / c)) for n, c in zip(shape, chunks)]
    for index in itertools.product(*grid):
        fname = os.path.join(path, ".".join(str(i) for i in index) if index else "0")
        if not os.path.exists(fname):
            continue
        piece = np.fromfile(fname, dtype=dtype).reshape(chunks)
        sel = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(index, chunks, shape))
        out[sel] = piece[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out, dims, attrs


def read_dataset(path: str) -> Dict[str, tuple]:
    """name -> (dims, array, attrs) for every array of the group."""
    out = {}
    for name in sorted(os.listdir(path)):
        if os.path.exists(os.path.join(path, name, ".zarray")):
            data, dims, attrs = read_array(path, name)
            out[name] = (dims, data, attrs)
    return out
