"""Data formats either side of the path: the dependency-free zarr v2 writer, the sub-tile netCDF
reader and the ``store`` hook, on synthetic files (no GPU needed); the file -> device -> zarr pipeline under -m gpu."""
import datetime
import json
import os

import numpy as np
import pytest

from fv3net_amd.io import netcdf, zarr_v2


def test_zarr_v2_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    t = rng.normal(size=(6, 5, 12, 12)).astype(np.float32)
    cat = rng.integers(0, 9, (6, 12, 12)).astype(np.int32)
    store = str(tmp_path / "coarse.zarr")
    zarr_v2.write_dataset(store, {"T": (["tile", "z", "y", "x"], t, {"units": "K"}), "slmsk": (["tile", "y", "x"], cat)},
                          coords={"x": np.arange(1.0, 13.0)}, attrs={"title": "synthetic"})
    # the documents the zarr v2 spec asks for
    assert json.load(open(os.path.join(store, ".zgroup"))) == {"zarr_format": 2}
    meta = json.load(open(os.path.join(store, "T", ".zarray")))
    assert meta == {"zarr_format": 2, "shape": [6, 5, 12, 12], "chunks": [1, 5, 12, 12], "dtype": "<f4", "compressor": None,
                    "fill_value": "NaN", "order": "C", "filters": None}
    assert json.load(open(os.path.join(store, "T", ".zattrs"))) == {"_ARRAY_DIMENSIONS": ["tile", "z", "y", "x"], "units": "K"}
    assert sorted(n for n in os.listdir(os.path.join(store, "T")) if not n.startswith(".")) == [f"{i}.0.0.0" for i in range(6)]
    raw = np.fromfile(os.path.join(store, "T", "3.0.0.0"), dtype="<f4").reshape(5, 12, 12)
    np.testing.assert_array_equal(raw, t[3])
    consolidated = json.load(open(os.path.join(store, ".zmetadata")))
    assert consolidated["zarr_consolidated_format"] == 1 and "T/.zarray" in consolidated["metadata"] and ".zgroup" in consolidated["metadata"]
    back = zarr_v2.read_dataset(store)
    np.testing.assert_array_equal(back["T"][1], t)
    np.testing.assert_array_equal(back["slmsk"][1], cat)
    assert back["slmsk"][1].dtype == np.int32 and back["T"][0] == ["tile", "z", "y", "x"] and back["T"][2] == {"units": "K"}
    # ragged chunks: the edge chunk is stored at full size, the padding is not part of the array
    odd = rng.normal(size=(5, 7)).astype(np.float64)
    zarr_v2.create_group(str(tmp_path / "g"))
    zarr_v2.write_array(str(tmp_path / "g"), "odd", odd, ["a", "b"], chunks=(2, 4))
    assert os.path.getsize(tmp_path / "g" / "odd" / "2.1") == 2 * 4 * 8
    np.testing.assert_array_equal(zarr_v2.read_array(str(tmp_path / "g"), "odd")[0], odd)
    with pytest.raises(TypeError):
        zarr_v2.write_array(str(tmp_path / "g"), "s", np.array(["a"]), ["a"])


@pytest.mark.parametrize("layout", [(2, 2), (4, 4), (1, 3)])
def test_subtile_files_reassemble_to_the_tile(tmp_path, layout):
    """``{prefix}.tile{N}.nc.{NNNN}`` (coarsen.py:27): the rectangles go back where their coordinate values say
    (io.py:6-28), cell-centred and staggered dims alike, big-endian file data into native arrays / given buffers."""
    rng = np.random.default_rng(1)
    n = 24
    fields = {
        "T": (["Time", "zaxis_1", "yaxis_2", "xaxis_1"], rng.normal(size=(1, 4, n, n)).astype(np.float32)),
        "u": (["Time", "zaxis_1", "yaxis_1", "xaxis_1"], rng.normal(size=(1, 4, n + 1, n))),
        "v": (["Time", "zaxis_1", "yaxis_2", "xaxis_2"], rng.normal(size=(1, 4, n, n + 1))),
        "phis": (["Time", "yaxis_2", "xaxis_1"], rng.normal(size=(1, n, n))),
    }
    prefix = str(tmp_path / "fv_core.res")
    paths = netcdf.write_subtile_files(prefix, 2, fields, layout=layout)
    assert paths == netcdf.subtile_filenames(prefix, 2, layout[0] * layout[1])
    assert os.path.basename(paths[-1]) == f"fv_core.res.tile2.nc.{layout[0] * layout[1] - 1:04d}"
    assert len(netcdf.all_filenames(prefix, 16)) == 96
    tiles = netcdf.open_tile(prefix, 2, layout[0] * layout[1])
    assert tiles.shape("u") == (1, 4, n + 1, n)
    for name, (dims, data) in fields.items():
        got = tiles.read(name)
        assert got.dtype == data.dtype and got.dtype.isnative
        np.testing.assert_array_equal(got, data)
    buf = np.empty((1, 4, n, n), np.float32)
    assert tiles.read("T", out=buf) is buf
    np.testing.assert_array_equal(tiles.coords()["xaxis_2"], np.arange(1.0, n + 2))
    with pytest.raises(ValueError, match="shape"):
        tiles.read("T", out=np.empty((1, 4, n, n + 1), np.float32))
    tiles.close()
    ds = netcdf.read_tile_dataset(prefix, 2, layout[0] * layout[1], variables=["T", "phis"])
    assert ds["T"].dims == ("Time", "zaxis_1", "yaxis_2", "xaxis_1")
    np.testing.assert_array_equal(ds["phis"].values, fields["phis"][1])
    with pytest.raises(FileNotFoundError):
        netcdf.open_tile(prefix, 5, 4)


def test_store_hook_writes_zarr_and_netcdf_at_the_output_times(tmp_path):
    """monitor.py:221-305: output every ``output_freq_sec`` counted from the first call, evaluated at model_time + dt;
    fields squeezed, cast to float32 and transposed to [sample, z]; attrs from the metadata with the _input/_output
    suffix removed; one netCDF file per (time, rank); a zarr store with a time axis."""
    from scipy.io import netcdf_file

    from fv3net_amd.emulation.config import EmulationConfig
    from fv3net_amd.emulation.monitor import StorageHook

    meta = {"air_temperature": {"units": "K", "long_name": "temperature"}}
    hook = StorageHook(output_freq_sec=1800, dt_sec=900, metadata=meta, n_ranks=2, directory=str(tmp_path))
    rng = np.random.default_rng(2)
    stored = []
    for step in range(5):  # model times 00:00, 00:15, ...; model_time + dt hits a multiple of 30 min on odd steps
        minute = 15 * step
        for rank in (0, 1):
            state = {"air_temperature_input": rng.normal(size=(7, 10)), "surface_air_pressure": rng.normal(size=(10,)),
                     "model_time": [2016, 8, 1, 0, minute // 60, minute % 60], "rank": np.array([rank])}
            hook_r = hook if rank == 0 else hook2 if step else None
            if rank == 1 and step == 0:
                hook2 = StorageHook(output_freq_sec=1800, dt_sec=900, metadata=meta, n_ranks=2, directory=str(tmp_path))
                hook_r = hook2
            hook_r.store(state)
            if (minute + 15) % 30 == 0:
                stored.append((minute + 15, rank, state))
    times = sorted({m for m, _, _ in stored})
    assert times == [30, 60]
    root = str(tmp_path / "state_output.zarr")
    from fv3net_amd.io import zarr_v2

    data, dims, attrs = zarr_v2.read_array(root, "air_temperature_input")
    assert dims == ["time", "rank", "sample", "z"] and data.shape == (2, 2, 10, 7) and attrs["units"] == '"K"'
    for minute, rank, state in stored:
        np.testing.assert_array_equal(data[times.index(minute), rank], state["air_temperature_input"].astype(np.float32).T)
    tvals = zarr_v2.read_array(root, "time")[0]
    assert tvals[1] - tvals[0] == 1800.0
    names = sorted(os.listdir(tmp_path / "netcdf_output"))
    # labelled with the model time of the call, not with time + dt (monitor.py:273-281 passes `time` to the writers)
    assert names == ["state_20160801.001500_0.nc", "state_20160801.001500_1.nc", "state_20160801.004500_0.nc", "state_20160801.004500_1.nc"]
    assert tvals[0] == (np.datetime64("2016-08-01T00:15:00") - np.datetime64("1970-01-01T00:00:00")) / np.timedelta64(1, "s")
    f = netcdf_file(str(tmp_path / "netcdf_output" / names[1]), "r", mmap=False)
    assert f.variables["air_temperature_input"].dimensions == ("sample", "z") and f.tile == 1
    assert f.variables["surface_air_pressure"].units == b"unknown"
    f.close()
    # through the configuration, as get_hooks() builds it
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        store = EmulationConfig.from_dict({"storage": {"output_freq_sec": 900, "save_nc": False}}).build_storage_hook()
        store({"q": np.ones((3, 4)), "model_time": [2016, 8, 1, 0, 0, 0]})
    finally:
        os.chdir(cwd)


def _hook_rank_process(directory, rank, n_steps, barrier):
    """One model rank in its own process (what MPI ranks are): the same store calls as the other ranks, at the same time."""
    import numpy as np

    from fv3net_amd.emulation.monitor import StorageHook

    hook = StorageHook(output_freq_sec=900, dt_sec=900, save_nc=False, n_ranks=4, directory=directory)
    for step in range(n_steps):
        minute = 15 * step
        barrier.wait()  # all ranks reach the output time together, as the time-stepping model's ranks do
        hook.store({"q": np.full((5, 6), 100.0 * rank + step), "ps": np.full((6,), float(rank)),
                    "model_time": [2016, 8, 1, 0, minute // 60, minute % 60], "rank": np.array([rank])})


def test_store_hook_ranks_as_concurrent_processes(tmp_path):
    """the ranks of a run are separate processes that create the same .zgroup / .zarray / .zattrs at the
    first output time -- with a shared staging name two of them could publish a truncated document or lose the rename.
    Four rank processes, released together at every step, twelve steps: every chunk and every metadata document intact."""
    import multiprocessing as mp

    from fv3net_amd.io import zarr_v2

    ctx = mp.get_context("spawn")
    n_ranks, n_steps = 4, 12
    barrier = ctx.Barrier(n_ranks)
    procs = [ctx.Process(target=_hook_rank_process, args=(str(tmp_path), r, n_steps, barrier)) for r in range(n_ranks)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    root = str(tmp_path / "state_output.zarr")
    data, dims, _ = zarr_v2.read_array(root, "q")
    assert dims == ["time", "rank", "sample", "z"] and data.shape == (n_steps, n_ranks, 6, 5)
    want = 100.0 * np.arange(n_ranks)[None, :] + np.arange(n_steps)[:, None]
    np.testing.assert_array_equal(data[:, :, 0, 0], want)
    assert zarr_v2.read_array(root, "time")[0].shape == (n_steps,)
    assert not [f for _, _, files in os.walk(root) for f in files if f.endswith(".tmp")]  # no staging file left behind


@pytest.mark.gpu
def test_files_to_device_to_zarr_pipeline(tmp_path):
    """Sub-tile files -> pinned buffers -> device block average -> coarse zarr equals the oracle on the same arrays."""
    import torch

    from fv3net_amd.io import coarsen_subtile_files_to_zarr
    from oracle import coarsen_np as onp

    rng = np.random.default_rng(3)
    n, nz, f = 32, 5, 4
    area = rng.uniform(0.5, 1, (6, n, n)).astype(np.float32)
    prefix = str(tmp_path / "atmos")
    truth = {}
    for tile in range(1, 7):
        fields = {"T": (["time", "pfull", "yaxis_1", "xaxis_1"], rng.normal(size=(2, nz, n, n)).astype(np.float32)),
                  "ps": (["time", "yaxis_1", "xaxis_1"], rng.normal(size=(2, n, n)).astype(np.float32)),
                  "u": (["time", "pfull", "yaxis_2", "xaxis_1"], rng.normal(size=(2, nz, n + 1, n)).astype(np.float32))}
        netcdf.write_subtile_files(prefix, tile, fields, layout=(2, 2), y_dims=("yaxis_1", "yaxis_2"))
        truth[tile] = fields
    stats = coarsen_subtile_files_to_zarr(prefix, str(tmp_path / "coarse.zarr"), area, f, num_subtiles=4)
    out = zarr_v2.read_dataset(str(tmp_path / "coarse.zarr"))
    assert set(out) == {"T", "ps", "tile"}  # `u` lives on other horizontal dims: skipped
    assert out["T"][0] == ["tile", "time", "pfull", "yaxis_1", "xaxis_1"] and out["T"][1].shape == (6, 2, nz, n // f, n // f)
    for tile in range(1, 7):
        want = onp.weighted_block_average(truth[tile]["T"][1], area[tile - 1][None, None], f)
        np.testing.assert_allclose(out["T"][1][tile - 1], want, rtol=1e-5, atol=1e-6)
    assert stats["bytes_in"] == 6 * (2 * nz + 2) * n * n * 4 and torch.cuda.is_available()
