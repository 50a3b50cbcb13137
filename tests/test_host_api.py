"""Host-side logic that needs no GPU: the C ABI exports what the header declares, the labelled
array layer, coordinate coarsening, the model io registry, spec (de)serialisation, the hook's
in-place contract, the partitioning helpers, and that the product path refuses to run without a
GPU instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

import fv3net_amd
from fv3net_amd import _lib
from fv3net_amd.xr_compat import DataArray, Dataset, assert_identical_including_dtype

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_in_the_header():
    header = open(os.path.join(ROOT, "include", "fv3hip.h")).read()
    declared = set(re.findall(r"\b(fv3hip_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fv3hip_abi_version() == _lib.ABI_VERSION == 3
    assert lib.fv3hip_mappm_workspace_bytes(10, 79) >= 5 * 79 * 10 * 4


def test_library_reports_errors_without_touching_the_gpu():
    lib = _lib.load()
    rc = lib.fv3hip_weighted_block_average(None, 7, None, 0, 1, 4, 4, 1, 2, None, None)
    assert rc == _lib.EINVAL
    assert b"dtype" in lib.fv3hip_last_error()
    # kord > 7 is served (cs_profile); only its iv = -2 branch, which reads the array mappm never sets, is refused
    rc = lib.fv3hip_mappm(None, None, None, 0, None, 1, 1, 79, 79, -2, 9, 0, 0, None, 0, None)
    assert rc == _lib.EUNSUPPORTED and b"cs_profile" in lib.fv3hip_last_error()
    rc = lib.fv3hip_mappm(None, None, None, 0, None, 1, 1, 79, 79, 1, 9, 0, 0, None, 0, None)
    assert rc == _lib.EINVAL and b"null pointer" in lib.fv3hip_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from fv3net_amd import mappm, ops
    from fv3net_amd.cubedsphere import weighted_block_average

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.weighted_block_average(torch.zeros(4, 4), torch.zeros(4, 4), 2)
    da = DataArray(np.zeros((4, 4)), dims=["y", "x"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        weighted_block_average(da, da, 2, x_dim="x", y_dim="y")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mappm.mappm(np.zeros((1, 6)), np.zeros((1, 5)), np.zeros((1, 6)), 1, 1, 1, 1, 0.0)


def test_dataarray_basics():
    da = DataArray(np.arange(24.0).reshape(2, 3, 4), dims=["t", "y", "x"], coords={"x": [1.0, 2, 3, 4]}, name="foo",
                   attrs={"units": "m"})
    assert da.sizes == {"t": 2, "y": 3, "x": 4}
    tr = da.transpose("x", "t", "y")
    assert tr.dims == ("x", "t", "y") and tr.shape == (4, 2, 3) and tr.attrs == {"units": "m"}
    assert da.transpose("x", ...).dims == ("x", "t", "y")
    sel = da.isel(t=0, x=slice(1, 3))
    assert sel.dims == ("y", "x") and list(sel.coords["x"]) == [2.0, 3.0]
    ds = da.to_dataset()
    assert list(ds) == ["foo"] and ds.dims == {"t": 2, "y": 3, "x": 4}
    assert da.rename("bar").name == "bar" and da.rename({"x": "lon"}).dims == ("t", "y", "lon")
    with pytest.raises(ValueError):
        DataArray(np.zeros((2, 2)), dims=["a"])
    assert_identical_including_dtype(da, da.copy())


def test_coordinate_coarsening():
    from fv3net_amd.cubedsphere import add_coordinates, coarsen_coords, coarsen_coords_coord_func

    # external/vcm/tests/test_cubedsphere.py:422-616: subtile coordinates, float32 and float64
    for dtype in (np.float32, np.float64):
        for start in (1, 49, 97):
            c = np.arange(start, start + 48, dtype=dtype)
            ref = DataArray(np.zeros(48), dims=["x"], coords={"x": c})
            out = coarsen_coords(2, ref, ["x"])["x"]
            assert out.dtype == np.float32
            np.testing.assert_array_equal(out, np.arange((start - 1) // 2 + 1, (start - 1) // 2 + 25))
            np.testing.assert_array_equal(coarsen_coords_coord_func(c.reshape(-1, 2)), out)
    coarse = DataArray(np.zeros(24), dims=["x"], name="a")
    ref = DataArray(np.zeros(48), dims=["x"], coords={"x": np.arange(1, 49, dtype=np.float32)})
    assert add_coordinates(ref, coarse, 2, ["x"]).coords["x"][-1] == 24.0


def test_io_registry_and_constant_predictor(tmp_path):
    from fv3net_amd import fit

    model = fit.ConstantOutputPredictor(["air_temperature"], ["dQ1", "rain"])
    model.set_outputs(dQ1=np.arange(5.0), rain=2.5)
    fit.dump(model, str(tmp_path / "m"))
    assert (tmp_path / "m" / "name").read_text() == "constant-output"
    loaded = fit.load(str(tmp_path / "m"))
    X = Dataset({"air_temperature": DataArray(np.zeros((5, 3, 4)), dims=["z", "y", "x"])})
    out = loaded.predict(X)
    assert out["dQ1"].dims == ("z", "y", "x") and out["rain"].dims == ("y", "x")
    np.testing.assert_array_equal(out["dQ1"].values[:, 1, 2], np.arange(5.0))
    assert float(out["rain"].values[0, 0]) == 2.5
    with pytest.raises(ValueError, match="already registered"):
        fit.io.register("constant-output")
    (tmp_path / "tf").mkdir()
    (tmp_path / "tf" / "name").write_text("all-keras")
    with pytest.raises(ValueError, match="TensorFlow"):
        fit.load(str(tmp_path / "tf"))


def _spec(rng):
    from fv3net_amd.mlp import InputSpec, MlpSpec, OutputSpec, ResidualSpec

    return MlpSpec(
        inputs=[InputSpec("T", 5, center=rng.normal(size=5), scale=rng.uniform(1, 2, 5)),
                InputSpec("q", 3, start=2, transform="log", eps=1e-8)],
        hidden_kernels=[rng.normal(size=(8, 4)).astype(np.float32)], hidden_biases=[np.zeros(4, np.float32)],
        outputs=[OutputSpec("dQ1", 5, scale=np.ones(5), center=np.zeros(5), min=-1.0, mask=np.array([0, 1, 1, 1, 1.0]))],
        out_kernel=rng.normal(size=(4, 5)).astype(np.float32), out_bias=np.zeros(5, np.float32),
        residuals=[ResidualSpec("T_after", "T", "dQ1")],
    )


def test_spec_roundtrip_and_validation(tmp_path):
    from fv3net_amd import fit
    from fv3net_amd.mlp import MlpSpec

    spec = _spec(np.random.default_rng(0))
    spec.validate()
    assert spec.sources == ["T", "q"] and spec.source_nfeat() == {"T": 5, "q": 5}
    assert spec.output_names == ["dQ1", "T_after"]
    model = fit.HipDenseModel(["T", "q"], ["dQ1"], spec)
    fit.dump(model, str(tmp_path / "m"))
    assert (tmp_path / "m" / "name").read_text() == "hip-dense"
    back = fit.load(str(tmp_path / "m"))
    assert back.input_variables == ["T", "q"] and back.output_variables == ["dQ1"]
    m1, a1 = spec.to_arrays()
    m2, a2 = back.spec.to_arrays()
    assert m1 == m2 and set(a1) == set(a2)
    for k in a1:
        np.testing.assert_array_equal(a1[k], a2[k])
    bad = _spec(np.random.default_rng(0))
    bad.out_kernel = bad.out_kernel[:, :3]
    with pytest.raises(ValueError):
        bad.validate()
    with pytest.raises(ValueError, match="not produced"):
        fit.HipDenseModel(["T"], ["nope"], spec)


def test_hook_updates_state_in_place():
    # external/emulation/tests/test_microphysics.py:21-44: an "x + 1" model, [feature, sample] arrays
    from fv3net_amd.emulation import MicrophysicsHook

    def model(x):
        return {"air_temperature_output": x["air_temperature_input"] + 1}

    state = {"air_temperature_input": np.arange(6.0).reshape(3, 2), "model_time": [2016, 8, 1, 0, 0, 0], "rank": 0}
    hook = MicrophysicsHook(model)
    assert hook.microphysics(state) is None
    assert state["air_temperature_output"].shape == (3, 2)
    np.testing.assert_array_equal(state["air_temperature_output"], np.arange(6.0).reshape(3, 2) + 1)
    assert state["model_time"] == [2016, 8, 1, 0, 0, 0]


def test_emulation_config(tmp_path):
    from fv3net_amd.emulation.config import EmulationConfig, get_hooks

    gscond, micro, store = get_hooks(str(tmp_path / "missing.yml"))
    state = {"a": np.zeros(3)}
    assert micro(state) is None and gscond(state) is None and store(state) is None and list(state) == ["a"]
    # transforms that build from nothing are accepted (tests/test_host_transforms.py); one that has to be fitted is not
    with pytest.raises(NotImplementedError, match="ConditionallyScaled"):
        EmulationConfig.from_dict({"model": {"path": "x", "tensor_transform": [{"to": "a", "source": "b", "condition_on": "T", "bins": 10}]}})
    with pytest.raises(ValueError, match="unknown tensor transform"):
        EmulationConfig.from_dict({"model": {"path": "x", "tensor_transform": [{"to": "a", "source": "b"}]}})
    assert EmulationConfig.from_dict({"model": {"path": "x", "classifier_path": "y"}}).model.classifier_path == "y"
    cfg = EmulationConfig.from_dict({"model": {"path": "x", "cloud_squash": 1e-6, "enforce_conservative": True,
                                               "ranges": {"total_precipitation": {"min": 0}},
                                               "mask_emulator_levels": {"air_temperature_after_precpd": {"start": 74}}}})
    # the reference's composition order (config.py:178-221): range, squash x2, conservation, level mask
    assert len(list(cfg.model._build_masks())) == 5
    # the reference's production configs carry a storage section (projects/microphysics/configs/*.yaml): it builds the
    # store hook (tests/test_io.py exercises it); unknown keys and TFRecord output are refused
    cfg2 = EmulationConfig.from_dict({"storage": {"output_freq_sec": 10800, "save_zarr": True}})
    assert cfg2.storage.output_freq_sec == 10800 and callable(cfg2.build_storage_hook())
    assert EmulationConfig.from_dict({}).build_storage_hook()({"a": np.zeros(3)}) is None
    with pytest.raises(ValueError, match="unknown"):
        EmulationConfig.from_dict({"storage": {"output_frequency": 1}})
    with pytest.raises(ValueError, match="TensorFlow"):
        EmulationConfig.from_dict({"storage": {"save_tfrecord": True}}).build_storage_hook()
    with pytest.raises(ValueError, match="mutually exclusive"):
        EmulationConfig.from_dict({"model": {"enforce_conservative": True, "enforce_conservative_phase_dependent": True}})
    with pytest.raises(ValueError, match="unknown"):
        EmulationConfig.from_dict({"model": {"pth": "x"}})


def test_partitioning():
    from fv3net_amd import parallel

    for n, w in [(884736, 8), (10, 3), (5, 8)]:
        ranges = [parallel.column_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    units = parallel.tile_bands(6, 3072, 8, 8)
    assert [len(u) for u in units] == [3] * 8
    flat = sorted(u for r in units for u in r)
    assert flat[0] == (0, 0, 768) and all((hi - lo) % 8 == 0 for _, lo, hi in flat)
    assert sum(hi - lo for _, lo, hi in flat) == 6 * 3072
    assert [len(u) for u in parallel.tile_bands(6, 384, 8, 6)] == [1] * 6
    assert parallel.world() == (0, 1)


def test_every_module_imports_on_its_own():
    """No import-order dependence between the subpackages (thermo <-> cubedsphere share device helpers)."""
    import subprocess
    import sys

    mods = ["fv3net_amd.thermo", "fv3net_amd.cubedsphere", "fv3net_amd.cubedsphere.coarsen_restarts", "fv3net_amd.emulation",
            "fv3net_amd.fit", "fv3net_amd.fit.streaming", "fv3net_amd.fit.derived", "fv3net_amd.parallel", "fv3net_amd.interpolate",
            "fv3net_amd.mappm", "fv3net_amd.local_mlp", "fv3net_amd.mlp", "fv3net_amd.ops"]
    code = "import importlib, sys\nfor m in sys.argv[1:]:\n    for k in [k for k in sys.modules if k.startswith('fv3net_amd')]:\n        del sys.modules[k]\n    importlib.import_module(m)\n"
    res = subprocess.run([sys.executable, "-c", code, *mods], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert res.returncode == 0, res.stderr[-2000:]


def test_interval_schedule_and_time_mask_known_answers():
    """external/emulation/tests/test_microphysics.py:47-66 and test_config.py:16-55 (cftime.DatetimeJulian there:
    plain (year, month, day, hour, minute, second) tuples or datetimes here)."""
    import datetime

    from fv3net_amd.emulation.config import EmulationConfig, ModelConfig
    from fv3net_amd.emulation.schedule import IntervalSchedule, TimeMask, julian_seconds

    scheduler = IntervalSchedule(datetime.timedelta(hours=3), (2000, 1, 1))
    assert scheduler((2000, 1, 1)) == 1
    assert scheduler((2000, 1, 1, 1)) == 1
    assert scheduler((2000, 1, 1, 1, 30)) == 0
    assert scheduler((2000, 1, 1, 2)) == 0
    assert scheduler((2000, 1, 20)) == 1
    assert scheduler(datetime.datetime(2000, 1, 1, 1, 30)) == 0
    # Julian calendar: 1900 is a leap year there (not in the Gregorian one); day numbers are consecutive
    assert julian_seconds((1900, 3, 1)) - julian_seconds((1900, 2, 28)) == 2 * 86400
    assert julian_seconds((2001, 1, 1)) - julian_seconds((2000, 1, 1)) == 366 * 86400
    for weight in (0.0, 0.5, 1.0):
        mask = TimeMask(schedule=lambda time: weight)
        assert mask({"a": 0.0, "model_time": [2021, 1, 1, 0, 0, 0]}, {"a": 1.0}) == {"a": 1 - weight}
    config = EmulationConfig.from_dict({"model": {"path": "some-path", "online_schedule": {
        "period": 60, "initial_time": datetime.datetime(2000, 2, 1)}}})
    assert config.model.online_schedule.period == datetime.timedelta(seconds=60)
    assert config.model.online_schedule.initial_time.month == 2
    assert len(list(ModelConfig(path="")._build_masks())) == 0

    def schedule(time):
        return 1.0

    time_masks = [m for m in ModelConfig(path="", online_schedule=schedule)._build_masks() if isinstance(m, TimeMask)]
    assert time_masks[0].schedule == schedule
    # the model's time tuple: fields 0, 1, 2, 4, 5 (_time.py:6-12)
    mask = TimeMask(IntervalSchedule(datetime.timedelta(hours=2), (2016, 8, 1)))
    assert mask({"a": 5.0, "model_time": [2016, 8, 1, 0, 0, 30]}, {"a": 1.0}) == {"a": 5.0}   # 00:30 -> first half: physics
    assert mask({"a": 5.0, "model_time": [2016, 8, 1, 0, 1, 30]}, {"a": 1.0}) == {"a": 1.0}   # 01:30 -> second half: emulator


def test_graphed_call_needs_a_gpu():
    """No silent eager fallback: without a ROCm device GraphedCall refuses."""
    import torch

    from fv3net_amd.graphs import GraphedCall

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="cuda"):
        GraphedCall(lambda: None)


def test_bench_launcher_never_touches_the_gpu_library(monkeypatch, tmp_path):
    """``python bench.py --gpus N`` starts its ranks from a parent that has made no GPU-library call (a
    parent that initialised HIP must not start children that exec): with every ``torch.cuda`` attribute raising,
    ``spawn_ranks`` still counts the GPUs (kernel-driver topology files), starts N children with the rank environment and
    leaves with their worst exit code."""
    import subprocess
    import sys

    import torch

    import bench

    class Forbidden:
        def __getattr__(self, name):
            raise AssertionError(f"torch.cuda.{name} used in the launcher parent")

    monkeypatch.setattr(torch, "cuda", Forbidden())
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
            self.code = 3 if env["RANK"] == "1" else 0

        def poll(self):
            return self.code

        def terminate(self):
            pass

    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env: FakeProc(cmd, env))
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 2)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as exit_info:
        bench.spawn_ranks(2)
    assert exit_info.value.code == 3
    assert [env["RANK"] for _, env in started] == ["0", "1"]
    assert all(env["WORLD_SIZE"] == "2" and env["MASTER_ADDR"] == "127.0.0.1" for _, env in started)
    # fewer GPUs than ranks: refused before anything starts; an unknown count leaves the check to the ranks
    started.clear()
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    with pytest.raises(SystemExit) as exit_info:
        bench.spawn_ranks(2)
    assert exit_info.value.code == 2 and not started
    monkeypatch.undo()
    assert bench.visible_gpu_count() in (None, 0) or bench.visible_gpu_count() >= 1  # (file reads only; no GPU here)
