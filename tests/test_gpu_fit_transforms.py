"""TransformedPredictor's named data transforms, the novelty detectors, the taper functions and OutOfSampleModel on the
device (fv3net_amd/fit/data_transform.py, novelty.py, transformed.py) against the numpy oracle
(oracle/data_transform_np.py, pinned in tests/test_oracle_data_transform.py) and against sklearn itself for the detectors;
the model-level cases restate external/fv3fit/tests/test_transformed_predictor.py, test_out_of_sample.py:24-71 and
test_taper.py.  Element-wise float64 results are the same bits as numpy's (same operations, same order); sums over the
column and ``pow`` are compared at 1e-13."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import data_transform_np as D

pytestmark = pytest.mark.gpu

TWO_D = {D.DLW_SFC, D.DSW_SFC, D.DSW_TOA, D.ULW_SFC, D.ULW_TOA, D.USW_SFC, D.USW_TOA, D.LHF, D.SHF, D.COL_T_NUDGE,
         "implied_downward_radiative_flux_at_surface", "implied_surface_precipitation_rate"}
BITWISE = {"Qm_from_Q1_Q2", "Q1_from_Qm_Q2", "Qm_from_Q1_Q2_temperature_dependent", "Q1_from_Qm_Q2_temperature_dependent",
           "Q1_from_dQ1_pQ1", "Q2_from_dQ2_pQ2", "Qm_flux_from_Qm_tendency", "Q2_flux_from_Q2_tendency", "Qm_tendency_from_Qm_flux",
           "Q2_tendency_from_Q2_flux", "cloud_water_mixing_ratio_from_incloud", "cloud_ice_mixing_ratio_from_incloud",
           "tapered_dQ1", "tapered_dQ2"}


def _dataset(rng, names, dtype=np.float64, shape3=(6, 4, 5), on_gpu=False):
    from fv3net_amd.xr_compat import DataArray, Dataset

    ds, raw = Dataset(), {}
    for name in names:
        if name == "cloud_amount":
            a = rng.choice([0.0, 5e-4, 1e-3, 2e-3, 5e-2, 0.3, 1.0], size=shape3)
        elif name == D.DELP:
            a = rng.uniform(300, 1500, shape3)
        elif name == "air_temperature":
            a = rng.uniform(200, 310, shape3)
        else:
            a = rng.normal(0, 1, shape3[1:] if name in TWO_D else shape3)
        a = a.astype(dtype)
        raw[name] = a
        data = torch.from_numpy(a).cuda() if on_gpu else a
        ds[name] = DataArray(data, dims=("y", "x") if name in TWO_D else ("z", "y", "x"))
    return ds, raw


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_every_registered_transform_against_the_oracle(dtype):
    from fv3net_amd.fit import DATA_TRANSFORM_REGISTRY, DataTransform

    rng = np.random.default_rng(0)
    for key, entry in DATA_TRANSFORM_REGISTRY.items():
        kwargs = {"rate": 2.5, "cutoff": 3} if key.startswith("tapered") else {}
        ds, raw = _dataset(rng, entry.inputs, dtype)
        out = DataTransform(key, kwargs).apply(ds)
        want = D.apply(key, raw, **kwargs)
        for name in entry.outputs:
            got = np.asarray(out[name].data)
            assert got.shape == want[name].shape and got.dtype == want[name].dtype, (key, name, got.dtype, want[name].dtype)
            if key in BITWISE:
                np.testing.assert_array_equal(got, want[name], err_msg=f"{key}: {name}")
            else:
                np.testing.assert_allclose(got, want[name], rtol=1e-13 if dtype == np.float64 else 1e-5, atol=1e-13 if dtype == np.float64 else 1e-4)
        for name in entry.inputs:  # inputs pass through untouched
            np.testing.assert_array_equal(np.asarray(out[name].data), raw[name])


def test_flux_transforms_other_layouts_rectify_and_device_resident_data():
    """z last ([y, x, z]), rectified surface fluxes, unrectified round trip, NaN columns, data already on the device."""
    from fv3net_amd.fit import DataTransform
    from fv3net_amd.xr_compat import DataArray, Dataset

    rng = np.random.default_rng(5)
    ds, raw = _dataset(rng, ["Q2", D.DELP, D.LHF], shape3=(9, 7, 3))
    raw["Q2"] = np.abs(raw["Q2"]) * 50  # strong drying: negative closure fluxes -> rectified to 0
    raw["Q2"][:, 2, 1] = np.nan
    zlast = Dataset({"Q2": DataArray(torch.from_numpy(np.ascontiguousarray(raw["Q2"].transpose(1, 2, 0))).cuda(), dims=("y", "x", "z")),
                     D.DELP: DataArray(torch.from_numpy(np.ascontiguousarray(raw[D.DELP].transpose(1, 2, 0))).cuda(), dims=("y", "x", "z")),
                     D.LHF: DataArray(torch.from_numpy(raw[D.LHF]).cuda(), dims=("y", "x"))})
    out = DataTransform("Q2_flux_from_Q2_tendency").apply(zlast)
    want = D.apply("Q2_flux_from_Q2_tendency", raw)
    assert out["Q2_flux"].dims == ("y", "x", "z") and isinstance(out["Q2_flux"].data, torch.Tensor) and out["Q2_flux"].data.is_cuda
    np.testing.assert_array_equal(out["Q2_flux"].data.cpu().numpy(), want["Q2_flux"].transpose(1, 2, 0))
    down = out["implied_surface_precipitation_rate"].data.cpu().numpy()
    np.testing.assert_array_equal(down, want["implied_surface_precipitation_rate"])
    assert (down == 0).sum() > 5 and down[2, 1] == 0  # where(x >= 0, 0): a NaN closure flux becomes 0 too
    closure = DataTransform("implied_surface_precipitation_rate", {"rectify": False}).apply(zlast)["implied_surface_precipitation_rate"]
    np.testing.assert_allclose(closure.data.cpu().numpy(), D.apply("implied_surface_precipitation_rate", raw, rectify=False)[
        "implied_surface_precipitation_rate"], rtol=1e-13, equal_nan=True)
    unrect = DataTransform("Q2_flux_from_Q2_tendency", {"rectify_surface_precipitation_rate": False}).apply(zlast)
    back = DataTransform("Q2_tendency_from_Q2_flux").apply(Dataset({k: unrect[k] for k in ("Q2_flux", "implied_surface_precipitation_rate", D.DELP, D.LHF)}))
    np.testing.assert_allclose(back["Q2"].data.cpu().numpy(), raw["Q2"].transpose(1, 2, 0), rtol=1e-9, equal_nan=True)


def test_cloud_and_taper_known_answers():
    from fv3net_amd.fit import DataTransform, get_taper_function, taper_decay, taper_mask, taper_ramp
    from fv3net_amd.xr_compat import DataArray, Dataset

    ds = Dataset({"cloud_amount": DataArray(np.array([1.0e-3, 1.0e-2, 1.0e-1]), dims=("x",)),
                  "incloud_water_mixing_ratio": DataArray(np.array([1.0e-2, 1.0e-2, 1.0e-2]), dims=("x",))})
    out = DataTransform("cloud_water_mixing_ratio_from_incloud").apply(ds)
    np.testing.assert_allclose(out["cloud_water_mixing_ratio"].data, [1.0e-2, 5.0e-4, 1.0e-3])  # vcm/tests/test_calc_clouds.py:60-75
    score = DataArray(np.array([[1, 3, 5], [6, 4, 2]], dtype=np.float64), dims=("y", "x"))
    np.testing.assert_array_equal(get_taper_function(taper_mask.__name__, {"cutoff": 3})(score).data, [[1, 1, 0], [0, 0, 1]])
    np.testing.assert_almost_equal(get_taper_function(taper_ramp.__name__, {"ramp_min": 2, "ramp_max": 5})(score).data, [[1, 2 / 3, 0], [0, 1 / 3, 1]])
    np.testing.assert_almost_equal(get_taper_function(taper_decay.__name__, {"threshold": 2, "rate": 0.5})(score).data,
                                   [[1, 2 ** -1, 2 ** -3], [2 ** -4, 2 ** -2, 1]])
    rng = np.random.default_rng(1)
    s = rng.normal(0, 2, (50, 40))
    s[3, 4] = np.nan
    sd = DataArray(s, dims=("y", "x"))
    np.testing.assert_array_equal(taper_ramp(sd, ramp_min=-0.5, ramp_max=1.7).data, D.taper_ramp(s, -0.5, 1.7))
    np.testing.assert_allclose(taper_decay(sd, threshold=0.3, rate=0.7).data, D.taper_decay(s, 0.3, 0.7), rtol=1e-14)
    np.testing.assert_array_equal(taper_mask(sd, cutoff=0.1).data, D.taper_mask(s, 0.1))
    with pytest.raises(ValueError, match="Incorrect tapering name"):
        get_taper_function("taper_nothing")


def test_transformed_predictor(tmp_path):
    """external/fv3fit/tests/test_transformed_predictor.py."""
    import fv3net_amd.fit as fit
    from fv3net_amd.xr_compat import DataArray, Dataset, assert_identical_including_dtype

    transforms = [fit.DataTransform("Qm_from_Q1_Q2")]
    base = fit.ConstantOutputPredictor(["input"], ["Q1", "Q2"])
    base.set_outputs(Q1=1.0, Q2=2.0)
    model = fit.TransformedPredictor(base, transforms)
    x = Dataset({"input": DataArray(np.array([0.0, 1.0, 2.0]), dims=("x",))})
    out = model.predict(x)
    np.testing.assert_array_equal(out["Qm"].data, D.moist_static_energy_tendency(np.full(3, 1.0), np.full(3, 2.0)))
    assert set(out) == {"Q1", "Q2", "Qm"} and set(x) == {"input"}
    # a required input the base model does not predict comes from X and is not returned
    model2 = fit.TransformedPredictor(fit.ConstantOutputPredictor(["input"], ["Q1"]), transforms)
    out2 = model2.predict(Dataset({"input": DataArray(np.array([0.0, 1.0, 2.0]), dims=("x",)), "Q2": DataArray(np.array([0.0, 1.0, 2.0]), dims=("x",))}))
    assert "Qm" in out2 and "Q2" not in out2
    # the inputs contain an output of the transform (offline diagnostics): the prediction's value is used
    base3 = fit.ConstantOutputPredictor(["input"], ["Q1"])
    base3.set_outputs(Q1=np.array([5.0, 6.0, 7.0]))
    x3 = Dataset({"input": DataArray(np.array([0.0, 1.0, 2.0]), dims=("z",)), "Q2": DataArray(np.array([0.0, 1.0, 2.0]), dims=("z",)),
                  "Qm": DataArray(np.array([3.0, 4.0, 5.0]), dims=("z",))})
    out3 = fit.TransformedPredictor(base3, transforms).predict(x3)
    assert "Q2" not in out3
    np.testing.assert_array_equal(out3["Qm"].data, D.moist_static_energy_tendency(np.array([5.0, 6.0, 7.0]), np.array([0.0, 1.0, 2.0])))
    fit.dump(model, str(tmp_path / "m"))
    loaded = fit.load(str(tmp_path / "m"))
    assert isinstance(loaded, fit.TransformedPredictor)
    assert_identical_including_dtype(loaded.predict(x), out)


@pytest.mark.parametrize("base_value,novelty_cutoff,output", [(1, -1, 0), (1, 1, 1)])
def test_out_of_sample_model(base_value, novelty_cutoff, output, tmp_path):
    """external/fv3fit/tests/test_out_of_sample.py:24-71, test_taper.py:56-102."""
    import fv3net_amd.fit as fit
    from fv3net_amd.xr_compat import DataArray, Dataset

    base = fit.ConstantOutputPredictor(["shared_input", "base_input"], ["output"])
    base.set_outputs(output=base_value)
    detector = fit.ConstantOutputNoveltyDetector(["shared_input", "novelty_input"])
    model = fit.OutOfSampleModel(base, detector, novelty_cutoff)
    ds_in = Dataset({"shared_input": DataArray(np.zeros([3, 3, 5]), dims=("x", "y", "z")), "base_input": DataArray(np.ones([3, 3]), dims=("x", "y")),
                     "novelty_input": DataArray(np.ones([3, 3, 5]), dims=("x", "y", "z"))})
    out = model.predict(ds_in)
    assert len(list(out)) == 5
    np.testing.assert_array_equal(out["is_novelty"].data, 1 - np.asarray(out["taper_values"].data))
    np.testing.assert_almost_equal(np.asarray(out["output"].data), output)
    fit.dump(base, str(tmp_path / "base"))
    fit.dump(detector, str(tmp_path / "novelty"))
    for tapering, scores, want in ((None, [-1e-5, 1e-5], [1, 0]), ({"name": "taper_ramp", "ramp_min": -1, "ramp_max": 2}, [-1, 0.5, 2], [1, 0.5, 0])):
        config = {"base_model_path": str(tmp_path / "base"), "novelty_detector_path": str(tmp_path / "novelty")}
        if tapering:
            config["tapering_function"] = tapering
        with open(tmp_path / fit.OutOfSampleModel._CONFIG_FILENAME, "w") as f:
            yaml.safe_dump(config, f)
        loaded = fit.OutOfSampleModel.load(str(tmp_path))
        np.testing.assert_allclose(loaded.taper(DataArray(np.array(scores, dtype=np.float64), dims=("x",))).data, want)


def test_out_of_sample_tapers_vertical_outputs_column_by_column():
    """A [z, y, x] tendency times the [y, x] taper of a ramp; float32 tendencies come back float64 as xarray's product does."""
    import fv3net_amd.fit as fit
    from fv3net_amd.xr_compat import DataArray, Dataset

    rng = np.random.default_rng(4)
    nz, ny, nx = 7, 5, 6
    train = rng.normal(0, 1, (nz, 40, 40))
    X = Dataset({"T": DataArray(rng.normal(0, 1.3, (nz, ny, nx)), dims=("z", "y", "x"))})
    detector = fit.MinMaxNoveltyDetector.fit(["T"], Dataset({"T": DataArray(train, dims=("z", "y", "x"))}))
    base = fit.ConstantOutputPredictor(["T"], ["dQ1", "flux"])
    dq1 = rng.normal(0, 1, nz).astype(np.float32)
    base.set_outputs(dQ1=dq1, flux=3.0)
    taper = fit.get_taper_function("taper_ramp", {"ramp_min": 0.0, "ramp_max": 0.2})
    out = fit.OutOfSampleModel(base, detector, cutoff=0.0, taper=taper).predict(X)
    score = np.asarray(out["novelty_score"].data)
    keep = D.taper_ramp(score, 0.0, 0.2)
    assert 0 < (keep < 1).sum() < keep.size
    np.testing.assert_array_equal(out["taper_values"].data, keep)
    base_out = base.predict(X)
    assert out["dQ1"].dims == base_out["dQ1"].dims
    zaxis = out["dQ1"].dims.index("z")
    np.testing.assert_array_equal(np.moveaxis(np.asarray(out["dQ1"].data), zaxis, 0), dq1.astype(np.float64)[:, None, None] * keep[None])
    np.testing.assert_array_equal(out["flux"].data, 3.0 * keep)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_minmax_novelty_detector_against_sklearn(dtype, tmp_path):
    import fv3net_amd.fit as fit
    from sklearn.preprocessing import MinMaxScaler
    from fv3net_amd.xr_compat import DataArray, Dataset

    rng = np.random.default_rng(7)
    nz, ny, nx = 11, 9, 13
    train = {"T": rng.normal(250, 20, (nz, 30, 30)), "q": rng.gamma(2.0, 1e-3, (nz, 30, 30)), "ps": rng.normal(1e5, 1e3, (30, 30))}
    test = {"T": rng.normal(250, 25, (nz, ny, nx)), "q": rng.gamma(2.0, 1.2e-3, (nz, ny, nx)), "ps": rng.normal(1e5, 1.5e3, (ny, nx))}
    test["T"][3, 2, 5] = np.nan
    clip = {"q": {"start": 2, "stop": 9}}
    names = ["T", "q", "ps"]

    def pack(d):  # stack + pack (stacking.py:7-37, packer.py): [sample, features of T | clipped q | ps]
        return np.concatenate([d["T"].reshape(nz, -1).T, d["q"][2:9].reshape(7, -1).T, d["ps"].reshape(-1, 1)], axis=1)

    def ds(d, cast=None):
        return Dataset({k: DataArray(v.astype(cast) if cast else v, dims=("z", "y", "x") if v.ndim == 3 else ("y", "x")) for k, v in d.items()})

    scaler = MinMaxScaler().fit(pack(train))
    detector = fit.MinMaxNoveltyDetector.from_sklearn(names, scaler, clip)
    fitted = fit.MinMaxNoveltyDetector.fit(names, ds(train), clip)
    np.testing.assert_allclose(fitted.scale_, scaler.scale_, rtol=1e-15)
    np.testing.assert_allclose(fitted.min_, scaler.min_, rtol=1e-15)
    out = detector.predict(ds(test, dtype))
    want = D.minmax_score(scaler.transform(pack({k: v.astype(dtype) for k, v in test.items()}).astype(np.float64))).reshape(ny, nx)
    assert out["novelty_score"].dims == ("y", "x") and np.asarray(out["novelty_score"].data).dtype == np.float64
    np.testing.assert_allclose(out["novelty_score"].data, want, rtol=1e-14, atol=1e-15, equal_nan=True)
    np.testing.assert_array_equal(out["centered_score"].data, out["novelty_score"].data)
    assert np.isnan(want[2, 5]) and (want[np.isfinite(want)] > 0).any() and (want == 0).any()
    _, diag = detector.predict_novelties(ds(test, dtype), cutoff=0.05)
    np.testing.assert_array_equal(diag["is_novelty"].data, np.where(want > 0.05, 1, 0))
    fit.dump(detector, str(tmp_path / "mm"))
    assert sorted(os.listdir(tmp_path / "mm")) == ["arrays.npz", "metadata.yaml", "name"]
    again = fit.load(str(tmp_path / "mm")).predict(ds(test, dtype))
    np.testing.assert_array_equal(again["novelty_score"].data, out["novelty_score"].data)


def test_ocsvm_novelty_detector_against_sklearn(tmp_path):
    import fv3net_amd.fit as fit
    from sklearn.pipeline import make_pipeline
    from sklearn.preprocessing import StandardScaler
    from sklearn.svm import OneClassSVM
    from fv3net_amd.xr_compat import DataArray, Dataset

    rng = np.random.default_rng(8)
    nz, ny, nx = 19, 17, 23
    train = {"T": rng.normal(250, 20, (nz, 25, 25)), "q": rng.gamma(2.0, 1e-3, (nz, 25, 25))}
    test = {"T": rng.normal(250, 30, (nz, ny, nx)), "q": rng.gamma(2.0, 1.5e-3, (nz, ny, nx))}

    def pack(d):
        return np.concatenate([d["T"].reshape(nz, -1).T, d["q"].reshape(nz, -1).T], axis=1)

    pipeline = make_pipeline(StandardScaler(), OneClassSVM(kernel="rbf", gamma=1.0 / (2 * nz) / 4, nu=0.1))
    pipeline.fit(pack(train))
    max_train = float(np.max(-1 * pipeline.score_samples(pack(train))))
    detector = fit.OCSVMNoveltyDetector.from_sklearn(["T", "q"], pipeline, max_train)
    x = Dataset({k: DataArray(v, dims=("z", "y", "x")) for k, v in test.items()})
    out = detector.predict(x)
    want = (-1 * pipeline.score_samples(pack(test))).reshape(ny, nx)
    assert detector.support_vectors_.shape[0] > 20
    np.testing.assert_allclose(out["novelty_score"].data, want, rtol=1e-12)
    np.testing.assert_allclose(out["centered_score"].data, want - max_train, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    fit.dump(detector, str(tmp_path / "svm"))
    again = fit.load(str(tmp_path / "svm")).predict(x)
    np.testing.assert_array_equal(again["centered_score"].data, out["centered_score"].data)
