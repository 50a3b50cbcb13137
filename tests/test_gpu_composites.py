"""fv3fit's composite predictors on the device, written as the reference's own tests are
(external/fv3fit/tests/test_tapered_model.py, test_ensemble.py, test_squashed_output_model.py) plus
compositions over the HIP dense predictor."""
import os

import numpy as np
import pytest
import yaml

from fv3net_amd import fit
from fv3net_amd.fit import (CombinedOutputModel, ConstantOutputPredictor, EnsembleModel, SquashedOutputConfig,
                            SquashedOutputModel, TaperConfig, TaperedModel)
from fv3net_amd.xr_compat import DataArray, Dataset

pytestmark = pytest.mark.gpu


def test_vertical_tapering_scale_factors_known_answer():
    # external/vcm/vcm/calc/calc.py:52-56: exp((z - cutoff) / rate) for z < cutoff, then ones
    got = fit.vertical_tapering_scale_factors(5, 2, 2.0)
    np.testing.assert_allclose(got, [np.exp(-1.0), np.exp(-0.5), 1, 1, 1], rtol=1e-15)


def test_TaperedModel():
    # test_tapered_model.py:12-30
    model = ConstantOutputPredictor(input_variables=["in0", "in1"], output_variables=["out0", "out1"])
    model.set_outputs(out1=np.ones(10), out0=np.ones(10))
    taper_config0 = TaperConfig(cutoff=3, rate=5.0, taper_dim="z")
    taper_config1 = TaperConfig(cutoff=6, rate=3.0, taper_dim="z")
    tapered_model = TaperedModel(model, {"out0": taper_config0, "out1": taper_config1})
    da = DataArray(np.ones((5, 10)), dims=["x", "z"])
    X = Dataset({"in0": da, "in1": da})
    tapered_prediction = tapered_model.predict(X)
    np.testing.assert_array_equal(tapered_prediction["out0"].values, taper_config0.apply(model.predict(X)["out0"]).values)
    np.testing.assert_array_equal(tapered_prediction["out1"].values, taper_config1.apply(model.predict(X)["out1"]).values)
    # and the numbers themselves: scaling * data in float64
    want = np.ones((5, 10)) * fit.vertical_tapering_scale_factors(10, 3, 5.0)
    assert tapered_prediction["out0"].values.dtype == np.float64
    np.testing.assert_array_equal(tapered_prediction["out0"].values, want)
    with pytest.raises(KeyError):
        TaperedModel(model, {"nope": taper_config0})


def test_TaperedModel_load(tmp_path):
    # test_tapered_model.py:33-62
    model = ConstantOutputPredictor(input_variables=["in0", "in1"], output_variables=["out0", "out1"])
    model.set_outputs(out1=np.ones(10), out0=np.ones(10))
    base = str(tmp_path / "predictor")
    fit.dump(model, base)
    out = tmp_path / "tapered_model"
    os.mkdir(out)
    with open(out / "tapered_model.yaml", "w") as f:
        yaml.dump({"tapering": {"out0": {"cutoff": 3, "rate": 5}, "out1": {"cutoff": 2, "rate": 6}}, "model": base}, f)
    with open(out / "name", "w") as f:
        print("tapered_model", file=f)
    tapered_model = fit.load(str(out))
    assert isinstance(tapered_model, TaperedModel)
    da = DataArray(np.ones((5, 10)), dims=["x", "z"])
    pred = tapered_model.predict(Dataset({"in0": da, "in1": da}))
    assert np.mean(pred["out0"].values) < 1.0


@pytest.mark.parametrize("values, reduction, output", [((0.0, 3.0, 5.0), "median", 3.0), ((0.0, 3.0, 5.0), "mean", 8.0 / 3)])
def test_ensemble_model(values, reduction, output):
    # test_ensemble.py:8-28
    models = tuple(ConstantOutputPredictor(["input"], ["output"]) for _ in values)
    for i, m in enumerate(models):
        m.set_outputs(output=values[i])
    ensemble = EnsembleModel(models, reduction=reduction)
    ds_out = ensemble.predict(Dataset({"input": DataArray(np.zeros([3, 3, 5]), dims=["x", "y", "z"])}))
    assert list(ds_out) == ["output"]
    np.testing.assert_almost_equal(ds_out["output"].values, output)
    with pytest.raises(NotImplementedError):
        EnsembleModel(models, reduction="max")


def test_ensemble_of_hip_dense_models_and_nan_members():
    from test_gpu_api import _dense_model, _state

    rng = np.random.default_rng(0)
    members = [_dense_model(np.random.default_rng(s), nz=20, width=8, depth=2) for s in range(4)]
    X = _state(rng, nz=20, ny=6, nx=5)
    preds = [m.predict(X) for m in members]
    for reduction, fn in (("mean", np.mean), ("median", np.median)):
        out = EnsembleModel(members, reduction).predict(X)
        for name in out:
            stack = np.stack([p[name].values for p in preds])
            assert out[name].dims == preds[0][name].dims
            np.testing.assert_allclose(out[name].values, fn(stack, axis=0), rtol=2e-7, atol=1e-30)
    # NaN members are skipped, all-NaN stays NaN (xarray's skipna default for floats)
    from fv3net_amd import ops
    import torch

    a = rng.normal(0, 1, (5, 1000)).astype(np.float64)
    a[rng.random(a.shape) < 0.3] = np.nan
    a[:, :3] = np.nan
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_mean, want_med = np.nanmean(a, axis=0), np.nanmedian(a, axis=0)
    dev = [torch.from_numpy(r.copy()).cuda() for r in a]
    np.testing.assert_allclose(ops.member_reduce(dev, "mean").cpu().numpy(), want_mean, rtol=1e-14, equal_nan=True)
    np.testing.assert_array_equal(ops.member_reduce(dev, "median").cpu().numpy(), want_med)


OUTPUT_DICT = {"a": np.array([[-2.0, -1.0, 0.0, 1.0, 2.0]]), "b": np.array([[-0.2, -0.1, 0.0, 0.1, 0.2]])}


@pytest.mark.parametrize(
    ["additional_targets", "squash_to", "squash_threshold", "expected"],
    [
        (["b"], 0.0, 1.5, {"a": [[0.0, 0.0, 0.0, 0.0, 2.0]], "b": [[0.0, 0.0, 0.0, 0.0, 0.2]]}),
        ([], 0.0, 1.5, {"a": [[0.0, 0.0, 0.0, 0.0, 2.0]], "b": OUTPUT_DICT["b"]}),
        (["b"], 0.0, -1.5, {"a": [[0.0, -1.0, 0.0, 1.0, 2.0]], "b": [[0.0, -0.1, 0.0, 0.1, 0.2]]}),
        (["b"], 0.1, 1.5, {"a": [[0.1, 0.1, 0.1, 0.1, 2.0]], "b": [[0.1, 0.1, 0.1, 0.1, 0.2]]}),
    ],
)
def test_squashed_output_model_predict(additional_targets, squash_to, squash_threshold, expected):
    # test_squashed_output_model.py:62-118
    base_model = ConstantOutputPredictor(input_variables=["n"], output_variables=["a", "b"])
    base_model.set_outputs(**{k: v.squeeze() for k, v in OUTPUT_DICT.items()})
    squashing = [SquashedOutputConfig(squash_by_name="a", additional_squash_target_names=additional_targets,
                                      squash_threshold=squash_threshold, squash_to=squash_to)]
    squashed_model = SquashedOutputModel(base_model, squashing)
    predictions = squashed_model.predict(Dataset({"n": DataArray(np.zeros((1, 5)), dims=["x", "z"])}))
    for name in predictions:
        np.testing.assert_allclose(predictions[name].values, expected[name])


def test_squashed_output_model_validation():
    # test_squashed_output_model.py:13-59
    S = SquashedOutputConfig
    with pytest.raises(ValueError):
        SquashedOutputModel._validate([S(squash_by_name="a", squash_threshold=0.08)], output_variables=["b"])
    with pytest.raises(ValueError):
        SquashedOutputModel._validate([S("a", 0.08), S("a", 0.02)], output_variables=["a"])
    with pytest.raises(ValueError):
        SquashedOutputModel._validate([S("a", 0.02, additional_squash_target_names=["c"]),
                                       S("b", 0.02, additional_squash_target_names=["c"])], output_variables=["a", "b", "c"])
    with pytest.raises(ValueError):
        SquashedOutputModel._validate([S("a", 0.02, additional_squash_target_names=["b"]), S("b", 0.02)],
                                      output_variables=["a", "b"])


def test_combined_output_model_and_nesting(tmp_path):
    m1 = ConstantOutputPredictor(["in0"], ["out0"])
    m1.set_outputs(out0=np.arange(10.0))
    m2 = ConstantOutputPredictor(["in1"], ["out1"])
    m2.set_outputs(out1=2.0)
    combined = CombinedOutputModel([m1, m2])
    assert tuple(combined.input_variables) == ("in0", "in1") and tuple(combined.output_variables) == ("out0", "out1")
    da = DataArray(np.ones((5, 10)), dims=["x", "z"])
    X = Dataset({"in0": da, "in1": da})
    out = combined.predict(X)
    np.testing.assert_array_equal(out["out0"].values, np.broadcast_to(np.arange(10.0), (5, 10)))
    np.testing.assert_array_equal(out["out1"].values, np.full(5, 2.0))
    with pytest.raises(ValueError, match="different outputs"):
        CombinedOutputModel([m1, m1])
    # a tapered, squashed combination loaded from a directory tree, the way composite models are deployed
    for name, m in (("m1", m1), ("m2", m2)):
        fit.dump(m, str(tmp_path / name))
    os.mkdir(tmp_path / "combined")
    (tmp_path / "combined" / "name").write_text("combined_output_model")
    (tmp_path / "combined" / "combined_output_model.yaml").write_text(yaml.dump({"models": [str(tmp_path / "m1"), str(tmp_path / "m2")]}))
    os.mkdir(tmp_path / "squashed")
    (tmp_path / "squashed" / "name").write_text("squashed_output_model")
    (tmp_path / "squashed" / "squashed_output_model.yaml").write_text(yaml.dump(
        {"base_model_path": str(tmp_path / "combined"), "squashing": [{"squash_by_name": "out0", "squash_threshold": 4.5, "squash_to": -1.0}]}))
    loaded = fit.load(str(tmp_path / "squashed"))
    out = loaded.predict(X)
    np.testing.assert_array_equal(out["out0"].values, np.broadcast_to(np.where(np.arange(10.0) > 4.5, np.arange(10.0), -1.0), (5, 10)))


# ------------------------------------------------------------------------------------------------
# DerivedModel (external/fv3fit/tests/test_derived_model.py)
# ------------------------------------------------------------------------------------------------
_SW = "override_for_time_adjusted_total_sky_downward_shortwave_flux_at_surface"


def _sw_base(value=1.0, extra=()):
    m = ConstantOutputPredictor(["input"], [_SW, *extra])
    m.set_outputs(**{_SW: value, **{k: 1.0 for k in extra}})
    return m


def test_derived_model_wraps_another_derived_model():
    from fv3net_amd.fit import DerivedModel

    base_outputs = [_SW, "dQ2"]
    derived_model_0 = DerivedModel(_sw_base(extra=["dQ2"]), derived_output_variables=["net_shortwave_sfc_flux_derived"])
    derived_model_1 = DerivedModel(derived_model_0, derived_output_variables=["Q2"])
    assert not isinstance(derived_model_1.base_model, DerivedModel)
    assert set(derived_model_1.input_variables) == {"input", "surface_diffused_shortwave_albedo",
                                                    "pressure_thickness_of_atmospheric_layer", "pQ2"}
    assert set(derived_model_1.output_variables) == set(base_outputs) | {"Q2", "net_shortwave_sfc_flux_derived"}
    arr = DataArray(np.zeros(10), dims=["x"])
    outputs = derived_model_1.predict(Dataset({var: arr for var in derived_model_1.input_variables}))
    assert set(outputs) == set(derived_model_1.output_variables)


def test_derived_model_prediction_inputs_errors_and_io(tmp_path):
    from fv3net_amd.fit import DerivedMapping, DerivedModel

    derived_model = DerivedModel(_sw_base(4.0), derived_output_variables=["net_shortwave_sfc_flux_derived"])
    # the base model's own output is not an additional input (test_derived_model.py:53-60)
    assert derived_model._additional_input_variables == ["surface_diffused_shortwave_albedo"]
    albedo = np.random.default_rng(0).uniform(0, 1, (3, 3))
    ds_in = Dataset({"input": DataArray(np.zeros([3, 3, 5]), dims=["x", "y", "z"]),
                     "surface_diffused_shortwave_albedo": DataArray(albedo, dims=["x", "y"])})
    prediction = derived_model.predict(ds_in)
    # (1 - albedo) * downward flux (derived_mapping.py:194-195)
    np.testing.assert_array_equal(prediction["net_shortwave_sfc_flux_derived"].values, (1 - albedo) * 4.0)
    with pytest.raises(KeyError):
        derived_model.predict(Dataset({"input": ds_in["input"]}))
    with pytest.raises(ValueError):
        DerivedModel(_sw_base(), derived_output_variables=["variable_not_in_DerivedMapping"])
    fit.dump(derived_model, str(tmp_path / "derived"))
    loaded = fit.load(str(tmp_path / "derived"))
    after = loaded.predict(ds_in)
    for name in prediction:
        np.testing.assert_array_equal(after[name].values, prediction[name].values)
    # Q2 = dQ2 + pQ2 when dQ2 is there, pQ2 = zeros_like(delp) unless given; transmissivity route
    delp = DataArray(np.full((4, 5), 3.0), dims=["x", "z"])
    m = DerivedMapping(Dataset({"dQ2": delp, "pressure_thickness_of_atmospheric_layer": delp}))
    np.testing.assert_array_equal(m["Q2"].values, np.full((4, 5), 3.0))
    np.testing.assert_array_equal(m["Q1"].values, np.zeros((4, 5)))
    toa, tr = DataArray(np.full((6,), 1000.0), dims=["x"]), DataArray(np.linspace(0, 1, 6), dims=["x"])
    m = DerivedMapping(Dataset({"total_sky_downward_shortwave_flux_at_top_of_atmosphere": toa,
                                "shortwave_transmissivity_of_atmospheric_column": tr,
                                "surface_diffused_shortwave_albedo": DataArray(np.full((6,), 0.25), dims=["x"])}))
    np.testing.assert_array_equal(m["net_shortwave_sfc_flux_via_transmissivity"].values, 0.75 * (np.linspace(0, 1, 6) * 1000.0))
    assert set(DerivedMapping.find_all_required_inputs(["Q1"])) == {"pQ1", "pressure_thickness_of_atmospheric_layer"}
