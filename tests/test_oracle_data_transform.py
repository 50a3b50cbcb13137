"""The numpy oracle of the output transforms (oracle/data_transform_np.py) against the known answers and properties the
reference's own tests hold (external/vcm/tests/test_calc_clouds.py:60-75, test_flux_form.py:19-46,
test_data_transform.py:113-131; external/fv3fit/tests/test_taper.py:20-53), and the host-side registry logic of the
product (data_transform.py:302-367; test_data_transform.py:84-110)."""
import numpy as np
import pytest

from oracle import data_transform_np as D


def test_incloud_to_gridcell_known_answers():
    cf = np.array([1.0e-3, 1.0e-2, 1.0e-1])
    incloud = np.array([1.0e-2, 1.0e-2, 1.0e-2])
    np.testing.assert_allclose(D.incloud_to_gridcell_condensate(cf, incloud), [1.0e-2, 5.0e-4, 1.0e-3])
    np.testing.assert_allclose(D.incloud_to_gridcell_condensate(cf, incloud, 1.0e-2, 5.0e-2), [1.0e-2, 1.0e-2, 1.0e-3])
    np.testing.assert_allclose(D.incloud_to_gridcell_condensate(cf, incloud, 1.0e-3, 1.0e-2), [1.0e-2, 1.0e-4, 1.0e-3])


def test_taper_known_answers():
    score = np.array([[1, 3, 5], [6, 4, 2]])
    np.testing.assert_almost_equal(D.taper_mask(score, cutoff=3), [[1, 1, 0], [0, 0, 1]])
    np.testing.assert_almost_equal(D.taper_ramp(score, ramp_min=2, ramp_max=5), [[1, 2 / 3, 0], [0, 1 / 3, 1]])
    np.testing.assert_almost_equal(D.taper_decay(score, threshold=2, rate=0.5), [[1, 2 ** -1, 2 ** -3], [2 ** -4, 2 ** -2, 1]])
    np.testing.assert_allclose(D.taper_mask(np.array([-1e-5, 1e-5])), [1, 0])
    np.testing.assert_allclose(D.taper_ramp(np.array([-1, 0.5, 2]), ramp_min=-1, ramp_max=2), [1, 0.5, 0])


def test_flux_form_round_trip_and_budget_closure():
    rng = np.random.default_rng(0)
    tend, delp = rng.random((6, 4, 4)), rng.random((6, 4, 4))
    toa, up = rng.random((4, 4)), rng.random((4, 4))
    flux, down = D.tendency_to_flux(tend, toa, up, delp, rectify=False)
    np.testing.assert_allclose(D.flux_to_tendency(flux, down, up, delp), tend)
    implied = D.tendency_to_implied_surface_downward_flux(tend, toa, up, delp, rectify=False)
    np.testing.assert_allclose((tend * delp / D.GRAVITY).sum(axis=0), toa + up - implied)
    np.testing.assert_allclose(implied, down)
    assert (D.tendency_to_flux(-tend, toa * 0, up * 0, delp, rectify=True)[1] >= 0).all()


@pytest.mark.parametrize("name,kwargs", [("Qm", {"rectify_downward_radiative_flux": False}), ("Q2", {"rectify_surface_precipitation_rate": False})])
def test_registered_flux_transforms_round_trip(name, kwargs):
    from fv3net_amd.fit.data_transform import DATA_TRANSFORM_REGISTRY

    rng = np.random.default_rng(1)
    forward, backward = f"{name}_flux_from_{name}_tendency", f"{name}_tendency_from_{name}_flux"
    ds = {v: rng.random((6, 4, 4)) if v in (name, D.DELP) else rng.random((4, 4)) for v in DATA_TRANSFORM_REGISTRY[forward].inputs}
    with_flux = {**ds, **D.apply(forward, ds, **kwargs)}
    del with_flux[name]
    np.testing.assert_allclose(D.apply(backward, with_flux)[name], ds[name])


def test_registry_names_inputs_and_outputs_match_the_oracle():
    from fv3net_amd.fit.data_transform import DATA_TRANSFORM_REGISTRY

    rng = np.random.default_rng(2)
    two_d = {D.DLW_SFC, D.DSW_SFC, D.DSW_TOA, D.ULW_SFC, D.ULW_TOA, D.USW_SFC, D.USW_TOA, D.LHF, D.SHF, D.COL_T_NUDGE,
             "implied_downward_radiative_flux_at_surface", "implied_surface_precipitation_rate"}
    assert len(DATA_TRANSFORM_REGISTRY) == 16
    for key, entry in DATA_TRANSFORM_REGISTRY.items():
        ds = {v: rng.random((4, 4)) if v in two_d else rng.random((6, 4, 4)) for v in entry.inputs}
        out = D.apply(key, ds, **({"rate": 1.0, "cutoff": 0} if key.startswith("tapered") else {}))
        assert sorted(out) == sorted(entry.outputs), key


def test_transform_inputs_outputs():
    from fv3net_amd.fit import ChainedDataTransform, DataTransform

    t = DataTransform("Qm_from_Q1_Q2")
    assert t.input_variables == ["Q1", "Q2"] and t.output_variables == ["Qm"]
    with pytest.raises(ValueError, match="unknown data transform"):
        DataTransform("Qm_from_nothing")
    for transforms, inputs, outputs in (
            ([], [], []),
            (["Q1_from_dQ1_pQ1", "Qm_from_Q1_Q2"], ["Q2", "dQ1", "pQ1"], ["Q1", "Qm"]),
            (["Q1_from_dQ1_pQ1", "Q2_from_dQ2_pQ2"], ["dQ1", "dQ2", "pQ1", "pQ2"], ["Q1", "Q2"])):
        chain = ChainedDataTransform([DataTransform(n) for n in transforms])
        assert chain.input_variables == inputs and chain.output_variables == outputs


def test_transformed_predictor_and_out_of_sample_variables():
    from fv3net_amd.fit import (ConstantOutputNoveltyDetector, ConstantOutputPredictor, DataTransform, OutOfSampleModel,
                                TransformedPredictor)

    transforms = [DataTransform("Qm_from_Q1_Q2")]
    m = TransformedPredictor(ConstantOutputPredictor(["input"], ["Q1", "Q2"]), transforms)
    assert m.input_variables == ["input"] and m.output_variables == ["Q1", "Q2", "Qm"]
    m = TransformedPredictor(ConstantOutputPredictor(["input"], ["Q1"]), transforms)
    assert m.input_variables == ["Q2", "input"] and m.output_variables == ["Q1", "Qm"]
    oos = OutOfSampleModel(ConstantOutputPredictor(["shared_input", "base_input"], ["output"]),
                           ConstantOutputNoveltyDetector(["shared_input", "novelty_input"]), 1)
    assert oos.input_variables == ("base_input", "novelty_input", "shared_input")
    assert oos.output_variables == ("centered_score", "is_novelty", "novelty_score", "output", "taper_values")


def test_minmax_score_against_sklearn():
    from sklearn.preprocessing import MinMaxScaler

    rng = np.random.default_rng(3)
    train, test = rng.normal(0, 1, (200, 7)), rng.normal(0, 1.5, (50, 7))
    scaler = MinMaxScaler().fit(train)
    score = D.minmax_score(scaler.transform(test))
    inside = ((test >= train.min(axis=0)) & (test <= train.max(axis=0))).all(axis=1)
    assert ((score == 0) == inside).all() and (score >= 0).all() and (~inside).any()
