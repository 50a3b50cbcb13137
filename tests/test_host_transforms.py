"""Config-level tensor transforms of the hook (fv3net_amd/emulation/transforms.py) on numpy arrays: the known answers of
the reference's own tests (external/fv3fit/tests/emulation/test_transform.py:27-31, 125-143, 261-275, 338-361)."""
import numpy as np
import pytest

from fv3net_amd.emulation.config import ModelConfig
from fv3net_amd.emulation.models import transform_model
from fv3net_amd.emulation.transforms import (ComposedTransform, CloudWaterDiffPrecpd, Difference, LimitValueTransform, LogTransform,
                                            TransformedVariableConfig, transform_from_dict)


def test_log_transform_round_trip():
    t = LogTransform()
    x = np.array([0.001], np.float32)
    np.testing.assert_allclose(t.backward(t.forward(x)), x, rtol=1e-6)
    assert t.forward(np.array([0.0]))[0] == pytest.approx(np.log(1e-30))  # the floor
    assert np.isnan(t.forward(np.array([np.nan]))[0])                     # tf.maximum keeps a NaN


def test_difference_forward_and_backward():  # test_transform.py:261-275
    diff = Difference("diff", "before", "after")
    assert diff.forward({"after": np.float64(1), "before": np.float64(0)}) == {"diff": 1, "after": 1, "before": 0}
    assert diff.backward({"diff": np.float64(1), "before": np.float64(0)}) == {"after": 1, "diff": 1, "before": 0}
    assert diff.backward({"diff": np.float64(1), "before": np.float64(0), "after": np.float64(1000)}) == {"after": 1, "before": 0, "diff": 1}


@pytest.mark.parametrize("lower,upper,expected", [(None, None, [-2, -1, 0, 1, 2, 3]), (0, None, [0, 0, 0, 1, 2, 3]), (None, 0, [-2, -1, 0, 0, 0, 0]),
                                                  (-2, 2, [0, -1, 0, 1, 0, 0]), (1, 1, [0, 0, 0, 0, 0, 0])])
def test_limit_value_transform(lower, upper, expected):  # test_transform.py:338-361
    x = np.array([-2, -1, 0, 1, 2, 3], np.float64)
    t = LimitValueTransform(lower=lower, upper=upper)
    np.testing.assert_array_equal(t.forward(x), x)
    np.testing.assert_array_equal(t.backward(x), np.array(expected, np.float64))


def test_composed_transform_skips_what_it_cannot_apply_and_runs_backward_in_reverse():
    class Rename:  # test_transform.py:125-143
        def __init__(self, a, b):
            self.a, self.b = a, b

        def forward(self, x):
            return {self.b: x[self.a]}

        def backward(self, y):
            return {self.a: y[self.b]}

    t = ComposedTransform([Rename("a", "b"), Rename("b", "c")])
    data = {"a": np.ones(1)}
    assert set(t.forward(data)) == {"c"} and set(t.backward(t.forward(data))) == {"a"}
    # a Difference whose `after` is not among the inputs does nothing on the way in (KeyError) and adds it on the way out
    d = ComposedTransform([Difference("dT", "T_in", "T_out")])
    x = {"T_in": np.array([1.0, 2.0])}
    assert set(d.forward(x)) == {"T_in"}
    np.testing.assert_array_equal(d.backward({**x, "dT": np.array([0.5, -1.0])})["T_out"], [1.5, 1.0])


def test_cloud_water_diff_precpd():
    t = CloudWaterDiffPrecpd("dq", "sphum_src", "cloud_in", "cloud_out")
    y = {"sphum_src": np.array([0.1, -0.2]), "cloud_in": np.array([1.0, 2.0]), "dq": np.array([0.05, 0.1])}
    # cloud after gscond = cloud_in - sphum_src; cloud_out = that + dq  (transforms.py:94-108)
    np.testing.assert_allclose(t.backward(y)["cloud_out"], [0.95, 2.3])
    np.testing.assert_allclose(t.forward({**y, "cloud_out": np.array([0.95, 2.3])})["dq"], [0.05, 0.1])


def test_model_config_builds_the_transforms_around_the_model():
    """zhao_carr_emulation.model.tensor_transform as fv3config.yml spells it: parsed into the transform objects, applied as
    emulation.models.transform_model does (forward, model, backward on inputs and predictions together)."""
    cfg = ModelConfig.from_dict({"tensor_transform": [
        {"source": "q", "transform": {"epsilon": 1e-8}, "to": "log_q"},
        {"to": "dT", "before": "T_in", "after": "T_out"},
        {"source": "precip", "transform": {"lower": 0.0}},
    ]})
    assert [type(t).__name__ for t in cfg.tensor_transform] == ["TransformedVariableConfig", "Difference", "TransformedVariableConfig"]
    seen = {}

    def model(x):
        seen.update(x)
        return {"dT": x["T_in"] * 0 + 2.0, "precip": np.array([-1.0, 3.0])}

    out = transform_model(model, ComposedTransform(cfg.tensor_transform))({"q": np.array([0.0, 1e-3]), "T_in": np.array([280.0, 290.0])})
    np.testing.assert_allclose(seen["log_q"], np.log([1e-8, 1e-3]))           # the model saw the transformed input
    np.testing.assert_array_equal(out["T_out"], [282.0, 292.0])               # Difference.backward
    np.testing.assert_array_equal(out["precip"], [0.0, 3.0])                  # LimitValueTransform.backward
    np.testing.assert_allclose(out["q"], [1e-8, 1e-3])                        # LogTransform.backward of log_q (models.py:56-65 returns it)
    hook = cfg.build()                                                        # no path: the identity model, still wrapped
    state = {"q": np.array([[0.0, 1e-3]]), "T_in": np.array([[280.0, 290.0]]), "model_time": [2016, 8, 1, 0, 0, 0]}
    hook.microphysics(state)
    np.testing.assert_allclose(state["log_q"], np.log([[1e-8, 1e-3]]))
    with pytest.raises(NotImplementedError, match="ConditionallyScaled"):
        transform_from_dict({"to": "a", "source": "b", "condition_on": "T", "bins": 50})
    with pytest.raises(ValueError, match="unknown tensor transform"):
        transform_from_dict({"foo": 1})
