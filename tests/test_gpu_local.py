"""The "dense-local" emulators, the classifier decode and ModelWithClassifier on the device
(external/fv3fit/fv3fit/emulation/layers/architecture.py:518-527, external/emulation/emulation/models.py:14-65),
against the float64 numpy oracle (oracle/mlp_np.py:forward_local; its transform pieces are pinned by the
reference's known answers in tests/test_oracle_mlp.py; the dense contraction itself is parity-unpinned
at the TensorFlow boundary like the "dense" architecture's).  Tolerance: 1e-5 of each output's magnitude
(north star), classes and masks bit-exact."""
import numpy as np
import pytest
import torch
import yaml

import local_cases as cases
from oracle import emulation_np as E
from oracle import mlp_np

from tolerances import assert_close_per_level

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["tiles-of-128", "feature-split"])
def _mlp_kernel(request, monkeypatch):
    """Every emulator test on both MLP kernels: the 128-sample-tile kernel and the feature-split kernel for small sample
    counts (which a model rank's column counts select by default)."""
    from fv3net_amd import mlp

    monkeypatch.setattr(mlp, "DEFAULT_SMALL_LIMIT", 0 if request.param == "tiles-of-128" else 1 << 40)
    return request.param


def _dev(st):
    return {k: torch.from_numpy(v).cuda() for k, v in st.items()}


def _check(got, truth, name, tol=1e-5, f32=None, **kw):
    """[level, column] arrays: the gate is per (variable, level), tests/tolerances.py."""
    assert got.shape == truth.shape, name
    assert_close_per_level(got.T, truth.T, None if f32 is None else f32.T, name, rel=tol, **kw)


@pytest.mark.parametrize("arithmetic", ["fp32", "split-bf16"])
@pytest.mark.parametrize("nz,ncol,dtype", [(79, 1024, np.float64), (79, 333, np.float32), (5, 64, np.float64), (1, 97, np.float32)])
def test_local_regressor_matches_oracle(nz, ncol, dtype, arithmetic):
    """... on the product kernel and on the opt-in split-bf16 arithmetic (1 output tile, a 12-feature first layer)."""
    from fv3net_amd.local_mlp import LocalMlpModel

    rng = np.random.default_rng(nz * 1000 + ncol)
    st = cases.state(rng, nz, ncol, dtype)
    # (the split kernel is built for the production width, 256; the product kernel is also exercised at 64)
    spec = cases.regressor(rng, st, nz, width=256 if arithmetic == "split-bf16" else 64, make=cases.product_makers())
    model = LocalMlpModel(spec, device="cuda", arithmetic=arithmetic)
    assert type(model._inner).__name__ == ("MlpModelSplitBf16" if arithmetic == "split-bf16" else "MlpModel")
    got = model.predict(_dev(st))
    truth = mlp_np.forward_local(spec, {k: v.T for k, v in st.items()}, dtype=np.float64)
    assert list(got) == spec.output_names
    # ... and no worse than a few times the float32 evaluation of the same graph, level by level
    f32 = mlp_np.forward_local(spec, {k: v.T for k, v in st.items()}, dtype=np.float32)
    for name in spec.output_names:
        assert got[name].dtype == torch.float32
        _check(got[name].cpu().numpy(), truth[name].T, name, f32=f32[name].T)


def test_conditional_bins_are_exact_at_the_edges():
    """piecewise(): a value equal to an edge belongs to the bin that starts there; below the first
    edge and above the last one the end bins are used (keras/math.py:5-23)."""
    from fv3net_amd.local_mlp import ConditionalScale, LocalInput, LocalMlpModel, LocalMlpSpec, LocalOutput

    edges = np.array([0.0, 1.0, 2.0], np.float32)
    cond = ConditionalScale("u", "on", edges, scale=np.array([1.0, 2.0, 4.0], np.float32), center=np.array([10.0, 20.0, 30.0], np.float32))
    spec = LocalMlpSpec(inputs=[LocalInput("x", "x")], hidden_kernels=[np.zeros((1, 32), np.float32)], hidden_biases=[np.zeros(32, np.float32)],
                        outputs=[LocalOutput("y", conditional=cond)], out_kernel=np.zeros((32, 1), np.float32), out_bias=np.ones(1, np.float32))
    on = np.array([[-2, 0.0, 0.5, 0.75, 1, 1.5, 2.0, 2.5, np.nextafter(np.float32(1), np.float32(0))]], np.float32)
    out = LocalMlpModel(spec).predict({"x": torch.zeros(on.shape, device="cuda"), "on": torch.from_numpy(on).cuda()})
    np.testing.assert_array_equal(out["y"].cpu().numpy(), np.ones_like(on))
    np.testing.assert_array_equal(out["u"].cpu().numpy(), [[11, 11, 11, 11, 22, 22, 34, 34, 11]])


def test_classifier_decode_is_bit_exact():
    from fv3net_amd.emulation.models import _get_classify_output
    from fv3net_amd.emulation.zhao_carr import CLASS_NAMES

    rng = np.random.default_rng(0)
    logits = rng.normal(0, 1, (4, 7, 130)).astype(np.float32)
    logits[:, 0, :10] = 0.25            # four-way ties: every class hot
    logits[1, 1, :] = logits[3, 1, :] = 9.0   # two-way ties
    want = E.classify(logits)
    got = _get_classify_output(torch.from_numpy(logits).cuda(), one_hot_axis=0)
    assert set(got) == set(CLASS_NAMES) | {"nontrivial_tendency"}
    for name in CLASS_NAMES:
        assert got[name].dtype == torch.bool
        np.testing.assert_array_equal(got[name].cpu().numpy(), want[name])
    np.testing.assert_array_equal(got["nontrivial_tendency"].cpu().numpy(), want["positive_tendency"] | want["negative_tendency"])
    # [sample, z, class] views (what the Keras classifier returns) decode along the last axis
    view = torch.from_numpy(logits).cuda().permute(2, 1, 0)
    got_t = _get_classify_output(view, one_hot_axis=-1)
    for name in CLASS_NAMES:
        assert tuple(got_t[name].shape) == (130, 7)
        np.testing.assert_array_equal(got_t[name].cpu().numpy(), want[name].T)
    # numpy in -> numpy out, same answer (zhao_carr.py:193-198)
    got_np = _get_classify_output(logits, one_hot_axis=0)
    for name in CLASS_NAMES:
        np.testing.assert_array_equal(got_np[name], want[name])


def test_model_with_classifier(tmp_path):
    """external/emulation/tests/test_models.py:7-52 with device models: the outputs hold the regressor's
    outputs, the logits and every decoded class; a singleton entry in the state is ignored."""
    from fv3net_amd.emulation.models import HipLocalEmulator, ModelWithClassifier, load_emulator
    from fv3net_amd.emulation.zhao_carr import CLASS_NAMES

    rng = np.random.default_rng(5)
    nz, ncol = 19, 200
    st = cases.state(rng, nz, ncol)
    reg = HipLocalEmulator(cases.regressor(rng, st, nz, make=cases.product_makers()))
    cls = HipLocalEmulator(cases.classifier(rng, st, nz, make=cases.product_makers()))
    reg.dump(str(tmp_path / "reg"))
    cls.dump(str(tmp_path / "cls"))
    assert yaml.safe_load(open(tmp_path / "cls" / "spec.yaml"))["architecture"] == "dense-local"
    reg2, cls2 = load_emulator(str(tmp_path / "reg")), load_emulator(str(tmp_path / "cls"))
    assert isinstance(reg2, HipLocalEmulator) and reg2.output_variables == reg.output_variables

    model = ModelWithClassifier(reg2, cls2, inputs_to_ignore=["singleton_vector"])
    x = {k: v.T for k, v in st.items()}  # [sample, feature] views, as the hook passes them
    x["singleton_vector"] = np.ones([1])
    out = model(x)
    assert set(out) >= set(CLASS_NAMES) | {"gscond_classes", "nontrivial_tendency"} | set(reg.output_variables)
    for v in out.values():
        assert isinstance(v, np.ndarray)
    assert out["gscond_classes"].shape == (ncol, nz, 4) and out["zero_cloud"].shape == (ncol, nz)
    truth_c = mlp_np.forward_local(cls.spec, {k: v.T for k, v in st.items()}, dtype=np.float64)["gscond_classes"]
    _check(out["gscond_classes"], truth_c, "gscond_classes")
    want = E.classify(np.moveaxis(out["gscond_classes"], -1, 0))
    for name in CLASS_NAMES:
        np.testing.assert_array_equal(out[name], want[name])
    truth_r = mlp_np.forward_local(reg.spec, {k: v.T for k, v in st.items()}, dtype=np.float64)
    for name in reg.output_variables:
        _check(out[name], truth_r[name], name)
    # without a classifier (test_models.py:38-52)
    out = ModelWithClassifier(reg2, classifier=None)({k: v.T for k, v in st.items()})
    assert set(out) == set(reg.output_variables)


def test_hook_with_classifier_and_classifier_masks(tmp_path):
    """The production gscond configuration in miniature (projects/microphysics/configs/gscond-and-precpd.yaml):
    dense-local regressor + classifier, the two classifier masks and the conservation fix, built from
    fv3config-style keys; the state is updated in place with [feature, sample] arrays."""
    from fv3net_amd.emulation.config import ModelConfig
    from fv3net_amd.emulation.models import HipLocalEmulator

    rng = np.random.default_rng(11)
    nz, ncol = 23, 320
    st = cases.state(rng, nz, ncol)
    reg_spec = cases.regressor(rng, st, nz, make=cases.product_makers())
    # the hook's masks work on the Fortran names of the gscond outputs
    reg_spec.outputs[0].after, reg_spec.outputs[1].after = E.QV_G, E.T_G
    HipLocalEmulator(reg_spec).dump(str(tmp_path / "reg"))
    cls_spec = cases.classifier(rng, st, nz, make=cases.product_makers())
    # balance the random classifier so that every class occurs
    raw = mlp_np.forward_local(cls_spec, {k: v.T for k, v in st.items()}, dtype=np.float64)["gscond_classes"]
    cls_spec.out_bias = (cls_spec.out_bias - raw.mean(axis=(0, 1))).astype(np.float32)
    HipLocalEmulator(cls_spec).dump(str(tmp_path / "cls"))
    cfg = ModelConfig.from_dict({"path": str(tmp_path / "reg"), "classifier_path": str(tmp_path / "cls"),
                                 "gscond_cloud_conservative": True, "mask_gscond_no_tend_classifier": True,
                                 "mask_gscond_zero_cloud_classifier": True, "enforce_conservative": True})
    hook = cfg.build()
    state = {k: v.copy() for k, v in st.items()}
    state[E.CLOUD_G] = st[cases.CLOUD_IN] * 1.1
    state["model_time"] = [2016, 8, 1, 0, 0, 0]
    state["rank"] = 0
    before = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in state.items()}
    hook.microphysics(state)

    # oracle: network (float64 truth), decode, then the masks in the reference's order (config.py:178-221)
    x = {k: v.T for k, v in st.items()}
    em = {k: v.T for k, v in mlp_np.forward_local(reg_spec, x, dtype=np.float64).items()}
    logits = np.moveaxis(mlp_np.forward_local(cls_spec, x, dtype=np.float64)["gscond_classes"], -1, 0).transpose(0, 2, 1)
    got_logits = state["gscond_classes"]
    assert got_logits.shape == (4, nz, ncol)
    _check(got_logits, logits, "gscond_classes")
    s64 = {k: v for k, v in before.items() if isinstance(v, np.ndarray)}
    em = {**em, "gscond_classes": got_logits.astype(np.float64)}  # the device's own logits decide the classes
    em[E.CLOUD_G] = E.infer_gscond_cloud_from_conservation(s64, em)[E.CLOUD_G]
    em = E.update_with_net_condensation(E.gscond_cloud_choice(s64, em, "class_zero_tend"), s64, em)
    em = E.update_with_net_condensation(E.gscond_cloud_choice(s64, em, "class_zero_cloud"), s64, em)
    em = E.update_with_net_condensation(em[E.CLOUD_G], s64, em)
    for name in (E.CLOUD_G, E.QV_G, E.T_G):
        assert state[name].shape == (nz, ncol), name
        _check(state[name], em[name], name)
    classes = E.classify(got_logits)
    zero_cloud = classes["zero_cloud"]
    assert zero_cloud.any() and not zero_cloud.all()
    assert np.all(state[E.CLOUD_G][zero_cloud & ~classes["zero_tendency"]] == 0.0)
    np.testing.assert_array_equal(state["zero_cloud"], zero_cloud)
    for k in st:  # inputs untouched
        np.testing.assert_array_equal(state[k], before[k])


@pytest.mark.parametrize("nz,ncol,dtype", [(79, 256, np.float64), (17, 333, np.float32)])
def test_rnn_emulator_on_split_bf16_arithmetic(nz, ncol, dtype):
    """The same recurrence with its cells on the opt-in split-bf16 kernel (hidden output = the new state, cells without an
    output layer, a cell with one): against the float64 oracle with the bar of the fp32 path."""
    from fv3net_amd.local_mlp import RnnModel

    rng = np.random.default_rng(nz + ncol)
    st = cases.state(rng, nz, ncol, dtype)
    spec = cases.precpd_rnn(rng, st, nz, channels=256, make=cases.product_makers())
    model = RnnModel(spec, device="cuda", arithmetic="split-bf16")
    assert all(type(c).__name__ == "MlpModelSplitBf16" for c in model._cells)
    got = model.predict(_dev(st))
    truth = mlp_np.forward_rnn(spec, {k: v.T for k, v in st.items()}, dtype=np.float64)
    f32 = mlp_np.forward_rnn(spec, {k: v.T for k, v in st.items()}, dtype=np.float32)
    for name in spec.output_names:
        g, t = got[name].cpu().numpy(), truth[name].T
        assert g.shape == t.shape, name
        _check(g, t, name, f32=f32[name].T, compounding=True)


@pytest.mark.parametrize("nz,ncol,dtype,channels", [(79, 256, np.float64, 256), (17, 333, np.float32, 64), (5, 64, np.float64, 32)])
def test_rnn_emulator_matches_oracle(nz, ncol, dtype, channels, tmp_path):
    """The production precpd architecture (stacked SimpleRNNs over the levels, surface-step single-level output,
    limited differences): every level step of every layer is one launch of the fused MLP kernel."""
    from fv3net_amd.emulation.models import HipLocalEmulator, load_emulator
    from fv3net_amd.local_mlp import RnnModel

    rng = np.random.default_rng(nz + ncol)
    st = cases.state(rng, nz, ncol, dtype)
    spec = cases.precpd_rnn(rng, st, nz, channels=channels, make=cases.product_makers())
    model = RnnModel(spec, device="cuda", use_graph=True)
    got = model.predict(_dev(st))
    # with use_graph the level sweep is captured into a HIP graph on the first call of a shape and replayed afterwards:
    # the replay (on other data, then on the same data again) gives exactly what launching the steps one by one gives
    eager = RnnModel(spec, device="cuda", use_graph=False)
    st2 = cases.state(np.random.default_rng(99), nz, ncol, dtype)
    for data in (st2, st):
        a, b = model.predict(_dev(data)), eager.predict(_dev(data))
        for name in spec.output_names:
            assert torch.equal(a[name], b[name]), name
    for name in spec.output_names:
        assert torch.equal(a[name], got[name]), name
    truth = mlp_np.forward_rnn(spec, {k: v.T for k, v in st.items()}, dtype=np.float64)
    assert list(got) == spec.output_names
    f32 = mlp_np.forward_rnn(spec, {k: v.T for k, v in st.items()}, dtype=np.float32)
    for name in spec.output_names:
        g, t = got[name].cpu().numpy(), truth[name].T
        assert g.shape == t.shape == ((1, ncol) if name == "total_precipitation" else (nz, ncol)), name
        # the recurrence compounds rounding over nz steps: no worse than a few times the float32 CPU evaluation, per level
        _check(g, t, name, f32=f32[name].T, compounding=True)
    assert (got["cloud_water_mixing_ratio_after_precpd"] >= 0).all() and (got["cloud_precpd_difference"] <= 0).all()
    # dump / load round trip through the emulator directory format
    HipLocalEmulator(spec).dump(str(tmp_path / "precpd"))
    loaded = load_emulator(str(tmp_path / "precpd"))
    assert yaml.safe_load(open(tmp_path / "precpd" / "spec.yaml"))["architecture"] == "rnn-v1-shared-weights"
    out = loaded({k: v.T for k, v in st.items()})
    for name in spec.output_names:
        np.testing.assert_array_equal(out[name].T, got[name].cpu().numpy())


def test_production_style_fv3config_both_hooks(tmp_path):
    """projects/microphysics/configs/gscond-and-precpd.yaml:268-287 in miniature, through ``get_hooks``: the gscond
    hook (dense-local regressor + classifier, conservation fixes, classifier masks, level masks filled from the
    inputs) and then the microphysics hook (RNN precpd model) update the same Fortran state in place."""
    from fv3net_amd.emulation.config import get_hooks
    from fv3net_amd.emulation.models import HipLocalEmulator

    rng = np.random.default_rng(13)
    nz, ncol = 20, 256
    st = cases.state(rng, nz, ncol)
    reg_spec = cases.regressor(rng, st, nz, make=cases.product_makers())
    reg_spec.outputs[0].after, reg_spec.outputs[1].after = E.QV_G, E.T_G
    cls_spec = cases.classifier(rng, st, nz, make=cases.product_makers())
    raw = mlp_np.forward_local(cls_spec, {k: v.T for k, v in st.items()}, dtype=np.float64)["gscond_classes"]
    cls_spec.out_bias = (cls_spec.out_bias - raw.mean(axis=(0, 1))).astype(np.float32)
    rnn_spec = cases.precpd_rnn(rng, st, nz, channels=32, make=cases.product_makers())
    for name, spec in (("gscond", reg_spec), ("classifier", cls_spec), ("precpd", rnn_spec)):
        HipLocalEmulator(spec).dump(str(tmp_path / name))
    start = nz - 5
    config = {"zhao_carr_emulation": {
        "gscond": {"path": str(tmp_path / "gscond"), "classifier_path": str(tmp_path / "classifier"),
                   "enforce_conservative": True, "gscond_cloud_conservative": True,
                   "mask_gscond_zero_cloud_classifier": True, "mask_gscond_no_tend_classifier": True,
                   "mask_emulator_levels": {
                       E.CLOUD_G: {"start": start, "fill_value": cases.CLOUD_IN},
                       E.QV_G: {"start": start, "fill_value": cases.QV_IN},
                       E.T_G: {"start": start, "fill_value": cases.T_IN}}},
        "model": {"path": str(tmp_path / "precpd")}}}
    (tmp_path / "fv3config.yml").write_text(yaml.safe_dump(config))
    gscond, microphysics, store = get_hooks(str(tmp_path / "fv3config.yml"))

    state = {k: v.copy() for k, v in st.items()}
    state[E.CLOUD_G] = st[cases.CLOUD_IN] * 1.1
    state["model_time"], state["rank"] = [2016, 8, 1, 0, 0, 0], 0
    assert gscond(state) is None
    # masked levels hold the inputs exactly (masks.py:46-76), the rest went through the conservation fix
    for out_name, in_name in ((E.CLOUD_G, cases.CLOUD_IN), (E.QV_G, cases.QV_IN), (E.T_G, cases.T_IN)):
        assert state[out_name].shape == (nz, ncol)
        np.testing.assert_array_equal(state[out_name][start:], st[in_name][start:])
        assert not np.array_equal(state[out_name][:start], st[in_name][:start])
    # water is conserved where the fix applied: cloud + vapour unchanged
    total_before = st[cases.CLOUD_IN] + st[cases.QV_IN]
    np.testing.assert_allclose(state[E.CLOUD_G] + state[E.QV_G], total_before, rtol=1e-12, atol=1e-18)
    assert (state[E.CLOUD_G] >= 0).all() and (state[E.QV_G] >= 0).all()
    assert state["gscond_classes"].shape == (4, nz, ncol)

    after_gscond = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in state.items()}
    assert microphysics(state) is None
    truth = mlp_np.forward_rnn(rnn_spec, {k: v.T for k, v in st.items()}, dtype=np.float64)
    for name in rnn_spec.output_names:
        _check(state[name], truth[name].T, name, tol=2e-5)
    assert state["total_precipitation"].shape == (1, ncol)
    for k, v in after_gscond.items():  # the second hook adds its outputs and leaves the rest alone
        if isinstance(v, np.ndarray):
            np.testing.assert_array_equal(state[k], v)
    assert store(state) is None


def test_online_schedule_switches_between_physics_and_emulator(tmp_path):
    """ModelConfig.online_schedule (config.py:83-88, 176-177): in the first half of every period the Fortran physics'
    own values are kept for the keys the emulator also produces, in the second half the emulator's."""
    import datetime

    from fv3net_amd.emulation.config import ModelConfig
    from fv3net_amd.emulation.models import HipLocalEmulator

    rng = np.random.default_rng(17)
    nz, ncol = 11, 96
    st = cases.state(rng, nz, ncol)
    spec = cases.regressor(rng, st, nz, make=cases.product_makers())
    spec.outputs[0].after, spec.outputs[1].after = E.QV_G, E.T_G
    HipLocalEmulator(spec).dump(str(tmp_path / "reg"))
    hook = ModelConfig.from_dict({"path": str(tmp_path / "reg"), "online_schedule": {
        "period": 7200, "initial_time": datetime.datetime(2016, 8, 1)}}).build()
    truth = mlp_np.forward_local(spec, {k: v.T for k, v in st.items()}, dtype=np.float64)
    fortran = {E.QV_G: st[cases.QV_IN] * 0.99, E.T_G: st[cases.T_IN] + 0.25}
    for minute, emulated in ((30, False), (90, True)):
        state = {**{k: v.copy() for k, v in st.items()}, **{k: v.copy() for k, v in fortran.items()}}
        # (call_py_fort hands the bookkeeping entries over as integer arrays: they stay on the host)
        state["model_time"] = np.array([2016, 8, 1, 0, minute // 60, minute % 60], dtype=np.int32)
        state["rank"] = np.array([3], dtype=np.int32)
        hook.microphysics(state)
        assert isinstance(state["model_time"], np.ndarray) and state["rank"][0] == 3
        for name in (E.QV_G, E.T_G):
            if emulated:
                _check(state[name], truth[name].T, name)
            else:
                np.testing.assert_array_equal(state[name], fortran[name])
        # outputs the Fortran state does not hold come from the emulator either way
        _check(state["humidity_gscond_difference"], truth["humidity_gscond_difference"].T, "humidity_gscond_difference")


@pytest.mark.parametrize("nz,ncol,dtype,channels,dense_width,dense_depth,go_backwards", [
    (79, 1024, np.float64, 256, 256, 1, True), (79, 333, np.float32, 64, 128, 2, False), (5, 64, np.float64, 32, 32, 0, True),
    (79, 200, np.float32, 128, 384, 1, True)])
def test_hybrid_rnn_emulator_matches_oracle(nz, ncol, dtype, channels, dense_width, dense_depth, go_backwards, tmp_path):
    """The "rnn" architecture (HybridRNN, architecture.py:78-147; VERDICT r02 missing #2): SimpleRNN final state ->
    dense head; a head without hidden layers and one wider than 256 go through the layered path.  Graph replay and eager
    launches give the same bits; the saved directory loads back through ``load_emulator``."""
    from fv3net_amd.emulation.models import HipLocalEmulator, load_emulator
    from fv3net_amd.local_mlp import HybridRnnModel

    rng = np.random.default_rng(nz * 1000 + ncol)
    st = cases.state(rng, nz, ncol, dtype)
    spec = cases.hybrid_rnn(rng, st, nz, channels=channels, dense_width=dense_width, dense_depth=dense_depth,
                            go_backwards=go_backwards, make=cases.product_makers())
    got = HybridRnnModel(spec, device="cuda", use_graph=True).predict(_dev(st))
    eager = HybridRnnModel(spec, device="cuda", use_graph=False).predict(_dev(st))
    src = {k: v.T for k, v in st.items()}
    truth = mlp_np.forward_hybrid_rnn(spec, src, dtype=np.float64)
    f32 = mlp_np.forward_hybrid_rnn(spec, src, dtype=np.float32)
    assert list(got) == spec.output_names
    for name in spec.output_names:
        assert torch.equal(got[name], eager[name]), name
        _check(got[name].cpu().numpy(), truth[name].T, name, f32=f32[name].T)
    assert float(got["total_precipitation"].min()) >= 0.0
    HipLocalEmulator(spec).dump(str(tmp_path / "hybrid"))
    assert yaml.safe_load(open(tmp_path / "hybrid" / "spec.yaml"))["architecture"] == "rnn"
    loaded = load_emulator(str(tmp_path / "hybrid"))
    again = loaded({k: v.T for k, v in st.items()})  # [sample, feature] numpy in, as the hook's models are called
    for name in spec.output_names:
        np.testing.assert_array_equal(again[name], got[name].cpu().numpy().T, err_msg=name)


def test_linear_architecture_loads_as_a_dense_emulator_without_hidden_layers(tmp_path):
    """architecture "linear" (architecture.py:285-302, MLPBlock(depth=0)): dumped and loaded like a dense model."""
    from fv3net_amd.emulation.models import HipEmulator, load_emulator
    from fv3net_amd.mlp import InputSpec, MlpSpec, OutputSpec

    rng = np.random.default_rng(3)
    nz, n = 79, 500
    spec = MlpSpec(inputs=[InputSpec("T", nz, center=rng.normal(250, 5, nz), scale=rng.uniform(5, 20, nz)),
                           InputSpec("q", nz, transform="log", eps=1e-8, center=rng.normal(-8, 1, nz), scale=rng.uniform(1, 2, nz))],
                   hidden_kernels=[], hidden_biases=[], outputs=[OutputSpec("dT", nz, scale=rng.uniform(0.5, 2, nz), center=rng.normal(0, 1, nz))],
                   out_kernel=(rng.normal(0, 1, (2 * nz, nz)) / np.sqrt(2 * nz)).astype(np.float32), out_bias=rng.normal(0, 0.1, nz).astype(np.float32))
    HipEmulator(spec).dump(str(tmp_path / "lin"))
    meta = yaml.safe_load(open(tmp_path / "lin" / "spec.yaml"))
    meta["architecture"] = "linear"
    yaml.safe_dump(meta, open(tmp_path / "lin" / "spec.yaml", "w"))
    model = load_emulator(str(tmp_path / "lin"))
    state = {"T": rng.uniform(200, 300, (n, nz)), "q": 10.0 ** rng.uniform(-8, -2, (n, nz))}
    got = model(state)["dT"]
    truth = mlp_np.forward(spec, state, dtype=np.float64)["dT"]
    f32 = mlp_np.forward(spec, state, dtype=np.float32)["dT"]
    assert_close_per_level(got, truth, f32, "dT")
