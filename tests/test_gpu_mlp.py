"""Parity of the fused MFMA MLP kernel (through the C ABI) with the numpy oracle.

The contraction is fp32 on the matrix cores (exact fp32 products, fp32 accumulation in k
order).  Truth is the oracle evaluated in float64; the bar is the north star's 1e-5 relative:
    max |gpu - truth| <= 1e-5 * max(|truth|)   per output variable
and the GPU must be no less accurate than the reference-style float32 CPU evaluation by more
than a small factor.  Zero-masked levels must be exactly 0.
"""
import numpy as np
import pytest
import torch

from oracle import mlp_np

from tolerances import assert_close_per_level, decades

pytestmark = pytest.mark.gpu


def _random_spec(rng, in_feats, width, n_hidden, out_feats, log_inputs=(), residual=None, limits=None, masks=None):
    from fv3net_amd.mlp import InputSpec, MlpSpec, OutputSpec, ResidualSpec

    inputs = []
    for name, (src, nf, start) in in_feats.items():
        inputs.append(
            InputSpec(
                source=src, nfeat=nf, start=start, transform="log" if name in log_inputs else "none",
                eps=1e-8 if name in log_inputs else 0.0,
                center=rng.normal(0, 1, nf).astype(np.float32), scale=rng.uniform(0.5, 2, nf).astype(np.float32),
            )
        )
    K = sum(i.nfeat for i in inputs)
    hk, hb = [], []
    fan = K
    for _ in range(n_hidden):
        hk.append((rng.normal(0, 1, (fan, width)) / np.sqrt(fan)).astype(np.float32))
        hb.append(rng.normal(0, 0.1, width).astype(np.float32))
        fan = width
    outputs = []
    for name, nf in out_feats.items():
        lim = (limits or {}).get(name, (None, None))
        outputs.append(
            # per-level standard deviations over 4.5 decades, centres of the levels' own size (VERDICT r02 #6a); outputs
            # with limits keep O(1) scales so that the limits of the tests still cut
            OutputSpec(name=name, nfeat=nf, scale=rng.uniform(0.5, 2, nf).astype(np.float32) if name in (limits or {}) else decades(rng, nf),
                       center=rng.normal(0, 1, nf).astype(np.float32) * (1 if name in (limits or {}) else decades(rng, nf, top=0.5)),
                       min=lim[0], max=lim[1], mask=(masks or {}).get(name))
        )
    F = sum(o.nfeat for o in outputs)
    residuals = [ResidualSpec(name=n, source=s, output=o) for n, (s, o) in (residual or {}).items()]
    return MlpSpec(inputs=inputs, hidden_kernels=hk, hidden_biases=hb, outputs=outputs,
                   out_kernel=(rng.normal(0, 1, (fan, F)) / np.sqrt(fan)).astype(np.float32),  # (no hidden layer: fan = K)
                   out_bias=rng.normal(0, 0.1, F).astype(np.float32), residuals=residuals)


def _check(spec, sources_sf, device, layout, src_dtype=np.float32, out_dtype=torch.float32):
    """sources_sf: name -> [sample, feature] numpy arrays."""
    from fv3net_amd.mlp import MlpModel

    dev_src = {}
    for k, v in sources_sf.items():
        a = v.astype(src_dtype)
        a = a if layout == "sample_feature" else np.ascontiguousarray(a.T)
        dev_src[k] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    truth = mlp_np.forward(spec, {k: v.astype(src_dtype) for k, v in sources_sf.items()}, dtype=np.float64)
    cpu32 = mlp_np.forward(spec, {k: v.astype(src_dtype) for k, v in sources_sf.items()}, dtype=np.float32)
    # both kernels on every case: the 128-sample-tile kernel (small_limit=0) and the feature-split kernel for small sample
    # counts (any count goes to it here); the results returned are the big kernel's
    results = {}
    for which, limit in (("mlp_fused_kernel", 0), ("mlp_small_kernel", 1 << 40)):
        model = MlpModel(spec, device=device, small_limit=limit)
        out = model.predict(dev_src, layout=layout, out_dtype=out_dtype)
        assert model.last_variant.startswith(which), model.last_variant
        assert set(out) == set(truth)
        for name in truth:
            got = out[name].cpu().numpy()
            got = got if layout == "sample_feature" else got.T
            assert got.shape == truth[name].shape, (name, got.shape, truth[name].shape)
            assert_close_per_level(got, truth[name], cpu32[name], f"{name} ({which})")
        results[which] = out
    out = results["mlp_fused_kernel"]
    return out, truth


@pytest.mark.parametrize("layout", ["feature_sample", "sample_feature"])
@pytest.mark.parametrize("width,n_hidden", [(8, 2), (12, 1), (40, 2), (100, 3), (256, 2)])
def test_dense_model_shapes(device, layout, width, n_hidden):
    """fv3fit DenseModel-like: T, q (79 levels) + cos_zenith (1) -> dQ1, dQ2 (79 each)."""
    rng = np.random.default_rng(width)
    n = 1000  # not a multiple of the 128-sample tile
    spec = _random_spec(
        rng, {"T": ("T", 79, 0), "q": ("q", 79, 0), "cosz": ("cosz", 1, 0)}, width, n_hidden,
        {"dQ1": 79, "dQ2": 79},
    )
    src = {"T": rng.uniform(200, 300, (n, 79)) / 100, "q": rng.uniform(0, 0.02, (n, 79)) * 50,
           "cosz": rng.uniform(0, 1, (n, 1))}
    _check(spec, src, device, layout)


def test_zhao_carr_emulator_shape(device):
    """K = 711 (9 x 79, three of them log-transformed views of raw fields), 2 x 256 hidden,
    396 outputs (1 + 5 x 79) and the five residual 'after = before + difference' outputs
    (projects/microphysics/train/dense.yaml:62-89)."""
    rng = np.random.default_rng(0)
    n = 2000
    raw = ["air_temperature_input", "specific_humidity_input", "cloud_water_mixing_ratio_input",
           "pressure_thickness_of_atmospheric_layer", "air_temperature_after_last_gscond",
           "specific_humidity_after_last_gscond"]
    in_feats = {r: (r, 79, 0) for r in raw}
    in_feats["log_cloud_input"] = ("cloud_water_mixing_ratio_input", 79, 0)
    in_feats["log_humidity_input"] = ("specific_humidity_input", 79, 0)
    in_feats["log_humidity_after_last_gscond"] = ("specific_humidity_after_last_gscond", 79, 0)
    in_feats = dict(sorted(in_feats.items()))  # combine_inputs sorts by key
    outs = {"total_precipitation": 1, "cloud_precpd_difference": 79, "temperature_precpd_difference": 79,
            "humidity_precpd_difference": 79, "temperature_gscond_difference": 79, "humidity_gscond_difference": 79}
    residual = {
        "air_temperature_after_gscond": ("air_temperature_input", "temperature_gscond_difference"),
        "specific_humidity_after_gscond": ("specific_humidity_input", "humidity_gscond_difference"),
        "cloud_water_mixing_ratio_after_precpd": ("cloud_water_mixing_ratio_input", "cloud_precpd_difference"),
        "air_temperature_after_precpd": ("air_temperature_input", "temperature_precpd_difference"),
        "specific_humidity_after_precpd": ("specific_humidity_input", "humidity_precpd_difference"),
    }
    spec = _random_spec(rng, in_feats, 256, 2, outs,
                        log_inputs=("log_cloud_input", "log_humidity_input", "log_humidity_after_last_gscond"),
                        residual=residual)
    src = {
        "air_temperature_input": rng.uniform(1.8, 3.1, (n, 79)),
        "specific_humidity_input": 10 ** rng.uniform(-8, -2, (n, 79)),
        "cloud_water_mixing_ratio_input": np.where(rng.random((n, 79)) < 0.7, 0.0, 10 ** rng.uniform(-10, -3, (n, 79))),
        "pressure_thickness_of_atmospheric_layer": rng.uniform(0.3, 1.5, (n, 79)),
    }
    src["air_temperature_after_last_gscond"] = src["air_temperature_input"] + rng.normal(0, 0.01, (n, 79))
    src["specific_humidity_after_last_gscond"] = src["specific_humidity_input"] * rng.uniform(0.9, 1.1, (n, 79))
    for layout in ("feature_sample", "sample_feature"):
        _check(spec, src, device, layout)
    # float64 sources (what the Fortran hook hands over) and float64 outputs
    _check(spec, src, device, "feature_sample", src_dtype=np.float64, out_dtype=torch.float64)


def test_limits_masks_and_clipped_inputs(device):
    """OutputLimit (output_limit.py:29-48), zero mask of clipped output levels (clip.py:33-46),
    input clip slices (clip.py:48-62): masked levels are exactly 0 (test_train.py:418-443)."""
    rng = np.random.default_rng(5)
    n = 513
    mask = np.ones(79, np.float32)
    mask[:10] = 0
    spec = _random_spec(rng, {"T": ("T", 60, 19), "q": ("q", 79, 0)}, 32, 2, {"dQ1": 79, "dQ2": 79},
                        limits={"dQ2": (-0.5, 0.75)}, masks={"dQ1": mask})
    src = {"T": rng.normal(0, 1, (n, 79)), "q": rng.normal(0, 1, (n, 79))}
    out, truth = _check(spec, src, device, "sample_feature")
    got = out["dQ1"].cpu().numpy()
    assert np.all(got[:, :10] == 0.0)
    q2 = out["dQ2"].cpu().numpy()
    assert q2.min() >= -0.5 and q2.max() <= 0.75
    assert (q2 == -0.5).any() and (q2 == 0.75).any()


def test_small_sample_kernel_column_results_do_not_depend_on_the_call(device):
    """The feature-split kernel for small sample counts (what a model rank's 2 304 columns select): a column's result is
    the same bit for bit whatever else is in the call -- 2 304 columns at once, the same columns in calls of 1 000 and
    1 304 or one by 37, in another order -- so a run does not depend on its domain decomposition while every rank stays on
    this kernel.  Against the 128-sample-tile kernel the results agree to rounding (both are checked against the float64
    oracle at the same tolerance), not bit for bit: `small_limit` pins one kernel where that matters."""
    import bench
    from fv3net_amd.mlp import MlpModel

    spec = bench.zc_spec(0, per_level_output_scale=decades)
    n = 2304
    src = bench.zc_inputs_numpy(np.random.default_rng(21), n)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v.T.astype(np.float64))).to(device) for k, v in src.items()}  # as the hook gets them
    model = MlpModel(spec, device=device)
    full = model.predict(dev)
    assert model.last_variant.startswith("mlp_small_kernel<true>"), model.last_variant
    truth = mlp_np.forward(spec, src, dtype=np.float64)
    cpu32 = mlp_np.forward(spec, src, dtype=np.float32)
    for name, t in truth.items():
        assert_close_per_level(full[name].cpu().numpy().T, t, cpu32[name], name)
    for lo, hi in ((0, 1000), (1000, 2304), (37, 74)):
        part = model.predict({k: v[:, lo:hi].contiguous() for k, v in dev.items()})
        for name in truth:
            assert torch.equal(part[name], full[name][:, lo:hi]), (name, lo, hi)
    perm = torch.randperm(n, device=device, generator=torch.Generator(device=device).manual_seed(3))
    shuffled = model.predict({k: v[:, perm].contiguous() for k, v in dev.items()})
    for name in truth:
        assert torch.equal(shuffled[name], full[name][:, perm]), name
    # the other kernel on the same columns: the same values to rounding
    big = MlpModel(spec, device=device, small_limit=0).predict(dev)
    for name, t in truth.items():
        assert_close_per_level(big[name].cpu().numpy().T, full[name].cpu().numpy().T, None, f"{name}: small against big", rel=1e-5)


def test_nan_input_propagates_only_to_its_sample(device):
    rng = np.random.default_rng(6)
    n = 300
    spec = _random_spec(rng, {"a": ("a", 20, 0)}, 16, 2, {"y": 5})
    from fv3net_amd.mlp import MlpModel

    a = rng.normal(0, 1, (n, 20)).astype(np.float32)
    a[17, 3] = np.nan
    model = MlpModel(spec, device=device)
    out = model.predict({"a": torch.from_numpy(a).to(device)}, layout="sample_feature")["y"].cpu().numpy()
    assert np.all(np.isnan(out[17]))
    assert not np.isnan(np.delete(out, 17, axis=0)).any()


def test_predict_does_not_mutate_inputs_and_is_deterministic(device):
    rng = np.random.default_rng(8)
    spec = _random_spec(rng, {"a": ("a", 79, 0), "b": ("b", 1, 0)}, 8, 2, {"y": 79})
    from fv3net_amd.mlp import MlpModel

    model = MlpModel(spec, device=device)
    a = torch.from_numpy(rng.normal(0, 1, (79, 700)).astype(np.float32)).to(device)
    b = torch.from_numpy(rng.normal(0, 1, (700,)).astype(np.float32)).to(device)
    a0, b0 = a.clone(), b.clone()
    y1 = model.predict({"a": a, "b": b})["y"].clone()
    y2 = model.predict({"a": a, "b": b})["y"]
    assert torch.equal(a, a0) and torch.equal(b, b0)
    assert torch.equal(y1, y2)


def test_unsupported_configurations_fail_loudly(device):
    """What is still refused: a recurrent cell (hidden-output model) outside the fused kernels' range."""
    from fv3net_amd._lib import Fv3HipError
    from fv3net_amd.mlp import MlpModel

    rng = np.random.default_rng(9)
    spec = _random_spec(rng, {"a": ("a", 8, 0)}, 300, 1, {"y": 4})
    spec.hidden_output = "h"
    with pytest.raises(Fv3HipError, match="hidden-output"):
        MlpModel(spec, device=device)


def _check_layered(spec, sources_sf, device, layout="feature_sample", src_dtype=np.float32, out_dtype=torch.float32):
    from fv3net_amd.mlp import MlpModel

    dev_src = {}
    for k, v in sources_sf.items():
        a = v.astype(src_dtype)
        dev_src[k] = torch.from_numpy(np.ascontiguousarray(a if layout == "sample_feature" else a.T)).to(device)
    cast = {k: v.astype(src_dtype) for k, v in sources_sf.items()}
    truth = mlp_np.forward(spec, cast, dtype=np.float64)
    cpu32 = mlp_np.forward(spec, cast, dtype=np.float32)
    model = MlpModel(spec, device=device)
    out = model.predict(dev_src, layout=layout, out_dtype=out_dtype)
    assert model.last_variant.startswith("layered"), model.last_variant
    assert set(out) == set(truth)
    for name in truth:
        got = out[name].cpu().numpy()
        got = got if layout == "sample_feature" else got.T
        assert got.shape == truth[name].shape
        assert_close_per_level(got, truth[name], cpu32[name], f"{name} (layered)")
    return model, dev_src, out


@pytest.mark.parametrize("layout", ["feature_sample", "sample_feature"])
@pytest.mark.parametrize("width,n_hidden,n", [(0, 0, 1000), (0, 0, 37), (320, 1, 257), (512, 3, 1000), (300, 2, 70000)])
def test_layered_path_networks_outside_the_fused_kernels(device, layout, width, n_hidden, n):
    """No hidden layer (the reference's "linear" architecture, architecture.py:285-302) and hidden layers wider than 256
    (VERDICT r02 missing #2): layer by layer through HBM, against the same oracle at the same per-level tolerance; 70 000
    samples cross the 65 536-sample slab of the scratch."""
    rng = np.random.default_rng(1000 * width + n)
    spec = _random_spec(rng, {"T": ("T", 79, 0), "q": ("q", 79, 0), "cosz": ("cosz", 1, 0), "qlog": ("q", 79, 0)}, width, n_hidden,
                        {"dQ1": 79, "dQ2": 79, "flux": 1}, log_inputs=("qlog",), residual={"q_after": ("q", "dQ2")},
                        limits={"flux": (0.0, 0.6)}, masks={"flux": np.ones(1, np.float32)})
    src = {"T": rng.uniform(200, 300, (n, 79)) / 100, "q": rng.uniform(0, 0.02, (n, 79)) * 50, "cosz": rng.uniform(0, 1, (n, 1))}
    _check_layered(spec, src, device, layout)


def test_layered_path_float64_sources_and_outputs_linear_activation_many_inputs(device):
    """float64 sources read as they lie, float64 outputs, hidden layers without activation, more inputs (2 400) than the
    fused kernels' input table holds."""
    rng = np.random.default_rng(77)
    n = 600
    spec = _random_spec(rng, {"a": ("a", 1200, 0), "b": ("b", 1200, 0)}, 64, 2, {"y": 79, "z": 3})
    _check_layered(spec, {"a": rng.normal(0, 1, (n, 1200)), "b": rng.normal(0, 1, (n, 1200))}, device, src_dtype=np.float64,
                   out_dtype=torch.float64)
    spec = _random_spec(rng, {"a": ("a", 40, 0)}, 96, 2, {"y": 79})
    spec.activation = "linear"
    _check_layered(spec, {"a": rng.normal(0, 1, (n, 40))}, device, src_dtype=np.float64)


def test_layered_path_column_results_do_not_depend_on_the_call(device):
    """A column's outputs are the same bits whatever call (slab, position) it arrives in; NaNs stay in their sample."""
    rng = np.random.default_rng(5)
    n = 66000
    spec = _random_spec(rng, {"a": ("a", 50, 0)}, 288, 2, {"y": 79})
    a = rng.normal(0, 1, (n, 50)).astype(np.float32)
    a[65540, 7] = np.nan
    model, dev_src, full = _check_layered(spec, {"a": np.nan_to_num(a)}, device)
    full = model.predict({"a": torch.from_numpy(np.ascontiguousarray(a.T)).to(device)})["y"]
    bad = torch.isnan(full).any(dim=0)
    assert int(bad.sum()) == 1 and bool(bad[65540])
    for lo, hi in ((0, 300), (65500, 66000), (1234, 1235)):
        part = model.predict({"a": torch.from_numpy(np.ascontiguousarray(a[lo:hi].T)).to(device)})["y"]
        assert torch.equal(torch.nan_to_num(part), torch.nan_to_num(full[:, lo:hi])), (lo, hi)


def test_full_size_c384_properties(device):
    """All 884 736 C384 columns in one call: every column must equal the same column
    evaluated in a small batch (the result may not depend on tile position), and a permutation
    of the columns permutes the outputs."""
    rng = np.random.default_rng(1)
    from fv3net_amd.mlp import MlpModel

    spec = _random_spec(rng, {"a": ("a", 79, 0), "b": ("b", 79, 0)}, 64, 2, {"y": 79, "z": 1})
    model = MlpModel(spec, device=device, small_limit=0)  # (one kernel for both calls: bit-identity holds within a kernel)
    n = 6 * 384 * 384
    g = torch.Generator(device=device).manual_seed(0)
    a = torch.randn((79, n), device=device, generator=g)
    b = torch.randn((79, n), device=device, generator=g)
    full = model.predict({"a": a, "b": b})
    idx = torch.randint(0, n, (4096,), device=device, generator=g)
    sub = model.predict({"a": a[:, idx].contiguous(), "b": b[:, idx].contiguous()})
    for name in ("y", "z"):
        assert torch.equal(full[name][:, idx], sub[name])
    perm = torch.randperm(n, device=device, generator=g)
    permuted = model.predict({"a": a[:, perm].contiguous(), "b": b[:, perm].contiguous()})
    assert torch.equal(permuted["y"], full["y"][:, perm])


@pytest.mark.parametrize("n", [4096, 147456])
@pytest.mark.parametrize("residuals", [True, False])
def test_timed_kernel_against_oracle(device, n, residuals):
    """The instantiation bench.py times -- float32 [feature, sample] sources, fast I/O, K = 711 -> 256 -> 256 -> 396 with
    the fast-log chunks, the tile-major output layer and the branch-free side epilogue ("plain"; "residual" with the
    five ``after = before + difference`` outputs of dense.yaml:22-36) -- on bench.py's own model and input
    distributions against the float64 oracle (microphysics.py:123-139).  The launch must really be that kernel."""
    import bench
    from fv3net_amd.mlp import MlpModel

    # (output standard deviations per level, falling by 4.5 decades from the surface up: the gate below is per level)
    spec = bench.zc_spec(0, residuals=residuals, per_level_output_scale=decades)
    src = bench.zc_inputs_numpy(np.random.default_rng(11), n)
    model = MlpModel(spec, device=device, small_limit=0)  # (the kernel of the full-size call, whatever n is here)
    dev_src = {k: torch.from_numpy(np.ascontiguousarray(v.T)).to(device) for k, v in src.items()}
    out = model.predict(dev_src)
    assert model.last_variant == ("mlp_fused_kernel<8,false,true,false,false,false> epilogue="
                                  + ("residual" if residuals else "plain")), model.last_variant
    truth = mlp_np.forward(spec, src, dtype=np.float64)
    cpu32 = mlp_np.forward(spec, src, dtype=np.float32)
    assert set(out) == set(truth) and len(truth) == (11 if residuals else 6)
    for name, t in truth.items():
        got = out[name].cpu().numpy().T
        assert_close_per_level(got, t, cpu32[name], name)
    if residuals:  # the residual rows are exactly before + the difference row the kernel stored
        for name, (before, diff) in bench.RESIDUALS.items():
            assert torch.equal(out[name], dev_src[before] + out[diff]), name
    # the same columns inside a full C384-sized call (other tile positions, the persistent loop wrapped around)
    if n == 4096:
        big = 6 * 384 * 384
        reps = big // n
        full = model.predict({k: v.repeat(1, reps) for k, v in dev_src.items()})
        for name in truth:
            assert torch.equal(full[name][:, -n:], out[name]), name


def test_timed_kernel_float64_sources_take_general_epilogue(device):
    """float64 sources (what call_py_fort hands the hook) with residual outputs: same results as float32 sources to
    rounding of the inputs, through the f64 instantiation."""
    import bench
    from fv3net_amd.mlp import MlpModel

    spec = bench.zc_spec(0)
    src = bench.zc_inputs_numpy(np.random.default_rng(12), 4096)
    model = MlpModel(spec, device=device, small_limit=0)
    out = model.predict({k: torch.from_numpy(np.ascontiguousarray(v.T.astype(np.float64))).to(device) for k, v in src.items()})
    assert model.last_variant.startswith("mlp_fused_kernel<8,true,true,false,false,false>"), model.last_variant
    truth = mlp_np.forward(spec, src, dtype=np.float64)
    for name, t in truth.items():
        assert np.max(np.abs(out[name].cpu().numpy().T - t)) <= 1e-5 * np.max(np.abs(t)), name


@pytest.mark.parametrize("n", [4096, 147456 + 32])
@pytest.mark.parametrize("residuals", [True, False])
def test_split_bf16_kernel_against_oracle(device, n, residuals):
    """EXPERIMENTAL path (csrc/mlp_bf16x3.hip): the Zhao-Carr network with every fp32 operand split into three bf16 pieces
    and the contraction on the bf16 matrix cores, against the same float64 oracle and the same bar as the fp32 kernel --
    and no less accurate than the float32 CPU evaluation by more than a small factor."""
    import bench
    from fv3net_amd.mlp import MlpModelSplitBf16

    spec = bench.zc_spec(0, residuals=residuals, per_level_output_scale=decades)
    src = bench.zc_inputs_numpy(np.random.default_rng(13), n)
    model = MlpModelSplitBf16(spec, device=device)
    out = model.predict({k: torch.from_numpy(np.ascontiguousarray(v.T)).to(device) for k, v in src.items()})
    truth = mlp_np.forward(spec, src, dtype=np.float64)
    cpu32 = mlp_np.forward(spec, src, dtype=np.float32)
    assert set(out) == set(truth)
    for name, t in truth.items():
        got = out[name].cpu().numpy().T
        assert_close_per_level(got, t, cpu32[name], name, slack32=2e-7)


@pytest.mark.parametrize("n_hidden,out_feats,n,pad", [(1, {"y": 79, "z": 5}, 1000, 37), (3, {"a": 1, "b": 79, "c": 79}, 257, 37),
                                                      (2, {"y": 70}, 128, 37), (1, {"y": 79, "z": 5}, 1000, 0),
                                                      (3, {"a": 1, "b": 79, "c": 79}, 260, 0)])
def test_split_bf16_kernel_other_architectures(device, n_hidden, out_feats, n, pad):
    """The split-bf16 kernel on other shapes: 1 and 3 hidden layers (the weight stream's chunk types change at different
    k-steps), 3 / 5 output tiles, inputs that are feature ranges of a shared source (start > 0), log inputs, a residual
    output, sample counts that end inside a 128-sample tile and feature rows longer than the sample count (strided views:
    the 4-byte epilogue; pad = 0: aligned rows, the 16-byte epilogue)."""
    from fv3net_amd.mlp import MlpModelSplitBf16

    rng = np.random.default_rng(5)
    first = next(iter(out_feats))
    res_out = [k for k, v in out_feats.items() if v >= 19][0]
    nres = out_feats[res_out]
    in_feats = {"T": ("state", nres, 0), "q": ("state", 30, nres), "logq": ("state", 30, nres), "ps": ("ps", 1, 0)}
    spec = _random_spec(rng, in_feats, 256, n_hidden, out_feats, log_inputs=("logq",), residual={"T_after": ("state", res_out)})
    # (pad > 0: rows are views into wider arrays, feature stride > n)
    state = np.concatenate([rng.normal(0, 1, (nres, n + pad)), 10 ** rng.uniform(-8, -2, (30, n + pad))]).astype(np.float32)
    ps = rng.normal(0, 1, (1, n + pad)).astype(np.float32)
    dev_src = {"state": torch.from_numpy(state).to(device)[:, :n], "ps": torch.from_numpy(ps).to(device)[:, :n]}
    model = MlpModelSplitBf16(spec, device=device)
    out = model.predict(dev_src)
    host = {"state": np.ascontiguousarray(state[:, :n].T), "ps": np.ascontiguousarray(ps[:, :n].T)}
    truth = mlp_np.forward(spec, host, dtype=np.float64)
    cpu32 = mlp_np.forward(spec, host, dtype=np.float32)
    assert set(out) == set(truth) and first in out
    for name, t in truth.items():
        got = out[name].cpu().numpy().T
        assert_close_per_level(got, t, cpu32[name], name, slack32=2e-7)


def test_split_bf16_kernel_adversarial_inputs(device):
    """the domain of the split-bf16 arithmetic, pinned.  x = hi + mid + lo in bf16 (8 mantissa bits each, the
    float32 exponent range) is exact while lo stays a normal bf16 number, |x| >~ 2^-102; of the nine partial products six are
    kept, the dropped ones are <= 2^-25 |x w| -- under one float32 rounding per product.  Checked against the float64 oracle
    per (variable, level) and against the fp32 kernel sample for sample:
      * features whose magnitudes span 2^-60 .. 2^60 with standard deviations (folded into the weights) to match;
      * log inputs at and below their floor (0, 1e-12, exactly eps);
      * NaN in a sample -> NaN in every output of that sample, in both kernels, nowhere else;
      * +-Inf in a sample: the fp32 kernel yields Inf or NaN by IEEE rules; the split kernel yields NaN or Inf (Inf - hi is
        NaN in the mid piece) -- non-finite either way, never a finite number; other samples untouched."""
    from fv3net_amd.mlp import InputSpec, MlpModel, MlpModelSplitBf16, MlpSpec, OutputSpec

    rng = np.random.default_rng(21)
    nf, n, w = 79, 1024, 256
    expo = rng.integers(-60, 61, nf)
    mag = (2.0 ** expo).astype(np.float32)
    x = (rng.normal(0, 1, (n, nf)) * mag).astype(np.float32)
    q = np.where(rng.random((n, nf)) < 0.3, 0.0, 10.0 ** rng.uniform(-13, -3, (n, nf))).astype(np.float32)
    q[:8, :] = 1e-10  # exactly the floor
    q[8:16, :] = 1e-12
    inputs = [InputSpec("x", nf, center=(rng.normal(0, 0.1, nf) * mag).astype(np.float32), scale=mag),
              InputSpec("q", nf, transform="log", eps=1e-10, center=np.full(nf, -15.0, np.float32), scale=np.float32(4.0))]
    glorot = lambda a, b: rng.uniform(-1, 1, (a, b)).astype(np.float32) * np.float32(np.sqrt(6.0 / (a + b)))
    spec = MlpSpec(inputs=inputs, hidden_kernels=[glorot(2 * nf, w), glorot(w, w)],
                   hidden_biases=[rng.normal(0, 0.01, w).astype(np.float32) for _ in range(2)],
                   outputs=[OutputSpec("y", nf, scale=decades(rng, nf), center=(rng.normal(0, 1, nf) * decades(rng, nf)).astype(np.float32))],
                   out_kernel=glorot(w, nf), out_bias=rng.normal(0, 0.01, nf).astype(np.float32))
    split, fp32 = MlpModelSplitBf16(spec, device=device), MlpModel(spec, device=device)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(device)
    src = {"x": x, "q": q}
    truth = mlp_np.forward(spec, src, dtype=np.float64)["y"]
    cpu32 = mlp_np.forward(spec, src, dtype=np.float32)["y"]
    got_s = split.predict({k: dev(v) for k, v in src.items()})["y"].cpu().numpy().T
    got_f = fp32.predict({k: dev(v) for k, v in src.items()})["y"].cpu().numpy().T
    assert_close_per_level(got_f, truth, cpu32, "y (fp32 kernel)")
    assert_close_per_level(got_s, truth, cpu32, "y (split-bf16 kernel)", slack32=2e-7)
    # non-finite inputs
    bad = x.copy()
    bad[100, 5], bad[200, 40], bad[300, 7], bad[301, 70] = np.nan, np.inf, -np.inf, np.inf
    out_s = split.predict({"x": dev(bad), "q": dev(q)})["y"].cpu().numpy().T
    out_f = fp32.predict({"x": dev(bad), "q": dev(q)})["y"].cpu().numpy().T
    touched = np.zeros(n, bool)
    touched[[100, 200, 300, 301]] = True
    np.testing.assert_array_equal(out_s[~touched], got_s[~touched])  # other samples: bit for bit what they were
    np.testing.assert_array_equal(out_f[~touched], got_f[~touched])
    assert np.isnan(out_s[100]).all() and np.isnan(out_f[100]).all()
    assert not np.isfinite(out_f[[200, 300, 301]]).any() and not np.isfinite(out_s[[200, 300, 301]]).any()


def test_split_bf16_kernel_rows_of_256_mib(device):
    """Feature rows of 256 MiB and more (the dense-local emulators run 79 x 884 736 points per call): 16 rows no longer fit
    32-bit offsets, the offsets of the padding rows of a k-step are forced out of range instead of wrapping into it."""
    from fv3net_amd.mlp import MlpModelSplitBf16

    rng = np.random.default_rng(3)
    n = 68 * 1024 * 1024  # 272 MiB per row
    spec = _random_spec(rng, {"a": ("a", 2, 0)}, 256, 2, {"y": 1})
    x = torch.randn((2, n), device=device, generator=torch.Generator(device=device).manual_seed(1))
    out = MlpModelSplitBf16(spec, device=device).predict({"a": x})["y"]
    for lo in (0, n // 2 - 17, n - 4096):
        host = {"a": np.ascontiguousarray(x[:, lo:lo + 4096].T.cpu().numpy())}
        truth = mlp_np.forward(spec, host, dtype=np.float64)["y"]
        got = out[:, lo:lo + 4096].cpu().numpy().T
        assert np.max(np.abs(got - truth)) <= 1e-5 * np.max(np.abs(truth))


def test_split_bf16_kernel_refuses_what_it_does_not_implement(device):
    from fv3net_amd._lib import Fv3HipError
    from fv3net_amd.mlp import MlpModelSplitBf16

    rng = np.random.default_rng(1)
    spec = _random_spec(rng, {"a": ("a", 79, 0)}, 64, 2, {"y": 79})
    with pytest.raises(Fv3HipError, match="width 256"):
        MlpModelSplitBf16(spec, device=device)
    spec = _random_spec(rng, {"a": ("a", 79, 0)}, 256, 2, {"y": 79}, limits={"y": (0.0, None)})
    with pytest.raises(Fv3HipError, match="limits"):
        MlpModelSplitBf16(spec, device=device)
    # sources the C entry point could not check (it sees pointers, strides and n): refused before the launch
    spec = _random_spec(rng, {"a": ("a", 79, 0), "b": ("b", 40, 0)}, 256, 2, {"y": 79})
    model = MlpModelSplitBf16(spec, device=device)
    good = {"a": torch.zeros((79, 256), device=device), "b": torch.zeros((40, 256), device=device)}
    assert model.predict(good)["y"].shape == (79, 256)
    with pytest.raises(ValueError, match="has 39 features, the model needs 40"):
        model.predict({**good, "b": torch.zeros((39, 256), device=device)})
    with pytest.raises(ValueError, match="differ in their number of samples"):
        model.predict({**good, "b": torch.zeros((40, 128), device=device)})
    with pytest.raises(ValueError, match="must be 1-D or 2-D"):
        model.predict({**good, "b": torch.zeros((40, 16, 16), device=device)})


@pytest.mark.parametrize("in_dtype,n_buffers", [(np.float32, 3), (np.float64, 2)])
def test_snapshot_stream_matches_direct_predict(device, in_dtype, n_buffers):
    """Streaming snapshots through pinned host buffers on three HIP streams gives, for every snapshot and
    in order, bit-identically what one synchronous predict on the same data gives (buffer reuse across
    more snapshots than buffers included)."""
    from fv3net_amd.fit.streaming import SnapshotStream
    from fv3net_amd.mlp import MlpModel

    rng = np.random.default_rng(7)
    spec = _random_spec(rng, {"T": ("T", 19, 0), "q": ("q", 19, 0), "logq": ("q", 19, 0), "ps": ("ps", 1, 0)}, 64, 2,
                        {"dT": 19, "dq": 19, "pr": 1}, log_inputs=("logq",), residual={"T_after": ("T", "dT")})
    model = MlpModel(spec, device=device)
    n, n_snap = 4096, 8
    data = [{"T": rng.normal(0, 1, (19, n)).astype(in_dtype), "q": (10 ** rng.uniform(-8, -2, (19, n))).astype(in_dtype),
             "ps": rng.normal(0, 1, (1, n)).astype(in_dtype)} for _ in range(n_snap)]
    want = []
    for d in data:
        out = model.predict({k: torch.from_numpy(v).to(device) for k, v in d.items()})
        want.append({k: v.cpu().numpy() for k, v in out.items()})
    stream = SnapshotStream(model, n, in_dtype=in_dtype, n_buffers=n_buffers)
    seen = []

    def fill(i, inputs):
        for k, v in inputs.items():
            v[...] = data[i][k]

    def consume(i, outputs):
        seen.append(i)
        assert set(outputs) == set(spec.output_names)
        for k, v in outputs.items():
            np.testing.assert_array_equal(v, want[i][k], err_msg=f"snapshot {i} {k}")

    stream.run(n_snap, fill, consume)
    assert seen == list(range(n_snap))
    stream.run(3, fill, consume)  # reusable
    assert seen[-3:] == [0, 1, 2]


@pytest.mark.parametrize("width", [16, 64, 100, 128, 256])
@pytest.mark.parametrize("n_samples,with_outputs", [(4096, True), (1000, True), (4096, False), (77, False)])
def test_hidden_output_models(device, width, n_samples, with_outputs):
    """Models whose last hidden layer is an output of its own (the cells of the RNN emulators): every hidden
    tiling (width 16 ... 256), the fast-I/O and the general kernels (sample counts that are / are not multiples of
    32), with and without ordinary outputs, against the float64 oracle."""
    from fv3net_amd.mlp import MlpModel

    rng = np.random.default_rng(width + n_samples)
    spec = _random_spec(rng, {"a": ("a", 21, 0), "b": ("b", 40, 0)}, width, 2, {"y": 7, "z": 33} if with_outputs else {})
    spec.hidden_output = "h"
    src = {"a": rng.normal(0, 1, (n_samples, 21)).astype(np.float32), "b": rng.normal(0, 1, (n_samples, 40)).astype(np.float32)}
    model = MlpModel(spec, device=device)
    got = model.predict({k: torch.from_numpy(np.ascontiguousarray(v.T)).to(device) for k, v in src.items()})
    truth = mlp_np.forward(spec, src, dtype=np.float64)
    assert list(got) == (["y", "z"] if with_outputs else []) + ["h"]
    assert got["h"].shape == (width, n_samples)
    assert (got["h"] >= 0).all()
    for name in got:
        t = truth[name].T
        assert np.max(np.abs(got[name].cpu().numpy() - t)) <= 1e-5 * np.max(np.abs(t)), name
    # strided outputs (a level's slice of a larger array, as the RNN sweep passes them) and reuse of the handle
    big = torch.zeros((width, 3, n_samples), device=device)
    outs = {"h": big[:, 1]}
    if with_outputs:
        outs.update({"y": torch.empty((7, n_samples), device=device), "z": torch.empty((33, n_samples), device=device)})
    model.predict({k: torch.from_numpy(np.ascontiguousarray(v.T)).to(device) for k, v in src.items()}, out=outs)
    assert torch.equal(big[:, 1], got["h"]) and not big[:, 0].any() and not big[:, 2].any()


@pytest.mark.parametrize("width", [16, 64, 100, 128, 256])
@pytest.mark.parametrize("k_in,n_out,n_samples", [(12, 2, 4096), (16, 4, 2048), (40, 1, 4096), (12, 3, 1000), (300, 4, 4096)])
def test_small_output_and_short_first_chunk_variants(device, width, k_in, n_out, n_samples):
    """Models with at most 4 outputs take the v_mfma_f32_4x4x1 output path on fast-I/O launches (and, with at most 16
    inputs, the 8-slot first chunk); ragged sample counts fall back to the ordinary kernels of the same handle.
    Every hidden tiling, against the float64 oracle; with and without a hidden output."""
    from fv3net_amd.mlp import MlpModel

    rng = np.random.default_rng(width * 7 + k_in + n_out)
    outs = {f"y{i}": 1 for i in range(n_out)}
    src = {"a": rng.normal(0, 1, (n_samples, k_in)).astype(np.float32)}
    dev_src = {"a": torch.from_numpy(np.ascontiguousarray(src["a"].T)).to(device)}
    for hidden_output in (None, "h"):
        spec = _random_spec(rng, {"a": ("a", k_in, 0)}, width, 2, outs)
        spec.hidden_output = hidden_output
        got = MlpModel(spec, device=device).predict(dev_src)
        truth = mlp_np.forward(spec, src, dtype=np.float64)
        assert list(got) == list(outs) + ([hidden_output] if hidden_output else [])
        for name in got:
            t = truth[name].T
            assert got[name].shape == t.shape
            assert np.max(np.abs(got[name].cpu().numpy() - t)) <= 1e-5 * max(np.max(np.abs(t)), 1e-30), (name, hidden_output)
