"""The MLP oracle pinned where the reference's own tests allow it (normalisation regtest outputs,
OutputLimit known answers) and cross-checked against an independent torch-CPU implementation.
The dense contraction itself is parity-unpinned against TensorFlow (see oracle/mlp_np.py)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from oracle import mlp_np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "normalization_regtest.json")


def test_standard_normalize_regtest():
    # external/fv3fit/tests/keras/test_shared_utils.py:7-24 and its _regtest_outputs
    g = json.load(open(GOLDEN))
    sample = np.arange(20).reshape(10, 2).astype(np.float32)
    mean, std = mlp_np.fit_mean_per_feature(sample), mlp_np.fit_std_per_feature(sample)
    got = mlp_np.norm_forward(sample, mean, std, epsilon=1e-7)
    np.testing.assert_allclose(got, np.array(g["normalize"], np.float32), rtol=2e-7)
    got = mlp_np.norm_backward(sample, mean, std)
    np.testing.assert_allclose(got, np.array(g["denormalize"], np.float32), rtol=2e-7)


@pytest.mark.parametrize(
    "vmin, vmax, expected",
    [(None, None, [-2.0, -1.0, 0.0, 1.0, 2.0]), (None, 1.5, [-2.0, -1.0, 0.0, 1.0, 1.5]),
     (None, 1.0, [-2.0, -1.0, 0.0, 1.0, 1.0]), (-1.5, None, [-1.5, -1.0, 0.0, 1.0, 2.0]),
     (-1.0, None, [-1.0, -1.0, 0.0, 1.0, 2.0]), (-1.5, 1.5, [-1.5, -1.0, 0.0, 1.0, 1.5])],
)
def test_output_limit_known_answers(vmin, vmax, expected):
    # external/fv3fit/tests/keras/test_output_limit.py:6-26
    out = mlp_np.limit_output(np.array([-2.0, -1.0, 0.0, 1.0, 2.0]), vmin, vmax)
    assert np.array_equal(expected, out)


def _spec(rng, k, w, nh, f):
    ns = types.SimpleNamespace
    hk, hb, fan = [], [], k
    for _ in range(nh):
        hk.append((rng.normal(0, 1, (fan, w)) / np.sqrt(fan)).astype(np.float32))
        hb.append(rng.normal(0, 0.1, w).astype(np.float32))
        fan = w
    return ns(
        inputs=[ns(source="a", nfeat=k, start=0, transform="none", eps=0.0, center=rng.normal(0, 1, k).astype(np.float32),
                   scale=rng.uniform(0.5, 2, k).astype(np.float32))],
        hidden_kernels=hk, hidden_biases=hb,
        outputs=[ns(name="y", nfeat=f, scale=rng.uniform(0.5, 2, f).astype(np.float32),
                    center=rng.normal(0, 1, f).astype(np.float32), min=None, max=None, mask=None)],
        out_kernel=(rng.normal(0, 1, (w, f)) / np.sqrt(w)).astype(np.float32), out_bias=rng.normal(0, 0.1, f).astype(np.float32),
        residuals=[],
    )


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
def test_forward_against_torch_cpu(dtype, tol):
    rng = np.random.default_rng(0)
    spec = _spec(rng, 40, 16, 2, 7)
    x = rng.normal(0, 1, (300, 40)).astype(np.float32)
    got = mlp_np.forward(spec, {"a": x}, dtype=dtype)["y"]
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    i = spec.inputs[0]
    h = (torch.from_numpy(x).to(tdt) - torch.from_numpy(i.center).to(tdt)) / torch.from_numpy(i.scale).to(tdt)
    for kern, b in zip(spec.hidden_kernels, spec.hidden_biases):
        lin = torch.nn.Linear(kern.shape[0], kern.shape[1]).to(tdt)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(kern.T).to(tdt))
            lin.bias.copy_(torch.from_numpy(b).to(tdt))
            h = torch.relu(lin(h))
    with torch.no_grad():
        y = h @ torch.from_numpy(spec.out_kernel).to(tdt) + torch.from_numpy(spec.out_bias).to(tdt)
        o = spec.outputs[0]
        y = y * torch.from_numpy(o.scale).to(tdt) + torch.from_numpy(o.center).to(tdt)
    np.testing.assert_allclose(got, y.numpy(), rtol=tol, atol=tol)


# ------------------------------------------------------------------------------------------------
# dense-local architecture, piecewise interpolation, conditional scaling
# ------------------------------------------------------------------------------------------------
def test_piecewise_known_answers():
    # external/fv3fit/tests/keras/test_math.py:7-15
    x = np.array([0.0, 1, 2], np.float32)
    y = 2.0 * x
    got = mlp_np.piecewise(x, y, np.array([-2, 0.5, 0.75, 1, 1.5, 2.5], np.float32))
    np.testing.assert_array_equal(got, y[[0, 0, 0, 1, 1, 2]])
    assert mlp_np.piecewise(x, y, np.ones((10, 10), np.float32)).shape == (10, 10)  # test_math.py:18-23
    # external/fv3fit/tests/emulation/test_transform.py:145-153: bins 0,1,2 with bin means [1, 2]; the interpolant is
    # piecewise(edges[:-1], values, .)
    got = mlp_np.piecewise(np.array([0.0, 1.0]), np.array([1.0, 2.0]), np.array([-1, 0.0, 1, 3]))
    np.testing.assert_array_equal(got, [1, 1, 2, 2])


@pytest.mark.parametrize("min_scale", [0, 1, 2, 3, 4])
def test_conditionally_scaled_backward_known_answers(min_scale):
    # external/fv3fit/tests/emulation/test_transform.py:168-220: scale(.) = 2, center(.) = 0; the forward
    # transform of `expected * max(scale, min_scale)` is `expected` = 1, and backward undoes it
    on = np.zeros((3, 4), np.float32)
    y = np.ones((3, 4), np.float32)
    got = mlp_np.conditionally_scaled_backward(y, on, np.array([-1.0], np.float32), np.array([2.0], np.float32),
                                               np.array([0.0], np.float32), min_scale)
    np.testing.assert_array_almost_equal(got, 1.0 * max(2.0, min_scale))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-5)])
def test_forward_local_against_torch_cpu(dtype, tol):
    """The dense-local graph against an independent torch-CPU statement: Linear layers on a
    [sample, z, channel] tensor, Conv1d(kernel_size=1) heads as in RNNOutput(share_conv_weights=True)."""
    import local_cases

    rng = np.random.default_rng(3)
    nz, ncol = 9, 40
    st = local_cases.state(rng, nz, ncol)
    for spec in (local_cases.regressor(rng, st, nz), local_cases.classifier(rng, st, nz)):
        got = mlp_np.forward_local(spec, {k: v.T for k, v in st.items()}, dtype=dtype)
        td = torch.float64 if dtype == np.float64 else torch.float32
        cols = []
        for i in spec.inputs:
            x = torch.from_numpy(np.atleast_2d(st[i.source]).T.astype(np.float32)).to(td)  # [sample, nz or 1]
            if i.transform == "log":
                x = torch.log(torch.clamp(x, min=float(np.float32(i.eps))))
            x = (x - torch.from_numpy(np.atleast_1d(i.center)).to(td)) / torch.tensor(float(i.scale), dtype=td)
            cols.append(x.expand(ncol, nz).unsqueeze(-1))
        h = torch.cat(cols, -1)
        for kern, b in zip(spec.hidden_kernels, spec.hidden_biases):
            h = torch.relu(torch.nn.functional.linear(h, torch.from_numpy(kern.T.copy()).to(td), torch.from_numpy(b).to(td)))
        conv_w = torch.from_numpy(spec.out_kernel.T.copy()).to(td).unsqueeze(-1)  # [out_channels, in_channels, 1]
        y = torch.nn.functional.conv1d(h.transpose(1, 2), conv_w, torch.from_numpy(spec.out_bias).to(td)).transpose(1, 2)
        c0 = 0
        for o in spec.outputs:
            ref = y[..., c0:c0 + o.channels]
            c0 += o.channels
            if o.channels != 1:
                np.testing.assert_allclose(got[o.name], ref.numpy(), rtol=tol, atol=tol)
                continue
            ref = ref[..., 0] * torch.from_numpy(o.scale).to(td) + torch.from_numpy(o.center).to(td)
            np.testing.assert_allclose(got[o.name], ref.numpy(), rtol=tol, atol=tol * float(ref.abs().max()))
            c = o.conditional
            t_in = torch.from_numpy(st[c.on].T.astype(np.float32))
            idx = torch.clamp(torch.searchsorted(torch.from_numpy(c.edges), t_in.contiguous(), right=True) - 1, min=0)
            ref = ref * torch.clamp(torch.from_numpy(c.scale)[idx], min=c.min_scale).to(td) + torch.from_numpy(c.center)[idx].to(td)
            np.testing.assert_allclose(got[c.name], ref.numpy(), rtol=tol, atol=tol * float(ref.abs().max()))
            ref = torch.from_numpy(st[o.before].T.astype(np.float32)).to(td) + ref
            np.testing.assert_allclose(got[o.after], ref.numpy(), rtol=tol, atol=tol * float(ref.abs().max()))


def test_limit_value_backward_known_answers():
    # external/fv3fit/fv3fit/emulation/transforms/transforms.py:131-158 (keras relu with a threshold, then the upper mask)
    x = np.array([-2.0, -1.0, 0.0, 1.0, 2.0, np.nan], np.float32)
    np.testing.assert_array_equal(mlp_np.limit_value_backward(x, 0.0, None), [0, 0, 0, 1, 2, np.nan])
    np.testing.assert_array_equal(mlp_np.limit_value_backward(x, None, 0.0), [-2, -1, 0, 0, 0, np.nan])
    np.testing.assert_array_equal(mlp_np.limit_value_backward(x, -1.0, 2.0), [0, -1, 0, 1, 0, np.nan])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-5)])
def test_forward_rnn_against_torch_cpu(dtype, tol):
    """RNNBlock against torch.nn.RNN(nonlinearity='relu') -- the same recurrence as Keras' SimpleRNN -- run from
    the last level to the first, two stacked layers, Conv1d(kernel_size=1) heads."""
    import local_cases

    rng = np.random.default_rng(9)
    nz, ncol, ch = 11, 30, 16
    st = local_cases.state(rng, nz, ncol)
    spec = local_cases.precpd_rnn(rng, st, nz, channels=ch)
    got = mlp_np.forward_rnn(spec, {k: v.T for k, v in st.items()}, dtype=dtype)
    td = torch.float64 if dtype == np.float64 else torch.float32
    cols = []
    for i in spec.inputs:
        x = torch.from_numpy(np.atleast_2d(st[i.source]).T.astype(np.float32)).to(td)
        if i.transform == "log":
            x = torch.log(torch.clamp(x, min=float(np.float32(i.eps))))
        x = (x - torch.from_numpy(np.atleast_1d(i.center)).to(td)) / torch.tensor(float(i.scale), dtype=td)
        cols.append(x.expand(ncol, nz).unsqueeze(-1))
    seq = torch.flip(torch.cat(cols, -1), dims=[1])
    for layer in spec.layers:
        rnn = torch.nn.RNN(layer.kernel.shape[0], ch, nonlinearity="relu", batch_first=True).to(td)
        with torch.no_grad():
            rnn.weight_ih_l0.copy_(torch.from_numpy(layer.kernel.T.copy()))
            rnn.weight_hh_l0.copy_(torch.from_numpy(layer.recurrent_kernel.T.copy()))
            rnn.bias_ih_l0.copy_(torch.from_numpy(layer.bias))
            rnn.bias_hh_l0.zero_()
            seq, _ = rnn(seq)
    h = torch.flip(seq, dims=[1])
    y = (h @ torch.from_numpy(spec.out_kernel).to(td) + torch.from_numpy(spec.out_bias).to(td)).detach()
    o = spec.outputs
    ref = y[:, 0:1, 0] * float(o[0].scale) + float(o[0].center)
    np.testing.assert_allclose(got["total_precipitation"], ref.numpy(), rtol=tol, atol=tol * float(ref.abs().max()))
    assert got["total_precipitation"].shape == (ncol, 1)
    d = y[..., 1] * torch.from_numpy(o[1].scale).to(td) + torch.from_numpy(o[1].center).to(td)
    d = torch.where(d < 0, d, torch.zeros_like(d))
    np.testing.assert_allclose(got["cloud_precpd_difference"], d.numpy(), rtol=tol, atol=tol * float(d.abs().max()))
    after = torch.relu(torch.from_numpy(st[local_cases.CLOUD_IN].T.astype(np.float32)).to(td) + d)
    np.testing.assert_allclose(got["cloud_water_mixing_ratio_after_precpd"], after.numpy(), rtol=tol, atol=tol * float(after.abs().max()))
    assert (got["cloud_water_mixing_ratio_after_precpd"] >= 0).all() and (got["cloud_precpd_difference"] <= 0).all()
    assert (got["humidity_precpd_difference"] >= 0).all()


@pytest.mark.parametrize("go_backwards", [True, False])
@pytest.mark.parametrize("dense_depth", [0, 2])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-5)])
def test_forward_hybrid_rnn_against_torch_cpu(dtype, tol, dense_depth, go_backwards):
    """HybridRNN (architecture.py:78-147) against torch.nn.RNN(nonlinearity='relu')'s FINAL state (the sequence flipped
    for go_backwards, as Keras feeds it), torch.nn.Linear hidden layers and heads.  TensorFlow itself is not
    installed: the recurrence is pinned to an independent implementation, not to the reference's own output."""
    import local_cases

    rng = np.random.default_rng(19)
    nz, ncol, ch = 9, 25, 16
    st = local_cases.state(rng, nz, ncol)
    spec = local_cases.hybrid_rnn(rng, st, nz, channels=ch, dense_width=24, dense_depth=dense_depth, go_backwards=go_backwards)
    got = mlp_np.forward_hybrid_rnn(spec, {k: v.T for k, v in st.items()}, dtype=dtype)
    td = torch.float64 if dtype == np.float64 else torch.float32
    cols = []
    for i in spec.inputs:
        x = torch.from_numpy(np.atleast_2d(st[i.source]).T.astype(np.float32)).to(td)
        if i.transform == "log":
            x = torch.log(torch.clamp(x, min=float(np.float32(i.eps))))
        x = (x - torch.from_numpy(np.atleast_1d(i.center)).to(td)) / torch.tensor(float(i.scale), dtype=td)
        cols.append(x.expand(ncol, nz).unsqueeze(-1))
    seq = torch.cat(cols, -1)
    if go_backwards:
        seq = torch.flip(seq, dims=[1])
    rnn = torch.nn.RNN(len(spec.inputs), ch, nonlinearity="relu", batch_first=True).to(td)
    with torch.no_grad():
        rnn.weight_ih_l0.copy_(torch.from_numpy(spec.rnn.kernel.T.copy()))
        rnn.weight_hh_l0.copy_(torch.from_numpy(spec.rnn.recurrent_kernel.T.copy()))
        rnn.bias_ih_l0.copy_(torch.from_numpy(spec.rnn.bias))
        rnn.bias_hh_l0.zero_()
        _, h = rnn(seq)
    h = h[0]
    for kern, b in zip(spec.head.hidden_kernels, spec.head.hidden_biases):
        h = torch.relu(h @ torch.from_numpy(kern).to(td) + torch.from_numpy(b).to(td))
    y = (h @ torch.from_numpy(spec.head.out_kernel).to(td) + torch.from_numpy(spec.head.out_bias).to(td)).detach()
    o = spec.head.outputs
    dq = y[:, :nz] * torch.from_numpy(o[0].scale).to(td) + torch.from_numpy(o[0].center).to(td)
    dt = y[:, nz:2 * nz] * torch.from_numpy(o[1].scale).to(td) + torch.from_numpy(o[1].center).to(td)
    pr = torch.clamp(y[:, 2 * nz:] * float(o[2].scale[0]) + float(o[2].center[0]), min=0.0)
    for name, ref in (("humidity_difference", dq), ("temperature_difference", dt), ("total_precipitation", pr),
                      ("specific_humidity_after", torch.from_numpy(st[local_cases.QV_IN].T.astype(np.float32)).to(td) + dq),
                      ("air_temperature_after", torch.from_numpy(st[local_cases.T_IN].T.astype(np.float32)).to(td) + dt)):
        np.testing.assert_allclose(got[name], ref.numpy(), rtol=tol, atol=tol * float(ref.abs().max()), err_msg=name)
    assert set(got) == set(spec.head.outputs[i].name for i in range(3)) | {"specific_humidity_after", "air_temperature_after"}
