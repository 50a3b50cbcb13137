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
