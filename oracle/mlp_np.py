"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the column-MLP predict graph.  Arrays are [sample, feature] as in Keras.

Follows (paths relative to the reference checkout):
  external/fv3fit/fv3fit/keras/_models/dense.py:239-310            build_model (predict_model)
  external/fv3fit/fv3fit/keras/_models/shared/utils.py:34-105      standard_normalize / denormalize
  external/fv3fit/fv3fit/emulation/layers/normalization.py:21-49,117-164  NormLayer, fit of mean/std
  external/fv3fit/fv3fit/keras/_models/shared/dense_network.py:59-81  Dense(relu) * (depth-1)
  external/fv3fit/fv3fit/keras/_models/shared/clip.py:21-62, output_limit.py:29-48
  external/fv3fit/fv3fit/emulation/layers/fields.py:33-66, architecture.py:27-50,262-269,332-343
  external/fv3fit/fv3fit/emulation/transforms/transforms.py:17-58 (Difference), 111-129 (LogTransform)

The dense contraction itself is TensorFlow 2.8 / Keras 2.8 (constraints.txt:143,282), which is
not in the reference tree and not installed: its published semantics are restated
(``Dense``: ``activation(x @ kernel + bias)``, kernel [in, out]).  PARITY UNPINNED for the
contraction: no reference test holds a TensorFlow-free golden output on fixed weights.  The
normalisation pieces ARE pinned by the reference's regtest outputs (tests/golden/).
``dtype=np.float64`` gives the high-precision truth the fp32 paths are measured against.
"""
import numpy as np


def fit_mean_per_feature(sample):
    return np.asarray(sample).mean(axis=tuple(range(np.ndim(sample) - 1))).astype(np.float32)


def fit_std_per_feature(sample):
    return np.asarray(sample).std(axis=tuple(range(np.ndim(sample) - 1))).astype(np.float32)


def fit_std_all(sample):
    """StdDevMethod.all: one std over all features, centred per feature."""
    s = np.asarray(sample)
    mean = s.mean(axis=tuple(range(s.ndim - 1))).astype(np.float32)
    return np.sqrt(np.mean((s - mean) ** 2)).astype(np.float32)


def norm_forward(x, center, scale, epsilon=None):
    x = np.asarray(x)
    fs = np.float32(scale) if epsilon is None else (np.asarray(scale, np.float32) + np.float32(epsilon))
    return (x - np.asarray(center, x.dtype)) / np.asarray(fs, x.dtype)


def norm_backward(y, center, scale):
    y = np.asarray(y)
    return y * np.asarray(scale, y.dtype) + np.asarray(center, y.dtype)


def limit_output(y, vmin=None, vmax=None):
    """OutputLimit._limit_activation."""
    x = y
    if vmin is not None:
        x = np.where(y < vmin, np.asarray(vmin, y.dtype), x)
        if vmax is not None:
            x = np.where((y >= vmin) & (y < vmax), y, x)
    if vmax is not None:
        x = np.where(y >= vmax, np.asarray(vmax, y.dtype), x)
    return x


def forward(spec, sources, dtype=np.float32, internal=()):
    """Evaluate an ``fv3net_amd.mlp.MlpSpec``-shaped description.

    ``spec`` is duck-typed (attributes inputs / hidden_kernels / hidden_biases / outputs /
    out_kernel / out_bias / residuals), so that the oracle does not import the product.
    ``sources``: name -> [sample, feature] (or [sample]) arrays.  Returns name -> [sample, feature].
    """
    cols = []
    src = {}
    for name, arr in sources.items():
        a = np.asarray(arr)
        if a.ndim == 1:
            a = a[:, None]
        # Keras casts float64 inputs to the layer dtype (float32) first (``internal``: names that are no model inputs but
        # values handed on inside a graph -- a recurrent state -- and stay as they are)
        src[name] = a.astype(dtype) if name in internal else a.astype(np.float32).astype(dtype)
    for i in spec.inputs:
        x = src[i.source][:, i.start:i.start + i.nfeat]
        if i.transform == "log":
            x = np.log(np.maximum(x, np.asarray(i.eps, np.float32).astype(dtype)))
        center = np.zeros(i.nfeat, np.float32) if i.center is None else np.broadcast_to(np.asarray(i.center, np.float32), (i.nfeat,))
        scale = np.ones(i.nfeat, np.float32) if i.scale is None else np.broadcast_to(np.asarray(i.scale, np.float32), (i.nfeat,))
        cols.append((x - center.astype(dtype)) / scale.astype(dtype))
    h = np.concatenate(cols, axis=1)
    for kern, b in zip(spec.hidden_kernels, spec.hidden_biases):
        h = h @ np.asarray(kern, np.float32).astype(dtype) + np.asarray(b, np.float32).astype(dtype)
        if getattr(spec, "activation", "relu") == "relu":  # ("linear": no activation)
            h = np.maximum(h, 0)
    yhat = h @ np.asarray(spec.out_kernel, np.float32).astype(dtype) + np.asarray(spec.out_bias, np.float32).astype(dtype)
    out = {}
    if getattr(spec, "hidden_output", None):  # the last hidden layer's activations as an output of their own
        out[spec.hidden_output] = h
    f0 = 0
    for o in spec.outputs:
        y = yhat[:, f0:f0 + o.nfeat]
        f0 += o.nfeat
        scale = np.ones(o.nfeat, np.float32) if o.scale is None else np.broadcast_to(np.asarray(o.scale, np.float32), (o.nfeat,))
        center = np.zeros(o.nfeat, np.float32) if o.center is None else np.broadcast_to(np.asarray(o.center, np.float32), (o.nfeat,))
        y = y * scale.astype(dtype) + center.astype(dtype)
        y = limit_output(y, o.min, o.max)
        if o.mask is not None:
            y = y * np.asarray(o.mask, np.float32).astype(dtype)
        out[o.name] = y
    for r in getattr(spec, "residuals", []):
        out[r.name] = src[r.source][:, : out[r.output].shape[1]] + out[r.output]
    return out


# ------------------------------------------------------------------------------------------------
# "dense-local" architecture and the transforms the reference bakes into its saved models
# ------------------------------------------------------------------------------------------------
def piecewise(x, y, xg):
    """0th-order interpolation with constant extrapolation (external/fv3fit/fv3fit/keras/math.py:5-23):
    ``y[max(searchsorted(x, xg, side='right') - 1, 0)]``."""
    xg = np.asarray(xg)
    index = np.maximum(np.searchsorted(np.asarray(x), xg.reshape(-1), side="right") - 1, 0)
    return np.asarray(y)[index].reshape(xg.shape)


def conditionally_scaled_backward(y, on, edges, scale, center, min_scale=0.0):
    """ConditionallyScaledTransform.backward (transforms/transforms.py:209-224) with ``scale``/``center``
    = ``piecewise(edges[:-1], values, .)`` as ``fit_conditional`` builds them (factories.py:81-93);
    ``edges`` here are those left bin edges."""
    limited = np.maximum(piecewise(edges, scale, on), np.float32(min_scale))  # tf.maximum(float32 tensor, python float)
    return y * limited.astype(y.dtype) + piecewise(edges, center, on).astype(y.dtype)


def limit_value_backward(x, lower=None, upper=None):
    """LimitValueTransform.backward (transforms/transforms.py:148-158): ``relu(x, threshold=lower)`` (x where
    x >= lower, else 0), then ``cast(x < upper) * x``."""
    if lower is not None:
        x = np.where(x < np.asarray(lower, x.dtype), np.zeros((), x.dtype), x)
    if upper is not None:
        x = np.where(~(x < np.asarray(upper, x.dtype)) & ~np.isnan(x), np.zeros((), x.dtype), x)
    return x


def _local_inputs(spec, sources, dtype):
    """layers/architecture.py:53-75 + layers/fields.py:6-41 + transforms.py:111-129: [sample, nz, n_inputs]."""
    src = {}
    nz = 1
    for name, arr in sources.items():
        a = np.asarray(arr)
        if a.ndim == 1:
            a = a[:, None]
        src[name] = a.astype(np.float32).astype(dtype)  # Keras casts inputs to float32
        nz = max(nz, a.shape[1])
    cols = []
    for i in spec.inputs:
        x = src[i.source]
        if i.transform == "log":
            x = np.log(np.maximum(x, np.asarray(i.eps, np.float32).astype(dtype)))
        center = np.zeros(1, np.float32) if i.center is None else np.atleast_1d(np.asarray(i.center, np.float32))
        scale = np.ones(1, np.float32) if i.scale is None else np.atleast_1d(np.asarray(i.scale, np.float32))
        x = (x - center.astype(dtype)) / scale.astype(dtype)
        cols.append(np.broadcast_to(x, (x.shape[0], nz))[..., None])
    return np.concatenate(cols, axis=-1), src


def _local_outputs(spec, yhat, src, dtype):
    """layers/fields.py:44-66, then the backward transforms in reverse order of the training config:
    conditional un-scaling (transforms.py:219-224), value limits (:148-158), differences (:55-58), limits on
    the ``after`` values."""
    out = {}
    c0 = 0
    for o in spec.outputs:
        y = yhat[..., c0:c0 + o.channels]
        c0 += o.channels
        if o.channels != 1:
            out[o.name] = y
            continue
        y = y[..., 0]
        if getattr(o, "single_level", False):
            y = y[:, 0:1]  # rnn_outputs[..., 0:1, :] (architecture.py:403-407)
        scale = np.ones(1, np.float32) if o.scale is None else np.atleast_1d(np.asarray(o.scale, np.float32))
        center = np.zeros(1, np.float32) if o.center is None else np.atleast_1d(np.asarray(o.center, np.float32))
        y = y * scale.astype(dtype) + center.astype(dtype)
        value_limit = getattr(o, "value_limit", (None, None))
        after_limit = getattr(o, "after_limit", (None, None))
        if o.conditional is not None:
            out[o.name] = y
            c = o.conditional
            y = conditionally_scaled_backward(y, src[c.on], np.asarray(c.edges, np.float32), np.asarray(c.scale, np.float32),
                                              np.asarray(c.center, np.float32), c.min_scale)
            y = limit_value_backward(y, *value_limit)
            out[c.name] = y
        else:
            y = limit_value_backward(y, *value_limit)
            out[o.name] = y
        if o.after is not None:
            out[o.after] = limit_value_backward(src[o.before] + y, *after_limit)
    return out


def forward_local(spec, sources, dtype=np.float32):
    """Evaluate an ``fv3net_amd.local_mlp.LocalMlpSpec``-shaped description (duck-typed).

    Follows layers/architecture.py:53-75 (combine_sequence_inputs: inputs sorted by name, [sample, 1]
    inputs repeated over the levels, stacked on a last axis), :228-282 (MLPBlock on the last axis),
    :346-417 (RNNOutput with shared kernel-size-1 convolutions: a Dense on the last axis, squeezed when
    it has one channel), layers/fields.py:6-66 (per-level normalisation), transforms.py:111-129
    (log(max(x, eps))), :219-224, :55-58 (backward transforms, reverse order).
    ``sources``: name -> [sample, nz] (or [sample] / [sample, 1]).  Returns name -> [sample, nz]
    (``[sample, nz, channels]`` for multi-channel outputs)."""
    h, src = _local_inputs(spec, sources, dtype)  # [sample, nz, n_inputs]
    for kern, b in zip(spec.hidden_kernels, spec.hidden_biases):
        h = np.maximum(h @ np.asarray(kern, np.float32).astype(dtype) + np.asarray(b, np.float32).astype(dtype), 0)
    yhat = h @ np.asarray(spec.out_kernel, np.float32).astype(dtype) + np.asarray(spec.out_bias, np.float32).astype(dtype)
    return _local_outputs(spec, yhat, src, dtype)


def forward_rnn(spec, sources, dtype=np.float32):
    """Evaluate an ``fv3net_amd.local_mlp.RnnSpec``-shaped description: layers/architecture.py:149-226 (RNNBlock:
    the level axis reversed, stacked ``SimpleRNN(channels, activation='relu', return_sequences=True)`` --
    ``h_t = relu(x_t W + h_{t-1} U + b)``, zero initial state, Keras 2.8 -- and reversed back) and :346-417
    (RNNOutput; outputs of feature length 1 read level 0)."""
    seq, src = _local_inputs(spec, sources, dtype)
    seq = seq[:, ::-1, :]
    for layer in spec.layers:
        w = np.asarray(layer.kernel, np.float32).astype(dtype)
        u = np.asarray(layer.recurrent_kernel, np.float32).astype(dtype)
        b = np.asarray(layer.bias, np.float32).astype(dtype)
        state = np.zeros((seq.shape[0], w.shape[1]), dtype)
        steps = []
        for t in range(seq.shape[1]):
            state = np.maximum(seq[:, t] @ w + state @ u + b, 0)
            steps.append(state)
        seq = np.stack(steps, axis=1)
    rnn_out = seq[:, ::-1, :]
    yhat = rnn_out @ np.asarray(spec.out_kernel, np.float32).astype(dtype) + np.asarray(spec.out_bias, np.float32).astype(dtype)
    return _local_outputs(spec, yhat, src, dtype)


def forward_hybrid_rnn(spec, sources, dtype=np.float32):
    """Evaluate an ``fv3net_amd.local_mlp.HybridRnnSpec``-shaped description: layers/architecture.py:78-147 (HybridRNN:
    ``SimpleRNN(channels, activation='relu', go_backwards)`` returning its last state -- Keras 2.8: with
    ``go_backwards`` the sequence is fed reversed, ``h_t = relu(x_t W + h_{t-1} U + b)`` from a zero state -- then
    ``MLPBlock``) and :285-343 (StandardOutput dense heads), followed by the head's de-normalisation / limits /
    residuals as in ``forward``.  ``sources``: name -> [sample, nz] (or [sample] / [sample, 1])."""
    seq, src = _local_inputs(spec, sources, dtype)  # [sample, nz, n_inputs]
    if spec.go_backwards:
        seq = seq[:, ::-1, :]
    w = np.asarray(spec.rnn.kernel, np.float32).astype(dtype)
    u = np.asarray(spec.rnn.recurrent_kernel, np.float32).astype(dtype)
    b = np.asarray(spec.rnn.bias, np.float32).astype(dtype)
    state = np.zeros((seq.shape[0], w.shape[1]), dtype)
    for t in range(seq.shape[1]):
        state = np.maximum(seq[:, t] @ w + state @ u + b, 0)
    head_sources = {"rnn_state": state}
    for r in spec.head.residuals:
        head_sources[r.source] = sources[r.source]
    return forward(spec.head, head_sources, dtype=dtype, internal=("rnn_state",))
