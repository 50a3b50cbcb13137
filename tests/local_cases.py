"""Shared by the oracle test and the GPU tests: a synthetic "dense-local" gscond regressor and
classifier shaped like the reference's production configs
(projects/microphysics/configs/models/gscond.yaml, classifier.yaml): 12-13 inputs at one point,
width x depth hidden layers, per-level normalisation, temperature-conditional un-scaling and
Difference outputs; a 4-class classifier."""
import types

import numpy as np

NS = types.SimpleNamespace
T_IN, QV_IN, CLOUD_IN = "air_temperature_input", "specific_humidity_input", "cloud_water_mixing_ratio_input"


def state(rng, nz, ncol, dtype=np.float64):
    """[nz, ncol] (call_py_fort layout) arrays with SURVEY 8d config 2 style distributions."""
    qv = 10.0 ** rng.uniform(-8, -2, (nz, ncol))
    cloud = np.where(rng.uniform(size=(nz, ncol)) < 0.7, 0.0, 10.0 ** rng.uniform(-10, -3, (nz, ncol)))
    t = rng.uniform(180, 310, (nz, ncol))
    s = {
        T_IN: t, QV_IN: qv, CLOUD_IN: cloud,
        "pressure_thickness_of_atmospheric_layer": rng.uniform(300, 1500, (nz, ncol)),
        "air_temperature_after_last_gscond": t + rng.normal(0, 1, (nz, ncol)),
        "specific_humidity_after_last_gscond": qv * rng.uniform(0.9, 1.1, (nz, ncol)),
        "air_pressure": np.sort(rng.uniform(300, 101000, (nz, ncol)), axis=0),
        "surface_air_pressure": rng.uniform(95000, 103000, (ncol,)),
        "surface_air_pressure_after_last_gscond": rng.uniform(95000, 103000, (1, ncol)),
    }
    return {k: np.ascontiguousarray(v.astype(dtype)) for k, v in s.items()}


def _network(rng, k, width, depth, channels):
    hk, hb, fan = [], [], k
    for _ in range(depth):
        hk.append((rng.normal(0, 1, (fan, width)) / np.sqrt(fan)).astype(np.float32))
        hb.append(rng.normal(0, 0.1, width).astype(np.float32))
        fan = width
    return hk, hb, (rng.normal(0, 1, (width, channels)) / np.sqrt(width)).astype(np.float32), rng.normal(0, 0.1, channels).astype(np.float32)


def _inputs(rng, st, nz, make_input):
    raw = [(n, n, "none", 0.0) for n in st] + [
        ("log_cloud_input", CLOUD_IN, "log", 1e-10), ("log_humidity_input", QV_IN, "log", 1e-8),
        ("log_humidity_after_last_gscond", "specific_humidity_after_last_gscond", "log", 1e-8)]
    inputs = []
    for name, source, transform, eps in sorted(raw):
        x = st[source].astype(np.float64)
        x = np.atleast_2d(x)
        if transform == "log":
            x = np.log(np.maximum(x, eps))
        # center per level, one scale for the variable (scale=all, center=per_feature)
        center = x.mean(axis=1).astype(np.float32)
        scale = np.float32(np.sqrt(np.mean((x - x.mean(axis=1, keepdims=True)) ** 2)) + 1e-7)
        inputs.append(make_input(name=name, source=source, transform=transform, eps=eps, center=center, scale=scale))
    return inputs


def regressor(rng, st, nz, width=64, depth=2, make=None):
    """Outputs: humidity / temperature differences, temperature-conditionally scaled, as Differences."""
    m = make or NS(input=NS, output=NS, cond=NS, spec=NS)
    inputs = _inputs(rng, st, nz, m.input)
    hk, hb, ok, ob = _network(rng, len(inputs), width, depth, 2)
    edges = np.linspace(180, 310, 51)[:-1].astype(np.float32)
    outs = []
    for name, unscaled, after, before, mag in (
            ("humidity_gscond_difference_tscaled", "humidity_gscond_difference", "specific_humidity_after_gscond", QV_IN, 1e-5),
            ("temperature_gscond_difference_tscaled", "temperature_gscond_difference", "air_temperature_after_gscond", T_IN, 0.5)):
        cond = m.cond(name=unscaled, on=T_IN, edges=edges, scale=(mag * rng.uniform(0.0, 2.0, 50)).astype(np.float32),
                      center=(mag * rng.normal(0, 0.1, 50)).astype(np.float32), min_scale=float(mag * 0.2))
        outs.append(m.output(name=name, channels=1, scale=rng.uniform(0.5, 2, nz).astype(np.float32),
                             center=rng.normal(0, 0.1, nz).astype(np.float32), conditional=cond, after=after, before=before))
    return m.spec(inputs=inputs, hidden_kernels=hk, hidden_biases=hb, outputs=outs, out_kernel=ok, out_bias=ob)


def classifier(rng, st, nz, width=64, depth=2, make=None):
    m = make or NS(input=NS, output=NS, cond=NS, spec=NS)
    inputs = _inputs(rng, st, nz, m.input)
    hk, hb, ok, ob = _network(rng, len(inputs), width, depth, 4)
    outs = [m.output(name="gscond_classes", channels=4, scale=None, center=None, conditional=None, after=None, before=None)]
    return m.spec(inputs=inputs, hidden_kernels=hk, hidden_biases=hb, outputs=outs, out_kernel=ok, out_bias=ob)


def precpd_rnn(rng, st, nz, channels=64, depth=2, make=None):
    """The production precpd model in miniature (projects/microphysics/configs/models/precpd.yaml): stacked
    SimpleRNNs over the levels, a single-level output (total_precipitation), limited differences added to the
    after-gscond state, the cloud limited to be non-negative."""
    m = make or NS(input=NS, output=NS, cond=NS, spec=NS, rnn_spec=NS, rnn_layer=NS)
    inputs = _inputs(rng, st, nz, m.input)
    layers, fan = [], len(inputs)
    for _ in range(depth):
        layers.append(m.rnn_layer(kernel=(rng.normal(0, 1, (fan, channels)) / np.sqrt(fan)).astype(np.float32),
                                  recurrent_kernel=(rng.normal(0, 0.6, (channels, channels)) / np.sqrt(channels)).astype(np.float32),
                                  bias=rng.normal(0, 0.1, channels).astype(np.float32)))
        fan = channels
    common = dict(channels=1, conditional=None)
    outs = [
        m.output(name="total_precipitation", scale=np.float32(1e-3), center=np.float32(2e-4), after=None, before=None,
                 value_limit=(None, None), after_limit=(None, None), single_level=True, **common),
        m.output(name="cloud_precpd_difference", scale=(1e-4 * rng.uniform(0.5, 2, nz)).astype(np.float32),
                 center=rng.normal(0, 1e-5, nz).astype(np.float32), after="cloud_water_mixing_ratio_after_precpd",
                 before=CLOUD_IN, value_limit=(None, 0.0), after_limit=(0.0, None), single_level=False, **common),
        m.output(name="humidity_precpd_difference", scale=(1e-4 * rng.uniform(0.5, 2, nz)).astype(np.float32),
                 center=rng.normal(0, 1e-5, nz).astype(np.float32), after="specific_humidity_after_precpd", before=QV_IN,
                 value_limit=(0.0, None), after_limit=(None, None), single_level=False, **common),
    ]
    return m.rnn_spec(inputs=inputs, layers=layers, outputs=outs,
                      out_kernel=(rng.normal(0, 1, (channels, 3)) / np.sqrt(channels)).astype(np.float32),
                      out_bias=rng.normal(0, 0.1, 3).astype(np.float32))


def hybrid_rnn(rng, st, nz, channels=64, dense_width=64, dense_depth=1, go_backwards=True, make=None):
    """A "rnn" (HybridRNN, architecture.py:78-147) emulator: the final state of one SimpleRNN over the levels feeds
    ``dense_depth`` hidden layers and whole-column dense heads -- per-level de-normalised differences added to the
    input state, one limited single-value output."""
    m = make or NS(input=NS, rnn_layer=NS, hybrid_spec=NS, head_spec=NS, head_input=NS, head_output=NS, head_residual=NS)
    inputs = _inputs(rng, st, nz, m.input)
    fan = len(inputs)
    layer = m.rnn_layer(kernel=(rng.normal(0, 1, (fan, channels)) / np.sqrt(fan)).astype(np.float32),
                        recurrent_kernel=(rng.normal(0, 0.6, (channels, channels)) / np.sqrt(channels)).astype(np.float32),
                        bias=rng.normal(0, 0.1, channels).astype(np.float32))
    hk, hb, k = [], [], channels
    for _ in range(dense_depth):
        hk.append((rng.normal(0, 1, (k, dense_width)) / np.sqrt(k)).astype(np.float32))
        hb.append(rng.normal(0, 0.1, dense_width).astype(np.float32))
        k = dense_width
    outs = [m.head_output(name="humidity_difference", nfeat=nz, scale=(1e-4 * rng.uniform(0.5, 2, nz)).astype(np.float32),
                          center=rng.normal(0, 1e-5, nz).astype(np.float32), min=None, max=None, mask=None),
            m.head_output(name="temperature_difference", nfeat=nz, scale=rng.uniform(0.5, 2, nz).astype(np.float32),
                          center=rng.normal(0, 0.1, nz).astype(np.float32), min=None, max=None, mask=None),
            m.head_output(name="total_precipitation", nfeat=1, scale=np.array([1e-3], np.float32), center=np.array([2e-4], np.float32),
                          min=0.0, max=None, mask=None)]
    f = 2 * nz + 1
    head = m.head_spec(inputs=[m.head_input(source="rnn_state", nfeat=channels, start=0, transform="none", eps=0.0, center=None, scale=None)],
                       hidden_kernels=hk, hidden_biases=hb, outputs=outs,
                       out_kernel=(rng.normal(0, 1, (k, f)) / np.sqrt(k)).astype(np.float32), out_bias=rng.normal(0, 0.1, f).astype(np.float32),
                       residuals=[m.head_residual(name="specific_humidity_after", source=QV_IN, output="humidity_difference"),
                                  m.head_residual(name="air_temperature_after", source=T_IN, output="temperature_difference")],
                       activation="relu", hidden_output=None)
    return m.hybrid_spec(inputs=inputs, rnn=layer, head=head, go_backwards=go_backwards)


def product_makers():
    from fv3net_amd.local_mlp import ConditionalScale, HybridRnnSpec, LocalInput, LocalMlpSpec, LocalOutput, RnnLayer, RnnSpec
    from fv3net_amd.mlp import InputSpec, MlpSpec, OutputSpec, ResidualSpec

    return NS(input=LocalInput, output=LocalOutput, cond=ConditionalScale, spec=LocalMlpSpec, rnn_spec=RnnSpec, rnn_layer=RnnLayer,
              hybrid_spec=HybridRnnSpec, head_spec=MlpSpec, head_input=InputSpec, head_output=OutputSpec, head_residual=ResidualSpec)
