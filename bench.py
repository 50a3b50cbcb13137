#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json configs[1]): columns/s of the Zhao-Carr microphysics MLP emulator
(K = 711 inputs, 2 x 256 hidden, 396 outputs; projects/microphysics/train/dense.yaml:62-89 of
the reference) over one C384 snapshot (6 x 384 x 384 = 884 736 columns, 79 levels) per GPU.
A "step" is one pass of the fused kernel over the rank's snapshot, inputs resident in HBM.
With N > 1 every rank owns its own snapshot (tile/snapshot sharding, no data-path collective):
weak scaling; value = columns of all ranks / max-over-ranks time.

The JSON line also carries
  roofline     -- the MLP kernel against the fp32 MFMA peak, timed with HIP events on the launch stream
  cpu_baseline -- the numpy oracle of the same network on a bounded sample, on this box's host cores
  secondary    -- the HBM-bound coarsening kernels (C3072 -> C384 weighted_block_average, C384 mappm)
                  against the HBM roofline
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E peak

RAW_FIELDS = [
    "air_temperature_input",
    "specific_humidity_input",
    "cloud_water_mixing_ratio_input",
    "pressure_thickness_of_atmospheric_layer",
    "air_temperature_after_last_gscond",
    "specific_humidity_after_last_gscond",
]
LOG_FIELDS = {
    "log_cloud_input": ("cloud_water_mixing_ratio_input", 1e-10),
    "log_humidity_input": ("specific_humidity_input", 1e-8),
    "log_humidity_after_last_gscond": ("specific_humidity_after_last_gscond", 1e-8),
}
OUTPUTS = {
    "total_precipitation": 1,
    "cloud_precpd_difference": 79,
    "temperature_precpd_difference": 79,
    "humidity_precpd_difference": 79,
    "temperature_gscond_difference": 79,
    "humidity_gscond_difference": 79,
}
# Difference.backward (fv3fit/emulation/transforms/transforms.py:54-58; projects/microphysics/train/dense.yaml:22-36):
# after = before + difference -- the five fields the Fortran model reads back from the hook
RESIDUALS = {
    "air_temperature_after_gscond": ("air_temperature_input", "temperature_gscond_difference"),
    "specific_humidity_after_gscond": ("specific_humidity_input", "humidity_gscond_difference"),
    "cloud_water_mixing_ratio_after_precpd": ("cloud_water_mixing_ratio_input", "cloud_precpd_difference"),
    "air_temperature_after_precpd": ("air_temperature_input", "temperature_precpd_difference"),
    "specific_humidity_after_precpd": ("specific_humidity_input", "humidity_precpd_difference"),
}
NZ = 79


def zc_inputs_numpy(rng, n):
    """SURVEY.md 8(d) config 2 distributions, [sample, feature] float32."""
    t = rng.uniform(180, 310, (n, NZ))
    q = 10 ** rng.uniform(-8, -2, (n, NZ))
    c = np.where(rng.random((n, NZ)) < 0.7, 0.0, 10 ** rng.uniform(-10, -3, (n, NZ)))
    dp = rng.uniform(300, 1500, (n, NZ))
    src = {
        "air_temperature_input": t,
        "specific_humidity_input": q,
        "cloud_water_mixing_ratio_input": c,
        "pressure_thickness_of_atmospheric_layer": dp,
        "air_temperature_after_last_gscond": t + rng.normal(0, 0.5, (n, NZ)),
        "specific_humidity_after_last_gscond": q * rng.uniform(0.9, 1.1, (n, NZ)),
    }
    return {k: v.astype(np.float32) for k, v in src.items()}


def _per_level_output(OutputSpec, rng, name, nfeat, scale_fn):
    scale = np.asarray(scale_fn(rng, nfeat), dtype=np.float32)
    return OutputSpec(name, nfeat, scale=scale, center=(rng.normal(0, 1, nfeat) * scale).astype(np.float32))


def zc_spec(seed=0, residuals=True, per_level_output_scale=None):
    """Random-init weights of the reference architecture; normalisation fitted on a sample the
    way MicrophysicsConfig does (center per feature, one std over all features).  ``residuals``: the
    production graph of dense.yaml, whose five difference outputs also leave as ``after = before +
    difference`` (396 + 395 output rows); False: the 396 direct outputs only.
    ``per_level_output_scale(rng, nfeat) -> [nfeat]``: per-feature output standard deviations (StdDevMethod.per_feature,
    normalization.py:105-114) instead of one per variable -- the parity tests let them span > 4 decades over the levels;
    the arithmetic and the kernel instantiation are the same (scale and centre are folded into the output weights)."""
    from fv3net_amd.mlp import InputSpec, MlpSpec, OutputSpec, ResidualSpec

    def fit_mean_per_feature(data):  # MeanMethod.per_feature (emulation/layers/normalization.py:117-128)
        return data.mean(axis=0).astype(np.float32)

    def fit_std_all(data):  # StdDevMethod.all: one standard deviation over all features, centred per feature
        return np.sqrt(np.mean((data - data.mean(axis=0).astype(np.float32)) ** 2)).astype(np.float32)

    rng = np.random.default_rng(seed)
    sample = zc_inputs_numpy(rng, 4096)
    names = sorted(RAW_FIELDS + list(LOG_FIELDS))  # combine_inputs sorts by key
    inputs = []
    for name in names:
        if name in LOG_FIELDS:
            source, eps = LOG_FIELDS[name]
            data = np.log(np.maximum(sample[source], np.float32(eps)))
            inputs.append(InputSpec(source, NZ, transform="log", eps=eps,
                                    center=fit_mean_per_feature(data), scale=fit_std_all(data)))
        else:
            data = sample[name]
            inputs.append(InputSpec(name, NZ, center=fit_mean_per_feature(data), scale=fit_std_all(data)))
    k, w = NZ * len(inputs), 256
    f = sum(OUTPUTS.values())
    glorot = lambda a, b: rng.uniform(-1, 1, (a, b)).astype(np.float32) * np.float32(np.sqrt(6.0 / (a + b)))
    return MlpSpec(
        inputs=inputs,
        hidden_kernels=[glorot(k, w), glorot(w, w)],
        hidden_biases=[rng.normal(0, 0.01, w).astype(np.float32) for _ in range(2)],
        outputs=[OutputSpec(n_, nf, scale=np.float32(rng.uniform(0.5, 2)), center=rng.normal(0, 1, nf).astype(np.float32))
                 if per_level_output_scale is None else _per_level_output(OutputSpec, rng, n_, nf, per_level_output_scale)
                 for n_, nf in OUTPUTS.items()],
        out_kernel=glorot(w, f),
        out_bias=rng.normal(0, 0.01, f).astype(np.float32),
        residuals=[ResidualSpec(name, before, diff) for name, (before, diff) in RESIDUALS.items()] if residuals else [],
    )


def parity_slice_predict(model, src, n=4096):
    """One FULL-SIZE call of the timed model object on the timed inputs -- the very launch the loop repeats, not a smaller
    one (which the library would hand to another kernel) -- of which the first ``n`` columns are kept for the oracle.
    Returns what ``parity_max_rel`` needs; the device work happens here, before the warm-up, the CPU oracle after the
    timed loop -- a second of host work between warm-up and timing would let the GPU clocks drop."""
    got = model.predict(src)
    return {"variant": model.last_variant, "got": {k: v[:, :n].cpu().numpy().T for k, v in got.items()},
            "host": {k: v[:, :n].T.contiguous().cpu().numpy() for k, v in src.items()}}


def parity_max_rel(spec, sliced):
    """max over the output variables of max|gpu - truth| / max|truth| against the float64 oracle."""
    from oracle import mlp_np

    truth = mlp_np.forward(spec, sliced["host"], dtype=np.float64)
    return max(float(np.max(np.abs(sliced["got"][name] - t)) / np.max(np.abs(t))) for name, t in truth.items())


def zc_inputs_device(dev, n, seed):
    """The same distributions generated on the device, [feature, sample] float32."""
    g = torch.Generator(device=dev).manual_seed(seed)
    u = lambda lo, hi: torch.rand((NZ, n), device=dev, generator=g) * (hi - lo) + lo
    t = u(180, 310)
    q = 10 ** u(-8, -2)
    c = torch.where(torch.rand((NZ, n), device=dev, generator=g) < 0.7, torch.zeros((), device=dev), 10 ** u(-10, -3))
    return {
        "air_temperature_input": t,
        "specific_humidity_input": q,
        "cloud_water_mixing_ratio_input": c,
        "pressure_thickness_of_atmospheric_layer": u(300, 1500),
        "air_temperature_after_last_gscond": t + torch.randn((NZ, n), device=dev, generator=g) * 0.5,
        "specific_humidity_after_last_gscond": q * u(0.9, 1.1),
    }


PMC_TRAFFIC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")


def pmc_traffic(kernel_key):
    """(HBM bytes per launch, where the figure comes from) for the kernel named ``kernel_key`` -- the name the
    library reports for the launch it made (``fv3hip_mlp_last_variant``), so a changed dispatch finds no entry.  The
    figure is a RECORDED measurement of the same launch (profiles/r0N_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs of this script, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes for gfx950),
    not collected in this run; (None, reason) where absent or uncalibrated."""
    for name in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                value = json.load(f)["kernels"][kernel_key]["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
        return value, f"recorded: profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script), not measured in this run"
    return None, f"no recorded PMC pass for {kernel_key!r}"


def pmc_valu(kernel_key):
    """VALU issue-bound fraction of a remap kernel from the committed PMC passes (profiles/r01_pmc_mappm_valu.json), a
    recorded measurement (rocprofv3 --pmc passes of benchmarks/mappm_one.py) like ``pmc_traffic``; None if absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_pmc_mappm_valu.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel_key]["valu_issue_bound_fraction"]
    except (OSError, KeyError, ValueError):
        return None


def _timed_reps(fn, budget_s, max_reps=100):
    fn()  # warm
    t0, reps = time.perf_counter(), 0
    while True:
        fn()
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= max_reps:
            break
    return reps, time.perf_counter() - t0


def cpu_baseline(spec, budget_s=6.0):
    """The oracle on bounded samples of the same workloads, on this box's host cores (rank 0, N = 1 only; a reported
    baseline, not a target).  Headline entry: the numpy float32 restatement of the network with all BLAS threads;
    ``others``: the same single-threaded, and the two coarse-graining kernels: numpy
    ``weighted_block_average`` on one C384 field and the C port of ``mappm`` (and, when the compiled reference Fortran
    travelled with the repo, that too) on a column sample, single-threaded and over a thread pool."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import coarsen_np, mappm_c, mlp_np

    cores = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threadpool_limits, blas_threads = None, cores
    n = 16384
    src = zc_inputs_numpy(np.random.default_rng(1), n)
    pool_threads = min(cores, 32)
    # three CPU configurations of the same float32 oracle; `value` is the BEST of them (an oversubscribed
    # BLAS is not "all cores"), the others are listed
    trials = []
    reps, dt = _timed_reps(lambda: mlp_np.forward(spec, src), budget_s / 3)
    trials.append((n * reps / dt, int(blas_threads), f"{reps} passes over {n} columns, one call, BLAS threads = {blas_threads}"))
    if threadpool_limits is not None:
        try:
            n1 = 4096
            src1 = {k: v[:n1] for k, v in src.items()}
            with threadpool_limits(limits=1):
                reps, dt = _timed_reps(lambda: mlp_np.forward(spec, src1), budget_s / 3)
                trials.append((n1 * reps / dt, 1, f"{reps} passes over {n1} columns, BLAS limited to 1 thread"))
                pieces = [{k: v[i:i + 512] for k, v in src.items()} for i in range(0, n, 512)]
                with ThreadPoolExecutor(pool_threads) as ex:
                    reps, dt = _timed_reps(lambda: list(ex.map(lambda part: mlp_np.forward(spec, part), pieces)), budget_s / 3)
                trials.append((n * reps / dt, pool_threads, f"{reps} passes over {n} columns in pieces of 512 over a pool of "
                                                           f"{pool_threads} workers, 1 BLAS thread each"))
        except Exception as err:  # noqa: BLE001
            trials.append((0.0, 0, f"failed: {type(err).__name__}: {err}"))
    best = max(trials, key=lambda t: t[0])
    out = {
        "value": best[0],
        "unit": "columns/s",
        "cores": int(cores),
        "threads": best[1],
        "kind": "port",
        "sample": f"numpy float32 oracle of the same network on {cores} host cores, best of {len(trials)} configurations: {best[2]}",
        "others": [{"workload": "MLP, another CPU configuration", "value": v, "unit": "columns/s", "threads": t, "kind": "port", "sample": what}
                   for v, t, what in trials if (v, t, what) != best],
    }
    others = out["others"]
    pool_threads = min(cores, 32)
    try:  # weighted_block_average, one C384 -> C48 field (config 3), float32, 2-D area weights
        rng = np.random.default_rng(2)
        obj = rng.uniform(-1000, 1000, (6, NZ, 384, 384)).astype(np.float32)
        area = rng.uniform(0.5, 1, (6, 1, 384, 384)).astype(np.float32)
        alg = 4 * obj.size * (1 + 1 / 64) + 4 * obj.size / NZ
        reps, dt = _timed_reps(lambda: coarsen_np.weighted_block_average(obj, area, 8), budget_s / 2, 20)
        others.append({"workload": "weighted_block_average C384->C48, one [6,79,384,384] f32 field, single thread",
                       "value": alg * reps / dt / 1e9, "unit": "GB/s (algorithmic)", "threads": 1, "kind": "port",
                       "ms": dt / reps * 1e3, "sample": f"{reps} passes, numpy reshape-sum oracle"})
        parts = [(t, z0) for t in range(6) for z0 in range(0, NZ, 10)]
        with ThreadPoolExecutor(pool_threads) as ex:
            run = lambda: list(ex.map(lambda tz: coarsen_np.weighted_block_average(
                obj[tz[0], tz[1]:tz[1] + 10], area[tz[0]], 8), parts))
            reps, dt = _timed_reps(run, budget_s / 2, 50)
        others.append({"workload": f"weighted_block_average C384->C48, the same field over a pool of {pool_threads} threads",
                       "value": alg * reps / dt / 1e9, "unit": "GB/s (algorithmic)", "threads": pool_threads, "kind": "port",
                       "ms": dt / reps * 1e3, "sample": f"{reps} passes, {len(parts)} (tile, 10-level) pieces"})
        del obj, area
    except Exception as err:  # noqa: BLE001
        others.append({"workload": "weighted_block_average", "error": f"{type(err).__name__}: {err}"})
    try:  # mappm, km = kn = 79, [column, level] as the f2py module takes them
        rng = np.random.default_rng(3)
        ncol = 65536
        pe1 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, NZ)), 1)], 1).astype(np.float32)
        pe2 = np.concatenate([np.full((ncol, 1), 300.0), 300 + np.cumsum(rng.uniform(300, 1500, (ncol, NZ)), 1)], 1).astype(np.float32)
        q = rng.uniform(-1000, 1000, (ncol, NZ)).astype(np.float32)
        impls = [("port", "C restatement of mappm.f90 (-O2)", mappm_c.mappm)]
        if mappm_c.have_reference():
            impls.append(("reference", "the reference's own mappm.f90 compiled with amdflang -O2 (oracle/_ref)", mappm_c.reference_mappm))
        for kind, what, fn in impls:
            reps, dt = _timed_reps(lambda: fn(pe1, q, pe2), budget_s / 4, 20)
            others.append({"workload": f"mappm iv=1 kord=1 km=kn=79, single thread: {what}", "value": ncol * reps / dt,
                           "unit": "columns/s", "threads": 1, "kind": kind, "sample": f"{reps} passes over {ncol} of the 884736 C384 columns"})
            pieces = [slice(i, i + 2048) for i in range(0, ncol, 2048)]
            with ThreadPoolExecutor(pool_threads) as ex:
                run = lambda: list(ex.map(lambda sl: fn(pe1[sl], q[sl], pe2[sl]), pieces))
                reps, dt = _timed_reps(run, budget_s / 4, 50)
            others.append({"workload": f"mappm, pool of {pool_threads} threads: {what}", "value": ncol * reps / dt,
                           "unit": "columns/s", "threads": pool_threads, "kind": kind,
                           "sample": f"{reps} passes over {ncol} columns in pieces of 2048"})
    except Exception as err:  # noqa: BLE001
        others.append({"workload": "mappm", "error": f"{type(err).__name__}: {err}"})
    return out


def time_kernel(fn, steps, dev, warm=3):
    """Mean HIP-event time of ``steps`` calls after ``warm`` untimed ones (clocks ramp over the first launches after a pause)."""
    from fv3net_amd.ops import HipTimer

    for _ in range(warm):
        fn()
    timer = HipTimer()
    timer.start(dev)
    for _ in range(steps):
        fn()
    timer.stop(dev)
    return timer.elapsed_ms() / steps


def _guarded(out, fn, *fn_args):
    """A secondary workload never costs the line: a failure is recorded in place of its numbers."""
    try:
        out.extend(fn(*fn_args))
    except Exception as err:  # noqa: BLE001
        out.append({"kernel": fn.__name__, "error": f"{type(err).__name__}: {err}"})
    torch.cuda.empty_cache()


def plain_network_benchmark(dev, steps):
    """The headline network without its residual outputs (the round-1 headline): 396 output rows, "plain" epilogue."""
    from fv3net_amd.mlp import MlpModel

    ncol = 6 * 384 * 384
    model = MlpModel(zc_spec(0, residuals=False), device=dev)
    src = zc_inputs_device(dev, ncol, seed=1000)
    fn = lambda: model.predict(src)
    fn()
    torch.cuda.synchronize(dev)
    ms = time_kernel(fn, max(3, min(steps, 20)), dev)
    achieved = model.flops_per_sample * ncol / (ms * 1e-3) / 1e12
    traffic, source = pmc_traffic("mlp_fused_kernel<8,false,true,false,false,false>")
    return [{"kernel": model.last_variant, "workload": "the headline network with its 396 direct outputs only (no Difference residuals)",
             "ms": ms, "columns_per_s": ncol / ms * 1e3,
             "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                          "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": source}}]


def wavg_benchmark(dev, steps, label, n, tiles=6):
    """weighted_block_average (f = 8) of one 3-D float32 field with 2-D area weights: the second north-star metric at
    C3072 -> C384 (17.9 GB per field), BASELINE configs[2]'s at C384 -> C48."""
    from fv3net_amd import ops

    g = torch.Generator(device=dev).manual_seed(0)
    try:
        obj = torch.rand((tiles, NZ, n, n), device=dev, generator=g) * 2000 - 1000
        area = torch.rand((tiles, n, n), device=dev, generator=g) * 0.5 + 0.5
        fn = lambda: ops.weighted_block_average(obj, area, 8)
        fn()
        torch.cuda.synchronize(dev)
        ms = time_kernel(fn, max(3, min(steps, 10)), dev)
        nel = obj.numel()
        alg_bytes = 4 * nel * (1 + 1 / 64) + 4 * nel / NZ
        traffic, source = pmc_traffic("wavg_block_kernel<float,float,8> C3072->C384") if n == 3072 else (None, None)
        return [{"kernel": "weighted_block_average (wavg_block_kernel<float,float,8>)",
                 "workload": f"{label} f=8, one [6,79,{n},{n}] f32 field, 2-D area weights",
                 "ms": ms, "roofline": {"bound": "hbm", "achieved": alg_bytes / ms / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                        "frac": alg_bytes / ms / 1e6 / PEAK_HBM_GBPS, "traffic": traffic, "traffic_source": source}}]
    except torch.cuda.OutOfMemoryError:
        return [{"kernel": "weighted_block_average", "workload": label, "error": "out of memory"}]


def remap_benchmark(dev, steps):
    """mappm at C384: 884 736 columns, km = kn = 79, native [tile, z, y, x] layout, iv = 1, kord = 1.  Target grid: the one
    the pipeline produces (fine delp ~ U(300, 1500) iid per cell -- the lanes of a wave drift ~16 target
    rows apart at the bottom of the column, the divergence worst case for a lockstep sweep; `smooth` shrinks the spread
    to a tenth, closer to a real atmosphere) read on its own coarse grid through (y // 8, x // 8), as the pipelines launch
    it.  float32 single field / 4 fields per sweep; float64 inputs 4 fields per sweep (what the float64 restarts hand
    over); both arithmetic modes (exact = the library default, bit-identical to the compiled Fortran)."""
    from fv3net_amd import ops

    n, ncol = 384, 6 * 384 * 384
    reps = max(3, min(steps, 10))
    out = []
    for data_label, noise in (("configs[2] data (iid delp)", 1.0), ("smooth delp (a tenth of the spread)", 0.1)):
        g = torch.Generator(device=dev).manual_seed(0)
        delp = 900 + (torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * noise
        area = torch.rand((6, n, n), device=dev, generator=g, dtype=torch.float64) * 0.5 + 0.5
        pe1 = ops.pressure_at_interface(delp, 300.0, 1)
        pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, 8), 300.0, 1)
        qs = [torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) * 2000 - 1000 for _ in range(4)]
        for dt, dname, e in ((torch.float32, "float", 4), (torch.float64, "double", 8)):
            p1, p2, fs = pe1.to(dt), pe2c.to(dt), [q.to(dt) for q in qs]
            for nf in ((1, 4) if dt == torch.float32 else (4,)):
                if noise != 1.0 and nf == 1:
                    continue
                for arith in ("exact", "fast"):
                    fn = lambda: ops.mappm_multi_coarse_target(p1, fs[:nf], p2, 8, z_axis=1, arith=arith)
                    fn()
                    torch.cuda.synchronize(dev)
                    ms = time_kernel(fn, reps, dev)
                    alg = ncol * ((NZ + 1) * e + nf * NZ * e + nf * NZ * 4) + p2.numel() * e
                    name = f"mappm_sweep_kernel<{dname}, {1 if nf == 1 else 2}, {1 if nf == 1 else 2}, {'true' if arith == 'fast' else 'false'}, true, false>"
                    out.append({
                        "kernel": f"{name} (arith={arith})",
                        "workload": f"C384 884736 columns x {nf} field(s), km=kn=79, coarse-pressure target, {data_label}",
                        "ms": ms, "ms_per_field": ms / nf, "columns_per_s": nf * ncol / ms * 1e3,
                        "roofline": {"bound": "hbm", "achieved": alg / ms / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                     "frac": alg / ms / 1e6 / PEAK_HBM_GBPS, "traffic": None,
                                     "note": "latency / VALU bound, not HBM bound: counters in profiles/r03_pmc_mappm.json"}})
        del delp, pe1, pe2c, qs
        torch.cuda.empty_cache()
    return out


def block_mean_benchmark(dev, steps):
    """The fused remap + masked block mean (fv3hip_mappm_block_mean; the pipelines take it where it pays: FV3NET_AMD_FUSED_BLOCK_MEAN)
    beside the three launches it replaces, 4 float64 fields, on both data sets: one entry, times in ms."""
    from fv3net_amd import ops

    n, ncol = 384, 6 * 384 * 384
    reps = max(3, min(steps, 10))
    table = {}
    for data_label, noise in (("configs[2] data (iid delp)", 1.0), ("smooth delp (a tenth of the spread)", 0.1)):
        g = torch.Generator(device=dev).manual_seed(0)
        delp = 900 + (torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * noise
        area = (torch.rand((6, n, n), device=dev, generator=g, dtype=torch.float64) * 0.5 + 0.5).float()
        pe1 = ops.pressure_at_interface(delp, 300.0, 1)
        pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, 8), 300.0, 1)
        qs = [torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) * 2000 - 1000 for _ in range(4)]

        def three_calls(arith):
            q2 = ops.mappm_multi_coarse_target(pe1, qs, pe2c, 8, z_axis=1, arith=arith)
            return ops.weighted_block_average_multi(q2, ops.mask_weights(area, pe2c, pe1, 1, coarse_factor=8), 8)

        for arith in ("exact", "fast"):
            fused, plain = (lambda: ops.mappm_block_mean(pe1, qs, pe2c, area, arith=arith)), (lambda: three_calls(arith))
            same = all(torch.equal(a, b) for a, b in zip(fused(), plain()))
            torch.cuda.synchronize(dev)
            table[f"{data_label}, arith={arith}"] = {"fused_ms": time_kernel(fused, reps, dev), "three_calls_ms": time_kernel(plain, reps, dev),
                                                    "bit_identical": bool(same)}
        del delp, pe1, pe2c, qs
        torch.cuda.empty_cache()
    alg = ncol * ((NZ + 1) * 8 + 4 * NZ * 8 + 4)
    best = min(v["fused_ms"] for v in table.values())
    return [{"kernel": "mappm_sweep_kernel<double, 2, 2, *, true, MEAN> + mean_rest_kernel (fused remap + masked 8x8 block mean; the pipelines' route where the data allow)",
             "workload": "C384 -> C48: 884736 columns x 4 float64 fields, km=kn=79, against mappm_multi_coarse_target + mask_weights + weighted_block_average",
             "ms": best, "calls": table,
             "roofline": {"bound": "hbm", "achieved": alg / best / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": alg / best / 1e6 / PEAK_HBM_GBPS,
                          "traffic": None, "note": "fastest of the four fused calls; inputs only (the fine result is never written)"}}]


def pipelines_benchmark(dev):
    """The three restart pipelines end to end, in the remap's exact arithmetic (the library default) and in the fast one."""
    from fv3net_amd import ops

    out = []
    keep = ops.MAPPM_ARITHMETIC
    try:
        # smooth thicknesses (a tenth of configs[2]'s spread, closer to a real restart file): the adaptive route of the
        # cell-centred fields settles on the fused remap + block-mean kernel here, on the three launches on the iid data below
        for entry in restart_pipeline_benchmark(dev, which=("pressure",), graph=False, delp_spread=0.1):
            entry["kernel"] += ", remap arith=exact, SMOOTH thicknesses (FV3NET_AMD_FUSED_BLOCK_MEAN=auto -> fused)"
            out.append(entry)
        for arith in ("exact", "fast"):
            ops.MAPPM_ARITHMETIC = arith
            which = ("sigma", "pressure", "blended") if arith == "fast" else ("pressure", "blended")  # (sigma has no remap)
            for entry in restart_pipeline_benchmark(dev, which=which):
                entry["kernel"] += f", remap arith={arith}" if "sigma" not in entry["kernel"] else ""
                out.append(entry)
    finally:
        ops.MAPPM_ARITHMETIC = keep
    return out


def secondary_benchmarks(dev, steps):
    """Everything besides the headline, ORDERED so that the end of the line -- what a truncated record keeps -- holds the
    north star's second metric and its callers: exploratory work first, then the emulators and the I/O paths, then the
    remap, the restart pipelines and, last, the C3072 -> C384 weighted_block_average."""
    out = []
    _guarded(out, split_bf16_benchmark, dev)                       # exploratory
    _guarded(out, dense_local_benchmark, dev, steps)               # C384 emulators (incl. their exploratory split-bf16 runs)
    _guarded(out, streaming_benchmark, dev)                        # PCIe-inclusive
    _guarded(out, io_pipeline_benchmark, dev)                      # file I/O inclusive
    _guarded(out, rank_latency_benchmark, dev)                     # one model rank's columns per call
    _guarded(out, block_mean_benchmark, dev, steps)                # fused remap + block mean
    _guarded(out, plain_network_benchmark, dev, steps)
    _guarded(out, wavg_benchmark, dev, steps, "C384->C48", 384)
    _guarded(out, remap_benchmark, dev, steps)
    _guarded(out, pipelines_benchmark, dev)
    _guarded(out, wavg_benchmark, dev, steps, "C3072->C384", 3072)
    return out


def restart_pipeline_benchmark(dev, n=384, f=8, reps=6, tiles=tuple(range(6)), which=("sigma", "pressure", "blended"), sync=None,
                               graph=True, delp_spread=1.0):
    """BASELINE configs[2] end to end: the three restart coarse-graining pipelines (vcm coarsen_restarts_on_sigma /
    _on_pressure / _via_blended_method, all four restart categories, 'complex' surface method) C384 -> C48 on float64
    restarts resident in HBM, through the drop-in Python API.  Wall time of a whole pipeline call.  ``tiles``: the cube
    tiles this rank owns (tile-sharded runs: the halo rows of the D-grid winds' edge pressures then come from the
    all-gather of parallel.exchange_edge_rows); ``sync``: fence to use instead of a device synchronise (a barrier)."""
    from fv3net_amd.cubedsphere import (coarsen_restarts_on_pressure, coarsen_restarts_on_sigma,
                                        coarsen_restarts_via_blended_method)
    from fv3net_amd.xr_compat import DataArray, Dataset

    nt = len(tiles)
    sync_is_default = sync is None
    sync = sync or (lambda: torch.cuda.synchronize(dev))
    g = torch.Generator(device=dev).manual_seed(3 + 17 * int(tiles[0]))
    u = lambda lo, hi, *shape: torch.rand(shape, device=dev, generator=g, dtype=torch.float64) * (hi - lo) + lo
    zc = ["tile", "Time", "zaxis_1", "yaxis_2", "xaxis_1"]
    core = Dataset({
        "u": DataArray(u(-30, 30, nt, 1, NZ, n + 1, n), dims=["tile", "Time", "zaxis_1", "yaxis_1", "xaxis_1"]),
        "v": DataArray(u(-30, 30, nt, 1, NZ, n, n + 1), dims=["tile", "Time", "zaxis_1", "yaxis_2", "xaxis_2"]),
        **{k: DataArray(u(lo, hi, nt, 1, NZ, n, n), dims=zc) for k, (lo, hi) in
           {"W": (-1, 1), "T": (200, 300), "delp": (300, 1500), "DZ": (-500, -50), "ua": (-30, 30), "va": (-30, 30)}.items()},
        "phis": DataArray(u(0, 1e4, nt, 1, n, n), dims=["tile", "Time", "yaxis_2", "xaxis_1"]),
    })
    tracers = ["sphum", "liq_wat", "rainwat", "ice_wat", "snowwat", "graupel", "o3mr", "sgs_tke", "cld_amt"]
    tracer = Dataset({k: DataArray(u(0, 0.02, nt, 1, NZ, n, n), dims=["tile", "Time", "zaxis_1", "yaxis_1", "xaxis_1"]) for k in tracers})
    srf = Dataset({k: DataArray(u(-10, 10, nt, 1, n, n), dims=["tile", "Time", "yaxis_1", "xaxis_1"]) for k in ("u_srf", "v_srf")})
    # surface data: the variables, dims and value ranges of the reference's regression schema (tests/golden metadata)
    with np.load(os.path.join(ROOT, "tests", "golden", "coarsen_restarts_reference.npz")) as z:
        meta = json.loads(bytes(z["meta_json"]).decode())
    rng = np.random.default_rng(0)
    sfc = Dataset()
    for name, info in meta["inputs"]["sfc_data"].items():
        lo, hi = meta["ranges"].get(name, meta["default_range"])
        shape = [nt if d == "tile" else sz for d, sz in zip(info["dims"], info["shape"])][:-2] + [n, n]
        sfc[name] = DataArray(torch.from_numpy(rng.uniform(lo, hi, shape).astype(info["dtype"])).to(dev), dims=info["dims"])
    grid = Dataset({"area": DataArray(u(0.5, 1, nt, n, n).float(), dims=["tile", "grid_yt", "grid_xt"]),
                    "dx": DataArray(u(0.5, 1, nt, n + 1, n).float(), dims=["tile", "grid_y", "grid_xt"]),
                    "dy": DataArray(u(0.5, 1, nt, n, n + 1).float(), dims=["tile", "grid_yt", "grid_x"])})
    if delp_spread != 1.0:   # (the remap benchmark's `smooth` data: that fraction of configs[2]'s spread around 900 Pa)
        core["delp"] = DataArray(900 + (core["delp"].data - 900) * delp_spread, dims=zc)
    restarts = {"fv_core.res": core, "fv_tracer.res": tracer, "fv_srf_wnd.res": srf, "sfc_data": sfc}
    nbytes = sum(v.data.numel() * v.data.element_size() for ds in restarts.values() for v in ds.values())
    out = []
    fns = {"sigma": ("coarsen_restarts_on_sigma", lambda: coarsen_restarts_on_sigma(f, grid, restarts, coarsen_agrid_winds=True)),
           "pressure": ("coarsen_restarts_on_pressure", lambda: coarsen_restarts_on_pressure(f, grid, 300.0, restarts, coarsen_agrid_winds=True)),
           "blended": ("coarsen_restarts_via_blended_method",
                       lambda: coarsen_restarts_via_blended_method(f, grid, 300.0, restarts, coarsen_agrid_winds=True))}
    for key in which:
        label, fn = fns[key]
        for _ in range(6):   # (memory handed between the pipelines' streams returns to the allocator an event later: a few calls until its pools have settled)
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        entry = {"kernel": f"{label} (whole pipeline, Python API)",
                 "workload": f"C{n}->C{n // f}, 4 restart categories, {nt} tile(s), {nbytes / 1e9:.2f} GB of float64 restarts in HBM",
                 "ms": ms, "bytes": nbytes,
                 "roofline": {"bound": "hbm", "achieved": nbytes / ms / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                              "frac": nbytes / ms / 1e6 / PEAK_HBM_GBPS, "traffic": None,
                              "note": "input bytes / wall time"}}
        if graph and nt == 6 and sync_is_default:  # the same call captured as a HIP graph and replayed (fv3net_amd.graphs)
            try:
                from fv3net_amd.graphs import GraphedCall

                gc = GraphedCall(fn, device=dev)
                for _ in range(3):
                    gc.replay()
                sync()
                t0 = time.perf_counter()
                for _ in range(reps * 2):
                    gc.replay()
                sync()
                gms = (time.perf_counter() - t0) / (reps * 2) * 1e3
                entry["graph_replay_ms"] = gms
                entry["graph_replay_frac"] = nbytes / gms / 1e6 / PEAK_HBM_GBPS
                del gc
            except Exception as err:  # noqa: BLE001
                entry["graph_replay_error"] = f"{type(err).__name__}: {err}"
        out.append(entry)
    return out


def io_pipeline_benchmark(dev, n=384, f=8):
    """SURVEY 8f rank 4 end to end: synthetic C384 diagnostics written as the model writes them (6 tiles x 16 sub-tile
    netCDF classic files, coarsen.py:27) -> pinned host buffers -> device block average -> coarse zarr store.  File I/O
    included, so never the headline; the reads of tile t + 1 overlap the device work of tile t."""
    import shutil
    import tempfile

    from fv3net_amd.io import coarsen_subtile_files_to_zarr, netcdf

    root = tempfile.mkdtemp(prefix="fv3net_amd_io_")
    try:
        rng = np.random.default_rng(7)
        area = rng.uniform(0.5, 1, (6, n, n)).astype(np.float32)
        prefix = os.path.join(root, "atmos_8xdaily")
        t0 = time.perf_counter()
        for tile in range(1, 7):
            netcdf.write_subtile_files(prefix, tile, {
                "T": (["time", "pfull", "yaxis_1", "xaxis_1"], rng.standard_normal((1, NZ, n, n), dtype=np.float32)),
                "PRATEsfc": (["time", "yaxis_1", "xaxis_1"], rng.standard_normal((1, n, n), dtype=np.float32))}, layout=(4, 4))
        t_make = time.perf_counter() - t0
        coarsen_subtile_files_to_zarr(prefix, os.path.join(root, "warm.zarr"), area, f)  # page cache + pinned pools warm
        t0 = time.perf_counter()
        st = coarsen_subtile_files_to_zarr(prefix, os.path.join(root, "coarse.zarr"), area, f)
        wall = time.perf_counter() - t0
        return [{
            "kernel": "sub-tile netCDF files -> pinned -> weighted_block_average -> zarr (fv3net_amd.io.coarsen_subtile_files_to_zarr)",
            "workload": f"C{n}->C{n // f}: 96 classic-netCDF sub-tile files, {st['bytes_in'] / 1e6:.0f} MB in (page cache warm), "
                        f"{st['bytes_out'] / 1e6:.1f} MB of uncompressed zarr out; synthetic files took {t_make:.1f} s to write",
            "ms": wall * 1e3, "read_ms": st["read_s"] * 1e3, "device_ms_incl_transfers": st["device_s"] * 1e3, "write_ms": st["write_s"] * 1e3,
            "GBps_end_to_end": st["bytes_in"] / wall / 1e9,
            "roofline": {"bound": "file I/O", "achieved": None, "peak": None, "unit": "GB/s", "frac": None, "traffic": None},
        }]
    finally:
        shutil.rmtree(root, ignore_errors=True)


def split_bf16_benchmark(dev):
    """EXPLORATORY (the headline stays fp32): the whole network on the bf16 matrix cores with every fp32 operand split into
    three bf16 pieces.  (The standalone single-layer study that preceded it is benchmarks/bf16split/run.py; its numbers are
    in DESIGN.md section 10 and no longer in the bench line.)"""
    return split_bf16_network_benchmark(dev)


PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense v_mfma_f32_32x32x16_bf16 peak (MI355X_MICROARCH.md)


def split_bf16_network_benchmark(dev, steps=10):
    """EXPLORATORY: the whole headline network (same graph, same inputs, same launch shape) through the fused split-bf16 kernel
    (``fv3hip_mlp3_*``, csrc/mlp_bf16x3.hip) -- time, fp32-equivalent TFLOP/s and the error against the float64 oracle next to
    the fp32 kernel's on the same 4096 columns."""
    from fv3net_amd.mlp import MlpModel, MlpModelSplitBf16

    out = []
    ncol = 6 * 384 * 384
    src = zc_inputs_device(dev, ncol, seed=1000)
    for residuals in (False, True):
        spec = zc_spec(0, residuals=residuals)
        try:
            row = {}
            for label, cls in (("fp32", MlpModel), ("split", MlpModelSplitBf16)):
                model = cls(spec, device=dev)
                sliced = parity_slice_predict(model, src)
                ms = time_kernel(lambda: model.predict(src), steps, dev, warm=6)
                row[label] = (ms, model.flops_per_sample, sliced)
                del model
            ms, flops, sliced = row["split"]
            tf = flops * ncol / (ms * 1e-3) / 1e12
            out.append({
                "kernel": "mlp3_kernel<13> (EXPLORATORY split-bf16 arithmetic, 6 bf16 MFMAs per product; not the product path)",
                "workload": "the headline network, 884736 columns, " + ("791 output rows (with the Difference residuals)" if residuals
                                                                         else "396 direct outputs only"),
                "ms": ms, "columns_per_s": ncol / ms * 1e3, "fp32_kernel_ms_same_run": row["fp32"][0],
                "max_rel_err_vs_f64": parity_max_rel(spec, sliced), "fp32_kernel_max_rel_err_vs_f64": parity_max_rel(spec, row["fp32"][2]),
                "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS / 6, "unit": "TFLOP/s (fp32-equivalent)",
                             "frac": 6 * tf / PEAK_BF16_MFMA_TFLOPS, "traffic": None,
                             "note": "against what the kernel issues: six bf16 MFMAs per fp32 product, so the roof is the dense "
                                     "bf16 peak / 6 = 419 fp32-equivalent TFLOP/s (VERDICT r02 #12); for scale, the fp32 MFMA "
                                     f"peak is {PEAK_FP32_MFMA_TFLOPS}"}})
        except Exception as err:  # noqa: BLE001
            out.append({"kernel": "mlp3_kernel (exploratory)", "error": f"{type(err).__name__}: {err}"})
    return out


def streaming_benchmark(dev, n_snapshots=24):
    """BASELINE configs[3] on one GPU's share: a sequence of C384 tile snapshots (147 456 columns) that start and
    end in (pinned) host memory, H2D / kernel / D2H of consecutive snapshots overlapped on three HIP streams
    (fit/streaming.py).  PCIe-inclusive, so never the headline value."""
    from fv3net_amd.fit.streaming import SnapshotStream
    from fv3net_amd.mlp import MlpModel

    ncol = 384 * 384
    model = MlpModel(zc_spec(0), device=dev)
    out = []
    for in_dtype in (np.float32, np.float64):
        stream = SnapshotStream(model, ncol, in_dtype=in_dtype, n_buffers=3)
        src = zc_inputs_numpy(np.random.default_rng(5), ncol)
        for slot in range(stream.n_buffers):  # synthetic snapshots already sit in the pinned buffers
            for k, v in stream.host_inputs(slot).items():
                v[...] = np.ascontiguousarray(src[k].T)
        nothing = lambda i, arrays: None
        stream.run(4, nothing, nothing)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        stream.run(n_snapshots, nothing, nothing)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out.append({
            "kernel": "SnapshotStream (H2D + mlp_fused_kernel + D2H on three streams)",
            "workload": f"{n_snapshots} C384 tile snapshots of {ncol} columns from/to pinned host memory, {np.dtype(in_dtype).name} "
                        f"inputs ({stream.bytes_in / 1e6:.0f} MB in, {stream.bytes_out / 1e6:.0f} MB out per snapshot)",
            "ms": dt / n_snapshots * 1e3, "columns_per_s": ncol * n_snapshots / dt,
            "pcie_GBps": {"h2d": stream.bytes_in * n_snapshots / dt / 1e9, "d2h": stream.bytes_out * n_snapshots / dt / 1e9},
            "roofline": {"bound": "pcie", "achieved": None, "peak": None, "unit": "GB/s", "frac": None, "traffic": None},
        })
        del stream
    return out


def emulator_state(dev, ncol, seed=2, dtype=torch.float32):
    """[79, ncol] state arrays of the gscond / precpd emulators' inputs (device)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    u = lambda lo, hi, shape=(NZ, ncol): torch.rand(shape, device=dev, generator=g) * (hi - lo) + lo
    qv = torch.pow(10.0, u(-8, -2))
    t = u(180, 310)
    st = {
        "air_temperature_input": t, "specific_humidity_input": qv,
        "cloud_water_mixing_ratio_input": torch.where(u(0, 1) < 0.7, torch.zeros_like(t), torch.pow(10.0, u(-10, -3))),
        "pressure_thickness_of_atmospheric_layer": u(300, 1500), "air_temperature_after_last_gscond": t + u(-1, 1),
        "specific_humidity_after_last_gscond": qv * u(0.9, 1.1), "air_pressure": u(300, 101000),
        "surface_air_pressure": u(95000, 103000, (ncol,)), "surface_air_pressure_after_last_gscond": u(95000, 103000, (ncol,)),
    }
    return {k: v.to(dtype) for k, v in st.items()}


def emulator_specs(width=256):
    """(gscond "dense-local" regressor spec, precpd "rnn-v1-shared-weights" spec, number of inputs): the reference's
    production architectures (projects/microphysics/configs/models/{gscond,precpd}.yaml) with random-init weights."""
    from fv3net_amd.local_mlp import ConditionalScale, LocalInput, LocalMlpSpec, LocalOutput, RnnLayer, RnnSpec

    rng = np.random.default_rng(0)
    two_d = {"surface_air_pressure", "surface_air_pressure_after_last_gscond"}
    names = ["air_temperature_input", "specific_humidity_input", "cloud_water_mixing_ratio_input",
             "pressure_thickness_of_atmospheric_layer", "air_temperature_after_last_gscond", "specific_humidity_after_last_gscond",
             "air_pressure", "surface_air_pressure", "surface_air_pressure_after_last_gscond"]
    raw = [(n, n, "none", 0.0) for n in names] + [(k, v[0], "log", v[1]) for k, v in LOG_FIELDS.items()]
    inputs = [LocalInput(name, source, tr, eps, center=rng.normal(0, 1, NZ).astype(np.float32) if source not in two_d else
                         np.float32(rng.normal()), scale=np.float32(rng.uniform(0.5, 2))) for name, source, tr, eps in sorted(raw)]
    k = len(inputs)
    edges = np.linspace(180, 310, 51)[:-1].astype(np.float32)
    outs = [LocalOutput(f"{v}_gscond_difference_tscaled", scale=rng.uniform(0.5, 2, NZ).astype(np.float32),
                        center=rng.normal(0, 0.1, NZ).astype(np.float32),
                        conditional=ConditionalScale(f"{v}_gscond_difference", "air_temperature_input", edges,
                                                     rng.uniform(0, 2, 50).astype(np.float32), rng.normal(0, 0.1, 50).astype(np.float32), 0.1),
                        after=f"{src}_after_gscond", before=f"{src}_input")
            for v, src in (("humidity", "specific_humidity"), ("temperature", "air_temperature"))]
    spec = LocalMlpSpec(
        inputs=inputs,
        hidden_kernels=[(rng.normal(0, 1, (k, width)) / np.sqrt(k)).astype(np.float32),
                        (rng.normal(0, 1, (width, width)) / np.sqrt(width)).astype(np.float32)],
        hidden_biases=[rng.normal(0, 0.1, width).astype(np.float32) for _ in range(2)], outputs=outs,
        out_kernel=(rng.normal(0, 1, (width, 2)) / np.sqrt(width)).astype(np.float32), out_bias=rng.normal(0, 0.1, 2).astype(np.float32))
    layers, fan = [], k
    for _ in range(2):
        layers.append(RnnLayer((rng.normal(0, 1, (fan, width)) / np.sqrt(fan)).astype(np.float32),
                               (rng.normal(0, 0.6, (width, width)) / np.sqrt(width)).astype(np.float32),
                               rng.normal(0, 0.1, width).astype(np.float32)))
        fan = width
    routs = [LocalOutput("total_precipitation", scale=np.float32(1e-3), center=np.float32(1e-4), single_level=True)] + [
        LocalOutput(f"{v}_precpd_difference", scale=rng.uniform(0.5, 2, NZ).astype(np.float32), center=rng.normal(0, 0.1, NZ).astype(np.float32),
                    after=f"{src}_after_precpd", before=f"{src}_input", value_limit=lim)
        for v, src, lim in (("cloud", "cloud_water_mixing_ratio", (None, 0.0)), ("temperature", "air_temperature", (None, 0.0)),
                            ("humidity", "specific_humidity", (0.0, None)))]
    rspec = RnnSpec(inputs=inputs, layers=layers, outputs=routs,
                    out_kernel=(rng.normal(0, 1, (width, 4)) / np.sqrt(width)).astype(np.float32), out_bias=rng.normal(0, 0.1, 4).astype(np.float32))
    return spec, rspec, k


def dense_local_benchmark(dev, steps):
    """The reference's production gscond emulator (projects/microphysics/configs/models/gscond.yaml: "dense-local",
    12 inputs at a point -> 2 x 256 -> 2 outputs, temperature-conditional un-scaling + Difference outputs) over one
    C384 snapshot: 79 x 884 736 points.  MFMA-bound like the headline kernel, plus two HBM-bound passes."""
    from fv3net_amd.local_mlp import LocalMlpModel

    ncol, width = 6 * 384 * 384, 256
    st = emulator_state(dev, ncol)
    spec, rspec, k = emulator_specs(width)
    model = LocalMlpModel(spec, device=dev)
    fn = lambda: model.predict(st)
    fn()
    torch.cuda.synchronize(dev)
    ms = time_kernel(fn, max(3, min(steps, 5)), dev, warm=1)
    flops = 2 * (k * width + width * width + width * 2)
    achieved = flops * NZ * ncol / (ms * 1e-3) / 1e12
    out = [{
        "kernel": "dense-local emulator (local_pack + mlp_fused_kernel + local_unpack)",
        "workload": f"C384 gscond regressor: {k} inputs -> 2 x {width} -> 2 outputs at each of 79 x {ncol} points, float32 state",
        "ms": ms, "columns_per_s": ncol / ms * 1e3, "flops_per_point": flops,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None},
    }]
    del model
    try:  # the same emulator on the opt-in split-bf16 arithmetic (EXPLORATORY)
        model = LocalMlpModel(spec, device=dev, arithmetic="split-bf16")
        fn = lambda: model.predict(st)
        fn()
        torch.cuda.synchronize(dev)
        ms3 = time_kernel(fn, max(3, min(steps, 5)), dev, warm=1)
        tf = flops * NZ * ncol / (ms3 * 1e-3) / 1e12
        out.append({
            "kernel": "dense-local emulator on mlp3_kernel<1> (EXPLORATORY split-bf16 arithmetic, opt-in)",
            "workload": out[0]["workload"], "ms": ms3, "columns_per_s": ncol / ms3 * 1e3, "fp32_kernel_ms_same_run": ms,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS / 6, "unit": "TFLOP/s (fp32-equivalent)",
                         "frac": 6 * tf / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "note": "roof = dense bf16 peak / 6 (six bf16 MFMAs per fp32 product)"}})
        del model
    except Exception as err:  # noqa: BLE001
        out.append({"kernel": "dense-local emulator on mlp3_kernel<1>", "error": f"{type(err).__name__}: {err}"})
    # the production precpd emulator (configs/models/precpd.yaml: "rnn-v1-shared-weights", 2 stacked SimpleRNN(256) over the
    # 79 levels, 4 outputs): 79 x 2 launches of the fused kernel, the states ping-ponging in HBM
    from fv3net_amd.local_mlp import RnnModel

    rmodel = RnnModel(rspec, device=dev)
    fn = lambda: rmodel.predict(st)
    fn()
    torch.cuda.synchronize(dev)
    ms = time_kernel(fn, 2, dev, warm=1)
    flops = 2 * (k * width + 3 * width * width + width * 4)  # the recurrences themselves, without the identity output layers
    achieved = flops * NZ * ncol / (ms * 1e-3) / 1e12
    ms_rnn_fp32 = ms
    out.append({
        "kernel": "RNN emulator (local_pack + 79 x 2 mlp_fused_kernel launches + local_unpack)",
        "workload": f"C384 precpd regressor: {k} inputs -> 2 x SimpleRNN({width}) over 79 levels -> 4 outputs, {ncol} columns, float32 state",
        "ms": ms, "columns_per_s": ncol / ms * 1e3, "flops_per_point": flops,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None},
    })
    del rmodel
    try:  # the same recurrence on the opt-in split-bf16 arithmetic (EXPLORATORY)
        rmodel = RnnModel(rspec, device=dev, arithmetic="split-bf16")
        fn = lambda: rmodel.predict(st)
        fn()
        torch.cuda.synchronize(dev)
        ms3 = time_kernel(fn, 2, dev, warm=1)
        tf = flops * NZ * ncol / (ms3 * 1e-3) / 1e12
        out.append({
            "kernel": "RNN emulator on mlp3_kernel<1> cells (EXPLORATORY split-bf16 arithmetic, opt-in)",
            "workload": out[-1]["workload"], "ms": ms3, "columns_per_s": ncol / ms3 * 1e3, "fp32_kernel_ms_same_run": ms_rnn_fp32,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS / 6, "unit": "TFLOP/s (fp32-equivalent)",
                         "frac": 6 * tf / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "note": "roof = dense bf16 peak / 6 (six bf16 MFMAs per fp32 product)"}})
        del rmodel
    except Exception as err:  # noqa: BLE001
        out.append({"kernel": "RNN emulator on mlp3_kernel<1> cells", "error": f"{type(err).__name__}: {err}"})
    return out


def rank_latency_benchmark(dev, column_counts=(2304, 9216), reps=20):
    """The shape of the prognostic loop (north_star: drops into prognostic_c48_run): ONE model rank's columns per call --
    2 304 (a 48 x 48 subdomain at C48 on 6 ranks) and 9 216 -- not a C384 snapshot.  Per emulator and column count the wall
    time of a call with both MLP kernels: the feature-split kernel for small sample counts (the library's choice here) and
    the 128-sample-tile kernel pinned (`small_limit=0`; what rounds 1-2 ran), plus the number of library launches per call.
      * MicrophysicsHook.microphysics(state) as the Fortran model calls it: float64 [79, ncol] numpy arrays in, state
        updated in place (upload, Zhao-Carr dense network, squash / range / level masks, download);
      * the same emulator on device-resident float64 arrays (no PCIe: the network + masks alone);
      * the gscond dense-local regressor and the precpd RNN (2 x SimpleRNN(256) over 79 levels, HIP-graph replay) on
        device-resident float32 state."""
    from fv3net_amd import _lib, mlp
    from fv3net_amd.emulation import HipEmulator, MicrophysicsHook
    from fv3net_amd.emulation.config import ModelConfig
    from fv3net_amd.local_mlp import LocalMlpModel, RnnModel
    from fv3net_amd.mlp import ResidualSpec

    out = []
    zc = zc_spec(0)
    zc.residuals = [
        ResidualSpec("air_temperature_after_precpd", "air_temperature_input", "temperature_precpd_difference"),
        ResidualSpec("specific_humidity_after_precpd", "specific_humidity_input", "humidity_precpd_difference"),
        ResidualSpec("cloud_water_mixing_ratio_after_precpd", "cloud_water_mixing_ratio_input", "cloud_precpd_difference"),
    ]
    cfg = ModelConfig.from_dict({"cloud_squash": 1e-8, "ranges": {"total_precipitation": {"min": 0.0}},
                                 "mask_emulator_levels": {"air_temperature_after_precpd": {"start": 74, "fill_value": "air_temperature_input"}}})
    gspec, rspec, _ = emulator_specs(256)

    def wall(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e3

    def launches(fn):
        count, real = [0], _lib.call_on

        def counting(*a, **k):
            count[0] += 1
            return real(*a, **k)

        _lib.call_on = counting
        try:
            fn()
        finally:
            _lib.call_on = real
        return count[0]

    for ncol in column_counts:
        host = {k: np.ascontiguousarray(v.T.astype(np.float64)) for k, v in zc_inputs_numpy(np.random.default_rng(5), ncol).items()}
        host["model_time"], host["rank"] = [2016, 8, 1, 0, 15, 0], 0
        dev64t = {k: torch.from_numpy(v).to(dev).t() for k, v in host.items() if isinstance(v, np.ndarray)}  # [sample, feature] views, as the hook hands them to the model
        st32 = emulator_state(dev, ncol, seed=7)
        entry = {"kernel": "one model rank's columns per call (MicrophysicsHook / emulators)", "workload": f"{ncol} columns x 79 levels", "calls": {}}
        for label, limit in (("feature-split kernel (default)", None), ("128-sample tiles pinned", 0)):
            mlp.DEFAULT_SMALL_LIMIT = limit
            try:
                hook = MicrophysicsHook(model=HipEmulator(zc), mask=cfg._build_mask())
                rnn = RnnModel(rspec, device=dev, use_graph=True)
                loc = LocalMlpModel(gspec, device=dev)
                emu = hook.model
                res = {
                    "hook_wall_ms (float64 numpy state in, updated in place)": wall(lambda: hook.microphysics(dict(host))),
                    "zhao_carr_dense_device_resident_ms": wall(lambda: emu(dev64t)),
                    "gscond_dense_local_device_resident_ms": wall(lambda: loc.predict(st32)),
                    "precpd_rnn_device_resident_ms (HIP-graph replay)": wall(lambda: rnn.predict(st32)),
                    "library_launches": {"hook": launches(lambda: hook.microphysics(dict(host))), "dense_local": launches(lambda: loc.predict(st32))},
                }
                entry["calls"][label] = res
                del hook, rnn, loc
            except Exception as err:  # noqa: BLE001
                entry["calls"][label] = {"error": f"{type(err).__name__}: {err}"}
            finally:
                mlp.DEFAULT_SMALL_LIMIT = None
        out.append(entry)
    torch.cuda.empty_cache()
    return out


def sharded_secondary_benchmarks(dev, steps, rank, world, rehearsal):
    """N > 1: the second metric across the GPUs (BASELINE configs[4]).  Every rank takes part (collective).
      * one C3072 -> C384 float32 field sharded by (tile, row band) over ALL ranks (parallel.tile_bands: 6 tiles x 4 bands on
        8 GPUs), no data-path collective: per-rank and aggregate algorithmic GB/s against N x 8 TB/s;
      * the pressure-level restart pipeline C384 -> C48 sharded by whole tiles over min(N, 6) ranks, the edge pressures of
        the D-grid winds taking their halo rows from the all-gather of parallel.exchange_edge_rows (RCCL over xGMI)."""
    from fv3net_amd import parallel

    host = torch.device("cpu")

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=host if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    out = []
    try:
        n = int(os.environ.get("FV3_BENCH_SHARDED_N", "3072"))
        units = parallel.units_of_rank(6, n, 8, world, rank)
        g = torch.Generator(device=dev).manual_seed(200 + rank)
        objs = [torch.rand((NZ, r1 - r0, n), device=dev, generator=g) * 2000 - 1000 for (_, r0, r1) in units]
        areas = [torch.rand((r1 - r0, n), device=dev, generator=g) * 0.5 + 0.5 for (_, r0, r1) in units]
        fn = lambda: parallel.weighted_block_average_banded(objs, areas, 8)
        fn()
        fence()
        reps = max(3, min(steps, 10))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        mine_ms = (time.perf_counter() - t0) / reps * 1e3
        fence()
        ms = reduce_max(mine_ms)
        nel_rank = sum(o.numel() for o in objs)
        nel = 6 * NZ * n * n
        alg = lambda e: 4 * e * (1 + 1 / 64) + 4 * e / NZ
        out.append({
            "kernel": "weighted_block_average, (tile, row-band) sharded",
            "workload": f"C{n}->C{n // 8} f=8, one [6,79,{n},{n}] f32 field over {world} ranks: {len(units)} band(s) of "
                        f"{units[0][2] - units[0][1]} rows per rank, no collective",
            "ms": ms, "rank0_ms": mine_ms, "rank0_GBps": alg(nel_rank) / mine_ms / 1e6,
            "roofline": {"bound": "hbm", "achieved": alg(nel) / ms / 1e6, "peak": PEAK_HBM_GBPS * world, "unit": "GB/s",
                         "frac": alg(nel) / ms / 1e6 / (PEAK_HBM_GBPS * world), "traffic": None,
                         "note": "aggregate algorithmic bytes of the whole field / max-over-ranks time, against N x 8 TB/s"},
        })
        del objs, areas
    except Exception as err:  # noqa: BLE001
        out.append({"kernel": "weighted_block_average, (tile, row-band) sharded", "error": f"{type(err).__name__}: {err}"})
    torch.cuda.empty_cache()
    try:
        owners = min(world, 6)
        group = dist.new_group(list(range(owners))) if owners < world else None  # (every rank calls new_group)
        res = None
        if rank < owners:
            parallel.use_group(group)
            tiles = parallel.tiles_of_rank(owners, rank)
            sync = (lambda: (torch.cuda.synchronize(dev), dist.barrier(group=group), torch.cuda.synchronize(dev)))
            res = restart_pipeline_benchmark(dev, tiles=tuple(tiles), which=("pressure",), sync=sync)[0]
            parallel.use_group(None)
        fence()
        ms = reduce_max(res["ms"] if res else 0.0)
        if rank == 0:
            total = res["bytes"] * 6 / len(parallel.tiles_of_rank(owners, 0))
            res.update({"kernel": res["kernel"] + f", tile-sharded over {owners} of {world} ranks, halo rows by all_gather",
                        "ms": ms, "roofline": {"bound": "hbm", "achieved": total / ms / 1e6, "peak": PEAK_HBM_GBPS * owners,
                                               "unit": "GB/s", "frac": total / ms / 1e6 / (PEAK_HBM_GBPS * owners), "traffic": None,
                                               "note": "input bytes of the whole cube / max-over-ranks wall time"}})
            out.append(res)
    except Exception as err:  # noqa: BLE001
        out.append({"kernel": "coarsen_restarts_on_pressure, tile-sharded", "error": f"{type(err).__name__}: {err}"})
    torch.cuda.empty_cache()
    return out


def visible_gpu_count():
    """GPU agents of the KFD topology (a node with a non-zero simd_count is a GPU; CPUs have simd_count 0), narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  Pure file reads -- no HIP / HSA / amdsmi call.  None if unknown."""
    import glob

    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as fh:
                for line in fh:
                    key, _, val = line.partition(" ")
                    if key == "simd_count" and int(val) > 0:
                        count += 1
        except OSError:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            count = min(count, len([v for v in val.split(",") if v.strip() != ""]))
    return count


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher (WORLD_SIZE unset): start the N ranks here, as fresh child
    processes, BEFORE anything in this process touches the GPU, and leave with their worst exit code.  Rank 0's
    stdout (the JSON line) passes through.  If any rank fails the others are stopped: never a silent 1-GPU line."""
    import socket
    import subprocess

    rehearsal = os.environ.get("FV3_BENCH_REHEARSAL") == "1"
    # No GPU-library call in this parent (`torch.cuda.device_count()` falls back to hipGetDeviceCount where
    # amdsmi is unusable, and a parent that has initialised HIP must not start children that exec): the GPUs are counted
    # from the kernel driver's topology files; where those are unreadable the count is unknown and rank r itself exits 2
    # when cuda:r does not exist (the loop below propagates it).
    have = visible_gpu_count()
    if not rehearsal and have is not None and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0:
                worst = worst or (code if code > 0 else 1)
                for other in live:  # a rank died: the others would wait at the barrier forever
                    other.terminate()
    sys.exit(worst)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)  # (the clocks ramp over the first ~6 launches after a pause)
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the first 4096 timed columns")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box: FV3_BENCH_REHEARSAL=1 puts every rank on cuda:0 and rendezvouses over gloo
    # (two ranks cannot share a device under RCCL).  Real runs: one rank per GPU, backend nccl = RCCL over xGMI.
    rehearsal = os.environ.get("FV3_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():  # (a rank's own process may ask; the launcher parent never does)
        print(f"bench.py: rank {rank} needs cuda:{local_rank}, {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from fv3net_amd import _lib, ops
    from fv3net_amd.mlp import MlpModel

    _lib.load()
    spec = zc_spec(0)  # the production graph: 396 direct + 395 residual output rows
    model = MlpModel(spec, device=dev)
    ncol = 6 * 384 * 384
    src = zc_inputs_device(dev, ncol, seed=1000 + rank)
    n_out_rows = sum(OUTPUTS.values()) + NZ * len(RESIDUALS)

    def step():
        return model.predict(src)

    # parity of the timed object on a slice of the timed inputs (rank 0; the oracle is the checker, never the thing timed):
    # device part now, oracle part after the timed loop
    sliced, parity, parity_variant = None, None, None
    if rank == 0 and not args.no_parity:
        try:
            sliced = parity_slice_predict(model, src)
            parity_variant = sliced["variant"]
        except Exception as err:  # noqa: BLE001
            parity_variant = f"failed: {type(err).__name__}: {err}"
    for _ in range(args.warmup):
        step()
    variant = model.last_variant

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    from fv3net_amd.ops import HipTimer

    timer = HipTimer()
    t0 = time.perf_counter()
    timer.start(dev)
    for _ in range(args.steps):
        step()
    timer.stop(dev)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = timer.elapsed_ms() / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = None
    if rank == 0:
        if sliced is not None:
            try:
                parity = parity_max_rel(spec, sliced)
            except Exception as err:  # noqa: BLE001
                parity_variant = f"failed: {type(err).__name__}: {err}"
        total_cols = ncol * world * args.steps
        flops = model.flops_per_sample
        achieved = flops * ncol / (kernel_ms * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(variant)
        line = {
            "metric": "columns/s ML-tendency inference at C384x79 (Zhao-Carr dense emulator)",
            "value": total_cols / elapsed,
            "unit": "columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (SURVEY 8d config 2 distributions, random-init weights of the dense.yaml architecture)",
            "config": {
                "workload": "C384 Zhao-Carr microphysics MLP emulator, one 6x384x384x79 snapshot per GPU "
                            "(BASELINE configs[1]), the production graph of projects/microphysics/train/dense.yaml: "
                            f"K=711 -> 256 -> 256 -> 396 direct outputs + the five Difference residuals after = before + "
                            f"difference ({n_out_rows} output rows), float32 [feature, sample] inputs in HBM",
                "columns_per_gpu": ncol,
                "flops_per_column": flops,
                "output_rows": n_out_rows,
                "parallelism": f"snapshot/tile sharding over {world} GPU(s), no collective on the data path",
                "parity_max_rel": parity,
                "parity_note": "max over the 11 outputs of max|gpu - f64 oracle| / max|oracle| on the first 4096 columns of a "
                               f"full-size call of the timed model object ({parity_variant}); bar 1e-5",
            },
            "parity_max_rel": parity,
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": variant,
                "kernel_ms": kernel_ms,
            },
        }
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline(spec)
            except Exception as err:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "columns/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {type(err).__name__}: {err}"}
    del src
    torch.cuda.empty_cache()
    if not args.no_secondary:
        # (never lose the headline line to a secondary workload)
        if world == 1:
            try:
                sec = secondary_benchmarks(dev, args.steps)
            except Exception as err:  # noqa: BLE001
                sec = [{"kernel": "secondary_benchmarks", "error": f"{type(err).__name__}: {err}"}]
        else:
            sec = sharded_secondary_benchmarks(dev, args.steps, rank, world, rehearsal)  # collective: every rank takes part
        if line is not None:
            line["secondary"] = sec
    if rank == 0:
        info = ops.device_info()
        line["device"] = {"name": info["name"], "arch": info["arch"], "compute_units": info["compute_units"]}
        # One JSON line.  Key order: the secondary workloads first (their exploratory entries leading), the headline fields
        # with `roofline` and `cpu_baseline` LAST, so that a record which keeps only the end of the output still holds both
        # north-star metrics (the last secondaries are the remap, the pipelines and the C3072 -> C384 coarsening).
        tail_keys = [k for k in line if k not in ("secondary", "device")]
        ordered = {k: line[k] for k in ("device", "secondary") if k in line}
        ordered.update({k: line[k] for k in tail_keys})
        print(json.dumps(ordered), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
