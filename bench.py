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


def zc_spec(seed=0):
    """Random-init weights of the reference architecture; normalisation fitted on a sample the
    way MicrophysicsConfig does (center per feature, one std over all features)."""
    from fv3net_amd.mlp import InputSpec, MlpSpec, OutputSpec

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
                 for n_, nf in OUTPUTS.items()],
        out_kernel=glorot(w, f),
        out_bias=rng.normal(0, 0.01, f).astype(np.float32),
    )


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


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this script, FETCH_SIZE corrected as
    MI355X_MICROARCH.md prescribes for gfx950) -- a recorded measurement of the same launch, not
    collected live; None where the counters are uncalibrated or the file is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel_key]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def pmc_valu(kernel_key):
    """VALU issue-bound fraction of a remap kernel from the committed PMC passes (profiles/r01_pmc_mappm_valu.json), a
    recorded measurement like ``pmc_traffic``; None if absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_mappm_valu.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel_key]["valu_issue_bound_fraction"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(spec, budget_s=15.0):
    """The oracle (numpy float32, [sample, feature]) on a bounded sample of the same workload."""
    from oracle import mlp_np

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    n = 16384
    src = zc_inputs_numpy(np.random.default_rng(1), n)
    mlp_np.forward(spec, src)  # warm
    t0, reps = time.perf_counter(), 0
    while True:
        mlp_np.forward(spec, src)
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 100:
            break
    dt = time.perf_counter() - t0
    return {
        "value": n * reps / dt,
        "unit": "columns/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"{reps} passes over {n} of the 884736 C384 columns, numpy float32 oracle (BLAS threads = {threads})",
    }


def time_kernel(fn, steps, dev):
    from fv3net_amd.ops import HipTimer

    timer = HipTimer()
    timer.start(dev)
    for _ in range(steps):
        fn()
    timer.stop(dev)
    return timer.elapsed_ms() / steps


def secondary_benchmarks(dev, steps):
    """HBM-bound kernels of the coarse-graining path against the 8 TB/s roofline."""
    from fv3net_amd import ops

    out = []
    g = torch.Generator(device=dev).manual_seed(0)
    # C3072 -> C384 (f = 8) weighted_block_average of one 3-D float32 field, 2-D area weights
    for label, n, tiles in (("C3072->C384", 3072, 6), ("C384->C48", 384, 6)):
        try:
            obj = torch.rand((tiles, NZ, n, n), device=dev, generator=g) * 2000 - 1000
            area = torch.rand((tiles, n, n), device=dev, generator=g) * 0.5 + 0.5
            fn = lambda: ops.weighted_block_average(obj, area, 8)
            fn()
            torch.cuda.synchronize(dev)
            ms = time_kernel(fn, max(3, min(steps, 10)), dev)
            nel = obj.numel()
            alg_bytes = 4 * nel * (1 + 1 / 64) + 4 * nel / NZ
            out.append({
                "kernel": "weighted_block_average", "workload": f"{label} f=8, one [6,79,{n},{n}] f32 field, 2-D area weights",
                "ms": ms, "roofline": {"bound": "hbm", "achieved": alg_bytes / ms / 1e6, "peak": PEAK_HBM_GBPS,
                                       "unit": "GB/s", "frac": alg_bytes / ms / 1e6 / PEAK_HBM_GBPS,
                                       "traffic": pmc_traffic("wavg_block_kernel<float,float,8> C3072->C384") if n == 3072 else None},
            })
            del obj, area
        except torch.cuda.OutOfMemoryError:
            out.append({"kernel": "weighted_block_average", "workload": label, "error": "out of memory"})
        torch.cuda.empty_cache()
    # mappm at C384: 884 736 columns, km = kn = 79, native [tile, z, y, x] layout.  Two target grids:
    # the one the pipeline produces (fine delp ~ U(300, 1500), target = interface
    # pressures of its area-weighted f = 8 block mean, upsampled) and an independent random grid
    # (every lane's merge sweep takes a different path: the divergence worst case).
    n = 384
    delp = torch.rand((6, NZ, n, n), device=dev, generator=g) * 1200 + 300
    delp2 = torch.rand((6, NZ, n, n), device=dev, generator=g) * 1200 + 300
    area = torch.rand((6, n, n), device=dev, generator=g) * 0.5 + 0.5
    q = torch.rand((6, NZ, n, n), device=dev, generator=g) * 2000 - 1000
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    targets = (
        ("coarse-pressure target (config 3)", ops.pressure_at_interface(
            ops.block_upsample(ops.weighted_block_average(delp, area, 8), 8), 300.0, 1)),
        ("independent random target", ops.pressure_at_interface(delp2, 300.0, 1)),
    )
    ncol = 6 * n * n
    alg_bytes = ncol * 1272.0
    for label, pe2 in targets:
        fn = lambda: ops.mappm(pe1, q, pe2, z_axis=1)
        fn()
        torch.cuda.synchronize(dev)
        ms = time_kernel(fn, max(3, min(steps, 10)), dev)
        out.append({
            "kernel": "mappm", "workload": f"C384 884736 columns, km=kn=79, iv=1 kord=1, [tile,z,y,x] f32, {label}",
            "ms": ms, "columns_per_s": ncol / ms * 1e3,
            "roofline": {"bound": "hbm", "achieved": alg_bytes / ms / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": alg_bytes / ms / 1e6 / PEAK_HBM_GBPS, "traffic": None,
                         "note": "VALU-issue-bound (bit-exact IEEE divisions): see valu_issue_bound_frac",
                         "valu_issue_bound_frac": pmc_valu("mappm_merge_kernel<float>")},
        })
    # the same remap for 4 fields that share their pressures (one fv_core / tracer group of the pipeline)
    qs = [q] + [torch.rand((6, NZ, n, n), device=dev, generator=g) * 2000 - 1000 for _ in range(3)]
    pe2 = targets[0][1]
    fn = lambda: ops.mappm_multi(pe1, qs, pe2, z_axis=1)
    fn()
    torch.cuda.synchronize(dev)
    ms = time_kernel(fn, max(3, min(steps, 10)), dev)
    alg_bytes = ncol * (160 * 4 + 4 * 158 * 4.0)
    out.append({
        "kernel": "mappm_multi (4 fields per sweep)", "workload": "C384 884736 columns x 4 fields sharing pe1/pe2, km=kn=79, "
        "iv=1 kord=1, coarse-pressure target (config 3)", "ms": ms, "columns_per_s": 4 * ncol / ms * 1e3,
        "roofline": {"bound": "hbm", "achieved": alg_bytes / ms / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                     "frac": alg_bytes / ms / 1e6 / PEAK_HBM_GBPS, "traffic": None,
                     "valu_issue_bound_frac": pmc_valu("mappm_merge_multi_kernel<float,4>")},
    })
    del qs
    # (the secondary workloads never cost the headline line: a failure is recorded in place of the numbers)
    for fn, fn_args in ((dense_local_benchmark, (dev, steps)), (streaming_benchmark, (dev,)), (restart_pipeline_benchmark, (dev,))):
        try:
            out.extend(fn(*fn_args))
        except Exception as err:  # noqa: BLE001
            out.append({"kernel": fn.__name__, "error": f"{type(err).__name__}: {err}"})
        torch.cuda.empty_cache()
    return out


def restart_pipeline_benchmark(dev, n=384, f=8, reps=3):
    """BASELINE configs[2] end to end: the three restart coarse-graining pipelines (vcm coarsen_restarts_on_sigma /
    _on_pressure / _via_blended_method, all four restart categories, 'complex' surface method) C384 -> C48 on float64
    restarts resident in HBM, through the drop-in Python API.  Wall time of a whole pipeline call."""
    from fv3net_amd.cubedsphere import (coarsen_restarts_on_pressure, coarsen_restarts_on_sigma,
                                        coarsen_restarts_via_blended_method)
    from fv3net_amd.xr_compat import DataArray, Dataset

    g = torch.Generator(device=dev).manual_seed(3)
    u = lambda lo, hi, *shape: torch.rand(shape, device=dev, generator=g, dtype=torch.float64) * (hi - lo) + lo
    zc = ["tile", "Time", "zaxis_1", "yaxis_2", "xaxis_1"]
    core = Dataset({
        "u": DataArray(u(-30, 30, 6, 1, NZ, n + 1, n), dims=["tile", "Time", "zaxis_1", "yaxis_1", "xaxis_1"]),
        "v": DataArray(u(-30, 30, 6, 1, NZ, n, n + 1), dims=["tile", "Time", "zaxis_1", "yaxis_2", "xaxis_2"]),
        **{k: DataArray(u(lo, hi, 6, 1, NZ, n, n), dims=zc) for k, (lo, hi) in
           {"W": (-1, 1), "T": (200, 300), "delp": (300, 1500), "DZ": (-500, -50), "ua": (-30, 30), "va": (-30, 30)}.items()},
        "phis": DataArray(u(0, 1e4, 6, 1, n, n), dims=["tile", "Time", "yaxis_2", "xaxis_1"]),
    })
    tracers = ["sphum", "liq_wat", "rainwat", "ice_wat", "snowwat", "graupel", "o3mr", "sgs_tke", "cld_amt"]
    tracer = Dataset({k: DataArray(u(0, 0.02, 6, 1, NZ, n, n), dims=["tile", "Time", "zaxis_1", "yaxis_1", "xaxis_1"]) for k in tracers})
    srf = Dataset({k: DataArray(u(-10, 10, 6, 1, n, n), dims=["tile", "Time", "yaxis_1", "xaxis_1"]) for k in ("u_srf", "v_srf")})
    # surface data: the variables, dims and value ranges of the reference's regression schema (tests/golden metadata)
    with np.load(os.path.join(ROOT, "tests", "golden", "coarsen_restarts_reference.npz")) as z:
        meta = json.loads(bytes(z["meta_json"]).decode())
    rng = np.random.default_rng(0)
    sfc = Dataset()
    for name, info in meta["inputs"]["sfc_data"].items():
        lo, hi = meta["ranges"].get(name, meta["default_range"])
        shape = list(info["shape"][:-2]) + [n, n]
        sfc[name] = DataArray(torch.from_numpy(rng.uniform(lo, hi, shape).astype(info["dtype"])).to(dev), dims=info["dims"])
    grid = Dataset({"area": DataArray(u(0.5, 1, 6, n, n).float(), dims=["tile", "grid_yt", "grid_xt"]),
                    "dx": DataArray(u(0.5, 1, 6, n + 1, n).float(), dims=["tile", "grid_y", "grid_xt"]),
                    "dy": DataArray(u(0.5, 1, 6, n, n + 1).float(), dims=["tile", "grid_yt", "grid_x"])})
    restarts = {"fv_core.res": core, "fv_tracer.res": tracer, "fv_srf_wnd.res": srf, "sfc_data": sfc}
    nbytes = sum(v.data.numel() * v.data.element_size() for ds in restarts.values() for v in ds.values())
    out = []
    for label, fn in (("coarsen_restarts_on_sigma", lambda: coarsen_restarts_on_sigma(f, grid, restarts, coarsen_agrid_winds=True)),
                      ("coarsen_restarts_on_pressure", lambda: coarsen_restarts_on_pressure(f, grid, 300.0, restarts, coarsen_agrid_winds=True)),
                      ("coarsen_restarts_via_blended_method",
                       lambda: coarsen_restarts_via_blended_method(f, grid, 300.0, restarts, coarsen_agrid_winds=True))):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / reps * 1e3
        out.append({"kernel": f"{label} (whole pipeline, Python API)",
                    "workload": f"C{n}->C{n // f}, 4 restart categories, {nbytes / 1e9:.2f} GB of float64 restarts in HBM",
                    "ms": ms, "roofline": {"bound": "hbm", "achieved": nbytes / ms / 1e6, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                           "frac": nbytes / ms / 1e6 / PEAK_HBM_GBPS, "traffic": None,
                                           "note": "input bytes / wall time; the pressure-level pipelines are bound by the VALU-limited remap"}})
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


def dense_local_benchmark(dev, steps):
    """The reference's production gscond emulator (projects/microphysics/configs/models/gscond.yaml: "dense-local",
    12 inputs at a point -> 2 x 256 -> 2 outputs, temperature-conditional un-scaling + Difference outputs) over one
    C384 snapshot: 79 x 884 736 points.  MFMA-bound like the headline kernel, plus two HBM-bound passes."""
    from fv3net_amd.local_mlp import ConditionalScale, LocalInput, LocalMlpModel, LocalMlpSpec, LocalOutput

    rng = np.random.default_rng(0)
    ncol, width = 6 * 384 * 384, 256
    g = torch.Generator(device=dev).manual_seed(2)
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
    raw = [(n, n, "none", 0.0) for n in st] + [(k, v[0], "log", v[1]) for k, v in LOG_FIELDS.items()]
    inputs = [LocalInput(name, source, tr, eps, center=rng.normal(0, 1, NZ).astype(np.float32) if st[source].dim() == 2 else
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
    model = LocalMlpModel(spec, device=dev)
    fn = lambda: model.predict(st)
    fn()
    torch.cuda.synchronize(dev)
    ms = time_kernel(fn, max(3, min(steps, 5)), dev)
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
    # the production precpd emulator (configs/models/precpd.yaml: "rnn-v1-shared-weights", 2 stacked SimpleRNN(256) over the
    # 79 levels, 4 outputs): 79 x 2 launches of the fused kernel, the states ping-ponging in HBM
    from fv3net_amd.local_mlp import RnnLayer, RnnModel, RnnSpec

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
    rmodel = RnnModel(rspec, device=dev)
    fn = lambda: rmodel.predict(st)
    fn()
    torch.cuda.synchronize(dev)
    ms = time_kernel(fn, 2, dev)
    flops = 2 * (k * width + 3 * width * width + width * 4)  # the recurrences themselves, without the identity output layers
    achieved = flops * NZ * ncol / (ms * 1e-3) / 1e12
    out.append({
        "kernel": "RNN emulator (local_pack + 79 x 2 mlp_fused_kernel launches + local_unpack)",
        "workload": f"C384 precpd regressor: {k} inputs -> 2 x SimpleRNN({width}) over 79 levels -> 4 outputs, {ncol} columns, float32 state",
        "ms": ms, "columns_per_s": ncol / ms * 1e3, "flops_per_point": flops,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None},
    })
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a one-GPU box: FV3_BENCH_REHEARSAL=1 puts every rank on cuda:0 and rendezvouses over gloo
    # (two ranks cannot share a device under RCCL).  Real runs: one rank per GPU, backend nccl = RCCL over xGMI.
    rehearsal = os.environ.get("FV3_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    assert world == max(args.gpus, 1) or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from fv3net_amd import _lib, ops
    from fv3net_amd.mlp import MlpModel

    _lib.load()
    spec = zc_spec(0)
    model = MlpModel(spec, device=dev)
    ncol = 6 * 384 * 384
    src = zc_inputs_device(dev, ncol, seed=1000 + rank)

    def step():
        return model.predict(src)

    for _ in range(args.warmup):
        step()

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
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_cols = ncol * world * args.steps
        flops = model.flops_per_sample
        achieved = flops * ncol / (kernel_ms * 1e-3) / 1e12
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
                            "(BASELINE configs[1]): K=711 -> 256 -> 256 -> 396, float32 [feature, sample] inputs in HBM",
                "columns_per_gpu": ncol,
                "flops_per_column": flops,
                "parallelism": f"snapshot/tile sharding over {world} GPU(s), no collective on the data path",
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "traffic": pmc_traffic("mlp_fused_kernel<8,false,true,false,false,false>"),
                "kernel": "mlp_fused_kernel<8,false,true,false,false,false>",
                "kernel_ms": kernel_ms,
            },
        }
        if world == 1 and not args.no_cpu:
            try:
                line["cpu_baseline"] = cpu_baseline(spec)
            except Exception as err:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "columns/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {type(err).__name__}: {err}"}
        if world == 1 and not args.no_secondary:
            del src
            torch.cuda.empty_cache()
            try:
                line["secondary"] = secondary_benchmarks(dev, args.steps)
            except Exception as err:  # noqa: BLE001  (never lose the headline line to a secondary workload)
                line["secondary"] = [{"kernel": "secondary_benchmarks", "error": f"{type(err).__name__}: {err}"}]
        info = ops.device_info()
        line["device"] = {"name": info["name"], "arch": info["arch"], "compute_units": info["compute_units"]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
