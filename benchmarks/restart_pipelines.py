"""Whole restart pipelines at C384 -> C48 on device-resident inputs (float64 restarts, as the model writes them),
timed end to end; run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from fv3net_amd.cubedsphere import coarsen_restarts_on_pressure, coarsen_restarts_on_sigma, coarsen_restarts_via_blended_method
from fv3net_amd.xr_compat import DataArray, Dataset

dev = torch.device('cuda:0')
n, nz, f = int(sys.argv[1]) if len(sys.argv) > 1 else 384, 79, 8
dt = torch.float64
g = torch.Generator(device=dev).manual_seed(0)
def u(lo, hi, *shape): return torch.rand(shape, device=dev, generator=g, dtype=dt) * (hi - lo) + lo
def da(t, dims): return DataArray(t, dims=dims)
core = Dataset({
    "u": da(u(-30, 30, 6, 1, nz, n + 1, n), ["tile", "Time", "zaxis_1", "yaxis_1", "xaxis_1"]),
    "v": da(u(-30, 30, 6, 1, nz, n, n + 1), ["tile", "Time", "zaxis_1", "yaxis_2", "xaxis_2"]),
    **{k: da(u(lo, hi, 6, 1, nz, n, n), ["tile", "Time", "zaxis_1", "yaxis_2", "xaxis_1"])
       for k, (lo, hi) in {"W": (-1, 1), "T": (200, 300), "delp": (300, 1500), "DZ": (-500, -50), "ua": (-30, 30), "va": (-30, 30)}.items()},
    "phis": da(u(0, 1e4, 6, 1, n, n), ["tile", "Time", "yaxis_2", "xaxis_1"]),
})
tracers = ["sphum", "liq_wat", "rainwat", "ice_wat", "snowwat", "graupel", "o3mr", "sgs_tke", "cld_amt"]
tracer = Dataset({k: da(u(0, 0.02, 6, 1, nz, n, n), ["tile", "Time", "zaxis_1", "yaxis_1", "xaxis_1"]) for k in tracers})
srf = Dataset({k: da(u(-10, 10, 6, 1, n, n), ["tile", "Time", "yaxis_1", "xaxis_1"]) for k in ("u_srf", "v_srf")})
import json, os
sys.path.insert(0, 'tests')  # (the surface-data schema lives with the fixtures)
import coarsen_restarts_cases as cases
meta, _ = cases.load()
sfc = Dataset()
rng = np.random.default_rng(0)
for name, info in meta["inputs"]["sfc_data"].items():
    lo, hi = meta["ranges"].get(name, meta["default_range"])
    shape = list(info["shape"][:-2]) + [n, n]
    sfc[name] = da(torch.from_numpy(rng.uniform(lo, hi, shape).astype(info["dtype"])).to(dev), info["dims"])
grid = Dataset({"area": da(u(0.5, 1, 6, n, n).float(), ["tile", "grid_yt", "grid_xt"]),
                "dx": da(u(0.5, 1, 6, n + 1, n).float(), ["tile", "grid_y", "grid_xt"]),
                "dy": da(u(0.5, 1, 6, n, n + 1).float(), ["tile", "grid_yt", "grid_x"])})
restarts = {"fv_core.res": core, "fv_tracer.res": tracer, "fv_srf_wnd.res": srf, "sfc_data": sfc}
nbytes = sum(v.data.numel() * v.data.element_size() for ds in restarts.values() for v in ds.values())
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["sigma", "pressure", "blended"]
for label, fn in (("sigma", lambda: coarsen_restarts_on_sigma(f, grid, restarts, coarsen_agrid_winds=True)),
                  ("pressure", lambda: coarsen_restarts_on_pressure(f, grid, 300.0, restarts, coarsen_agrid_winds=True)),
                  ("blended", lambda: coarsen_restarts_via_blended_method(f, grid, 300.0, restarts, coarsen_agrid_winds=True))):
    if label not in which:
        continue
    out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    assert isinstance(out["fv_core.res"]["T"].data, torch.Tensor) and out["fv_core.res"]["T"].data.is_cuda
    print(f"{label}: C{n}->C{n // f}, {nbytes / 1e9:.2f} GB of restarts: {ms:.1f} ms wall = {nbytes / ms / 1e6:.0f} GB/s of input", flush=True)
