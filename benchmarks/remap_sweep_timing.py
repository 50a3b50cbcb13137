"""Times the instantiations of mappm_sweep_kernel the restart pipelines launch (C384, km = kn = 79, coarse-pressure target
read through the (y // 8, x // 8) map) with HIP events and checks fast against exact.

    python benchmarks/remap_sweep_timing.py [--reps 20] [--label NAME] [--only f64x4,f32x1]

One JSON line per run; FV3HIP_LIBRARY selects an alternative build of the library (A/B of kernel variants on one box).
Algorithmic bytes per column: pe1 (km + 1) e + NF km e + NF kn 4 (+ the coarse pe2, 1/64 of a fine plane per level)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fv3net_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--label", default=os.environ.get("FV3HIP_LIBRARY", "in-tree"))
ap.add_argument("--only", default="")
ap.add_argument("--n", type=int, default=384)
ap.add_argument("--noise", type=float, default=1.0, help="1: delp ~ U(300, 1500) iid per cell (BASELINE configs[2]); < 1: that fraction of the spread around 900 Pa (lanes of a wave stay aligned)")
args = ap.parse_args()

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, NZ, F = args.n, 79, 8
delp = 900 + (torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * args.noise
area = torch.rand((6, n, n), device=dev, generator=g, dtype=torch.float64) * 0.5 + 0.5
pe1 = ops.pressure_at_interface(delp, 300.0, 1)
pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, F), 300.0, 1)
qs = [torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) * 2000 - 1000 for _ in range(4)]
ncol = 6 * n * n
out = {"label": args.label, "noise": args.noise, "columns": ncol, "cases": {}}
only = set(filter(None, args.only.split(",")))
for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
    p1, p2, fs = pe1.to(dt), pe2c.to(dt), [q.to(dt) for q in qs]
    e = 4 if dt == torch.float32 else 8
    for nf in (1, 4):
        name = f"{dname}x{nf}"
        if only and name not in only:
            continue
        res = {}
        for arith in ("fast", "exact"):
            run = lambda: ops.mappm_multi_coarse_target(p1, fs[:nf], p2, F, z_axis=1, arith=arith)
            for _ in range(3):
                r = run()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(args.reps):
                r = run()
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / args.reps
            nbytes = ncol * ((NZ + 1) * e + nf * NZ * e + nf * NZ * 4) + p2.numel() * e
            res[arith] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "of_hbm": round(nbytes / ms / 1e6 / 8000, 3)}
            res[arith + "_out"] = r
        d = max(float((a - b).abs().max()) for a, b in zip(res.pop("fast_out"), res.pop("exact_out")))
        res["max_abs_fast_minus_exact"] = d
        out["cases"][name] = res
print(json.dumps(out))
