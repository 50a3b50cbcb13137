"""Times the fused remap + masked block mean (fv3hip_mappm_block_mean) against the three launches it replaces (coarse-target
remap, masked weights, weighted block average) on the pipelines' shape: C384 -> C48, km = kn = 79, float64 restarts.

    python benchmarks/block_mean_timing.py [--reps 20] [--noise 1.0] [--fields 4] [--dtype f64]

One JSON line.  Algorithmic bytes of the fused call: pe1 (km + 1) e + NF km e per fine column (+ area 4 B, + the coarse
tables and means, 1/64 of a plane per level); the unfused route adds NF kn 4 written and read again and the masked weights."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fv3net_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--n", type=int, default=384)
ap.add_argument("--noise", type=float, default=1.0)
ap.add_argument("--fields", default="1,4")
ap.add_argument("--dtype", default="f64,f32")
ap.add_argument("--label", default=os.environ.get("FV3HIP_LIBRARY", "in-tree"))
args = ap.parse_args()

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, NZ, F = args.n, 79, 8
delp = 900 + (torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * args.noise
area = (torch.rand((6, n, n), device=dev, generator=g, dtype=torch.float64) * 0.5 + 0.5).float()
pe1 = ops.pressure_at_interface(delp, 300.0, 1)
pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, F), 300.0, 1)
qs = [torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) * 2000 - 1000 for _ in range(4)]
ncol = 6 * n * n


def timed(run, reps):
    for _ in range(3):
        r = run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        r = run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps, r


def unfused(p1, fs, p2, arith):
    q2 = ops.mappm_multi_coarse_target(p1, fs, p2, F, z_axis=1, arith=arith)
    mw = ops.mask_weights(area, p2, p1, 1, coarse_factor=F)
    return ops.weighted_block_average_multi(q2, mw, F) if len(q2) > 1 else [ops.weighted_block_average(q2[0], mw, F)]


out = {"label": args.label, "noise": args.noise, "columns": ncol, "cases": {}}
for dname in args.dtype.split(","):
    dt = torch.float32 if dname == "f32" else torch.float64
    p1, p2, fs = pe1.to(dt), pe2c.to(dt), [q.to(dt) for q in qs]
    e = 4 if dt == torch.float32 else 8
    for nf in [int(x) for x in args.fields.split(",")]:
        res = {}
        for arith in ("fast", "exact"):
            ms_f, rf = timed(lambda: ops.mappm_block_mean(p1, fs[:nf], p2, area, arith=arith), args.reps)
            ms_u, ru = timed(lambda: unfused(p1, fs[:nf], p2, arith), args.reps)
            same = all(torch.equal(a, b) for a, b in zip(rf, ru))
            nbytes = ncol * ((NZ + 1) * e + nf * NZ * e + 4) + (2 * p2.numel() * e + nf * NZ * ncol // 64 * 4)
            res[arith] = {"fused_ms": round(ms_f, 4), "unfused_ms": round(ms_u, 4), "bit_identical": same,
                          "fused_GBps": round(nbytes / ms_f / 1e6, 1), "fused_of_hbm": round(nbytes / ms_f / 1e6 / 8000, 3)}
        out["cases"][f"{dname}x{nf}"] = res
print(json.dumps(out))
