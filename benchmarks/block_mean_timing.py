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
ap.add_argument("--terrain", action="store_true", help="hybrid sigma-pressure thicknesses over a surface pressure with mountains instead of --noise")
ap.add_argument("--fields", default="1,4")
ap.add_argument("--dtype", default="f64,f32")
ap.add_argument("--label", default=os.environ.get("FV3HIP_LIBRARY", "in-tree"))
args = ap.parse_args()

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, NZ, F = args.n, 79, 8
delp = 900 + (torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * args.noise
if args.terrain:
    # A hybrid sigma-pressure column as the model has it -- delp(k) = dak(k) + dbk(k) ps -- over a surface pressure with terrain:
    # 40 Gaussian mountains per tile, 2 to 6 cells wide, up to 400 hPa deep (the Andes / Tibet at C384's 25 km), on 1000 hPa
    # with +-10 hPa of smooth weather.  Layers: pure pressure above 250 hPa, terrain-following below, thinnest at the surface.
    import math
    s01 = torch.linspace(0, 1, NZ + 1, device=dev, dtype=torch.float64)
    p_ref = 300.0 + (1.0e5 - 300.0) * torch.sin(0.5 * math.pi * s01) ** 1.5            # interfaces at ps = 1000 hPa: thin layers at both ends
    sig = torch.clamp((p_ref - 2.5e4) / 7.5e4, min=0) ** 1.3                           # bk: 0 above 250 hPa, 1 at the surface
    ak = p_ref - sig * 1.0e5                                                           # pe(ps) = ak + bk ps = p_ref + bk (ps - 1000 hPa)
    yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float64), torch.arange(n, device=dev, dtype=torch.float64), indexing="ij")
    ps = torch.full((6, n, n), 1.0e5, device=dev, dtype=torch.float64)
    cpu = torch.Generator().manual_seed(1)
    for t in range(6):
        for _ in range(40):
            cy, cx, w, depth = (torch.rand(4, generator=cpu).tolist())
            ps[t] -= (depth * 4.0e4) * torch.exp(-(((yy - cy * n) ** 2 + (xx - cx * n) ** 2) / (2 * (2 + 4 * w) ** 2)))
        ps[t] += 1.0e3 * torch.sin(2 * math.pi * (yy / n * 1.5 + t / 6)) * torch.cos(2 * math.pi * xx / n * 2.5)
    ps.clamp_(min=5.0e4)
    pe = ak.view(1, NZ + 1, 1, 1) + sig.view(1, NZ + 1, 1, 1) * ps.view(6, 1, n, n)
    delp = (pe[:, 1:] - pe[:, :-1]).contiguous()
    assert float(delp.min()) > 0
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


out = {"label": args.label, "noise": "terrain" if args.terrain else args.noise, "columns": ncol, "cases": {}}
if True:   # how many of the blocks run out of ring (the count the adaptive route looks at)
    counters = torch.zeros(4, dtype=torch.int32).pin_memory()
    ops.mappm_block_mean(pe1, qs, pe2c, area, arith="exact", counters=counters)
    torch.cuda.synchronize()
    out["blocks_that_gave_up_summing"] = [int(counters[2]), ncol // 64]
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
