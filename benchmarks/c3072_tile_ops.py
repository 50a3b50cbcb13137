"""The other coarse-graining ops on one C3072 tile (the per-GPU share of BASELINE configs[4]): each timed, and checked against
the same op on the tile's first 256 x 256 corner (blocks never straddle it) -- sizes far beyond the unit tests'."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fv3net_amd import ops

dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
n, nz, f, c = 3072, 79, 8, 256
r = lambda *shape, dtype=torch.float64: torch.rand(shape, device=dev, generator=g, dtype=dtype)


def timed(fn, reps=3):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def same(a, b):
    return bool(torch.equal(a, b) or torch.allclose(a, b, rtol=0, atol=0, equal_nan=True))


res = {}
delp, area = r(1, nz, n, n) * 1200 + 300, (r(1, n, n) * 0.5 + 0.5).float()
fields = [r(1, nz, n, n) for _ in range(4)]
ms, out = timed(lambda: ops.mass_weighted_block_average(fields, delp, area, f))
ref = ops.mass_weighted_block_average([q[..., :c, :c].contiguous() for q in fields], delp[..., :c, :c].contiguous(), area[..., :c, :c].contiguous(), f)
res["mass_weighted_block_average, 4 float64 fields"] = {"ms": round(ms, 2), "GBps": round(5 * delp.numel() * 8 / ms / 1e6), "corner_identical": all(same(a[..., :c // f, :c // f], b) for a, b in zip(out, ref))}
del fields
u, dx = r(1, nz, n + 1, n), (r(1, n + 1, n) * 0.5 + 0.5).float()
ms, out = timed(lambda: ops.edge_weighted_block_average(u, dx, f, "x"))
ref = ops.edge_weighted_block_average(u[..., :c + 1, :c].contiguous(), dx[..., :c + 1, :c].contiguous(), f, "x")
res["edge_weighted_block_average (u, dx)"] = {"ms": round(ms, 2), "corner_identical": same(out[..., :c // f + 1, :c // f], ref)}
del u
pe, out = timed(lambda: ops.pressure_at_interface(delp, 300.0, 1))
res["pressure_at_interface"] = {"ms": round(pe, 2), "GBps": round(2 * delp.numel() * 8 / pe / 1e6), "corner_identical": same(out[..., :c, :c], ops.pressure_at_interface(delp[..., :c, :c].contiguous(), 300.0, 1))}
cat = torch.floor(r(1, n, n) * 20)
for method in ("median", "mode", "max"):
    ms, out = timed(lambda: ops.block_reduce(cat, (f, f), None, method))
    res[f"block_reduce {method} (2-D)"] = {"ms": round(ms, 3), "corner_identical": same(out[..., :c // f, :c // f], ops.block_reduce(cat[..., :c, :c].contiguous(), (f, f), None, method))}
coarse = r(1, nz, n // f, n // f)
ms, out = timed(lambda: ops.block_upsample(coarse, f))
res["block_upsample"] = {"ms": round(ms, 2), "GBps": round(out.numel() * 8 / ms / 1e6), "corner_identical": same(out[..., :c, :c], ops.block_upsample(coarse[..., :c // f, :c // f].contiguous(), f))}
print(json.dumps(res, indent=1))
print("peak GB", round(torch.cuda.max_memory_allocated() / 1e9, 1))
