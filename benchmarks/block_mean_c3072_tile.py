"""The fused remap + masked block mean against the three launches on one C3072 tile (9.4 M columns: nine launches of 2^20 columns each),
4 float64 fields, smooth and iid thicknesses, both arithmetics; checks bit-identity at that size."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from fv3net_amd import ops
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
n, NZ, F = 3072, 79, 8
for noise in (0.1, 1.0):
    delp = 900 + (torch.rand((1, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * noise
    area = (torch.rand((1, n, n), device=dev, generator=g, dtype=torch.float64) * 0.5 + 0.5).float()
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, F), 300.0, 1)
    del delp
    qs = [torch.rand((1, NZ, n, n), device=dev, generator=g, dtype=torch.float64) * 2000 - 1000 for _ in range(4)]
    for arith in ("exact", "fast"):
        def unfused():
            q2 = ops.mappm_multi_coarse_target(pe1, qs, pe2c, F, z_axis=1, arith=arith)
            return ops.weighted_block_average_multi(q2, ops.mask_weights(area, pe2c, pe1, 1, coarse_factor=F), F)
        fused = lambda: ops.mappm_block_mean(pe1, qs, pe2c, area, arith=arith)
        res = {}
        for name, fn in (("fused", fused), ("three", unfused)):
            for _ in range(2): r = fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): r = fn()
            torch.cuda.synchronize(); res[name] = ((time.perf_counter() - t0) / 3 * 1e3, r)
        same = all(torch.equal(a, b) for a, b in zip(res["fused"][1], res["three"][1]))
        print(f"noise {noise} {arith}: fused {res['fused'][0]:.2f} ms, three launches {res['three'][0]:.2f} ms, bit-identical {same}")
    del pe1, pe2c, qs; torch.cuda.empty_cache()
