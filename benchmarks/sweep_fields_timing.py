"""The coarse-target sweep at C384 with 1 .. 6 float64 fields per launch and the 13 fields of the pipelines (6 + 4 + 3), both
arithmetic modes and data sets: what a field costs beside what a sweep costs.  `python benchmarks/sweep_fields_timing.py`"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fv3net_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, NZ, F = 384, 79, 8
out = {}
for noise in (1.0, 0.1):
    g = torch.Generator(device=dev).manual_seed(0)
    delp = 900 + (torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) - 0.5) * 1200 * noise
    area = torch.rand((6, n, n), device=dev, generator=g, dtype=torch.float64) * 0.5 + 0.5
    pe1 = ops.pressure_at_interface(delp, 300.0, 1)
    pe2c = ops.pressure_at_interface(ops.weighted_block_average(delp, area, F), 300.0, 1)
    qs = [torch.rand((6, NZ, n, n), device=dev, generator=g, dtype=torch.float64) * 2000 - 1000 for _ in range(13)]
    for arith in ("exact", "fast"):
        row = {}
        for nf in (1, 2, 3, 4, 5, 6, 13):
            run = lambda: ops.mappm_multi_coarse_target(pe1, qs[:nf], pe2c, F, z_axis=1, arith=arith)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                run()
            ev[1].record()
            torch.cuda.synchronize()
            row[nf] = round(ev[0].elapsed_time(ev[1]) / 10, 4)
        out[f"noise {noise}, {arith}"] = row
    del delp, pe1, pe2c, qs
    torch.cuda.empty_cache()
print(json.dumps(out))
