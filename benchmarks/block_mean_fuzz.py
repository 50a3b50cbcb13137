"""Randomised comparison of the fused remap + block mean with the three launches it replaces: shapes, field counts, dtypes,
spreads of the thicknesses (lanes far apart -> spill mode), mask kinds, layer counts, NaNs -- bit for bit.
`python benchmarks/block_mean_fuzz.py [cases] [seed]`"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fv3net_amd import ops

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    nt, nz = int(rng.integers(1, 4)), int(rng.integers(8, 64))
    ny, nx = 8 * int(rng.integers(1, 7)), 8 * int(rng.integers(1, 9))
    nf = int(rng.integers(1, 7))
    spread = float(rng.choice([0.05, 0.3, 1.0, 2.0]))
    dtype = rng.choice([np.float32, np.float64])
    extrapolate, arith = bool(rng.integers(0, 2)), str(rng.choice(["exact", "fast"]))
    kn = nz if rng.uniform() < 0.7 else int(rng.integers(4, 100))
    shape = (nt, nz, ny, nx)
    delp = np.maximum(rng.uniform(300, 1500, (nt, nz, 1, 1)) + spread * (rng.uniform(300, 1500, shape) - 900.0), 5.0)
    area = rng.uniform(0.5, 1.0, (nt, ny, nx)).astype(np.float32)
    fields = [rng.uniform(-1000, 1000, shape) for _ in range(nf)]
    if rng.uniform() < 0.15:
        fields[0][tuple(rng.integers(0, s) for s in shape)] = np.nan
    if rng.uniform() < 0.1:
        area[tuple(rng.integers(0, s) for s in area.shape)] = np.nan
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(dtype))).to(dev)
    delp_t, area_t = t(delp), torch.from_numpy(area).to(dev)
    delp_c = ops.weighted_block_average(delp_t, torch.ones_like(area_t), 8)
    if kn != nz:
        frac = torch.from_numpy(rng.dirichlet(np.ones(kn) * 8).astype(dtype)).to(dev).reshape(1, kn, 1, 1)
        delp_c = delp_c.sum(dim=1, keepdim=True) * frac
    pe1, pe2c = ops.pressure_at_interface(delp_t, 300.0, 1), ops.pressure_at_interface(delp_c, 300.0, 1)
    pfull = ops.pressure_at_midpoint_log(delp_c, 300.0, 1)
    qs = [t(f) for f in fields]
    got = ops.mappm_block_mean(pe1, qs, pe2c, area_t, level_coarse=pfull if extrapolate else None, arith=arith)
    q2 = ops.mappm_multi_coarse_target(pe1, qs, pe2c, 8, arith=arith)
    if kn == nz:
        mw = ops.mask_weights(area_t, pfull if extrapolate else pe2c, pe1, 1, extrapolate=extrapolate, coarse_factor=8)
    else:
        level = (pfull if extrapolate else pe2c[:, 1:]).repeat_interleave(8, dim=-2).repeat_interleave(8, dim=-1)
        mw = torch.where(level < pe1[:, -1:], area_t.unsqueeze(1), torch.zeros((), dtype=area_t.dtype, device=dev)).contiguous()
    want = ops.weighted_block_average_multi(q2, mw, 8) if nf > 1 else [ops.weighted_block_average(q2[0], mw, 8)]
    ok = got is not None and all(torch.equal(a, b) or bool(((a == b) | (a.isnan() & b.isnan())).all()) for a, b in zip(got, want))
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, nt=nt, nz=nz, ny=ny, nx=nx, nf=nf, spread=spread, dtype=dtype.__name__, extrapolate=extrapolate, arith=arith, kn=kn))
print(f"{n_cases} cases, {bad} mismatches")
