"""Single-field vs multi-field remap at C384 (884 736 columns, km = kn = 79), both arithmetic modes, on the pipeline's own
target grid and on an independent random one.  Run from the repo root on the GPU box."""
import sys, torch
sys.path.insert(0, '.')
from fv3net_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n, NZ = 384, 79
delp = torch.rand((6, NZ, n, n), device=dev, generator=g) * 1200 + 300
delp2 = torch.rand((6, NZ, n, n), device=dev, generator=g) * 1200 + 300
area = torch.rand((6, n, n), device=dev, generator=g) * 0.5 + 0.5
pe1 = ops.pressure_at_interface(delp, 300.0, 1)
pe2 = ops.pressure_at_interface(ops.block_upsample(ops.weighted_block_average(delp, area, 8), 8), 300.0, 1)
pe2r = ops.pressure_at_interface(delp2, 300.0, 1)
qs = [torch.rand((6, NZ, n, n), device=dev, generator=g) * 2000 - 1000 for _ in range(9)]
ncol = 6 * n * n
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    tm = ops.HipTimer(); tm.start(dev)
    for _ in range(reps): fn()
    tm.stop(dev); return tm.elapsed_ms() / reps
for arith in ("exact", "fast"):
    for label, p2 in (("coarse-pressure target", pe2), ("random target", pe2r)):
        ms = t(lambda: ops.mappm(pe1, qs[0], p2, z_axis=1, arith=arith))
        print("%-5s single field, %s: %.3f ms = %.2f TB/s algorithmic (1272 B/column)" % (arith, label, ms, ncol * 1272 / ms / 1e9))
    for nf in (2, 3, 4, 8, 9):
        ms = t(lambda: ops.mappm_multi(pe1, qs[:nf], pe2, z_axis=1, arith=arith))
        print("%-5s multi %d fields: %.3f ms (%.3f per field) = %.2f TB/s algorithmic" % (arith, nf, ms, ms / nf, ncol * (640 + nf * 632) / ms / 1e9))
a = ops.mappm_multi(pe1, qs, pe2, z_axis=1, arith="exact")
b = [ops.mappm(pe1, q, pe2, z_axis=1, arith="exact") for q in qs]
print("exact: multi identical to single:", all(torch.equal(x, y) for x, y in zip(a, b)))
c = ops.mappm_multi(pe1, qs, pe2, z_axis=1, arith="fast")
print("fast vs exact: max |diff| = %.3e (values up to 1000)" % max(float((x - y).abs().max()) for x, y in zip(a, c)))
