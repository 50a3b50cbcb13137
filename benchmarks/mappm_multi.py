import sys, torch
sys.path.insert(0, '.')
from fv3net_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n, NZ = 384, 79
delp = torch.rand((6, NZ, n, n), device=dev, generator=g) * 1200 + 300
area = torch.rand((6, n, n), device=dev, generator=g) * 0.5 + 0.5
pe1 = ops.pressure_at_interface(delp, 300.0, 1)
pe2 = ops.pressure_at_interface(ops.block_upsample(ops.weighted_block_average(delp, area, 8), 8), 300.0, 1)
qs = [torch.rand((6, NZ, n, n), device=dev, generator=g) * 2000 - 1000 for _ in range(9)]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    tm = ops.HipTimer(); tm.start(dev)
    for _ in range(reps): fn()
    tm.stop(dev); return tm.elapsed_ms() / reps
single = t(lambda: [ops.mappm(pe1, q, pe2, z_axis=1) for q in qs])
print("9 single-field calls: %.3f ms (%.3f per field)" % (single, single / 9))
for nf in (1, 2, 3, 4, 8, 9):
    ms = t(lambda: ops.mappm_multi(pe1, qs[:nf], pe2, z_axis=1))
    print("multi %d fields: %.3f ms (%.3f per field)" % (nf, ms, ms / nf))
a = ops.mappm_multi(pe1, qs, pe2, z_axis=1)
b = [ops.mappm(pe1, q, pe2, z_axis=1) for q in qs]
print("identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
