import sys, torch
sys.path.insert(0, '.')
from fv3net_amd import ops
arith, nf = sys.argv[1], int(sys.argv[2])
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n, NZ = 384, 79
delp = torch.rand((6, NZ, n, n), device=dev, generator=g) * 1200 + 300
area = torch.rand((6, n, n), device=dev, generator=g) * 0.5 + 0.5
pe1 = ops.pressure_at_interface(delp, 300.0, 1)
pe2 = ops.pressure_at_interface(ops.block_upsample(ops.weighted_block_average(delp, area, 8), 8), 300.0, 1)
qs = [torch.rand((6, NZ, n, n), device=dev, generator=g) * 2000 - 1000 for _ in range(nf)]
for _ in range(3):
    ops.mappm_multi(pe1, qs, pe2, z_axis=1, arith=arith) if nf > 1 else ops.mappm(pe1, qs[0], pe2, z_axis=1, arith=arith)
torch.cuda.synchronize()
