import torch, time, sys
sys.path.insert(0, '.')
from fv3net_amd import ops
dev = torch.device('cuda:0')
x = torch.rand((6, 79, 3072, 3072), device=dev)
area = torch.rand((6, 3072, 3072), device=dev) * 0.5 + 0.5
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    tm = ops.HipTimer(); tm.start(dev)
    for _ in range(reps): fn()
    tm.stop(dev); return tm.elapsed_ms() / reps
nb = x.numel() * 4
ms = t(lambda: x.sum()); print("torch sum: %.3f ms = %.2f TB/s" % (ms, nb / ms / 1e9))
ms = t(lambda: x.max()); print("torch max: %.3f ms = %.2f TB/s" % (ms, nb / ms / 1e9))
y = torch.empty_like(x[:3])
ms = t(lambda: y.copy_(x[:3])); print("torch copy (r+w): %.3f ms = %.2f TB/s" % (ms, 2 * y.numel() * 4 / ms / 1e9))
ms = t(lambda: ops.weighted_block_average(x, area, 8)); print("wavg f=8: %.3f ms = %.2f TB/s" % (ms, (nb * (1 + 1/64) + nb / 79) / ms / 1e9))
for f in (2, 4, 16):
    ms = t(lambda: ops.weighted_block_average(x, area, f)); print("wavg f=%d: %.3f ms = %.2f TB/s" % (f, ms, (nb * (1 + 1/f**2) + nb / 79) / ms / 1e9))
