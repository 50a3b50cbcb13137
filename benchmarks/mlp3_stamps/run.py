"""Per-phase cycle counts of the split-bf16 kernel on the headline network (build.sh first).  usage: run.py [1]   (1 = with the residual outputs)"""
import os, sys, ctypes, torch, numpy as np
os.environ["FV3HIP_LIBRARY"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfv3hip_stamps.so")
residuals = len(sys.argv) > 1 and sys.argv[1] == "1"
sys.path.insert(0, '.')
import bench
from fv3net_amd import _lib, ops
from fv3net_amd.mlp import MlpModelSplitBf16
dev = torch.device('cuda:0')
spec = bench.zc_spec(0, residuals=residuals)
model = MlpModelSplitBf16(spec, device=dev)
N = 6*384*384
src = bench.zc_inputs_device(dev, N, seed=1)
for _ in range(6): model.predict(src)
torch.cuda.synchronize()
t = ops.HipTimer(); t.start(dev)
for _ in range(10): model.predict(src)
t.stop(dev); print("ms", t.elapsed_ms() / 10)
stamps = torch.zeros((256*4, 8), dtype=torch.int64, device=dev)
lib = _lib.load()
lib.fv3hip_diag_set_mlp3_stamps.argtypes = [ctypes.c_void_p]
lib.fv3hip_diag_set_mlp3_stamps(ctypes.c_void_p(stamps.data_ptr()))
model.predict(src)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
tiles = N/128/256
names = ['tile start', 'layer1 ksteps', 'bias+relu', 'hidden ksteps', 'output ksteps', 'epilogue', 'hidden fences']
tot = s[:, :7].sum(1).mean()
for i, n in enumerate(names):
    print(f"{n:14s} {s[:, i].mean()/tiles:10.0f} ticks/tile  {100*s[:, i].mean()/tot:5.1f}%")
print('total/tile', tot/tiles)
