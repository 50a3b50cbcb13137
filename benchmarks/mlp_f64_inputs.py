"""The headline kernel on float64 sources (what call_py_fort hands the hook) next to float32 ones."""
import sys
import torch
sys.path.insert(0, '.')
import bench
from fv3net_amd.mlp import MlpModel
from fv3net_amd.ops import HipTimer

dev = torch.device('cuda:0')
model = MlpModel(bench.zc_spec(0), device=dev)
ncol = 6 * 384 * 384
src = bench.zc_inputs_device(dev, ncol, seed=1)
for label, s in (("float32", src), ("float64", {k: v.double() for k, v in src.items()})):
    for _ in range(3):
        model.predict(s)
    torch.cuda.synchronize()
    t = HipTimer(); t.start(dev)
    for _ in range(10):
        model.predict(s)
    t.stop(dev)
    ms = t.elapsed_ms() / 10
    print(f"{label} sources: {ms:.3f} ms per C384 snapshot = {ncol / ms * 1e3:.3e} columns/s, "
          f"{model.flops_per_sample * ncol / ms / 1e9:.1f} TFLOP/s")
