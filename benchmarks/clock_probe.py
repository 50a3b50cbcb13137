"""Shader clock and package power sampled with rocm-smi while the fp32 MLP kernel and then the split-bf16 kernel run back to back
for four seconds each on the production Zhao-Carr graph.  Run from the repo root on the GPU box."""
import sys, subprocess, threading, time, torch
sys.path.insert(0, '.')
import bench
from fv3net_amd.mlp import MlpModel, MlpModelSplitBf16
dev = torch.device('cuda:0')
src = bench.zc_inputs_device(dev, 6*384*384, seed=1)
def sample(tag, stop):
    vals = []
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = [l for l in out.splitlines() if "sclk" in l]
        pw = [l for l in out.splitlines() if "Power" in l and "W" in l]
        vals.append((sclk[0].split("(")[-1].rstrip(")") if sclk else "?", pw[0].split(":")[-1].strip() if pw else "?"))
        time.sleep(0.3)
    print(tag, vals[:8])
for name, cls in (("fp32 kernel", MlpModel), ("split-bf16 kernel", MlpModelSplitBf16)):
    m = cls(bench.zc_spec(0, residuals=True), device=dev)
    stop = threading.Event()
    t = threading.Thread(target=sample, args=(name, stop)); t.start()
    t0 = time.time()
    while time.time() - t0 < 4.0:
        for _ in range(50): m.predict(src)
        torch.cuda.synchronize()
    stop.set(); t.join()
