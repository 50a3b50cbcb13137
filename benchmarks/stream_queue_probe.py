"""Does the hardware queue a side stream lands on matter?  k dummy streams are created (and used once) before the pipelines
create theirs, shifting the round-robin stream -> queue assignment of the HIP runtime; the pressure pipeline (which creates the
'beside' and 'surface' streams) runs first, then sigma (which uses 'surface' only) -- eager ms of both, per k."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dummies = [torch.cuda.Stream(device=dev) for _ in range(k)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
r = bench.restart_pipeline_benchmark(dev, which=("pressure", "sigma"), graph=False)
print(json.dumps({"dummy_streams": k, "pressure_ms": round(r[0]["ms"], 3), "sigma_ms": round(r[1]["ms"], 3)}))
