import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
dev = torch.device("cuda:0")
for rep in range(3):
    r = bench.io_pipeline_benchmark(dev)[0]
    print(os.environ.get("FV3NET_AMD_IO_THREADS"), {k: round(v, 1) for k, v in r.items() if isinstance(v, float)})
