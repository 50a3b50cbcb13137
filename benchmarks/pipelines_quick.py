"""The three restart pipelines at C384 -> C48 in both remap arithmetics, as bench.py times them (a quick A/B of host-side
changes: `python benchmarks/pipelines_quick.py`)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
for e in bench.pipelines_benchmark(dev):
    print(json.dumps({"kernel": e["kernel"], "ms": round(e["ms"], 3), "graph_replay_ms": round(e.get("graph_replay_ms") or 0, 3)}))
