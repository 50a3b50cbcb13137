"""The pressure-level pipeline with the fused remap + block mean pinned off and on (FV3NET_AMD_FUSED_BLOCK_MEAN = 0 | 1; the default follows the data), on configs[2]'s
iid thicknesses and on smooth ones (a tenth of the spread), both remap arithmetics: `python benchmarks/pipeline_fused_ab.py`."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fv3net_amd import ops

dev = torch.device("cuda:0")
out = {}
for spread, label in ((1.0, "iid delp"), (0.1, "smooth delp")):
    for arith in ("exact", "fast"):
        ops.MAPPM_ARITHMETIC = arith
        for fused in ("0", "1"):
            os.environ["FV3NET_AMD_FUSED_BLOCK_MEAN"] = fused
            r = bench.restart_pipeline_benchmark(dev, which=("pressure",), graph=False, delp_spread=spread)
            out[f"{label}, {arith}, {'fused' if fused == '1' else 'three launches'}"] = round(r[0]["ms"], 3)
print(json.dumps(out, indent=1))
