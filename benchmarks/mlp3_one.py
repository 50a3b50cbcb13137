"""One warm launch pair of the split-bf16 network on the headline workload, for rocprofv3 counter passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 benchmarks/mlp3_one.py [1]
(1 = with the residual outputs)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from fv3net_amd.mlp import MlpModelSplitBf16  # noqa: E402

dev = torch.device("cuda:0")
res = len(sys.argv) > 1 and sys.argv[1] == "1"
model = MlpModelSplitBf16(bench.zc_spec(0, residuals=res), device=dev)
src = bench.zc_inputs_device(dev, 6 * 384 * 384, seed=1000)
for _ in range(3):
    model.predict(src)
torch.cuda.synchronize()
