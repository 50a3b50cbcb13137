"""Five timed calls of coarsen_restarts_on_pressure at C384 (the bench secondary) for rocprofv3 --kernel-trace --stats (profiles/r02_pressure_pipeline_kernel_stats.csv)."""
import sys, torch
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
dev = torch.device('cuda:0')
r = bench.restart_pipeline_benchmark(dev, reps=5, which=("pressure",))
print(r[0]["ms"])
