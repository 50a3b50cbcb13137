"""cProfile of the host side of coarsen_restarts_on_pressure at C384 (the eager call is bound by Python time per launch)."""
import cProfile, io, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
calls = {}
orig = bench.time.perf_counter
# reuse the benchmark's data builder: run it with reps=0 to get hold of the closure is not possible, so profile the whole thing
pr = cProfile.Profile()
r = bench.restart_pipeline_benchmark(dev, reps=2, which=("pressure",), graph=False)   # warm
pr.enable()
r = bench.restart_pipeline_benchmark(dev, reps=20, which=("pressure",), graph=False)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
print(r[0]["ms"])
