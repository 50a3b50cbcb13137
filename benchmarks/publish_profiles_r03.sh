#!/bin/bash
# gpurun_out/r03_profiles (benchmarks/collect_profiles_r03.sh on a GPU box) -> the summaries under profiles/ (run in the build container)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_profiles
P=$R/profiles
cp $O/r03_bench.json $P/r03_bench.json
newest() { ls -t $1 | head -1; }   # (gpurun merges every call's files into gpurun_out/: take the latest run of each pass)
cp $(newest "$O/stats_headline/*/*_kernel_stats.csv") $P/r03_bench_kernel_stats.csv
cp $(newest "$O/stats_all/*/*_kernel_stats.csv") $P/r03_bench_kernel_stats_with_secondary.csv
cp $(newest "$O/stats_pipeline/*/*_kernel_stats.csv") $P/r03_pressure_pipeline_kernel_stats.csv
cp $O/remap_timing_iid.json $P/r03_remap_timing_iid.json
cp $O/remap_timing_smooth.json $P/r03_remap_timing_smooth.json
grep -q "No such file" $O/r03_valu_issue_rates.txt || cp $O/r03_valu_issue_rates.txt $P/r03_valu_issue_rates.txt   # (the microbenchmark binary is built by hand: benchmarks/valu_ubench)
cp $O/pipelines_quick.jsonl $P/r03_pipelines_eager_and_graph.jsonl
python3 - $O $P <<'PY'
import csv, glob, json, sys
O, P = sys.argv[1:3]
# the headline's dispatches one by one: 10 warm-ups, then the 10 timed (bench.py --steps 10 --warmup 10 --no-parity)
import os
rows = [r for r in csv.DictReader(open(max(glob.glob(f"{O}/stats_headline/*/*_kernel_trace.csv"), key=os.path.getmtime))) if "mlp_fused_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(f"{P}/r03_bench_kernel_trace_headline.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["dispatch", "phase", "kernel", "start_ns", "end_ns", "duration_ms"])
    for i, r in enumerate(rows):
        name = r["Kernel_Name"].split("(")[0].replace("void fv3hip::(anonymous namespace)::", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        w.writerow([i + 1, "warm-up" if i < len(rows) - 10 else "timed", name, s, e, (e - s) / 1e6])
timed = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[-10:]]
print("headline: timed ten average", sum(timed) / len(timed), "ms over", len(rows), "dispatches")
json.dump({"command": "python benchmarks/block_mean_timing.py [--noise 0.1]  (HIP events, 20 repetitions, C384 -> C48, km = kn = 79; fused = fv3hip_mappm_block_mean, "
                      "unfused = mappm_multi_coarse_target + mask_weights + weighted_block_average(_multi))",
           "configs[2] data (iid delp)": json.load(open(f"{O}/block_mean_iid.json")),
           "smooth delp (a tenth of the spread)": json.load(open(f"{O}/block_mean_smooth.json"))}, open(f"{P}/r03_block_mean_timing.json", "w"), indent=1)
PY
F=$(newest "$O/pmc_fetch/*/*_counter_collection.csv"); W=$(newest "$O/pmc_write/*/*_counter_collection.csv")
python3 $R/benchmarks/make_pmc_json.py traffic $P/r03_pmc_traffic.json \
  "mlp_fused_kernel<8,false,true,false,false,false> epilogue=residual::=mlp_fused_kernel<8, false, true, false, false, false>:wide:max" \
  "wavg_block_kernel<float,float,8> C3072->C384::=wavg_block_kernel<float, float, 8>:wide:max" \
  "mass_wavg_block_kernel<double,float,8,4> C384::=mass_wavg_block_kernel<double, float, 8, 4>:wide:max" \
  "mappm_sweep_kernel<double,2,2,true,true>::=mappm_sweep_kernel<double, 2, 2, true, true, false>:max" -- $F $W
rm -rf /tmp/iid /tmp/smooth; mkdir -p /tmp/iid /tmp/smooth
for p in pmc_mappm1 pmc_mappm2 pmc_mappm3; do mkdir -p /tmp/iid/$p; cp $(newest "$O/$p/*/*_counter_collection.csv") /tmp/iid/$p/; done
mkdir -p /tmp/smooth/pmc_mappm1; cp $(newest "$O/pmc_mappm1s/*/*_counter_collection.csv") /tmp/smooth/pmc_mappm1/
python3 $R/benchmarks/make_pmc_mappm_json.py $P/r03_pmc_mappm.json "configs[2] data (iid delp)=/tmp/iid" "smooth delp (0.1 of the spread)=/tmp/smooth"
