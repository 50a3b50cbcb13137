#!/bin/bash
# Round-3 profile collection on one MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash benchmarks/collect_profiles_r03.sh'
# Everything lands under gpurun_out/r03_profiles/; the summaries judged are then copied to profiles/ (see profiles/README.md).
# Counter passes are separate runs with --kernel-trace only (no other trace domains), one counter group per pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -x
# 1. the bench line of this tree
python3 $R/bench.py --steps 20 --warmup 10 > $O/r03_bench.json 2> $O/bench.err || exit 1
# 2. kernel statistics of the headline alone: >= 10 warm-ups under the profiler, no parity call, per-dispatch trace kept
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -- python3 $R/bench.py --steps 10 --warmup 10 --no-cpu --no-secondary --no-parity > $O/stats_headline.log 2>&1 || exit 1
# 3. ... and with the secondary workloads
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_all -- python3 $R/bench.py --steps 10 --warmup 10 --no-cpu --no-parity > $O/stats_all.log 2>&1 || exit 1
# 4. HBM traffic of the headline and the coarsening kernels (FETCH_SIZE / WRITE_SIZE, separate passes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-parity > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-parity > $O/pmc_write.log 2>&1 || exit 1
# (gpurun merges at most 64 MiB back: of the two traffic passes keep the rows of the kernels that profiles/r03_pmc_traffic.json quotes)
for d in pmc_fetch pmc_write; do
  for f in $O/$d/*/*_counter_collection.csv; do
    (head -1 $f; grep -E 'mlp_fused_kernel<8, false, true, false, false, false>|wavg_block_kernel<float, float, 8>|mass_wavg_block_kernel<double, float, 8, 4>|mappm_sweep_kernel<double, 2, 2, true, true, false>' $f) > $f.kept && mv $f.kept $f
  done
  rm -f $O/$d/*/*_kernel_trace.csv
done
# 5. the remap sweep: occupancy / stall / instruction counters, three passes
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_mappm1 -- python3 $R/benchmarks/remap_sweep_timing.py --reps 3 > $O/pmc_mappm1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mappm2 -- python3 $R/benchmarks/remap_sweep_timing.py --reps 3 > $O/pmc_mappm2.log 2>&1 || exit 1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_mappm3 -- python3 $R/benchmarks/remap_sweep_timing.py --reps 3 > $O/pmc_mappm3.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_mappm1s -- python3 $R/benchmarks/remap_sweep_timing.py --reps 3 --noise 0.1 > $O/pmc_mappm1s.log 2>&1 || exit 1
# 6. the timing harness itself (no profiler), both data sets
python3 $R/benchmarks/remap_sweep_timing.py --label r03 > $O/remap_timing_iid.json 2>/dev/null
python3 $R/benchmarks/remap_sweep_timing.py --label r03 --noise 0.1 > $O/remap_timing_smooth.json 2>/dev/null
# 7. the pressure-level pipeline: kernel statistics of five calls
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pipeline -- python3 $R/benchmarks/pressure_pipeline_once.py > $O/stats_pipeline.log 2>&1 || exit 1
# 7b. the fused remap + block mean against the three launches, both data sets; the pipelines in both arithmetics
python3 $R/benchmarks/block_mean_timing.py > $O/block_mean_iid.json 2>/dev/null
python3 $R/benchmarks/block_mean_timing.py --noise 0.1 > $O/block_mean_smooth.json 2>/dev/null
python3 $R/benchmarks/pipelines_quick.py > $O/pipelines_quick.jsonl 2>/dev/null
# 8. vector-instruction issue rates
make -s -C $R/benchmarks/valu_ubench && $R/benchmarks/valu_ubench/valu_rate > $O/r03_valu_issue_rates.txt 2>&1
find $O -name "*.csv" | wc -l
du -sh $O
