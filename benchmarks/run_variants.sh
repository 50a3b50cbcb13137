cd $GRAFT_REPO_ROOT
for v in base w4; do
  for noise in 1.0 0.1; do
    FV3HIP_LIBRARY=$GRAFT_REPO_ROOT/gpurun_variants/libfv3hip_$v.so timeout -k 10 120 python benchmarks/block_mean_timing.py --fields 4 --dtype f64 --noise $noise --label $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['cases']['f64x4']
print(d['label'], d['noise'], 'fast', c['fast']['fused_ms'], c['fast']['unfused_ms'], c['fast']['bit_identical'], 'exact', c['exact']['fused_ms'], c['exact']['unfused_ms'], c['exact']['bit_identical'])"
  done
done
