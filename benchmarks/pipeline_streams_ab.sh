# A/B of the pipelines' stream structure inside full bench runs on one box (eager ms / graph replay ms per pipeline):
#   gpurun -- 'bash benchmarks/pipeline_streams_ab.sh'
cd $GRAFT_REPO_ROOT
run() {
python bench.py --no-cpu --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', ' '.join(f\"{s['ms']:.2f}/{s.get('graph_replay_ms',0):.2f}\" for s in d['secondary'] if 'pipeline' in s['kernel']))"
}
for rep in 1 2; do
FV3NET_AMD_PIPELINE_STREAMS=1 run "streams=1      "

FV3NET_AMD_PIPELINE_STREAMS=0 run "streams=0      "
done
