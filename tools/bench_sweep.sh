set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_bench_sweep; mkdir -p $O
for cfg in "8 1" "1 1" "4 1" "16 1" "8 2" "1 8"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --clips $1 --inflight $2 --no-cpu > $O/c2_clips$1_if$2.json 2> $O/c2_clips$1_if$2.err || echo "FAILED $cfg"
  python -c "
import json,sys
j=json.load(open('$O/c2_clips$1_if$2.json'))
print('clips $1 inflight $2:', round(j['value']/1e6,1),'M frames/s', round(j['ms_per_step'],3),'ms/step', j['roofline']['kernel'], round(j['roofline']['frac'],4), 'e2e', round(j['roofline_e2e']['frac'],4), j['config'].get('one_clip_per_step'), j['stage_ms'])
" || tail -3 $O/c2_clips$1_if$2.err
done
