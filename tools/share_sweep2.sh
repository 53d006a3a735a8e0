#!/bin/bash
# hardware queues x steps in flight for the smallest C4 shard -> gpurun_out/share_sweep2.log
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep2.log
: > $L
run() { echo "== Q=$GPU_MAX_HW_QUEUES $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 24 --warmup 8 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), j['config'].get('detector_tuning'), 'lat', j['config'].get('latency_ms_per_step'), json.dumps(j.get('stage_ms')))
" >> $L; }
for q in 16 32 64; do
  export GPU_MAX_HW_QUEUES=$q
  run --workload c4 --shard-of 8 --inflight 8 --tuning '{"concurrent_calls": 4}'
  run --workload c4 --shard-of 8 --inflight 12 --tuning '{"concurrent_calls": 4}'
done
export GPU_MAX_HW_QUEUES=32
run --workload c4 --shard-of 8 --inflight 16 --tuning '{"concurrent_calls": 4}'
run --workload c4 --shard-of 8 --inflight 16 --tuning '{"concurrent_calls": 2}'
run --workload c4 --shard-of 8 --inflight 12 --tuning '{"concurrent_calls": 2}'
run --workload c4 --shard-of 4 --inflight 6 --tuning '{"concurrent_calls": 2}'
run --workload c4 --shard-of 4 --inflight 8 --tuning '{"concurrent_calls": 4}'
run --workload c4 --shard-of 2 --inflight 4
cat $L
