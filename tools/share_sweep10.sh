#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep10.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 48 --warmup 16 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), j['config'].get('detector_tuning'), 'lat', j['config'].get('latency_ms_per_step'))
" >> $L; }
T='{"lane_merge": 1, "hp_dedupe": 1, "concurrent_calls": 4}'
T2='{"lane_merge": 1, "hp_dedupe": 1, "concurrent_calls": 2}'
T0='{"lane_merge": 1, "hp_dedupe": 1}'
run --workload c4 --shard-of 2 --inflight 6
run --workload c4 --shard-of 2 --inflight 6 --tuning "$T2"
run --workload c4 --shard-of 4 --inflight 12
run --workload c4 --shard-of 4 --inflight 8 --tuning "$T0"
run --workload c4 --shard-of 8 --inflight 16
run --workload c4 --shard-of 8 --inflight 12 --tuning "$T0"
run --workload c4 --shard-of 8 --inflight 12 --tuning '{"lane_merge": 1, "hp_dedupe": 1, "concurrent_calls": 8}'
cat $L
