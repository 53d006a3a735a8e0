#!/bin/bash
# C4 on one GPU: the whole batch and one rank's share of 2 / 4 / 8 GPUs, bench defaults -> gpurun_out/share_sweep9.log
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep9.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 36 --warmup 12 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), j['config'].get('detector_tuning'), 'lat', j['config'].get('latency_ms_per_step'))
" >> $L; }
run --workload c4
run --workload c4 --inflight 6
run --workload c4 --shard-of 2
run --workload c4 --shard-of 4
run --workload c4 --shard-of 8
run --workload c4 --inflight 1 --tuning '{}'
cat $L
