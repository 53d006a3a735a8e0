#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep7.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), 'lat', j['config'].get('latency_ms_per_step'))
" >> $L; }
run --clips 16 --inflight 4
run --clips 16 --inflight 6
run --clips 16 --inflight 8
run --clips 32 --inflight 3
run --clips 32 --inflight 4
run --clips 8 --inflight 8
run --workload c4 --inflight 4
run --workload c4 --inflight 6
run --workload c4 --shard-of 8
run --workload c4 --shard-of 4
run --workload c4 --shard-of 2
cat $L
