#!/bin/bash
# the driver's own command line (--steps 20 --warmup 5) for the C4 shards at several depths
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep11.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), 'lat', j['config'].get('latency_ms_per_step'))
" >> $L; }
for d in 4 6 8 12; do run --workload c4 --shard-of 8 --inflight $d; done
for d in 4 6 8 12; do run --workload c4 --shard-of 4 --inflight $d; done
for d in 3 4 6; do run --workload c4 --shard-of 2 --inflight $d; done
for d in 3 4 6; do run --workload c4 --inflight $d; done
for d in 4 6; do run --workload c2 --inflight $d; done
cat $L
