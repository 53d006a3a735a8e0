#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep4.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 24 --warmup 8 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), j['config'].get('detector_tuning'), 'lat', j['config'].get('latency_ms_per_step'), j.get('detector_passes'))
" >> $L; }
for t in '{"hp_early": 0}' '{"hp_early": -1}' '{"hp_chunk": 16384}' '{"hp_chunk": 16384, "hp_early": -1}' '{"hp_dedupe": -1, "hp_early": -1}' '{"hp_chunk": 16384, "hp_warm": 81920}'; do
run --workload c2 --clips 16 --inflight 4 --tuning "$t"
run --workload c4 --inflight 4 --tuning "$t"
done
cat $L
