#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep8.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 24 --warmup 6 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), 'lat', j['config'].get('latency_ms_per_step'), j.get('detector_passes'))
" >> $L; }
B='"lane_merge": 1, "hp_dedupe": 1'
for t in "{$B}" "{$B, \"mm_chunk\": 16384}" "{$B, \"mm_chunk\": 16384, \"ar_chunk\": 16384}" "{$B, \"mm_chunk\": 8192, \"ar_chunk\": 8192}" "{$B, \"mm_chunk\": 32768, \"ar_chunk\": 32768}" "{$B, \"mm_span\": 16}" "{$B, \"hp_warm\": 61440}" "{$B, \"hp_candidates\": 16}" "{$B, \"hp_chunk\": 65536}"; do
run --clips 16 --inflight 6 --tuning "$t"
done
cat $L
