#!/bin/bash
# One rank's share of C4 (and the C2 batch) with and without the concurrent_calls layout hint.
#   bash tools/share_sweep.sh   -> gpurun_out/share_sweep.log
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 24 --warmup 8 "$@" 2>>$L | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), j['config'].get('detector_tuning'))
" >> $L; }
for cc in 0 4 8; do run --workload c4 --shard-of 8 --inflight 8 --tuning "{\"concurrent_calls\": $cc}"; done
for cc in 8 12; do run --workload c4 --shard-of 8 --inflight 12 --tuning "{\"concurrent_calls\": $cc}"; done
for cc in 0 4; do run --workload c4 --shard-of 4 --inflight 4 --tuning "{\"concurrent_calls\": $cc}"; done
for cc in 0 3; do run --workload c4 --shard-of 2 --inflight 3 --tuning "{\"concurrent_calls\": $cc}"; done
for cc in 0 2 3; do run --workload c2 --clips 16 --inflight 3 --tuning "{\"concurrent_calls\": $cc}"; done
for cc in 0 3; do run --workload c4 --inflight 3 --tuning "{\"concurrent_calls\": $cc}"; done
cat $L
