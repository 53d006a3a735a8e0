#!/bin/bash
set -e
mkdir -p gpurun_out
L=gpurun_out/share_sweep6.log
: > $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu >> $L 2>&1 || { tail -30 $L; exit 1; }
run() { echo "== $*" >> $L; timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 24 --warmup 8 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step', j['config'].get('steps_in_flight_per_gpu'), j['config'].get('detector_tuning'), 'lat', j['config'].get('latency_ms_per_step'), json.dumps(j.get('stage_ms')))
" >> $L; }
for t in '{"lane_merge": 1, "fuse_elementwise": -1}' '{"lane_merge": 1, "fuse_elementwise": 1}'; do
run --workload c2 --clips 16 --inflight 4 --tuning "$t"
run --workload c4 --inflight 4 --tuning "$t"
done
for t in '{"fuse_elementwise": -1}' '{"fuse_elementwise": 1}'; do
run --workload c2 --clips 16 --inflight 1 --tuning "$t"
run --workload c2 --clips 1 --inflight 1 --tuning "$t"
run --workload c4 --inflight 1 --tuning "$t"
done
cat $L
