#!/bin/bash
# A sweep of bench.py configurations on one GPU -- one parameterised script for every "what if" table of DESIGN.md
# (steps in flight x shard size x detector tuning; rounds 2-3 kept eleven near-identical copies of this).
#
#   tools/share_sweep.sh <tag> [steps] [warmup] < cases.txt        one bench argument list per line, e.g.
#       --workload c4 --shard-of 8 --inflight 12 --tuning {"concurrent_calls":4}
#   tools/share_sweep.sh <tag> c4shards                            the C4 strong-scaling shards (whole batch, 1/2, 1/4, 1/8)
#   tools/share_sweep.sh <tag> inflight                            C2 x 16 at 1 / 2 / 4 / 6 / 8 steps in flight
# -> gpurun_out/share_sweep_<tag>.log
set -o pipefail
TAG=${1:-x}; PRESET=${2:-}
mkdir -p gpurun_out
L=gpurun_out/share_sweep_$TAG.log
: > $L
STEPS=24; WARM=8
run() { echo "== $*" >> $L; timeout -k 10 400 python bench.py --no-cpu --no-extras --steps $STEPS --warmup $WARM "$@" 2>>$L | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']
        print(round(j['value'] / 1e6, 1), 'M frames/s', round(j['ms_per_step'], 3), 'ms/step, in flight', c.get('steps_in_flight_per_gpu'), 'steps per call', c.get('steps_per_call'), c.get('detector_tuning'), 'latency', c.get('latency_ms_per_step'))
" >> $L; }
case "$PRESET" in
  c4shards) STEPS=20; WARM=5   # (the driver's K / W)
    run --workload c4; run --workload c4 --shard-of 2; run --workload c4 --shard-of 4; run --workload c4 --shard-of 8
    run --workload c4 --shard-of 2 --steps-per-call 1; run --workload c4 --shard-of 4 --steps-per-call 1; run --workload c4 --shard-of 8 --steps-per-call 1 ;;
  inflight) for d in 1 2 4 6 8; do run --inflight $d; done ;;
  *) [ -n "$PRESET" ] && STEPS=$PRESET; [ -n "$3" ] && WARM=$3
    while IFS= read -r line; do [ -n "$line" ] && eval run $line; done ;;
esac
cat $L
