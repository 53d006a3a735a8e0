#!/bin/bash
# in-flight bench with the stages on planar copies (-1), the rel side interleaved (1), both sides (0)
for v in -1 1 0 -1 1 0; do
  python bench.py --no-cpu --no-extras --steps 20 --warmup 5 --tuning "{\"lane_merge\":1,\"hp_dedupe\":1,\"interleaved\":$v}" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('interleaved=$v', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],2), 'ms', {k: round(v, 1) for k, v in j['stage_ms'].items()})
"
done
