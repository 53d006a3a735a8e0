#!/bin/bash
for v in 0 1 0 1; do
  OFP_MM_IL=$v python bench.py --no-cpu --no-extras --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('IL=$v', round(j['value']/1e6,1), 'M', round(j['ms_per_step'],2), 'ms', j['stage_ms'].get('mm'), j['stage_ms'].get('rel'))
"
done
