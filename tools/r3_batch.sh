#!/bin/bash
out=gpurun_out/r3_batch4; mkdir -p $out
run() { name=$1; shift; python bench.py --no-cpu --no-extras --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.load(open('$out/$name.json')); print('$name', round(d['value']/1e6,1), round(d['ms_per_step'],2), d['config']['latency_ms_per_step'])" || tail -3 $out/$name.err; }
run auto_c64d4 --clips 64 --inflight 4
run hp262k_c64d4 --clips 64 --inflight 4 --tuning '{"lane_merge":1,"hp_dedupe":1,"hp_chunk":262144}'
run auto_c48d6 --clips 48 --inflight 6
run auto_c64d5 --clips 64 --inflight 5
run auto_c16d6 --clips 16 --inflight 6
python tools/tune_detect.py c3 '[{}]' 2>&1 | grep -v amdgpu.ids | cut -c1-300
python tools/tune_detect.py c4 '[{}, {"lane_merge":1,"hp_dedupe":1}]' 2>&1 | grep -v amdgpu.ids | cut -c1-300
