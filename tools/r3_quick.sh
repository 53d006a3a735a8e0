#!/bin/bash
# round 3: the quick timing set after a kernel change: lone C2, 8 x C2 alone (both layouts), the in-flight bench, C3 detector
out=gpurun_out/r3_quick_$1
mkdir -p $out
python tools/perf_sweep.py "[dict()]" 2>&1 | grep -v amdgpu.ids | tee $out/lone.log
python tools/tune_detect.py c2x8 '[{}, {"lane_merge":1,"hp_dedupe":1}]' 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $out/c2x8.log
python bench.py --no-cpu --no-extras > $out/bench.json 2> $out/bench.err
python -c "import json; d=json.load(open('$out/bench.json')); print('bench', round(d['value']/1e6,1), round(d['ms_per_step'],2), d['stage_ms'], d.get('stage_ms_alone'))"
[ "$2" = c3 ] && python tools/tune_detect.py c3 '[{}]' 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $out/c3.log
