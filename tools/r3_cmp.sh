#!/bin/bash
# round 3: A/B of one tuning field (same bytes): lone C2, 8 x C2 alone, the in-flight bench (40 steps)
#   tools/r3_cmp.sh <tag> <field> <value on> <value off>
out=gpurun_out/r3_cmp_$1
mkdir -p $out
F=$2; ON=$3; OFF=$4
python tools/perf_sweep.py "[dict($F=$ON), dict($F=$OFF)]" 2>&1 | grep -v amdgpu.ids | tee $out/lone.log
python tools/tune_detect.py c2x8 "[{\"lane_merge\":1,\"hp_dedupe\":1,\"$F\":$ON}, {\"lane_merge\":1,\"hp_dedupe\":1,\"$F\":$OFF}]" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $out/c2x8.log
for v in $ON $OFF $ON $OFF; do
  python bench.py --no-cpu --no-extras --steps 40 --warmup 10 --tuning "{\"lane_merge\":1,\"hp_dedupe\":1,\"$F\":$v}" > $out/bench_$v.json 2> $out/bench_$v.err
  python -c "import json; d=json.load(open('$out/bench_$v.json')); print('$F=$v', round(d['value']/1e6,1), round(d['ms_per_step'],2), d['stage_ms'])"
done
