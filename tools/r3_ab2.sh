#!/bin/bash
# in-flight bench under the verification variants (host_verify 0: look-back + pre-enqueued IIR rounds; 1: host-verified;
# 2: host-verified IIR rounds only; 3: host-verified followers / tracker only)
out=gpurun_out/r3_ab2_$1
mkdir -p $out
for hv in 0 1 2 3; do
  python bench.py --no-cpu --no-extras --tuning "{\"lane_merge\":1,\"hp_dedupe\":1,\"host_verify\":$hv}" > $out/bench_hv$hv.json 2> $out/bench_hv$hv.err
  python -c "import json; d=json.load(open('$out/bench_hv$hv.json')); print($hv, round(d['value']/1e6,1), round(d['ms_per_step'],2), d['stage_ms'])"
done
python tools/perf_sweep.py "[dict(), dict(host_verify=1), dict(host_verify=2), dict(host_verify=3)]" 2>&1 | grep -v amdgpu.ids
