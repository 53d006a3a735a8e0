#!/bin/bash
# round 3: chain-local verification kernels against the host-verified pass groups (same bytes, timing only)
set -o pipefail
out=gpurun_out/r3_ab_$1
mkdir -p $out
python tools/perf_sweep.py "[dict(), dict(host_verify=1)]" > $out/lone_c2.log 2>&1
python tools/tune_detect.py c2x8 '[{}, {"host_verify":1}, {"lane_merge":1,"hp_dedupe":1}, {"lane_merge":1,"hp_dedupe":1,"host_verify":1}]' > $out/c2x8.log 2>&1
python tools/tune_detect.py c4 '[{}, {"host_verify":1}, {"lane_merge":1,"hp_dedupe":1}, {"lane_merge":1,"hp_dedupe":1,"host_verify":1}]' > $out/c4.log 2>&1
python bench.py --no-cpu > $out/bench.json 2> $out/bench.err
python bench.py --no-cpu --no-extras --tuning '{"lane_merge":1,"hp_dedupe":1,"host_verify":1}' > $out/bench_host_verify.json 2> $out/bench_hv.err
tail -n 3 $out/*.log; cat $out/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('one_clip_per_step'))"
cat $out/bench_host_verify.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
