#!/bin/bash
# the bench line, the one-clip line and the rocprofv3 kernel statistics of the default command -> gpurun_out/bench_set_<tag>/
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/bench_set_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
timeout -k 10 400 $B > $O/bench.json 2> $O/bench.err || echo "bench FAILED"
timeout -k 10 300 $B --clips 1 --inflight 1 --no-cpu --no-extras > $O/bench_one_clip_per_step.json 2>> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-cpu --no-extras > $O/bench_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<PY
import json
j = json.load(open("$O/bench.json")); print(round(j["value"] / 1e6, 1), j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"].get("alone"))
j = json.load(open("$O/bench_under_rocprof.json")); print("under rocprof", round(j["value"] / 1e6, 1), j["roofline"]["avg_launch_ms"])
PY
head -4 $O/kernel_stats.csv | cut -c1-60,180-330
