#!/bin/bash
# Kernel statistics (rocprofv3 --kernel-trace --stats) of the saturated regimes: a batch of 8 C2 clips
# per step and the C4 batch (512 clips x 4 ch x 10 s) on one GPU.  Output: gpurun_out/prof_regimes_$1/
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_regimes_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in "c2 --clips 8" "c4"; do
  set -- $W
  name=$1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu --no-extras > $O/$name.json 2> $O/$name.err || echo "FAILED $W"
  cp $(ls $O/$name/*/*kernel_stats.csv | tail -1) $O/${name}_kernel_stats.csv && rm -rf $O/$name
  python3 - <<PY
import json,csv
j=json.load(open("$O/$name.json"))
print("$name", round(j["value"]/1e6,1),"M frames/s", round(j["ms_per_step"],2),"ms/step", j["stage_ms"], j["detector_passes"])
rows=list(csv.DictReader(open("$O/${name}_kernel_stats.csv")))
for r in rows[:22]:
    print("  %-70s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
