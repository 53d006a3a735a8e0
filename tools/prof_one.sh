#!/bin/bash
# rocprofv3 kernel statistics of ONE bench configuration, one step at a time:
#   bash tools/prof_one.sh <tag> <bench args...>     -> gpurun_out/prof_one_<tag>.{json,csv,txt}
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1_$TAG -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-extras "$@" > $O/prof_one_$TAG.json 2> $O/prof_one_$TAG.err || { echo "FAILED"; tail -5 $O/prof_one_$TAG.err; exit 1; }
cp $(ls $O/p1_$TAG/*/*kernel_stats.csv | tail -1) $O/prof_one_$TAG.csv && rm -rf $O/p1_$TAG
python3 - <<PY | tee $O/prof_one_$TAG.txt
import json,csv
j=json.load(open("$O/prof_one_$TAG.json"))
print("$TAG", round(j["value"]/1e6,1),"M frames/s", round(j["ms_per_step"],2),"ms/step", j["stage_ms"], j.get("detector_passes"))
rows=list(csv.DictReader(open("$O/prof_one_$TAG.csv")))
for r in rows[:26]:
    print("  %-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
