#!/bin/bash
# rocprofv3 kernel statistics of tools/tune_detect.py <kind> <tunings>  -> gpurun_out/prof_tune_<tag>.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt_$TAG -- python3 $ROOT/tools/tune_detect.py "$@" > $O/prof_tune_$TAG.log 2> $O/prof_tune_$TAG.err || { echo FAILED; tail -5 $O/prof_tune_$TAG.err; exit 1; }
cp $(ls $O/pt_$TAG/*/*kernel_stats.csv | tail -1) $O/prof_tune_$TAG.csv && rm -rf $O/pt_$TAG
grep tuning $O/prof_tune_$TAG.log
python3 - <<PY | tee $O/prof_tune_$TAG.txt
import csv
rows=list(csv.DictReader(open("$O/prof_tune_$TAG.csv")))
for r in rows[:28]:
    print("  %-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
