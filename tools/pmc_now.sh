#!/bin/bash
# PMC traffic table of one bench configuration (one step at a time): bash tools/pmc_now.sh <tag> <bench args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pn_f_$TAG -- $B "$@" --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pn_f_$TAG.json 2> $O/pn_f_$TAG.err || { echo FAILED; tail -3 $O/pn_f_$TAG.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pn_w_$TAG -- $B "$@" --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pn_w_$TAG.json 2> $O/pn_w_$TAG.err
python3 $ROOT/tools/pmc_traffic.py $(ls $O/pn_f_$TAG/*/*counter_collection.csv | tail -1) $(ls $O/pn_w_$TAG/*/*counter_collection.csv | tail -1) > $O/pmc_now_$TAG.json
rm -rf $O/pn_f_$TAG $O/pn_w_$TAG
python3 - <<PY
import json
t = json.load(open("$O/pmc_now_$TAG.json"))
tot = sum(v["hbm_mb_per_launch"] * v["calls"] for v in t.values()) / 3
print("$TAG: PMC MB per step:", round(tot))
for k, v in list(t.items())[:22]:
    print("  %-22s calls/step %5.1f  MB/launch %8.0f  MB/step %8.0f" % (k, v["calls"] / 3, v["hbm_mb_per_launch"], v["hbm_mb_per_launch"] * v["calls"] / 3))
PY
