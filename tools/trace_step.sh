#!/bin/bash
# kernel timeline of the last step of ONE bench configuration (one step at a time)
#   bash tools/trace_step.sh <tag> <bench args...>   -> gpurun_out/trace_step_<tag>.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/ts_$TAG -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu --no-extras --inflight 1 "$@" > $O/trace_step_$TAG.json 2> $O/trace_step_$TAG.err || { echo FAILED; tail -5 $O/trace_step_$TAG.err; exit 1; }
python3 $ROOT/tools/trace_last.py $O/ts_$TAG 70 | tee $O/trace_step_$TAG.txt
rm -rf $O/ts_$TAG
