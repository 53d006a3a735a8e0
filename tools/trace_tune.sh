#!/bin/bash
# kernel timeline of the last detector call of tools/tune_detect.py <kind> <tunings>  -> gpurun_out/trace_tune_<tag>.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 700 rocprofv3 --kernel-trace --output-format csv -d $O/tt_$TAG -- python3 $ROOT/tools/tune_detect.py "$@" > $O/trace_tune_$TAG.log 2> $O/trace_tune_$TAG.err || { echo FAILED; tail -5 $O/trace_tune_$TAG.err; exit 1; }
grep tuning $O/trace_tune_$TAG.log
python3 $ROOT/tools/trace_last.py $O/tt_$TAG 90 | tee $O/trace_tune_$TAG.txt
rm -rf $O/tt_$TAG
