#!/bin/bash
# One 48-clip C2 call with the GPU to itself in the throughput settings, kernel by kernel (rocprofv3 kernel trace of
# `bench.py --inflight 1`, the last step):  tools/alone_trace.sh <tag> [extra bench args]  ->  gpurun_out/alone_<tag>.txt
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/alone_$TAG -- python3 $R/bench.py --inflight 1 --steps 3 --warmup 2 \
  --no-cpu --no-extras --tuning "${TUNING:-{\"lane_merge\":1,\"hp_dedupe\":1\}}" "$@" > $R/gpurun_out/alone_$TAG.log 2>&1
cd $R && python tools/trace_last_step.py gpurun_out/alone_$TAG > gpurun_out/alone_$TAG.txt
rm -rf gpurun_out/alone_$TAG
sed -n '/--- totals/,$p' gpurun_out/alone_$TAG.txt
