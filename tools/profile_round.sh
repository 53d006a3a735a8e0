#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel statistics of the same
# command, and the two PMC passes (counters only, separate runs) the roofline.traffic figure comes
# from.  Everything lands in gpurun_out/prof_$1; copy what is to be judged into profiles/.
set -e -o pipefail
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --inflight 1 --no-cpu > $O/bench_one_step_at_a_time.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
python tools/trace_concurrency.py $O/stats 20 > $O/concurrency.txt
# counters per launch do not depend on what else is in flight: one step at a time, with the tuning the default bench uses
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py --steps 2 --warmup 1 --no-cpu --inflight 1 --tuning '{"hp_span": 2, "mm_chunk": 8192}' > $O/pmc_f.json 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py --steps 2 --warmup 1 --no-cpu --inflight 1 --tuning '{"hp_span": 2, "mm_chunk": 8192}' > $O/pmc_w.json 2> $O/pmc_w.err
cp $(ls $O/pmc_f/*/*counter_collection.csv | tail -1) $O/pmc_fetch_size.csv
cp $(ls $O/pmc_w/*/*counter_collection.csv | tail -1) $O/pmc_write_size.csv
python tools/pmc_traffic.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv > $O/pmc_traffic_per_kernel.json
rm -rf $O/stats $O/pmc_f $O/pmc_w
head -c 600 $O/bench.json; echo; head -12 $O/kernel_stats.csv | cut -c1-150
