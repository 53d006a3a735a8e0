#!/bin/bash
# Round 3: the whole measurement set behind the bench line, on the GPU box through gpurun -> gpurun_out/prof_<tag>/
# (copy what is to be judged into profiles/r03/).  Nothing here spawns processes under the profiler: the clips of
# the profiled runs come from the cache the un-profiled bench run fills (bench.py: under_profiler / OFP_SYNTH_CACHE).
#   tools/profile_r3.sh <tag> [quick]
TAG=${1:-r03}; QUICK=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_$TAG
mkdir -p $O
export OFP_SYNTH_CACHE=/tmp/ofp_clip_cache GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
echo "== bench (default command; fills the clip cache)"; date
timeout -k 10 900 $B > $O/bench.json 2> $O/bench.err || { echo "bench FAILED"; tail -5 $O/bench.err; }
timeout -k 10 300 $B --clips 1 --inflight 1 --no-cpu --no-extras > $O/bench_one_clip_per_step.json 2>> $O/bench.err
echo "== kernel statistics of the in-flight regime (40 timed steps, 6 in flight)"; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-cpu --no-extras --steps 40 --warmup 6 > $O/bench_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/kernel_stats_all_launches.csv
python3 $ROOT/tools/trace_regime.py $O/stats > $O/kernel_stats_in_flight.json 2> $O/regime.err
rm -rf $O/stats
pmc() {  # name, counters..., then -- bench arguments
  name=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 400 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $O/pmc_$name -- $B "$@" --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pmc_$name.json 2> $O/pmc_$name.err
  f=$(ls $O/pmc_$name/*/*counter_collection.csv 2>/dev/null | tail -1)
  if [ -n "$f" ]; then cp $f $O/pmc_$name.csv; else echo "no counter file for $name"; tail -3 $O/pmc_$name.err; fi
  rm -rf $O/pmc_$name
}
TH='{"lane_merge": 1, "hp_dedupe": 1}'
CLIPS=${OFP_PROFILE_CLIPS:-48}   # (the bench's default batch: the per-launch counters belong to that size)
echo "{\"clips\": $CLIPS}" > $O/pmc_meta.json
echo "== HBM-side traffic per kernel (FETCH_SIZE / WRITE_SIZE, separate passes, one step at a time)"; date
pmc fetch_c2 FETCH_SIZE -- --clips $CLIPS --tuning "$TH"
pmc write_c2 WRITE_SIZE -- --clips $CLIPS --tuning "$TH"
python3 $ROOT/tools/pmc_traffic.py $O/pmc_fetch_c2.csv $O/pmc_write_c2.csv > $O/pmc_traffic_per_kernel.json
if [ -z "$QUICK" ]; then
  pmc fetch_c2x1 FETCH_SIZE -- --clips 1
  pmc write_c2x1 WRITE_SIZE -- --clips 1
  python3 $ROOT/tools/pmc_traffic.py $O/pmc_fetch_c2x1.csv $O/pmc_write_c2x1.csv > $O/pmc_traffic_per_kernel_c2x1.json
fi
echo "== SQ counters of every kernel of the step (detector + STFT/classifier), one step at a time"; date
pmc sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -- --clips $CLIPS --tuning "$TH"
pmc sq_cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -- --clips $CLIPS --tuning "$TH"
pmc sq_vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY -- --clips $CLIPS --tuning "$TH"
pmc sq_mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES -- --clips $CLIPS --tuning "$TH"
[ -f $O/pmc_sq_mfma.csv ] || pmc sq_mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES -- --clips $CLIPS --tuning "$TH"
python3 $ROOT/tools/pmc_sq_table.py $O > $O/pmc_sq_per_kernel.json 2> $O/sq_table.err
if [ -z "$QUICK" ]; then
  echo "== per-hop latency (BASELINE config 5)"; date
  timeout -k 10 120 python3 $ROOT/tools/stream_latency.py --config c5 --hops 10000 > $O/stream_latency_c5.json 2> $O/lat.err
  timeout -k 10 120 python3 $ROOT/tools/stream_latency.py --config realtime --hops 10000 > $O/stream_latency_realtime.json 2>> $O/lat.err
fi
date
head -c 1500 $O/bench.json; echo
python3 - <<PY
import json
t = json.load(open("$O/pmc_traffic_per_kernel.json"))
tot = sum(v["hbm_mb_per_launch"] * v["calls"] for v in t.values())
print("PMC MB per step (3 steps profiled):", round(tot / 3), {k: (v["calls"], round(v["hbm_mb_per_launch"])) for k, v in list(t.items())[:14]})
print(open("$O/kernel_stats_in_flight.json").read()[:1800])
print(open("$O/pmc_sq_per_kernel.json").read()[:2500])
PY
