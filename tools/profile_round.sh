#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel statistics of the same command,
# the two PMC passes (counters only, separate runs) behind roofline.traffic, the per-hop latencies of the
# streaming session.  Everything lands in gpurun_out/prof_$1; copy what is to be judged into profiles/.
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
timeout -k 10 400 $B > $O/bench.json 2> $O/bench.err || echo "bench FAILED"
timeout -k 10 300 $B --clips 1 --inflight 1 --no-cpu --no-extras > $O/bench_one_clip_per_step.json 2>> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-cpu --no-extras > $O/bench_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
# counters per launch do not depend on what else is in flight: one step at a time -- with the detector settings the
# bench uses when steps overlap (the batches), or the library defaults (the lone clip)
pmc() {  # name, then bench arguments
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$name -- $B "$@" --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pmc_f_$name.json 2> $O/pmc_f_$name.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$name -- $B "$@" --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pmc_w_$name.json 2> $O/pmc_w_$name.err
  cp $(ls $O/pmc_f_$name/*/*counter_collection.csv | tail -1) $O/pmc_fetch_size_$name.csv
  cp $(ls $O/pmc_w_$name/*/*counter_collection.csv | tail -1) $O/pmc_write_size_$name.csv
  python3 $ROOT/tools/pmc_traffic.py $O/pmc_fetch_size_$name.csv $O/pmc_write_size_$name.csv > $O/pmc_traffic_per_kernel_$name.json
  rm -rf $O/pmc_f_$name $O/pmc_w_$name
}
TH='{"lane_merge": 1, "hp_dedupe": 1}'
pmc c2x1 --clips 1
pmc c2x8 --clips 8 --tuning "$TH"
pmc c2x16 --clips 16 --tuning "$TH"
pmc c2x16_latency_settings --clips 16
rm -rf $O/stats
timeout -k 10 120 python3 $ROOT/tools/stream_latency.py --config c5 --hops 10000 > $O/stream_latency_c5.json 2> $O/lat.err
timeout -k 10 120 python3 $ROOT/tools/stream_latency.py --config realtime --hops 10000 > $O/stream_latency_realtime.json 2>> $O/lat.err
OFP_HOP_GRAPH=nodes timeout -k 10 120 python3 $ROOT/tools/stream_latency.py --config c5 --hops 5000 > $O/stream_latency_c5_five_node_graph.json 2>> $O/lat.err
head -c 900 $O/bench.json; echo; head -14 $O/kernel_stats.csv | cut -c1-150
python3 - <<PY
import json
for n in ("c2x1", "c2x8", "c2x16", "c2x16_latency_settings"):
    t = json.load(open("$O/pmc_traffic_per_kernel_%s.json" % n))
    tot = sum(v["hbm_mb_per_launch"] * v["calls"] for v in t.values())
    print(n, "PMC MB per step (3 steps profiled):", round(tot / 3), {k: (v["calls"], round(v["hbm_mb_per_launch"])) for k, v in list(t.items())[:12]})
for n in ("c5", "realtime", "c5_five_node_graph"):
    j = json.load(open("$O/stream_latency_%s.json" % n)); print(n, j["p50_us"], j["p99_us"])
PY
