#!/bin/bash
# Instruction-level PMC counters per kernel of the detector on a batch of 8 C2 clips (tools/tune_detect.py c2x8).
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_detect; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/tools/tune_detect.py c2x8 "[{}]" > $O/$tag.log 2>&1
  f=$(ls $O/$tag/*/*counter_collection.csv 2>/dev/null | tail -1)
  [ -n "$f" ] && cp $f $O/$tag.csv
  rm -rf $O/$tag
done
python3 - <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$O/*.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
samples = 8 * 8 * 2904000.0
rows = []
for k, d in acc.items():
    g = lambda c: (sum(d[c]) / max(1, len(d[c])), len(d[c])) if c in d else (0.0, 0)
    valu, n = g("SQ_INSTS_VALU"); wc, _ = g("SQ_WAVE_CYCLES"); busy, _ = g("SQ_BUSY_CYCLES"); wait, _ = g("SQ_WAIT_INST_ANY")
    rd, _ = g("SQ_INSTS_VMEM_RD"); wr, _ = g("SQ_INSTS_VMEM_WR"); act, _ = g("SQ_ACTIVE_INST_VALU"); waves, _ = g("SQ_WAVES")
    rows.append((valu * n, k, n, valu, waves, wc, busy, wait, rd, wr, act))
for tot, k, n, valu, waves, wc, busy, wait, rd, wr, act in sorted(rows, reverse=True)[:14]:
    print("%-18s launches %3d  VALU/launch %10.0f (%.1f per sample*)  waves %7.0f  wave_cycles %11.0f  busy %10.0f  wait_inst/wave_cycles %.2f  vmem rd %9.0f wr %9.0f" % (k, n, valu, valu * 64 / samples, waves, wc, busy, wait / max(wc, 1), rd, wr))
PY
