cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_stft; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/tools/perf_stft.py 1024 256 > $O/$tag.log 2>&1
  f=$(ls $O/$tag/*/*counter_collection.csv 2>/dev/null | tail -1)
  if [ -n "$f" ]; then python3 - <<PY
import csv
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open("$f")):
    if "k_stft_power" in r["Kernel_Name"]:
        key = "mlp" if "true" in r["Kernel_Name"] or "Lb1" in r["Kernel_Name"] else "nomlp"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (len(v), round(sum(v)/len(v))) for c, v in d.items()})
PY
  else echo "no csv for $set"; tail -3 $O/$tag.log; fi
done
