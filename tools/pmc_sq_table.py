"""Per-kernel table from the SQ counter passes of tools/profile_r3.sh (pmc_sq_*.csv in the given directory): per launch
averages of wave-instructions, wave cycles (quad-cycles), VALU-active / wait fractions, vector-memory instructions and
the MFMA counters of the kernels that use the matrix cores (the classifier epilogue of k_stft_power, k_mlp).

    python tools/pmc_sq_table.py <dir>  > pmc_sq_per_kernel.json
"""
import csv, glob, json, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_sq_*.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    g = lambda c: (sum(d[c]) / len(d[c])) if c in d and d[c] else None
    row = {"launches_profiled": max(len(v) for v in d.values())}
    for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES",
              "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM",
              "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA"):
        v = g(c)
        if v is not None:
            row[c] = round(v)
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        if g("SQ_ACTIVE_INST_VALU") is not None:
            row["valu_active_over_wave_cycles"] = round(g("SQ_ACTIVE_INST_VALU") / wc, 3)
        if g("SQ_WAIT_INST_ANY") is not None:
            row["wait_inst_over_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / wc, 3)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES"):
        row["mfma_busy_over_busy_cycles"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES"), 5)
    out[k] = row
out = dict(sorted(out.items(), key=lambda kv: -(kv[1].get("SQ_INSTS_VALU") or 0) * kv[1]["launches_profiled"]))
print(json.dumps(out, indent=1))
