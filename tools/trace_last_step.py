"""The kernels of the LAST step in a rocprofv3 kernel trace (from the call's zero fill on), in launch order:
start offset, duration, name -- what one call looks like with the GPU to itself (`bench.py --inflight 1`).

    python tools/trace_last_step.py <rocprof output dir>
"""
import csv, glob, re, sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = [int(r["Start_Timestamp"]) for r in rows if "k_zero(" in r["Kernel_Name"] or "k_zero<" in r["Kernel_Name"] or r["Kernel_Name"].endswith("k_zero")][-1]
tot = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?|__amd_[a-zA-Z_]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    print(f"{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:9.1f} us  {name}")
    tot[name] = tot.get(name, 0) + (e - s)
print("--- totals (us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v / 1e3:10.1f}  {k}")
