"""Print the kernel timeline of the LAST pipeline step from a rocprofv3 kernel trace:
start offset (from the step's k_transpose_in), duration and the gap to the previous kernel's end.

    python tools/trace_last.py <rocprof output dir> [n_kernels=80]
"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
last = max(i for i, r in enumerate(rows) if "k_transpose_in" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
prev_end = t0
for r in rows[last:last + n]:
    name = r["Kernel_Name"]
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_\w+)", name)
    short = m.group(1) if m else name[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %9.1f us  dur %8.1f us  gap %7.1f us" % (short, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
