"""Print every kernel of the last `ms` milliseconds of a rocprofv3 kernel trace with its queue,
start offset and duration (to see how concurrently running pipelines interleave on the GPU).

    python tools/trace_window.py <rocprof output dir> [ms=12] [skip_ms=0: end the window this long before the last kernel]
"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows = [r for r in rows if "k_" in r["Kernel_Name"] and "rocprim" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"]) - int((float(sys.argv[3]) if len(sys.argv) > 3 else 0.0) * 1e6)
t0 = t_end - int(ms * 1e6)
queues = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t_end:
        continue
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
    print("q%d %s%-20s start %9.1f us  dur %8.1f us  grid %s" % (q, "    " * q, m.group(1) if m else r["Kernel_Name"][:20],
                                                                (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "")))
