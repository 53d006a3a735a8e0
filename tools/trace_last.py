"""Print per-launch kernel durations of the LAST detect call from a rocprofv3 kernel trace."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
for r in rows[-n:]:
    name = r["Kernel_Name"]
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_\w+)", name)
    short = m.group(1) if m else name[:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-30s %9.1f us" % (short, d))
