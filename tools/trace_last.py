"""Print per-launch kernel durations of the LAST detect call from a rocprofv3 kernel trace."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
agg = {}
for r in rows[-n:]:
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1][:40]
    if "k_jacobi" in name:
        short = "k_jacobi" + ("_hp4" if "hp4" in name else "<" + name.split("k_jacobi<")[1].split("Stage")[0].split("::")[-1] + ">")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-34s %9.1f us" % (short, d))
