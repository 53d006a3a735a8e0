"""How full is the GPU while several steps are in flight?  From a rocprofv3 kernel trace: over the
middle part of the run, the fraction of time with at least one kernel running, the average number
of kernels running, and the same for the workgroup slots they ask for (grid / 256 threads).

    python tools/trace_concurrency.py <rocprof output dir> [skip_ms=20: cut this much at both ends]
"""
import csv, glob, re, sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f))]
skip = int((float(sys.argv[2]) if len(sys.argv) > 2 else 20.0) * 1e6)
ev = []
t_lo = min(int(r["Start_Timestamp"]) for r in rows if "k_hp_candidates" in r["Kernel_Name"]) + skip
t_hi = max(int(r["End_Timestamp"]) for r in rows if "k_hp_candidates" in r["Kernel_Name"]) - skip
per = defaultdict(float)
for r in rows:
    s, e = max(int(r["Start_Timestamp"]), t_lo), min(int(r["End_Timestamp"]), t_hi)
    if e <= s:
        continue
    waves = (int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) / 64.0
    ev.append((s, 1, waves))
    ev.append((e, -1, -waves))
    m = re.search(r"(k_[a-z_0-9]+|__amd_[a-zA-Z_]+)", r["Kernel_Name"])
    per[m.group(1) if m else r["Kernel_Name"][:24]] += (e - s)
ev.sort()
n = 0
w = 0.0
last = t_lo
busy = 0
area_n = 0.0
area_w = 0.0
hist = defaultdict(float)
for t, dn, dw in ev:
    dt = t - last
    if n > 0:
        busy += dt
    area_n += n * dt
    area_w += w * dt
    hist[min(n, 12)] += dt
    n += dn
    w += dw
    last = t
T = t_hi - t_lo
print("window %.1f ms; some kernel running %.1f%% of it; kernels running on average %.2f; waves resident on average %.0f (of 1024 SIMDs x up to 8)"
      % (T / 1e6, 100.0 * busy / T, area_n / T, area_w / T))
print("time share by number of kernels running:", {k: round(100 * v / T, 1) for k, v in sorted(hist.items())})
print("kernel-time share (sum of durations / window):")
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-28s %6.2f" % (k, v / T))
