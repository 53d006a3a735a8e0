"""Per-kernel statistics of ONE regime from a rocprofv3 kernel trace of bench.py: only the launches of the timed,
in-flight window (between the first and the last launch of the steady state: the first and last 15 % of the k_transpose_in
launches are cut off, which drops the warm-up steps that start together and the steps run alone afterwards), plus how
full the GPU was in that window.  The verdict of round 2 asked for this instead of the all-launches mixture.

    python tools/trace_regime.py <rocprof output dir>  > kernel_stats_in_flight.json
"""
import csv, glob, json, re, sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
anchor = sorted(int(r["Start_Timestamp"]) for r in rows if "k_transpose_in" in r["Kernel_Name"])
cut = max(1, int(0.15 * len(anchor)))
t_lo, t_hi = anchor[cut], anchor[-cut]
per = defaultdict(list)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t_lo or s >= t_hi:
        continue
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?|__amd_[a-zA-Z_]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:32]
    per[name].append(e - s)
    ev.append((s, 1))
    ev.append((min(e, t_hi), -1))
ev.sort()
n, last, busy, area = 0, t_lo, 0, 0.0
for t, d in ev:
    if n > 0:
        busy += t - last
    area += n * (t - last)
    n += d
    last = t
T = t_hi - t_lo
steps = sum(1 for a in anchor if t_lo <= a < t_hi)
out = {"window_ms": T / 1e6, "steps_started_in_window": steps, "ms_per_step": T / 1e6 / max(steps, 1),
       "some_kernel_running_frac": busy / T, "kernels_running_avg": area / T, "kernels": {}}
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    out["kernels"][k] = {"launches": len(v), "launches_per_step": round(len(v) / max(steps, 1), 2), "avg_us": round(sum(v) / len(v) / 1e3, 1),
                         "min_us": round(min(v) / 1e3, 1), "max_us": round(max(v) / 1e3, 1),
                         "kernel_ms_per_step": round(sum(v) / 1e6 / max(steps, 1), 3)}
print(json.dumps(out, indent=1))
