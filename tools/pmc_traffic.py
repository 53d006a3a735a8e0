"""Per-kernel HBM-side traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
separate runs, counters only), as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes:
both counters are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so fetched bytes =
2 x FETCH_SIZE x 1024 (an upper bound for lane-private 16-B loads), written bytes = WRITE_SIZE x 1024.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > traffic.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m:
            continue
        a = acc[m.group(1)]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    n = fetch[k][0]
    f_kb = fetch[k][1] / n
    w_kb = write[k][1] / write[k][0] if k in write and write[k][0] else 0.0
    out[k] = dict(calls=n, fetch_kb=f_kb, write_kb=w_kb, hbm_mb_per_launch=(2 * f_kb + w_kb) * 1024 / 1e6)
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_mb_per_launch"]))
print(json.dumps(out, indent=1))
