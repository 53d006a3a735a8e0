"""Sweep detector tuning on the C2 workload (or `channels seconds clips` given after the list);
prints stage times per configuration.

    python tools/perf_sweep.py "[dict(), dict(hp_quad=1)]" [channels=8] [seconds=60] [clips=1]
"""
import sys, time, itertools
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from onset_fingerprinting_amd import synth, detection

sr = 48000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
clips = int(sys.argv[4]) if len(sys.argv) > 4 else 1
x = synth.c2_drums(secs, C, sr, seed=1)
xd = torch.from_numpy(x).cuda().unsqueeze(0).repeat(clips, 1, 1).contiguous()
configs = eval(sys.argv[1]) if len(sys.argv) > 1 else [dict()]
for cfg in configs:
    bd = detection.BatchDetector(C, 256, sr=sr)
    if cfg:
        bd.set_tuning(**cfg)
    out = bd.detect(xd)
    torch.cuda.synchronize()
    best = None
    for it in range(3):
        t0 = time.perf_counter()
        out = bd.detect(xd, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    i = bd.last_info
    st = {k: round(v, 2) for k, v in i["stage_ms"].items()}
    print(cfg, "->", round(best * 1e3, 2), "ms", "passes", i["hp_passes"], i["ar_passes"], i["mm_passes"], "rep", i["repaired"], st, flush=True)
