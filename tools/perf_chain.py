"""Timing of the post-detection chain on a C4-style batch (clips x 4 ch x 10 s,
Poisson hit times are replaced by periodic hits so that groups exist): detect -> group ->
fix_onsets -> group windows -> CNN forward, all on the device.  Prints one JSON line.

    python tools/perf_chain.py [n_clips=64] [seconds=10]
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from onset_fingerprinting_amd import detection, model, synth

SR, C, B, W, PRE = 48000, 4, 256, 256, 32
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
base = [synth.drum_hits(C, secs, SR, seed=300 + i, period=0.23 + 0.01 * i) for i in range(8)]
x = torch.from_numpy(np.stack([base[i % 8] * (0.5 + 0.5 * ((i * 7) % 11) / 11) for i in range(n_clips)])).cuda()
x = x.contiguous()
bd = detection.BatchDetector(C, B, sr=SR)
torch.manual_seed(0)
cnn = model.CNN(W, 2, channels=C).eval()


def chain():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    out = bd.detect(x, want_rel=False)
    ev[1].record()
    groups, n_groups = detection.group_onsets_device(out, C, max_distance=1000, min_channels=C, cap_groups=64)
    ev[2].record()
    status = detection.fix_onsets_device(x, groups, d=1, take_abs=True, n_groups=n_groups, max_section=1200)
    ev[3].record()
    win, off = detection.group_windows_device(x, groups, n_groups, W, PRE)
    ev[4].record()
    logits = cnn(win)
    ev[5].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    return ms, out, n_groups, status, off, logits


chain()
best = None
for _ in range(5):
    ms, out, n_groups, status, off, logits = chain()
    if best is None or sum(ms) < sum(best):
        best = ms
n_on = int(out["counts"].sum())
n_g = int(n_groups.sum())
n_rows = int(off[-1])
frames = n_clips * C * synth.n_frames(x.shape[1], 1024, 256)
print(json.dumps(dict(
    workload=f"{n_clips} clips x {C} ch x {secs:g} s", onsets=n_on, groups=n_g, fixed=int((status == 0).sum()),
    window_rows=n_rows, cnn_rows=int(logits.shape[0]), hops=frames,
    ms=dict(detect=round(best[0], 3), group=round(best[1], 3), fix_onsets=round(best[2], 3),
            windows=round(best[3], 3), cnn=round(best[4], 3)),
    detect_Mhops_per_s=round(frames / best[0] / 1e3, 2), stage_info=bd.last_info["stage_ms"])))
