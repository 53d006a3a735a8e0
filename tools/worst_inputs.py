"""Inputs that defeat the speculation (no loud events to coalesce at): how slow does it get?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from onset_fingerprinting_amd import detection, synth
sr = 48000
rng = np.random.default_rng(0)
n = 30 * sr
cases = {
    "noise floor only": (1e-3 * rng.standard_normal((n, 8))).astype(np.float32),
    "steady 3 kHz tone + noise": (0.2 * np.sin(2 * np.pi * 3000 * np.arange(n) / sr)[:, None] + 1e-3 * rng.standard_normal((n, 8))).astype(np.float32),
    "one hit then 25 s of floor": synth.drum_hits(8, 30.0, sr, seed=3, period=25.0),
    "hits every 5 s": synth.drum_hits(8, 30.0, sr, seed=4, period=5.0),
    "drum hits 0.5 s (C2-like)": synth.drum_hits(8, 30.0, sr, seed=1, period=0.5),
}
bd = detection.BatchDetector(8, 256, sr=sr)
for name, x in cases.items():
    xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
    out = bd.detect(xd, want_rel=False)
    out = bd.detect(xd, want_rel=False)
    i = bd.last_info
    print(f"{name:32s} total {i['stage_ms']['total']:8.2f} ms  hp {i['stage_ms']['hp']:7.2f} ({i['hp_passes']} rounds)  "
          f"ar {i['stage_ms']['ar']:5.2f} ({i['ar_passes']})  mm {i['stage_ms']['mm']:6.2f} ({i['mm_passes']})  onsets {int(out['counts'][0])}", flush=True)
