"""How many verification rounds the IIR stage needs, over several inputs, for a list of tunings
(the round count is data-dependent: compare averages, not one clip).

    python tools/hp_pass_stats.py "[dict(hp_candidate_offset=4), dict(hp_candidate_offset=1021)]" [seconds=30]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from onset_fingerprinting_amd import detection, synth

sr = 48000
configs = eval(sys.argv[1]) if len(sys.argv) > 1 else [dict()]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
inputs = []
for seed in range(11, 17):
    inputs.append(synth.drum_hits(8, secs, sr, seed=seed, period=0.3 + 0.07 * (seed - 11)))
inputs.append(synth.drum_hits(8, secs, sr, seed=21, amp_log_uniform=(0.05, 0.9)))
inputs.append(synth.drum_hits(8, secs, sr, seed=22, poisson_rate=4.0))
inputs.append(synth.drum_hits(8, secs, sr, seed=23, poisson_rate=1.0, amp_log_uniform=(0.02, 0.5)))
for cfg in configs:
    bd = detection.BatchDetector(8, 256, sr=sr)
    if cfg:
        bd.set_tuning(**cfg)
    passes, rep, ms, mmp, mmms = [], [], [], [], []
    for x in inputs:
        xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
        out = bd.detect(xd, want_rel=False)
        out = bd.detect(xd, want_rel=False)
        i = bd.last_info
        passes.append(i["hp_passes"])
        rep.append(i["repaired"])
        ms.append(round(i["stage_ms"]["hp"], 2))
        mmp.append(i["mm_passes"])
        mmms.append(round(i["stage_ms"]["mm"], 2))
    print(cfg, "hp rounds", passes, "mean %.2f" % np.mean(passes), "hp ms", ms, "mean %.2f" % np.mean(ms), "| mm passes", mmp, "mm ms", mmms, "mean %.2f" % np.mean(mmms), flush=True)
