"""The tracker stage on an input with long stretches without a max reset (sparse hits of very
different loudness): prints the stage info; run under rocprofv3 --kernel-trace to see the kernels."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from onset_fingerprinting_amd import detection, synth
x = synth.drum_hits(8, 30.0, 48000, seed=23, poisson_rate=1.0, amp_log_uniform=(0.02, 0.5))
xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
bd = detection.BatchDetector(8, 256, sr=48000)
for _ in range(3):
    out = bd.detect(xd, want_rel=False)
print(bd.last_info)
