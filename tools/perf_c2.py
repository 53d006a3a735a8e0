"""Ad-hoc timing of the detector (and STFT) on the C2 workload; prints ms per call."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from onset_fingerprinting_amd import synth, detection

sr = 48000
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
x = synth.c2_drums(secs, 8, sr, seed=1)
xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
bd = detection.BatchDetector(8, 256, sr=sr)
out = bd.detect(xd)
torch.cuda.synchronize()
print("info", bd.last_info, "onsets", int(out["counts"][0]))
for it in range(3):
    t0 = time.perf_counter()
    out = bd.detect(xd, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = 8 * synth.n_frames(x.shape[0], 1024, 256)
    print(f"detect: {dt*1e3:.3f} ms  -> {frames/dt/1e6:.2f} M frames/s")
