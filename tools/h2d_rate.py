"""The PCIe-inclusive rate (never bench.py's `value`): the C2 clip (92 MB) uploaded from pinned host
memory before every step, (a) upload then step, one after the other, (b) uploads on their own
stream overlapping the previous step (double buffered).

    python tools/h2d_rate.py [steps=20]
"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from onset_fingerprinting_amd import synth
from onset_fingerprinting_amd.pipeline import FingerprintPipeline

SR, C, F, H = 48000, 8, 1024, 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = torch.from_numpy(synth.c2_drums(60.0, C, SR, seed=1)).unsqueeze(0).contiguous().pin_memory()
pipe = FingerprintPipeline(C, F, H, SR, 40, device=0)
frames = C * pipe.n_frames(x.shape[1])
dev = [torch.empty_like(x, device="cuda") for _ in range(2)]
dev[0].copy_(x, non_blocking=True)
pipe.run(dev[0])
torch.cuda.synchronize()

t0 = time.perf_counter()
for _ in range(steps):
    dev[0].copy_(x, non_blocking=True)
torch.cuda.synchronize()
h2d_ms = (time.perf_counter() - t0) / steps * 1e3

t0 = time.perf_counter()
for _ in range(steps):
    dev[0].copy_(x, non_blocking=True)
    pipe.run(dev[0])
torch.cuda.synchronize()
serial_ms = (time.perf_counter() - t0) / steps * 1e3

up = torch.cuda.Stream()
ev = [torch.cuda.Event(), torch.cuda.Event()]
with torch.cuda.stream(up):
    dev[0].copy_(x, non_blocking=True)
    ev[0].record(up)
t0 = time.perf_counter()
for i in range(steps):
    cur, nxt = i & 1, (i + 1) & 1
    with torch.cuda.stream(up):
        dev[nxt].copy_(x, non_blocking=True)
        ev[nxt].record(up)
    torch.cuda.current_stream().wait_event(ev[cur])
    pipe.run(dev[cur])
torch.cuda.synchronize()
overlap_ms = (time.perf_counter() - t0) / steps * 1e3
print(json.dumps({"clip_MB": round(x.numel() * 4 / 1e6, 1), "h2d_ms": round(h2d_ms, 3),
                  "h2d_GBps": round(x.numel() * 4 / 1e9 / (h2d_ms / 1e3), 1),
                  "upload_then_step_ms": round(serial_ms, 3), "frames_per_s_upload_then_step": round(frames / (serial_ms / 1e3)),
                  "upload_overlapping_previous_step_ms": round(overlap_ms, 3),
                  "frames_per_s_overlapped": round(frames / (overlap_ms / 1e3))}))
