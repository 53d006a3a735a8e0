"""BASELINE config 5: 2-ch 48 kHz stream, per-hop detect in a captured hipGraph.
Per hop: H2D of hop x C samples (pinned) -> k_stream -> D2H of the onset count.
Prints p50 / p99 latency per hop over N hops as one JSON line."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from onset_fingerprinting_amd import detection, synth  # noqa: E402

SR, C = 48000, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
x = synth.drum_hits(C, hops * B / SR + 0.1, SR, seed=4, period=0.31)
od = detection.AmplitudeOnsetDetector(C, B, sr=SR)
od.init_minmax_tracker(x[: int(0.5 * SR)])
host = torch.from_numpy(x[: hops * B].reshape(hops, B, C)).pin_memory()
hop_in = torch.empty((B, C), dtype=torch.float32, device="cuda")
rec = torch.empty((hops * C, 16), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
cnt_host = torch.zeros(1, dtype=torch.int64).pin_memory()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    od.process(hop_in, 1, 0, None, rec, cnt)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        od.process(hop_in, 1, 0, None, rec, cnt)
    lat = np.empty(hops)
    for i in range(hops):
        t0 = time.perf_counter()
        hop_in.copy_(host[i], non_blocking=True)
        g.replay()
        cnt_host.copy_(cnt, non_blocking=True)
        s.synchronize()
        lat[i] = time.perf_counter() - t0
print(json.dumps({"config": f"C5 streaming: {C} ch @ {SR} Hz, hop {B}, hipGraph per hop (H2D hop + k_stream + D2H count)",
                  "hops": hops, "p50_us": float(np.percentile(lat, 50) * 1e6),
                  "p99_us": float(np.percentile(lat, 99) * 1e6), "mean_us": float(lat.mean() * 1e6),
                  "hop_budget_us": B / SR * 1e6, "onsets": int(cnt_host.item())}))
