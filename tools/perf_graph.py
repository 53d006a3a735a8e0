"""One detector call as a captured hipGraph against the same call enqueued launch by launch (lone C2 clip by default).

    python tools/perf_graph.py [channels=8] [seconds=60] [clips=1] ['<tuning json>']
"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from onset_fingerprinting_amd import synth, detection

sr = 48000
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
clips = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tuning = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
x = torch.from_numpy(np.stack([synth.c2_drums(secs, C, sr, seed=1 + 7919 * i) for i in range(clips)])).cuda().contiguous()
bd = detection.BatchDetector(C, 256, sr=sr)
if tuning:
    bd.set_tuning(**tuning)
out = bd.detect(x, cap_per_clip=4096)
torch.cuda.synchronize()


def timed(fn, n=20):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def eager():
    bd.detect(x, out=out, cap_per_clip=4096)


def enq():
    bd.enqueue(x, out=out, cap_per_clip=4096)
    torch.cuda.synchronize()
    bd.complete(x, out)


g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    bd.enqueue(x, out=out, cap_per_clip=4096)


def graph():
    g.replay()
    torch.cuda.synchronize()
    bd.complete(x, out)


ref = out["rel"].clone()
res = dict(eager_ms=round(timed(eager), 3), enqueue_sync_complete_ms=round(timed(enq), 3), graph_replay_ms=round(timed(graph), 3))
res["same_bytes"] = bool(torch.equal(ref, out["rel"]))
res["info"] = {k: bd.last_info[k] for k in ("hp_passes", "ar_passes", "mm_passes", "repeated_host_verified")}
print(json.dumps(res))
