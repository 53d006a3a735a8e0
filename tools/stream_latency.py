"""BASELINE config 5: per-hop latency of the WHOLE captured graph (H2D hop -> detector -> ring write +
trailing-frame rFFT + mel + fused FCNN -> D2H of {count, records, logits, mel}) through ofp_hop_push.

    python tools/stream_latency.py [--config c5|realtime] [--hops N]

  c5        2 ch @ 48 kHz, hop 256, n_fft 1024, default detector arguments (BASELINE.json configs[4])
  realtime  3 ch @ 96 kHz, hop 128, n_fft 2048, the detector arguments of realtime/audio.py:39-52
            (realtime/config.py:15,24,36,53)

Latency = wall time of one ofp_hop_push call (host copy into the pinned hop buffer, graph launch,
stream synchronise, copy-out of the result block), measured around the ctypes call.  Prints one JSON
line; the onsets of the replay are checked against a second, untimed replay after a reset (the
timing loop must not disturb the result)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from onset_fingerprinting_amd import realtime, synth  # noqa: E402
from onset_fingerprinting_amd.pipeline import seeded_fcnn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c5", choices=["c5", "realtime"])
ap.add_argument("--hops", type=int, default=10000)
ap.add_argument("--no-classifier", action="store_true")
args = ap.parse_args()
if args.config == "c5":
    C, B, SR, F, kw = 2, 256, 48000, 1024, {}
else:
    C, B, SR, F, kw = 3, 128, 96000, 2048, dict(realtime.REALTIME_DETECTOR_KWARGS)
hops = args.hops
x = synth.drum_hits(C, hops * B / SR + 0.6, SR, seed=4, period=0.31)
clf = None if args.no_classifier else seeded_fcnn(40, 8)
sess = realtime.HopSession(C, B, sr=SR, n_fft=F, n_mels=40, classifier=clf, **kw)
blocks = np.ascontiguousarray(x[: hops * B].reshape(hops, B, C))


def run(timed):
    sess.reset()
    sess.init_minmax_tracker(x[: int(0.5 * SR)])
    lat = np.empty(hops)
    onsets = 0
    for i in range(hops):
        t0 = time.perf_counter()
        onsets += sess.push_raw(blocks[i])
        lat[i] = time.perf_counter() - t0
    return lat, onsets


run(False)  # warm: first-touch of every buffer, clocks up
lat, onsets = run(True)
_, onsets2 = run(False)
assert onsets == onsets2 and onsets > 0, (onsets, onsets2)
print(json.dumps({
    "config": f"{args.config}: {C} ch @ {SR} Hz, hop {B}, n_fft {F}, 40 mel, "
              f"{'FCNN(40-10-10-10-8)' if clf is not None else 'no classifier'}; one hipGraph per hop "
              + ("(H2D hop + k_hop_begin + k_stream_par + k_hop_spectral + D2H result block)"
                 if os.environ.get("OFP_HOP_GRAPH", "")[:1] == "n" else
                 "(ONE kernel node, k_hop_fused: detector workgroup + one spectral workgroup per channel, hop and "
                 "result block in pinned host memory)") +
              f", ring buffer {sess.ring_samples} rows on the device",
    "detector_kwargs": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()} or "reference defaults",
    "hops": hops, "p50_us": float(np.percentile(lat, 50) * 1e6), "p90_us": float(np.percentile(lat, 90) * 1e6),
    "p99_us": float(np.percentile(lat, 99) * 1e6), "max_us": float(lat.max() * 1e6), "mean_us": float(lat.mean() * 1e6),
    "hop_budget_us": B / SR * 1e6, "onsets": int(onsets)}))
