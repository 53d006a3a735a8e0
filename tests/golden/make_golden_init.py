#!/usr/bin/env python3
"""Golden vectors for AmplitudeOnsetDetector.init (detection.py:842-888), captured from the reference.

Run in the build container only:   python tests/golden/make_golden_init.py
  g15_init   thresholds / mins / maxs / noise_max after init(x), and the detector's per-block
             outputs on the audio that follows (pins the follower and filter state init leaves).
Only configurations the reference runs without reading past its buffers: len(x) and sr are
multiples of the block size (its follower calls always process block_size rows, detection.py:534-537).
Inputs are regenerated from seeds by synth.init_clip; only the reference's outputs are stored.
"""
import contextlib
import io
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from _refload import load_reference  # noqa: E402
from make_golden_init_cfg import G15  # noqa: E402
from onset_fingerprinting_amd import synth  # noqa: E402

warnings.filterwarnings("ignore")
det = load_reference().detection

out = {}
for name, cfg in G15.items():
    kw, sr, B, C = cfg["kw"], cfg["sr"], cfg["B"], cfg["C"]
    x, y = synth.init_clip(cfg["seed"], C, sr, cfg["seconds"], cfg["follow_blocks"] * B, amp=cfg.get("amp", 1.0))
    out[f"{name}/xsum"] = np.array([x.astype(np.float64).sum(), y.astype(np.float64).sum()])
    d = det.AmplitudeOnsetDetector(C, B, sr=sr, **kw)
    with contextlib.redirect_stdout(io.StringIO()) as msg:
        d.init(x)
    out[f"{name}/on"] = np.asarray(d.on_threshold)
    out[f"{name}/off"] = np.asarray(d.off_threshold)
    out[f"{name}/mins"] = np.asarray(d.mins)
    out[f"{name}/maxs"] = np.asarray(d.maxs)
    out[f"{name}/noise_max"] = np.asarray(d.noise_max)
    out[f"{name}/message"] = np.array(msg.getvalue())
    ch, de, blk, rel = [], [], [], []
    for j in range(cfg["follow_blocks"]):
        c, dl, r = d(y[j * B:(j + 1) * B])
        ch += list(c)
        de += list(dl)
        blk += [j] * len(c)
        rel.append(r.copy())
    out[f"{name}/ch"] = np.asarray(ch, np.int64)
    out[f"{name}/delta"] = np.asarray(de, np.int64)
    out[f"{name}/block"] = np.asarray(blk, np.int64)
    rel = np.concatenate(rel)
    out[f"{name}/rel_stride"] = rel[::31].copy()
    out[f"{name}/rel_sum"] = rel.astype(np.float64).sum(axis=0)
    print(name, "on", out[f"{name}/on"], "onsets", len(ch))
path = HERE / "g15_init.npz"
np.savez_compressed(path, **out)
print(f"g15_init: {path.stat().st_size / 1024:.1f} KiB")
