#!/usr/bin/env python3
"""Golden vectors added in round 2, captured from the reference (build container only):

    python tests/golden/make_golden_r2.py

  g17_realtime_sets   detect_onsets_amplitude with the realtime arguments of realtime/audio.py:39-52
                      (hipass_freq=0, slow_ar=(8000, 8000), thresholds 0.45 / 0.45, block 128 @ 96 kHz,
                      3 channels) at fast attacks >= 1 sample, where the result does not depend on the last
                      ulp of the host's log10 and must be reproduced exactly; plus
                      the per-block records of AmplitudeOnsetDetector.__call__ for one of them.
  g18_backtrack_py    AmplitudeOnsetDetector(backtrack=True).__call__ per block: the PYTHON backtracking loop
                      (detection.py:800-825, `i` advanced before the loop at :813) executed by the reference
                      itself on top of a ring-buffer STAND-IN for the absent loopmate.CircularArray
                      (tests/golden/_refload.py) -- the loop bound therefore stays "parity unpinned".
Only inputs' recipes (seeded generators of onset_fingerprinting_amd.synth) and the reference's outputs are stored.
"""
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from _refload import load_reference  # noqa: E402
from onset_fingerprinting_amd import synth  # noqa: E402

warnings.filterwarnings("ignore")
ref = load_reference(ring_stand_in=True)
det = ref.detection


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}: {path.stat().st_size / 1024:.1f} KiB")


from make_golden_r2_cfg import G17_CASES, G18_CASES, RT  # noqa: E402


def g17():
    out = {}
    for name, (kw, sr, B) in G17_CASES.items():
        x = synth.drum_hits(3, 4.0, sr, seed=170 + len(name), period=0.37)
        out[f"{name}_xsum"] = x.astype(np.float64).sum()
        c, o, rel = det.detect_onsets_amplitude(x, block_size=B, sr=sr, **RT, **kw)
        out[f"{name}_ch"] = np.array(c, dtype=np.int64)
        out[f"{name}_on"] = np.array(o, dtype=np.int64)
        out[f"{name}_rel"] = rel[::211].copy()
        out[f"{name}_relsum"] = rel.astype(np.float64).sum(axis=0)
        print(name, len(c), "onsets")
    # per block, the streaming form (what the PortAudio callback calls, realtime/audio.py:62-64)
    kw, sr, B = G17_CASES["a3"]
    x = synth.drum_hits(3, 1.5, sr, seed=177, period=0.21)
    od = det.AmplitudeOnsetDetector(3, B, sr=sr, **RT, **kw)
    od.init_minmax_tracker(x[: int(0.1 * sr)])
    recs = []
    for i in range(len(x) // B):
        c, d, r = od(x[i * B:(i + 1) * B].copy())
        recs += [(i, int(cc), int(dd)) for cc, dd in zip(c, d)]
    out["blk_xsum"] = x.astype(np.float64).sum()
    out["blk_records"] = np.array(recs, dtype=np.int64).reshape(-1, 3)
    print("blk", len(recs), "records")
    save("g17_realtime_sets", **out)


def g18():
    sr = 48000
    out = {}
    for name, (kw, C, B) in G18_CASES.items():
        x = synth.drum_hits(C, 2.0, sr, seed=180 + C + B, period=0.19)
        od = det.AmplitudeOnsetDetector(C, B, sr=sr, **kw)
        od.init_minmax_tracker(x[: int(0.1 * sr)])
        recs, plain = [], 0
        od2 = det.AmplitudeOnsetDetector(C, B, sr=sr, **dict(kw, backtrack=False))
        od2.init_minmax_tracker(x[: int(0.1 * sr)])
        for i in range(len(x) // B):
            blk = x[i * B:(i + 1) * B].copy()
            c, d, r = od(blk)
            c2, d2, _ = od2(blk)
            recs += [(i, int(cc), int(dd)) for cc, dd in zip(c, d)]
            plain += int(np.sum(np.asarray(d) != np.asarray(d2)))
        out[f"{name}_xsum"] = x.astype(np.float64).sum()
        out[f"{name}_records"] = np.array(recs, dtype=np.int64).reshape(-1, 3)
        print(name, len(recs), "records,", plain, "moved by backtracking")
    save("g18_backtrack_py", **out)


if __name__ == "__main__":
    g17()
    g18()
