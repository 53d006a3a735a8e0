"""CPU experiment (oracle arithmetic): fraction of chunk starts at which NONE of R staggered candidates with a
warm-up of W samples is in the true state (a "break"), for several R and W.

    python tools/cand_break_stats.py [seconds] [spacing]
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from onset_fingerprinting_amd import synth  # noqa: E402
from oracle import detector as od  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
SR = 48000
lib = ctypes.CDLL(str(Path(__file__).resolve().parents[1] / "oracle" / "libofp_oracle.so"))
b, a = od.butter_hp_f32(2000.0, 4, SR)
b = np.asarray(b, np.float32); a = np.asarray(a, np.float32)
x = synth.c2_drums(seconds, 8, SR, seed=1)
V = ctypes.c_void_p


def run(xs, z):
    xs = np.ascontiguousarray(xs, np.float32)
    y = np.empty_like(xs)
    lib.oracle_lfilter4(V(xs.ctypes.data), V(y.ctypes.data), V(b.ctypes.data), V(a.ctypes.data), V(z.ctypes.data), ctypes.c_long(len(xs)), 1)
    return z


Ws = [24576, 40960, 61440, 81920, 122880]
Rs = [1, 2, 4, 8, 16]
WMAX = max(Ws)
miss = {(R, W): 0 for R in Rs for W in Ws}
total = 0
for c in range(8):
    xc = np.ascontiguousarray(x[:, c])
    n = len(xc)
    truth = {}
    z = np.zeros(4, np.float32)
    for p in range(0, n - L + 1, L):
        truth[p] = z.copy()
        z = run(xc[p:p + L], z)
    for s in range((WMAX // L + 2) * L, n - L, L):
        total += 1
        t = truth[s].tobytes()
        for W in Ws:
            hit_at = None
            for r in range(max(Rs)):
                st = s - W - 8 * r
                z = run(xc[st:s], np.zeros(4, np.float32))
                if z.tobytes() == t:
                    hit_at = r
                    break
            for R in Rs:
                if hit_at is None or hit_at >= R:
                    miss[(R, W)] += 1
print(f"{total} chunk starts (8 ch x {seconds} s, every {L} samples); breaks per 1000 chunk starts:")
print("R \\ W " + "".join(f"{W:9d}" for W in Ws))
for R in Rs:
    print(f"{R:5d} " + "".join(f"{1000.0 * miss[(R, W)] / total:9.1f}" for W in Ws))
