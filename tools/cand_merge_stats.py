"""CPU experiment (oracle arithmetic): how fast do the R speculative IIR candidates of a chunk merge with EACH OTHER
during their warm-up, and with the true trajectory?  (How much of k_hp_candidates' work is spent on runs that are
already duplicates of a lower-numbered candidate.)

    python tools/cand_merge_stats.py [seconds] [R] [W] [L]
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from onset_fingerprinting_amd import synth  # noqa: E402
from oracle import detector as od  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 40960
L = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
SEG = 4096
SR = 48000
lib = ctypes.CDLL(str(Path(__file__).resolve().parents[1] / "oracle" / "libofp_oracle.so"))
b, a = od.butter_hp_f32(2000.0, 4, SR)
b = np.asarray(b, np.float32); a = np.asarray(a, np.float32)
x = synth.c2_drums(seconds, 8, SR, seed=1)


def run(xs, z):
    xs = np.ascontiguousarray(xs, np.float32)
    y = np.empty_like(xs)
    V = ctypes.c_void_p
    lib.oracle_lfilter4(V(xs.ctypes.data), V(y.ctypes.data), V(b.ctypes.data), V(a.ctypes.data), V(z.ctypes.data), ctypes.c_long(len(xs)), 1)
    return z


nseg = W // SEG
distinct = np.zeros(nseg + 1)
hit = 0
total = 0
merged_with_truth_at = np.zeros(nseg + 1)
for c in range(2):
    xc = np.ascontiguousarray(x[:, c])
    n = len(xc)
    # truth at every SEG boundary
    truth = {}
    z = np.zeros(4, np.float32)
    for p in range(0, n - SEG, SEG):
        truth[p] = z.copy()
        z = run(xc[p:p + SEG], z)
    for s in range(((W + 8 * R) // L + 1) * L, n - L, L):
        states = []
        for r in range(R):
            st = s - W - 8 * r
            z = np.zeros(4, np.float32)
            z = run(xc[st:s - W], z)  # bring every candidate to the common position s - W
            states.append(z)
        for j in range(nseg + 1):
            pos = s - W + j * SEG
            keys = {zz.tobytes() for zz in states}
            distinct[j] += len(keys)
            merged_with_truth_at[j] += truth[pos].tobytes() in keys
            if j < nseg:
                new = {}
                for k in keys:  # run each distinct state once
                    new[k] = run(xc[pos:pos + SEG], np.frombuffer(k, np.float32).copy())
                states = [new[zz.tobytes()] for zz in states]
        total += 1
print(f"{total} chunk starts, R={R}, W={W}, checkpoints every {SEG}")
print("position in warm-up | mean distinct candidates | fraction of chunks where a candidate equals the truth")
for j in range(nseg + 1):
    print(f"{j * SEG:7d}  {distinct[j] / total:6.2f}  {merged_with_truth_at[j] / total:6.3f}")
work_now = R * W
work_dedupe = sum(distinct[j] / total for j in range(nseg)) * SEG
print(f"warm-up steps per chunk: {work_now} now, {work_dedupe:.0f} with a dedupe every {SEG} = {work_dedupe / work_now:.2f} x")
