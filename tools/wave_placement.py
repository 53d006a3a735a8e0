"""Where and when the waves of the speculative IIR launch (k_hp_candidates) ran: per-wave start/end
(s_memtime) and hardware ids, recorded by the kernel itself when OFP_HP_PROBE names a file.

    python tools/wave_placement.py [tuning dict]      # one C2 detector step, then the summary
"""
import os, sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from onset_fingerprinting_amd import synth, detection

sr = 48000
x = synth.c2_drums(60.0, 8, sr, seed=1)
xd = torch.from_numpy(x).cuda().unsqueeze(0).contiguous()
bd = detection.BatchDetector(8, 256, sr=sr)
if len(sys.argv) > 1:
    bd.set_tuning(**eval(sys.argv[1]))
bd.detect(xd)
path = "/tmp/ofp_probe.txt"
os.environ["OFP_HP_PROBE"] = path
bd.detect(xd)
del os.environ["OFP_HP_PROBE"]
a = np.loadtxt(path, dtype=np.uint64)
t0, t1, hw, xcc = a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 3].astype(np.int64), a[:, 4].astype(np.int64)
base = t0.min()
dur = (t1 - t0)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
xc = xcc & 15
print("waves %d; kernel span %.0f ticks; wave duration min/median/max %.0f / %.0f / %.0f ticks; start spread %.0f ticks"
      % (len(a), (t1.max() - base), dur.min(), np.median(dur), dur.max(), (t0.max() - base)))
key = list(zip(xc, se, sh, cu, simd))
cnt = collections.Counter(key)
per_simd = collections.Counter(cnt.values())
print("waves per SIMD -> number of SIMDs:", dict(sorted(per_simd.items())))
cus = collections.Counter(zip(xc, se, sh, cu))
print("waves per CU -> number of CUs:", dict(sorted(collections.Counter(cus.values()).items())), "; CUs used", len(cus))
print("waves per XCC:", dict(sorted(collections.Counter(xc).items())))
# duration by how many waves share the SIMD
share = np.array([cnt[k] for k in key])
for n in sorted(set(share)):
    d = dur[share == n]
    print("  waves on a SIMD with %d wave(s): %d, duration median %.0f max %.0f" % (n, len(d), np.median(d), d.max()))
# durations by position in the launch (wave -> chain, chunk group) and by XCC
nw = len(a)
per_chain = nw // 8 if nw % 8 == 0 else None
print("median duration per XCC:", {int(k): int(np.median(dur[xc == k])) for k in sorted(set(xc))})
order = np.argsort(dur)
print("slowest 12 waves (wave, ticks, xcc, se, cu, simd):", [(int(i), int(dur[i]), int(xc[i]), int(se[i]), int(cu[i]), int(simd[i])) for i in order[-12:]])
print("fastest 6 waves:", [(int(i), int(dur[i])) for i in order[:6]])
q = np.percentile(dur, [5, 25, 50, 75, 90, 95, 99])
print("percentiles 5/25/50/75/90/95/99:", [int(v) for v in q])
if per_chain:
    w_in_chain = np.arange(nw) % per_chain
    bins = np.array_split(np.arange(per_chain), 8)
    print("median by position in the chain (8 bins):", [int(np.median(dur[np.isin(w_in_chain, b)])) for b in bins])
    print("median by chain:", [int(np.median(dur[np.arange(nw) // per_chain == c])) for c in range(8)])
