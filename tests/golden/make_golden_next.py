#!/usr/bin/env python3
"""Golden vectors for the SURVEY.md section 8(f) "next" rows, captured from the reference.

Run in the build container only:   python tests/golden/make_golden_next.py
  g5_groups   detection.find_onset_groups   (detection.py:131-189)
  g12_xcorr   detection.cross_correlation_lag (detection.py:195-268)
  g13_fix     detection.fix_onsets / adjust_onset (detection.py:299-352, 373-451)
  g14_model_variants  model.CNN / model.CCCNN with batch_norm / pool / groups / group
Only inputs and the reference's outputs are stored.
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

warnings.filterwarnings("ignore")
ref = load_reference()
det = ref.detection


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}: {path.stat().st_size / 1024:.1f} KiB")


def detector_like(rng, n_hits, C, period, spread, drop=0.1, dup=0.05, B=256):
    """Onset lists ordered as detect_onsets_amplitude orders them: by block, then channel."""
    recs = []
    for h in range(n_hits):
        t0 = 5000 + h * period + int(rng.integers(0, period // 4))
        for c in range(C):
            if rng.random() < drop:
                continue
            recs.append((t0 + int(rng.integers(0, spread)), c))
            if rng.random() < dup:
                recs.append((recs[-1][0] + int(rng.integers(300, 900)), c))
    recs.sort(key=lambda r: (r[0] // B, r[1]))
    return [r[0] for r in recs], [r[1] for r in recs]


def g5():
    rng = np.random.default_rng(55)
    out, k = {}, 0
    lists = [detector_like(rng, 40, 4, 9000, 400), detector_like(rng, 60, 8, 4000, 1500, drop=0.3),
             detector_like(rng, 25, 3, 20000, 90, drop=0.0, dup=0.3),
             detector_like(rng, 30, 5, 1200, 800, drop=0.2)]
    # not sorted at all: the function never sorts (abs() in detection.py:165)
    s = rng.integers(0, 50000, 120).tolist()
    lists.append((s, rng.integers(0, 6, 120).tolist()))
    # from the reference detector itself (the C2 slice of g4)
    g4 = np.load(HERE / "g4_end_to_end.npz")
    lists.append((g4["c2_on"].tolist(), g4["c2_ch"].tolist()))
    for onsets, channels in lists:
        for (md, mc, cc) in [(1000, 3, None), (1000, 3, 0), (400, 2, 1), (0, 1, None), (150, 1, 2),
                             (5000, 4, None), (1000, 9, None)]:
            if cc is not None and cc > max(channels):
                continue
            r = det.find_onset_groups(list(onsets), list(channels), max_distance=md, min_channels=mc,
                                      close_channel=cc)
            out[f"c{k}_onsets"] = np.asarray(onsets, np.int64)
            out[f"c{k}_channels"] = np.asarray(channels, np.int64)
            out[f"c{k}_args"] = np.asarray([md, mc, -1 if cc is None else cc], np.int64)
            out[f"c{k}_none"] = np.asarray(r is None)
            out[f"c{k}_groups"] = np.zeros((0, max(channels) + 1), np.int64) if r is None else r.astype(np.int64)
            k += 1
    out["n_cases"] = np.asarray(k)
    save("g5_groups", **out)


def lagged_pair(rng, n, lag, noise=0.05):
    base = np.zeros(n + 400, np.float32)
    t = np.arange(160)
    base[200:360] = (np.exp(-t / 35.0) * np.sin(2 * np.pi * t / 17.0)).astype(np.float32)
    x = base[100:100 + n] + noise * rng.standard_normal(n).astype(np.float32)
    y = base[100 - lag:100 - lag + n] + noise * rng.standard_normal(n).astype(np.float32)
    return x.astype(np.float32), y.astype(np.float32)


def g12():
    rng = np.random.default_rng(1212)
    out, k = {}, 0
    for n in (64, 200, 257, 600):
        for lag in (-20, 0, 7, 33):
            x, y = lagged_pair(rng, n, lag)
            for kw in (dict(legal_lags=(-40, 40)), dict(legal_lags=(0, 50), d=1, take_abs=True),
                       dict(onsets=(100, 100 + lag), onset_tolerance=30),
                       dict(onsets=(100, 104 + lag), onset_tolerance=12, d=2, normalization_cutoff=25),
                       dict(onsets=(100, 100 + lag), onset_tolerance=50, take_abs=True, d=1),
                       dict(legal_lags=(5, 5))):
                if n == 64 and "onsets" in kw and kw["onset_tolerance"] > 30:
                    continue
                r = det.cross_correlation_lag(x.copy(), y.copy(), **kw)
                out[f"c{k}_x"], out[f"c{k}_y"] = x, y
                ll = kw.get("legal_lags")
                on = kw.get("onsets")
                out[f"c{k}_args"] = np.asarray([
                    0 if ll is None else 1, ll[0] if ll else 0, ll[1] if ll else 0,
                    0 if on is None else 1, on[0] if on else 0, on[1] if on else 0,
                    kw.get("d", 0), kw.get("normalization_cutoff", 10), kw.get("onset_tolerance", 50),
                    int(kw.get("take_abs", False))], np.int64)
                out[f"c{k}_none"] = np.asarray(r is None)
                out[f"c{k}_lag"] = np.asarray(0 if r is None else int(r), np.int64)
                k += 1
    out["n_cases"] = np.asarray(k)
    save("g12_xcorr", **out)


def g13():
    from onset_fingerprinting_amd import synth
    out, k = {}, 0
    for case, kw in enumerate([dict(), dict(d=1, take_abs=True), dict(onset_direction="up", d=1),
                               dict(zero_left=True, onset_tolerance=20, normalization_cutoff=15),
                               dict(filter_size=3, shift_onsets=4, onset_direction="down", d=1),
                               dict(d=2, onset_tolerance=45)]):
        seed = 1313 + case
        audio, onsets = synth.sensor_hits(seed)  # regenerated by the tests from the seed
        fixed = det.fix_onsets(audio.copy(), onsets.copy(), **kw)
        out[f"c{k}_seed"] = np.asarray(seed)
        out[f"c{k}_xsum"] = np.asarray(np.float64(audio.astype(np.float64).sum()))
        out[f"c{k}_onsets"], out[f"c{k}_fixed"] = onsets, np.asarray(fixed, np.int64)
        out[f"c{k}_args"] = np.asarray([kw.get("filter_size", 5), kw.get("d", 0),
                                        {None: 0, "up": 1, "down": 2}[kw.get("onset_direction")],
                                        int(kw.get("take_abs", False)), int(kw.get("zero_left", False)),
                                        kw.get("normalization_cutoff", 10), kw.get("onset_tolerance", 30),
                                        kw.get("shift_onsets", 0)], np.int64)
        k += 1
    out["n_cases"] = np.asarray(k)
    save("g13_fix", **out)


from make_golden_next_cfg import G14  # noqa: E402


def g14():
    """model.CNN / model.CCCNN with the constructor options of model.py:62-67, 451-456."""
    import torch
    torch.manual_seed(14)
    out = {}
    for name, (cls, kw) in G14.items():
        m = getattr(ref.model, cls)(**kw)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 2.0)
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.GroupNorm)):
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)
        m.eval()
        x = torch.randn(5, kw["channels"], kw["input_size"])
        with torch.no_grad():
            y = m(x)
        for k, v in m.state_dict().items():
            out[f"{name}/{k}"] = v.numpy()
        out[f"{name}/x"], out[f"{name}/y"] = x.numpy(), y.numpy()
    save("g14_model_variants", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g5", "g12", "g13", "g14"]
    for w in which:
        globals()[w]()
