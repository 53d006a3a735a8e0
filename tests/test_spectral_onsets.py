"""Spectral-flux onset detector.  PARITY UNPINNED against the reference (its
arithmetic is librosa, absent): the oracle restates librosa's published definitions; these tests
pin the oracle with known answers (CPU) and the HIP path against the oracle (GPU): peak positions
exact, onset envelope within 1e-4 of its maximum."""
import numpy as np
import pytest

import oracle
from onset_fingerprinting_amd import synth

SR = 96000


def clicks(seconds=3.0, sr=SR, seed=0, period=0.25):
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    x = 1e-4 * rng.standard_normal(n)
    k = np.arange(400)
    times = []
    for s in np.arange(0.3, seconds - 0.1, period):
        i = int(s * sr) + int(rng.integers(0, 500))
        x[i:i + 400] += 0.5 * np.exp(-k / 60.0) * rng.standard_normal(400)
        times.append(i)
    return x.astype(np.float32), np.array(times)


def test_oracle_known_answers():
    # peak_pick on a hand-made envelope
    x = np.zeros(100, np.float32)
    x[[10, 13, 40, 41, 90]] = [1.0, 0.9, 0.5, 0.7, 0.3]
    assert oracle.peak_pick(x, 5, 2, 5, 3, 0.1, 3).tolist() == [10, 41, 90]
    assert oracle.peak_pick(x, 5, 5, 5, 3, 0.1, 40).tolist() == [10, 90]
    # the centred STFT: frame count, zero padding, periodic Hann gain on a bin-centred tone
    n_fft, hop = 256, 32
    t = np.arange(4096)
    tone = np.cos(2 * np.pi * 16 * t / n_fft).astype(np.float32)
    D = oracle.librosa_stft_mag(tone, n_fft, hop)
    assert D.shape == (n_fft // 2 + 1, 1 + len(tone) // hop)
    assert abs(D[16, 40] - n_fft / 4) < 1e-2 and D[40, 40] < 1e-3  # sum(hann)/2 = n_fft/4
    # the detector finds the planted clicks (hop resolution, STFT centring)
    x, times = clicks()
    peaks, oe = oracle.detect_onsets_spectral(x, return_oe=True)
    assert len(peaks) == len(times) and np.all(np.abs(peaks - times) <= 4 * 32)
    assert oe.dtype == np.float32 and abs(np.percentile(oe, 99.9) - 1.0) < 1e-6


@pytest.mark.gpu
def test_device_spectral_detector_matches_oracle():
    from onset_fingerprinting_amd import detection
    for seed, kw in ((0, dict()), (1, dict(n_fft=512, hop=64)), (2, dict(n_fft=256, hop=32, sr=48000))):
        x, times = clicks(seed=seed, sr=kw.get("sr", SR))
        peaks, oe = detection.detect_onsets_spectral(x, return_oe=True, **kw)
        rp, roe = oracle.detect_onsets_spectral(x, return_oe=True, **kw)
        assert oe.shape == roe.shape and np.abs(oe - roe).max() <= 1e-4 * roe.max()
        assert np.array_equal(peaks, rp) and len(peaks) == len(times)
    # a drum-hit clip through detect_onsets(method="spectral")
    x = synth.drum_hits(1, 2.0, 48000, seed=5, period=0.2)[:, 0]
    p = detection.detect_onsets(x, sr=48000, method="spectral")
    assert np.array_equal(p, oracle.detect_onsets_spectral(x, sr=48000)) and len(p) >= 8


@pytest.mark.gpu
def test_select_rank_and_peak_pick_kernels():
    import torch
    from onset_fingerprinting_amd import _lib, detection
    L = _lib.lib()
    rng = np.random.default_rng(4)
    v = np.abs(rng.standard_normal(50001)).astype(np.float32)
    v[:100] = 0.0
    v[100:200] = v[300]  # ties
    d = torch.from_numpy(v).cuda()
    out = torch.empty(1, dtype=torch.float32).cuda()
    st = detection._stream_ptr(d.device)
    s = np.sort(v)
    for r in (0, 99, 100, 150, 25000, 49950, 50000):
        _lib.check(L.ofp_select_rank(d.data_ptr(), len(v), r, out.data_ptr(), st))
        assert float(out.cpu()[0]) == s[r], r
    x = np.abs(rng.standard_normal(20000)).astype(np.float32) ** 4
    dx = torch.from_numpy(x).cuda()
    peaks = torch.empty(len(x), dtype=torch.int64).cuda()
    cnt = torch.zeros(1, dtype=torch.int64).cuda()
    fl = torch.empty(len(x), dtype=torch.uint8).cuda()
    for args in ((30, 5, 30, 6, 0.1, 20), (0, 1, 0, 1, 0.0, 0), (360, 30, 360, 31, 0.5, 210)):
        _lib.check(L.ofp_peak_pick(dx.data_ptr(), len(x), *args[:4], args[4], args[5], peaks.data_ptr(), len(x),
                                   cnt.data_ptr(), fl.data_ptr(), st))
        got = peaks[:int(cnt.cpu()[0])].cpu().numpy()
        assert np.array_equal(got, oracle.peak_pick(x, *args)), args
