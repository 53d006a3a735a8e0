"""The per-onset callables next to the built rows (VERDICT r1 "missing" 5): adjust_onset / adjust_onset_rel /
filter_data / detect_onset_region (detection.py:271-370, 454-484) and StretchFrameExtractor (data.py:195-223).
CPU: the oracle restatements against the reference's golden outputs (g19, tests/golden/make_golden_r2.py).
GPU (-m gpu): the HIP kernels through the C ABI against the same goldens and the oracle."""
import numpy as np
import pytest

import oracle
from onset_fingerprinting_amd import synth
from tests.conftest import load_golden


def _reg_audio(g):
    a = synth.c1_sine_clicks(4.0, 48000, seed=19)[:, 0]
    assert a.astype(np.float64).sum() == g["reg_xsum"], "synthetic generator drifted"
    return a


def _str_audio(g):
    a = synth.drum_hits(3, 1.0, 48000, seed=191, period=0.11)
    assert a.astype(np.float64).sum() == g["str_xsum"], "synthetic generator drifted"
    return a


def _shifts(L, ms, n):
    """The reference's draws (data.py:208-209) under the generator's seed."""
    np.random.seed(1900 + L)
    s = np.random.randint(1, int(L * ms), n)
    return s * np.random.choice((-1, 1), size=n)


REG = (("a", {}), ("b", dict(n=512, median_filter_size=9, threshold_factor=0.3)),
       ("c", dict(n=100, median_filter_size=3, threshold_factor=0.7)))
STR = (("s1", (256, 16, 0.03)), ("s2", (200, 8, 0.05)))


def test_oracle_postproc_matches_reference():
    g = load_golden("g19_postproc")
    for x, y, o, lag, mv in zip(g["adj_x"], g["adj_y"], g["adj_onsets"], g["adj_lag"], g["adj_moves"]):
        assert tuple(oracle.adjust_onset(list(o), x, y, int(lag))) == tuple(mv)
    for o, lag, want in zip(g["rel_onsets"], g["rel_lag"], g["rel_out"]):
        assert tuple(oracle.adjust_onset_rel(list(o), g["rel_x"], g["rel_y"], int(lag))) == tuple(want)
    assert np.array_equal(oracle.filter_data(g["fil_x"], "up"), g["fil_up"])
    assert np.array_equal(oracle.filter_data(g["fil_x"], "down"), g["fil_down"])
    assert np.array_equal(oracle.filter_data(g["fil_x"][:, 0], "up"), g["fil_1d_up"])
    a = _reg_audio(g)
    for name, kw in REG:
        got = [oracle.detect_onset_region(a, int(o), **kw) for o in g["reg_onsets"]]
        assert np.array_equal(got, g[f"reg_{name}"]), name
    audio, onsets = _str_audio(g), g["str_onsets"]
    for name, (L, pre, ms) in STR:
        ref = g[f"str_{name}"]
        got = oracle.stretch_frames(audio, onsets, _shifts(L, ms, len(onsets)), L, pre)
        assert got.shape == ref.shape and np.abs(got - ref).max() / np.abs(ref).max() < 1e-5
        ref1 = g[f"str_{name}_1d"]
        got1 = oracle.stretch_frames(audio[:, 1], onsets[:, 1], _shifts(L, ms, len(onsets)), L, pre)
        assert got1.shape == ref1.shape and np.abs(got1 - ref1).max() / np.abs(ref1).max() < 1e-5


@pytest.mark.gpu
def test_gpu_postproc_matches_reference():
    from onset_fingerprinting_amd import data, detection
    g = load_golden("g19_postproc")
    for x, y, o, lag, mv in zip(g["adj_x"], g["adj_y"], g["adj_onsets"], g["adj_lag"], g["adj_moves"]):
        assert detection.adjust_onset(list(o), x, y, int(lag)) == tuple(int(v) for v in mv)
    import torch
    moves = detection.adjust_onsets_device(torch.from_numpy(g["adj_x"]).cuda(), torch.from_numpy(g["adj_y"]).cuda(),
                                           torch.from_numpy(g["adj_onsets"].astype(np.int32)).cuda(),
                                           torch.from_numpy(g["adj_lag"].astype(np.int32)).cuda())
    assert np.array_equal(moves.cpu().numpy(), g["adj_moves"])
    for o, lag, want in zip(g["rel_onsets"], g["rel_lag"], g["rel_out"]):
        assert tuple(detection.adjust_onset_rel(list(o), g["rel_x"], g["rel_y"], int(lag))) == tuple(want)
    x = g["fil_x"].copy()
    assert detection.filter_data(x, "up") is x and np.array_equal(x, g["fil_up"])  # in place, as the reference
    assert np.array_equal(detection.filter_data(g["fil_x"].copy(), "down"), g["fil_down"])
    assert np.array_equal(detection.filter_data(g["fil_x"][:, 0].copy(), "up"), g["fil_1d_up"])
    with pytest.raises(RuntimeError):
        detection.filter_data(g["fil_x"].copy(), "sideways")
    with pytest.raises(TypeError):  # float32 only: a float64 array is not rounded behind the caller's back
        detection.filter_data(g["fil_x"].astype(np.float64), "up")
    xs = g["fil_x"].copy()[::2]     # a strided view is filtered in place as well
    want = oracle.filter_data(np.ascontiguousarray(xs), "down")
    assert detection.filter_data(xs, "down") is xs and np.array_equal(xs, want)
    a = _reg_audio(g)
    for name, kw in REG:
        got = [detection.detect_onset_region(a, int(o), **kw) for o in g["reg_onsets"]]
        assert np.array_equal(got, g[f"reg_{name}"]), name
    batch = detection.detect_onset_regions_device(torch.from_numpy(a).cuda(), torch.from_numpy(g["reg_onsets"]).cuda())
    assert np.array_equal(batch.cpu().numpy(), g["reg_a"])
    audio, onsets = _str_audio(g), g["str_onsets"]
    for name, (L, pre, ms) in STR:
        np.random.seed(1900 + L)
        got = data.StretchFrameExtractor(L, pre, ms)(audio, onsets)
        ref = g[f"str_{name}"]
        assert got.shape == ref.shape and got.dtype == np.float32
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-4   # north_star: 1e-4 relative (the reference resamples in fp32)
        np.random.seed(1900 + L)
        got1 = data.StretchFrameExtractor(L, pre, ms)(audio[:, 1].copy(), onsets[:, 1])
        assert got1.shape == g[f"str_{name}_1d"].shape
        assert np.abs(got1 - g[f"str_{name}_1d"]).max() / np.abs(g[f"str_{name}_1d"]).max() < 1e-4
    with pytest.raises(NotImplementedError):
        data.StretchFrameExtractor(256, 16, use_min_onset=False)
