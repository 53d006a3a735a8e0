"""GPU parity of framing / STFT / mel / MFCC against the numpy oracle and the
reference's golden vectors.  Floating point: the north-star tolerance is 1e-4
relative fp32; spectra are compared norm-wise (error relative to the largest
magnitude of the frame set), which is how an fp32 FFT's error is distributed."""
import ast

import numpy as np
import pytest
import torch

import oracle
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def data():
    from onset_fingerprinting_amd import data
    return data


def test_g6_stft_matches_reference_golden(data):
    g = load_golden("g6_stft")
    for k, case in enumerate(g["cases"]):
        name, method, L, hop, nfft, hep, onset = ast.literal_eval(str(case))
        S = data.stft(g[name], onset, L, hop, nfft, bool(hep), method)
        ref = g[f"S{k}"]
        assert S.shape == ref.shape and S.dtype == ref.dtype == np.complex64, (case, S.shape, ref.shape)
        err = np.abs(S - ref).max() / np.abs(ref).max()
        assert err < RTOL, (case, err)


def test_stft_frame(data):
    g = load_golden("g6_stft")
    w = oracle.hann_periodic(256)
    S = data.stft_frame(g["frame_x"], 256, w)
    ref = g["frame_S"]
    assert np.abs(S - ref).max() / np.abs(ref).max() < RTOL
    # n_fft > len(x): centre padding (data.py:588-589)
    w2 = oracle.pad_center(oracle.hann_periodic(256), 1024) if hasattr(oracle, "pad_center") else None
    from oracle.spectral import pad_center
    w2 = pad_center(oracle.hann_periodic(256), 1024)
    S2 = data.stft_frame(g["frame_x"], 1024, w2)
    ref2 = oracle.stft_frame(g["frame_x"], 1024, w2)
    assert np.abs(S2 - ref2).max() / np.abs(ref2).max() < RTOL


def test_stft_edges(data):
    rng = np.random.default_rng(5)
    a = rng.standard_normal(3000).astype(np.float32)
    # onset near the end: the slice audio[onset:onset+L] is short (numpy semantics)
    for method in ("zerozero", "prezero", "pre"):
        try:
            ref = oracle.stft(a, 2900, 256, 64, 256, False, method)
        except ValueError:
            continue
        S = data.stft(a, 2900, 256, 64, 256, False, method)
        assert S.shape == ref.shape
        if ref.size:  # method "pre" leaves less than one frame here: empty result, as numpy gives
            assert np.abs(S - ref).max() / max(np.abs(ref).max(), 1e-6) < RTOL
    with pytest.raises(ValueError):
        data.stft(a, 100, 256, 64, 100)  # unsupported n_fft


@pytest.mark.parametrize("n_fft,hop,C", [(1024, 256, 8), (2048, 512, 3), (256, 64, 2), (512, 128, 1), (4096, 1024, 2)])
def test_dense_power_matches_oracle(data, n_fft, hop, C):
    import torch
    rng = np.random.default_rng(n_fft)
    N = n_fft * 6 + 37
    x = (rng.standard_normal((2, N, C)) * np.exp(rng.uniform(-6, 0, (2, 1, C)))).astype(np.float32)
    P = data.stft_power_dense(torch.from_numpy(x).cuda(), n_fft, hop).cpu().numpy()
    for clip in range(2):
        ref = oracle.dense_power_frames(x[clip], n_fft, hop)  # [C, H, bins] float64
        assert P[clip].shape == ref.shape
        for c in range(C):
            err = np.abs(P[clip, c] - ref[c]).max() / ref[c].max()
            assert err < RTOL, (clip, c, err)


def elementwise_rel(got, ref, floor_frac=1e-5):
    """max |got - ref| / |ref| over the elements with |ref| >= floor_frac * max|ref|.  Why a floor: an
    fp32 FFT leaves an ABSOLUTE error of ~eps * |X|max on every bin, i.e. a relative error of
    ~2 eps / sqrt(P / Pmax) on a power bin -- 4e-5 at P = 1e-5 Pmax (measured: 1.1e-4 at 1e-6 Pmax),
    so 1e-4 relative is a meaningful bar for bins within 50 dB of the frame's largest and not below."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    big = np.abs(ref) >= floor_frac * np.abs(ref).max()
    return float((np.abs(got - ref)[big] / np.abs(ref)[big]).max()), float(big.mean())


@pytest.mark.parametrize("n_fft,hop,C", [(1024, 256, 8), (2048, 512, 3), (256, 64, 2), (512, 128, 1), (4096, 1024, 2)])
def test_dense_power_and_mel_elementwise(data, n_fft, hop, C):
    """north_star: "1e-4 relative fp32" -- ELEMENT-wise, not only norm-wise: every power bin that is at
    least 1e-5 of the largest bin of its channel (see elementwise_rel for the floor) and every mel band
    that is at least 1e-5 of the largest band (sums of non-negative terms: no cancellation beyond the
    bins' own error floor) within 1e-4 of the fp64 oracle."""
    from onset_fingerprinting_amd.data import MelBank, stft_power_mel_dense
    rng = np.random.default_rng(n_fft + 1)
    N = n_fft * 9 + 11
    # broadband noise + a strong tone + a hit: bins spread over ~8 decades
    t = np.arange(N)[:, None]
    x = (rng.standard_normal((N, C)) * 1e-3 + 0.5 * np.sin(2 * np.pi * 0.0731 * t) +
         (t > N // 2) * np.exp(-(t - N // 2) / 300.0) * rng.standard_normal((N, C))).astype(np.float32)
    mb = MelBank(48000, n_fft, 40)
    P, mel = stft_power_mel_dense(torch.from_numpy(x).cuda()[None], n_fft, hop, mb)
    P, mel = P[0].cpu().numpy(), mel[0].cpu().numpy()
    ref = oracle.dense_power_frames(x, n_fft, hop)
    fb = oracle.mel_filterbank(48000, n_fft, 40).astype(np.float64)
    for c in range(C):
        e, frac = elementwise_rel(P[c], ref[c])
        assert e < RTOL and frac > 0.05, (c, e, frac)  # (frac: the check is not vacuous)
        e, frac = elementwise_rel(mel[c], ref[c] @ fb.T)
        assert e < RTOL and frac > 0.05, ("mel", c, e, frac)


@pytest.mark.parametrize("n_fft,hop,C", [(1024, 256, 8), (2048, 512, 5), (256, 64, 2), (512, 128, 3), (4096, 1024, 2)])
def test_classifier_epilogue_is_bit_identical_to_the_separate_kernels(data, n_fft, hop, C):
    """ofp_stft_power_mel_mlp: power, mel and logits out of ONE kernel equal, bit for bit, what
    ofp_stft_power_mel followed by the FCNN gives -- for every frame count that leaves a partly
    filled 16-row tile, with and without the power / mel outputs."""
    from onset_fingerprinting_amd.data import MelBank, stft_power_mel_dense, stft_power_mel_mlp_dense
    from onset_fingerprinting_amd.pipeline import seeded_fcnn
    rng = np.random.default_rng(n_fft + 7)
    mb = MelBank(48000, n_fft, 40)
    m = seeded_fcnn(40, 8)
    mlp = m.device_mlp(0)
    for H in (1, 17, 203):
        N = n_fft + (H - 1) * hop + 3
        x = torch.from_numpy((rng.standard_normal((2, N, C)) * np.exp(rng.uniform(-5, 0, (2, 1, C)))).astype(np.float32)).cuda()
        P0, mel0 = stft_power_mel_dense(x, n_fft, hop, mb)
        log0 = m.forward_layerwise(mel0.reshape(-1, 40)).reshape(2, C, H, 8)
        P1, mel1, log1 = stft_power_mel_mlp_dense(x, n_fft, hop, mb, mlp, want_power=True, want_mel=True)
        assert torch.equal(P0, P1) and torch.equal(mel0, mel1)
        assert np.array_equal(log0.cpu().numpy().view(np.uint32), log1.cpu().numpy().view(np.uint32)), (n_fft, H)
        _, _, log2 = stft_power_mel_mlp_dense(x, n_fft, hop, mb, mlp, want_power=False, want_mel=False)
        assert torch.equal(log1, log2)


def test_dense_power_linearity_and_parseval_at_scale(data):
    """Size-independent properties on a large input (full C2 size is covered by
    bench.py's own check): Parseval against the windowed frame energy."""
    import torch
    rng = np.random.default_rng(9)
    N, C, F, hop = 480000, 4, 1024, 256
    x = rng.standard_normal((1, N, C)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    P = data.stft_power_dense(xd, F, hop)
    H = P.shape[2]
    w = oracle.hann_periodic(F)
    # Parseval for a real signal: sum_k c_k |X_k|^2 = F * sum_n (w x)^2, c = 1 at DC/Nyquist else 2
    wts = torch.full((F // 2 + 1,), 2.0, device="cuda")
    wts[0] = wts[-1] = 1.0
    lhs = (P[0] * wts).sum(-1).cpu().numpy()  # [C, H]
    for h in (0, 1, H // 2, H - 1):
        for c in range(C):
            seg = x[0, h * hop:h * hop + F, c].astype(np.float64) * w
            assert abs(lhs[c, h] - F * (seg ** 2).sum()) / (F * (seg ** 2).sum()) < RTOL
    P2 = data.stft_power_dense(2.0 * xd, F, hop)
    assert torch.allclose(P2, 4.0 * P, rtol=1e-5, atol=0)


def test_g7_frame_extractor_exact(data):
    g = load_golden("g7_frames")
    a, o = g["audio"], g["onsets"]
    assert np.array_equal(data.FrameExtractor(256, 16)(a, o), g["f1"])
    assert np.array_equal(data.FrameExtractor(256, 16, use_min_onset=False)(a, o), g["f2"])
    assert np.array_equal(data.FrameExtractor(128, 32, add_pre_samples=True)(a, o), g["f3"])
    assert np.array_equal(data.FrameExtractor(64, 8)(a[:, 0].copy(), o[:, 0]), g["f1d"])
    with pytest.raises(IndexError):
        data.FrameExtractor(256, 16)(a, np.array([[4900, 4900, 4900, 4900]]))


def test_g16_fast_frame_extractor_exact(data):
    """data.FastFrameExtractor (data.py:123-192) against the reference's outputs: 1-D / 2-D audio and
    onsets, add_pre_samples, and random shifts drawn from torch's CPU generator under the same seeds."""
    from tests.golden.make_golden_init_cfg import G16, g16_inputs
    g = load_golden("g16_fastframes")
    for name, cfg in G16.items():
        audio, onsets = g16_inputs(cfg)
        torch.manual_seed(cfg["seed"])
        fe = data.FastFrameExtractor(audio, onsets, **cfg["kw"])
        for call in range(2):
            torch.manual_seed(cfg["seed"] + 1 + call)
            got = fe()
            assert isinstance(got, torch.Tensor) and got.dtype == torch.float32 and got.device.type == "cpu"
            assert np.array_equal(got.numpy(), g[f"{name}/call{call}"]), (name, call)
    # device given: the frames stay in HBM
    audio, onsets = g16_inputs(G16["2d_min_onset"])
    fe = data.FastFrameExtractor(audio, onsets, device="cuda", **G16["2d_min_onset"]["kw"])
    assert fe().device.type == "cuda" and np.array_equal(fe().cpu().numpy(), g["2d_min_onset/call0"])
    with pytest.raises(IndexError):
        data.FastFrameExtractor(audio, np.array([[5990, 5990, 5990, 5990]]), 64, 8)


def test_mel_and_mfcc_match_oracle(data):
    """PARITY UNPINNED vs librosa (absent): compared with the oracle's restatement
    of librosa's published definition, plus known answers."""
    rng = np.random.default_rng(3)
    S = (rng.standard_normal((2, 513, 7)) + 1j * rng.standard_normal((2, 513, 7))).astype(np.complex64)
    S[:, :, 3] *= 1e-7  # exercises the top_db floor
    got = data.cspec_to_mfcc(S, 48000)
    ref = oracle.cspec_to_mfcc(S, 48000)
    assert got.shape == ref.shape == (2, 14, 7)
    assert np.abs(got - ref).max() / np.abs(ref).max() < RTOL
    fb = data.mel_filterbank(48000, 1024, 40)
    assert np.allclose(fb, oracle.mel_filterbank(48000, 1024, 40), rtol=1e-6, atol=1e-9)
    # known answers: every band is a triangle with Slaney area normalisation
    assert fb.shape == (40, 513) and (fb >= 0).all() and (fb.sum(1) > 0).all()
    S1 = data.cspec_to_mfcc(S[0], 48000, n_mels=20, n_mfcc=5, fmax=16000)
    assert np.abs(S1 - oracle.cspec_to_mfcc(S[0], 48000, n_mels=20, n_mfcc=5, fmax=16000)).max() < 1e-3


def test_window_contribution_weights(data):
    g = load_golden("g10_wcw")
    w = oracle.hann_periodic(1024)
    np.testing.assert_allclose(data.window_contribution_weights(w, 256), g["w1024_256"], rtol=1e-12)
    np.testing.assert_allclose(data.window_contribution_weights(w, 256, True), g["w1024_256_hep"], rtol=1e-12)


def test_fused_stft_mel_equals_the_two_kernels():
    """ofp_stft_power_mel == ofp_stft_power followed by ofp_mel, bit for bit (same summation
    order), for every supported n_fft, ragged frame counts, and with the power output dropped."""
    import torch
    from onset_fingerprinting_amd import data
    rng = np.random.default_rng(77)
    for n_fft, hop, C, N in ((256, 64, 3, 5000), (512, 128, 2, 7777), (1024, 256, 8, 40000), (2048, 512, 2, 30011),
                             (4096, 1024, 1, 20000)):
        x = torch.from_numpy(rng.standard_normal((2, N, C)).astype(np.float32)).cuda()
        mb = data.MelBank(48000, n_fft, 40)
        p0 = data.stft_power_dense(x, n_fft, hop)
        m0 = mb(p0)
        p1, m1 = data.stft_power_mel_dense(x, n_fft, hop, mb)
        assert torch.equal(p0, p1) and torch.equal(m0, m1), n_fft
        p2, m2 = data.stft_power_mel_dense(x, n_fft, hop, mb, want_power=False)
        assert p2 is None and torch.equal(m0, m2)
        xt = x.transpose(1, 2).contiguous()  # planar [clip][C][N]: same values, coalesced loads
        p3, m3 = data.stft_power_mel_dense(x, n_fft, hop, mb, planar=(xt.data_ptr(), N))
        assert torch.equal(p0, p3) and torch.equal(m0, m3)
        # series further apart than their length (the detector's copy has the warm-up in between)
        xw = torch.zeros((2, C, N + 1000), dtype=torch.float32, device="cuda")
        xw[:, :, 1000:] = xt
        p4, m4 = data.stft_power_mel_dense(x, n_fft, hop, mb, planar=(xw.data_ptr() + 4000, N + 1000))
        assert torch.equal(p0, p4) and torch.equal(m0, m4)


def test_sliding_frames_equal_the_plain_frame_mapping(monkeypatch):
    """Planar input with hop = n_fft / 4: a wave works through consecutive frames of a series and keeps the shared
    three quarters of the samples in registers (k_stft_power<.., SLIDE>).  Same arithmetic, so the same bits as the
    plain mapping (forced through OFP_STFT_NO_SLIDE), for runs that cross series boundaries, a frame count that
    does not fill the last workgroup, one-frame series and the classifier epilogue."""
    import torch
    from onset_fingerprinting_amd import data
    from onset_fingerprinting_amd.pipeline import seeded_fcnn
    rng = np.random.default_rng(78)
    mlp = seeded_fcnn(40, 8).device_mlp(0)
    for n_fft, C, n_clips, N in ((1024, 5, 3, 30000), (1024, 8, 16, 48000), (1024, 2, 1, 1024), (1024, 3, 7, 1280),
                                 (512, 4, 2, 20000), (256, 3, 2, 9000), (2048, 3, 2, 40000), (2048, 64, 1, 6144)):
        hop = n_fft // 4
        x = torch.from_numpy(rng.standard_normal((n_clips, N, C)).astype(np.float32)).cuda()
        xt = x.transpose(1, 2).contiguous()
        mb = data.MelBank(48000, n_fft, 40)
        outs = []
        for no_slide in (False, True):
            if no_slide:
                monkeypatch.setenv("OFP_STFT_NO_SLIDE", "1")
            else:
                monkeypatch.delenv("OFP_STFT_NO_SLIDE", raising=False)
            p, m = data.stft_power_mel_dense(x, n_fft, hop, mb, planar=(xt.data_ptr(), N))
            p2, m2, l2 = data.stft_power_mel_mlp_dense(x, n_fft, hop, mb, mlp, want_power=True, planar=(xt.data_ptr(), N))
            outs.append((p.clone(), m.clone(), p2.clone(), m2.clone(), l2.clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b), (n_fft, C, n_clips, N)
        assert torch.equal(outs[0][0], data.stft_power_dense(x, n_fft, hop))  # and as the interleaved input gives


def test_interleaved_sliding_frames_equal_the_plain_frame_mapping(monkeypatch):
    """The caller's interleaved input with 4 / 8 channels and hop = n_fft / 4 (k_stft_power<.., SLIDE, IL>): the C frame
    slots of a run share one block of hop x C new samples per frame, fetched once and handed over through an LDS stage.
    Same arithmetic, so the same bits as the plain mapping (OFP_STFT_NO_SLIDE) and as the planar sliding form -- for
    runs that cross clip boundaries, one- and two-frame clips, two runs per workgroup (C = 4; n_fft 256 with C = 8),
    the classifier epilogue, and an input that is not 16-byte aligned (which takes the plain mapping)."""
    import torch
    from onset_fingerprinting_amd import data
    from onset_fingerprinting_amd.pipeline import seeded_fcnn
    rng = np.random.default_rng(79)
    mlp = seeded_fcnn(40, 8).device_mlp(0)
    for n_fft, C, n_clips, N in ((1024, 8, 5, 30000), (1024, 4, 7, 21000), (1024, 8, 40, 1024), (1024, 4, 9, 1280),
                                 (1024, 8, 3, 1536), (512, 4, 3, 20000), (512, 8, 2, 9999), (256, 8, 3, 9000),
                                 (256, 4, 2, 5003), (1024, 8, 2, 300000)):
        hop = n_fft // 4
        base = torch.from_numpy(rng.standard_normal(n_clips * N * C + 4).astype(np.float32)).cuda()
        mb = data.MelBank(48000, n_fft, 40)
        for off in (0, 1):  # off = 1: the same values 4 bytes further on (not 16-byte aligned)
            x = base[off:off + n_clips * N * C].view(n_clips, N, C)
            if off:
                x.copy_(base[:n_clips * N * C].view(n_clips, N, C).clone())
            xt = x.transpose(1, 2).contiguous()
            outs = []
            for no_slide in (False, True):
                if no_slide:
                    monkeypatch.setenv("OFP_STFT_NO_SLIDE", "1")
                else:
                    monkeypatch.delenv("OFP_STFT_NO_SLIDE", raising=False)
                p, m = data.stft_power_mel_dense(x, n_fft, hop, mb)
                p2, m2, l2 = data.stft_power_mel_mlp_dense(x, n_fft, hop, mb, mlp, want_power=True)
                outs.append((p.clone(), m.clone(), p2.clone(), m2.clone(), l2.clone()))
            for a, b in zip(*outs):
                assert torch.equal(a, b), (n_fft, C, n_clips, N, off)
            monkeypatch.delenv("OFP_STFT_NO_SLIDE", raising=False)
            p3, m3 = data.stft_power_mel_dense(x, n_fft, hop, mb, planar=(xt.data_ptr(), N))
            assert torch.equal(outs[0][0], p3) and torch.equal(outs[0][1], m3)
