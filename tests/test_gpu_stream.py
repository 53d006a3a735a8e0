"""BASELINE config 5 on the GPU: the per-hop streaming session (ring buffer + ONE captured hipGraph
per hop: detector -> trailing-frame rFFT -> mel -> fused FCNN -> D2H) against the CPU oracle fed the
same hops, through the C ABI (ofp_hop_*).

Bars: onsets index-exact, relative envelope bit-exact (integer / canon arithmetic); mel bands and
logits within 1e-4 RELATIVE of the fp64 oracle, element-wise (mel: every band; logits: every value
above 1e-3 of the largest, plus norm-wise for all), and bit-identical to the dense GPU kernel on
the same stream.
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from onset_fingerprinting_amd import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def replay(sess, odet, x, B):
    """Feed x hop by hop to the session and to the oracle detector; returns the session's outputs
    and the oracle's (channels, absolute onsets, rel)."""
    nb = len(x) // B
    got = dict(ch=[], on=[], rel=[], mel=[], logits=[])
    exp = dict(ch=[], on=[], rel=[])
    for i in range(nb):
        hop = np.ascontiguousarray(x[i * B:(i + 1) * B])
        r = sess(hop)
        got["ch"] += [int(v) for v in r["channels"]]
        got["on"] += [int(v) for v in r["onsets"]]
        got["rel"].append(r["rel"])
        got["mel"].append(r["mel"])
        got["logits"].append(r["logits"])
        c, d, rel = odet(hop)
        exp["ch"] += [int(v) for v in c]
        exp["on"] += [i * B + int(v) for v in d]  # realtime/audio.py:65: current_index + delta
        exp["rel"].append(rel)
    return got, exp, nb


def spectral_reference(x, nb, B, F, sr, n_mels, classifier):
    """mel / logits of the trailing F samples after every hop (zeros before the stream starts)."""
    C = x.shape[1]
    xp = np.concatenate([np.zeros((F - B, C), np.float32), x[: nb * B]]) if F >= B else x[B - F: nb * B]
    P = oracle.dense_power_frames(xp, F, B)                       # [C, nb, bins] float64
    mel = P @ oracle.mel_filterbank(sr, F, n_mels).astype(np.float64).T
    sd = {k: v.numpy() for k, v in classifier.state_dict().items()}
    logits = oracle.fcnn_forward(sd, mel.reshape(-1, n_mels)).reshape(C, nb, -1)
    return xp, mel, logits


def check_spectral(got, mel_ref, log_ref):
    mel = np.stack(got["mel"], axis=1)        # [C, nb, n_mels]
    logits = np.stack(got["logits"], axis=1)  # [C, nb, n_out]
    assert mel.shape == mel_ref.shape and logits.shape == log_ref.shape
    live = mel_ref > 0  # (frames of pure zeros before the stream starts have zero bands)
    assert np.array_equal(mel[~live], mel_ref[~live])
    assert (np.abs(mel - mel_ref)[live] / mel_ref[live]).max() < RTOL
    scale = np.abs(log_ref).max()
    assert np.abs(logits - log_ref).max() / scale < RTOL
    big = np.abs(log_ref) >= 1e-3 * scale
    assert (np.abs(logits - log_ref)[big] / np.abs(log_ref)[big]).max() < RTOL
    return mel, logits


@pytest.mark.parametrize("cfg", [
    # BASELINE config 5 as stated: 2 ch @ 48 kHz, hop 256, the metric's 1024-point frame, default detector
    dict(C=2, B=256, sr=48000, F=1024, seconds=2.5, kw={}),
    # the reference's own realtime setup: 3 ch @ 96 kHz, hop 128, N_FFT 2048 (realtime/config.py:15,24,36,53)
    # with the detector arguments of realtime/audio.py:39-52
    dict(C=3, B=128, sr=96000, F=2048, seconds=2.1, kw="realtime"),
])
@pytest.mark.parametrize("graph", ["fused", "nodes"])
def test_hop_session_matches_the_oracle_hop_by_hop(cfg, graph, monkeypatch):
    """graph = "fused": the default, ONE kernel node per hop (detector workgroup + one spectral workgroup
    per channel, hop and result block in pinned host memory); "nodes": the five-node graph (H2D, begin,
    detector, spectral, D2H) that shapes too large for the fused kernel take."""
    from onset_fingerprinting_amd import realtime
    monkeypatch.setenv("OFP_HOP_GRAPH", graph)
    from onset_fingerprinting_amd.data import MelBank, stft_power_mel_mlp_dense
    from onset_fingerprinting_amd.pipeline import seeded_fcnn
    C, B, sr, F = cfg["C"], cfg["B"], cfg["sr"], cfg["F"]
    kw = dict(realtime.REALTIME_DETECTOR_KWARGS) if cfg["kw"] == "realtime" else {}
    x = synth.drum_hits(C, cfg["seconds"], sr, seed=4, period=0.23)
    clf = seeded_fcnn(40, 8)
    sess = realtime.HopSession(C, B, sr=sr, n_fft=F, n_mels=40, classifier=clf, want_rel=True, **kw)
    okw = {k: v for k, v in kw.items()}
    odet = oracle.OracleDetector(C, B, sr=sr, **okw)
    warm = x[: int(0.1 * sr)]
    sess.init_minmax_tracker(warm)
    odet.init_minmax_tracker(warm)
    got, exp, nb = replay(sess, odet, x, B)
    assert nb * B >= 2 * sr  # at least two seconds, hop by hop
    assert len(exp["on"]) > 5
    assert got["ch"] == exp["ch"] and got["on"] == exp["on"]
    assert np.array_equal(bits(np.concatenate(got["rel"])), bits(np.concatenate(exp["rel"])))
    xp, mel_ref, log_ref = spectral_reference(x, nb, B, F, sr, 40, clf)
    mel, logits = check_spectral(got, mel_ref, log_ref)
    # ... and bit for bit what the dense kernel gives on the same (zero-prefixed) stream
    _, dmel, dlog = stft_power_mel_mlp_dense(torch.from_numpy(xp).cuda()[None], F, B, MelBank(sr, F, 40),
                                             clf.device_mlp(0))
    assert np.array_equal(bits(dmel[0].cpu().numpy()), bits(mel))
    assert np.array_equal(bits(dlog[0].cpu().numpy()), bits(logits))
    # the ring buffer holds the stream (realtime/audio.py:97)
    assert np.array_equal(sess.audio(5 * B + 3), x[nb * B - (5 * B + 3): nb * B])
    assert sess.current_index == nb * B
    sess.close()


def test_ring_buffer_wraps_and_reset_restores_the_initial_state():
    from onset_fingerprinting_amd import realtime
    C, B, sr, F = 2, 64, 48000, 256
    x = synth.drum_hits(C, 0.4, sr, seed=9, period=0.05)
    sess = realtime.HopSession(C, B, sr=sr, n_fft=F, ring_seconds=(5 * B + 17) / sr, want_rel=True)
    R = sess.ring_samples
    assert R == 5 * B + 17  # not a multiple of the hop: writes straddle the end of the ring
    nb = len(x) // B
    first = [sess(np.ascontiguousarray(x[i * B:(i + 1) * B])) for i in range(nb)]
    assert np.array_equal(sess.audio(R), x[nb * B - R: nb * B])
    # the frame after a wrap reads across the seam: compare the last hop's mel with the oracle's
    P = oracle.dense_power_frames(x[nb * B - F: nb * B], F, B)[:, 0]
    mel_ref = P @ oracle.mel_filterbank(sr, F, 40).astype(np.float64).T
    assert (np.abs(first[-1]["mel"] - mel_ref) / mel_ref).max() < RTOL
    sess.reset()
    again = [sess(np.ascontiguousarray(x[i * B:(i + 1) * B])) for i in range(nb)]
    for a, b in zip(first, again):
        assert np.array_equal(a["onsets"], b["onsets"]) and np.array_equal(bits(a["rel"]), bits(b["rel"]))
        assert np.array_equal(bits(a["mel"]), bits(b["mel"]))
    assert first[0]["logits"] is None  # no classifier given
    sess.close()


def test_hop_session_argument_errors():
    from onset_fingerprinting_amd import realtime
    from onset_fingerprinting_amd._lib import OnsetFPError
    sess = realtime.HopSession(2, 64, sr=48000, n_fft=256, ring_seconds=0.1)
    with pytest.raises(ctypes.ArgumentError):
        sess(np.zeros((64, 2), np.float64))  # the reference's ndpointer check (detection.py:521-526)
    with pytest.raises(ValueError):
        sess(np.zeros((63, 2), np.float32))
    sess.submit(np.zeros((64, 2), np.float32))
    with pytest.raises(OnsetFPError):
        sess.submit(np.zeros((64, 2), np.float32))  # one hop in flight per session
    sess.collect()
    with pytest.raises(OnsetFPError):
        sess.collect()
    with pytest.raises(OnsetFPError):
        realtime.HopSession(2, 64, sr=48000, n_fft=300)
    sess.close()


@pytest.mark.parametrize("kw", [dict(), dict(hipass_freq=0, on_threshold=6.0, off_threshold=4.0),
                                dict(backtrack=True, backtrack_buffer_size=512), "realtime"])
def test_phase_split_block_kernel_equals_the_one_lane_per_channel_kernel(kw, monkeypatch):
    """k_stream_par (phases staged in LDS, recurrences on separate lanes) against k_stream (one lane per
    channel walks everything): same bytes for rel, same records, same carried state -- several blocks per
    call, odd channel counts, with and without the high-pass, manual thresholds, backtracking."""
    from onset_fingerprinting_amd import detection, realtime
    if kw == "realtime":
        kw = dict(realtime.REALTIME_DETECTOR_KWARGS)
    C, B, sr = 5, 96, 48000
    x = synth.drum_hits(C, 1.5, sr, seed=21, period=0.11)
    nb = len(x) // B
    xd = torch.from_numpy(x[: nb * B]).cuda()
    outs = []
    for mode in ("seq", "par"):
        monkeypatch.setenv("OFP_STREAM_KERNEL", mode)
        od = detection.AmplitudeOnsetDetector(C, B, sr=sr, **kw)
        od.init_minmax_tracker(x[: 4000])
        rel = torch.empty((nb * B, C), dtype=torch.float32, device="cuda")
        rec = torch.zeros((4096, 16), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        k = 0
        for n in (1, 7, 2, nb - 10):  # several calls of several blocks each: state hand-over both ways
            od.process(xd[k * B:(k + n) * B], n, k * B, rel[k * B:(k + n) * B], rec, cnt)
            k += n
        torch.cuda.synchronize()
        outs.append((rel.cpu().numpy().copy(), rec.cpu().numpy().copy(), int(cnt.item()), od._state.cpu().numpy().copy()))
    assert outs[0][2] == outs[1][2] and outs[0][2] > 10
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0]))
    assert np.array_equal(outs[0][1], outs[1][1])
    n_state = outs[0][3].size - B * C * 4  # (the trailing scratch block of the state is not state)
    assert np.array_equal(outs[0][3][:n_state], outs[1][3][:n_state])
    # and the oracle agrees
    odet = oracle.OracleDetector(C, B, sr=sr, **kw)
    odet.init_minmax_tracker(x[: 4000])
    exp = np.concatenate([odet(np.ascontiguousarray(x[i * B:(i + 1) * B]))[2] for i in range(nb - 0)][: k])
    assert np.array_equal(bits(outs[1][0][: k * B]), bits(exp))


@pytest.mark.parametrize("graph", ["fused", "nodes"])
def test_per_hop_onset_strength_matches_the_oracle_restatement(graph, monkeypatch):
    """N1, second half (realtime/recording.py:273-311): channel-mean frame, dB flux, tracked normalisation,
    moving max / mean, once per hop inside the session's graph.  PARITY UNPINNED: the trackers'
    arithmetic is assumed (loopmate is absent); the GPU is checked against the oracle's restatement of the
    same assumption at 1e-4 relative (+ 1e-6 absolute for values that are differences of dB terms)."""
    from onset_fingerprinting_amd import realtime
    monkeypatch.setenv("OFP_HOP_GRAPH", graph)
    C, B, sr, F = 3, 128, 96000, 2048
    x = synth.drum_hits(C, 0.6, sr, seed=31, period=0.09)
    sess = realtime.HopSession(C, B, sr=sr, n_fft=F, ring_seconds=1.0, onset_strength=dict(max_length=12, avg_length=40, ring=64),
                               **realtime.REALTIME_DETECTOR_KWARGS)
    ref = oracle.HopStrength(F, C, 12, 40, 64)
    worst = 0.0
    for i in range(len(x) // B):
        hop = np.ascontiguousarray(x[i * B:(i + 1) * B])
        got = sess(hop)["strength"]
        want = ref(hop)
        err = np.abs(got - want) / (np.abs(want) + 1e-2)   # (flux values are means of dB differences of O(1))
        worst = max(worst, float(err.max()))
        assert err.max() < RTOL, (i, got, want)
    assert worst > 0 and len(x) // B > 400
    sess.close()


@pytest.mark.parametrize("graph", ["fused", "nodes"])
def test_per_hop_tempogram_matches_the_reference_expression(graph, monkeypatch):
    """The third piece of the per-hop spectral worker (realtime/recording.py:313-327): the tempogram frame of every
    hop -- autocorrelation of the Hann-windowed last W entries of the normalised onset envelope -- against the
    reference's own expression (rfft / irfft of length 2 W - 1, float64) on the oracle's envelope history.  PARITY
    UNPINNED like the envelope (loopmate's trackers are assumed); the GPU sums the lags directly in float32."""
    from onset_fingerprinting_amd import realtime
    monkeypatch.setenv("OFP_HOP_GRAPH", graph)
    C, B, sr, F, W = 2, 128, 48000, 1024, 96
    x = synth.drum_hits(C, 0.8, sr, seed=41, period=0.07)
    sess = realtime.HopSession(C, B, sr=sr, n_fft=F, ring_seconds=1.0,
                               onset_strength=dict(max_length=12, avg_length=40, ring=128, tg_win_length=W))
    ref = oracle.HopStrength(F, C, 12, 40, 128, tg_win_length=W)
    worst = 0.0
    for i in range(len(x) // B):
        hop = np.ascontiguousarray(x[i * B:(i + 1) * B])
        got = sess(hop)
        ref(hop)
        want = ref.tempogram()
        assert got["tempogram"].shape == (W,)
        err = np.abs(got["tempogram"] - want).max()   # (normalised: the largest lag is 1)
        worst = max(worst, float(err))
        assert err < RTOL, (i, err)
        if i > W:
            assert abs(got["tempogram"][0] - 1.0) < 1e-5
    assert len(x) // B > 250 and worst > 0
    sess.close()
    # the upstream window (config.py:55: TG_WIN_LENGTH = 1024) on a short run
    sess = realtime.HopSession(C, B, sr=sr, n_fft=F, ring_seconds=1.0,
                               onset_strength=dict(max_length=12, avg_length=40, ring=1024, tg_win_length=1024))
    ref = oracle.HopStrength(F, C, 12, 40, 1024, tg_win_length=1024)
    for i in range(40):
        hop = np.ascontiguousarray(x[i * B:(i + 1) * B])
        got = sess(hop)["tempogram"]
        ref(hop)
        assert np.abs(got - ref.tempogram()).max() < RTOL
    sess.close()


def test_wide_session_takes_the_five_node_graph_and_the_one_lane_kernel():
    """200 channels x 64 samples: too many channels for the fused kernel's detector workgroup (2 C lanes) and too
    large a block for the phase-split kernel's LDS (3 x B x C floats) -- the session falls back to the five-node
    graph with the one-lane-per-channel block kernel and must give the same answers."""
    from onset_fingerprinting_amd import realtime
    C, B, sr, F = 200, 64, 48000, 256
    x = synth.drum_hits(C, 0.12, sr, seed=5, period=0.03)
    sess = realtime.HopSession(C, B, sr=sr, n_fft=F, ring_seconds=0.05, want_rel=True)
    odet = oracle.OracleDetector(C, B, sr=sr)
    got, exp, nb = replay(sess, odet, x, B)
    assert got["ch"] == exp["ch"] and got["on"] == exp["on"] and len(exp["on"]) > 20
    assert np.array_equal(bits(np.concatenate(got["rel"])), bits(np.concatenate(exp["rel"])))
    P = oracle.dense_power_frames(x[nb * B - F: nb * B], F, B)[:, 0]
    mel_ref = P @ oracle.mel_filterbank(sr, F, 40).astype(np.float64).T
    big = mel_ref >= 1e-5 * mel_ref.max()
    assert (np.abs(got["mel"][-1] - mel_ref)[big] / mel_ref[big]).max() < RTOL
    sess.close()
