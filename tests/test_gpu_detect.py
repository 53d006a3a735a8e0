"""GPU parity of the HIP onset detector against the CPU oracle (and through it
against the reference's golden vectors).  Every call goes through the C ABI of
libonsetfp.so.  Integer outputs must match exactly; the relative envelope must
match bit-for-bit because both sides evaluate include/ofp_math.h."""
import ctypes

import numpy as np
import pytest

import oracle
from onset_fingerprinting_amd import synth
from tests.conftest import load_golden
from tests.golden.make_golden_cfg import G3_CONFIGS
from tests.golden.make_golden_init_cfg import G15
from tests.test_oracle_golden import run_init_case

pytestmark = pytest.mark.gpu

SR = 48000


@pytest.fixture(scope="module")
def det():
    from onset_fingerprinting_amd import detection
    return detection


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def oracle_records(x, **kw):
    c, o, rel = oracle.detect_onsets_amplitude(x, **kw)
    return np.array(c, np.int64), np.array(o, np.int64), rel


def check_clip(det, x, tuning=None, **kw):
    recs, rel, info = det.detect_batch(x[None], tuning=tuning, **kw)
    c, o, orel = oracle_records(x, **kw)
    assert np.array_equal(recs[0]["channel"], c), (recs[0]["channel"][:20], c[:20])
    assert np.array_equal(recs[0]["sample"], o)
    assert rel[0].shape == orel.shape
    assert np.array_equal(bits(rel[0]), bits(orel)), "relative envelope not bit-identical"
    return info, len(c)


def test_c1_offline_matches_oracle_and_golden(det):
    x = synth.c1_sine_clicks(10.0, SR, seed=0)
    g = load_golden("g4_end_to_end")
    for B in (128, 256):
        c, o, rel = det.detect_onsets_amplitude(x, block_size=B, sr=SR)
        # the reference's own answer (captured in tests/golden)
        assert np.array_equal(np.array(c), g[f"c1_B{B}_ch"])
        assert np.array_equal(np.array(o), g[f"c1_B{B}_on"])
        oc, oo, orel = oracle_records(x, block_size=B, sr=SR)
        assert np.array_equal(bits(rel), bits(orel))
        assert isinstance(c, list) and rel.dtype == np.float32 and rel.shape == (480000 // B * B, 1)


def test_c2_slice_default_tuning(det):
    x = synth.c2_drums(10.0, 8, SR, seed=1)
    g = load_golden("g4_end_to_end")
    info, n = check_clip(det, x, block_size=256, sr=SR)
    assert n == len(g["c2_ch"]) and n > 100
    c, o, _ = det.detect_onsets_amplitude(x, block_size=256, sr=SR)
    assert np.array_equal(np.array(c), g["c2_ch"]) and np.array_equal(np.array(o), g["c2_on"])


@pytest.mark.parametrize("C", [4, 8, 64])
@pytest.mark.parametrize("kw", [
    dict(block_size=256, sr=SR),
    dict(block_size=100, sr=SR, cooldown=20),                       # rows per block not a power of two, not a multiple of 64 / C
    dict(block_size=32, sr=SR, hipass_freq=0.0, cooldown=0),
    dict(block_size=256, sr=SR, backtrack=True, backtrack_buffer_size=512, backtrack_smooth_size=5),
])
def test_interleaved_layout_matches_planar_and_oracle(det, C, kw):
    """Throughput layout with 4 / 8 / 64 channels: tracker, crossing pass and backtracking read the caller's interleaved `rel`
    (warm-up rows from a buffer of their own; the joint falls inside a chunk), no planar copy of it is written; on request
    (interleaved 2 / 3) the IIR stage reads the caller's interleaved audio, no planar copy of the input.  == the planar
    layout == the oracle, bit for bit.  (Without the high-pass -- the B = 32 case -- the input side keeps its planar
    copy: the dB pass reads it.)"""
    x = synth.c2_drums(2.7, C, SR, seed=11 + C)[: 129_003]   # (not a whole number of blocks)
    for tuning in (dict(lane_merge=1, hp_dedupe=1), dict(lane_merge=1, mm_chunk=1999, mm_warm=5000, mm_span=3),
                   dict(lane_merge=1, hp_dedupe=1, interleaved=3), dict(lane_merge=1, hp_dedupe=1, interleaved=2),
                   # the IIR stage on the interleaved audio (on request) with the restart of the stream (n_w) inside a
                   # chunk and a sub-chunk; few candidates and a short warm-up: whole re-runs and early joins
                   dict(hp_dedupe=1, hp_chunk=3072, hp_warm=8192, hp_candidates=4, hp_early=1, interleaved=3),
                   dict(hp_dedupe=1, hp_chunk=1001 * 4, hp_warm=9000, hp_candidate_offset=-1, interleaved=2),
                   dict(lane_merge=1, hp_dedupe=1, walk_through=-1),
                   dict(lane_merge=1, hp_dedupe=1, line_stores=-1),
                   # complete-line stores with whole re-runs, early joins and sub-chunk items in the same waves
                   dict(lane_merge=1, hp_chunk=4096, hp_warm=6000, hp_candidates=2, hp_early=1, ar_chunk=1024, ar_span=4),
                   dict(lane_merge=1, hp_chunk=8192, hp_warm=-1, hp_candidates=1, ar_chunk=2048, ar_warm=-1),
                   # walk-through chunks as their own pass 0, with spans and chunk lengths that put group ends, the joint
                   # of the two `rel` pieces and the stream's end in every position; no warm-up: the repairs run through them
                   dict(lane_merge=1, ar_chunk=2048, ar_span=4, mm_chunk=2048, mm_span=5, mm_warm=3000),
                   dict(lane_merge=1, ar_chunk=4096, ar_warm=-1, ar_span=8, mm_chunk=1024, mm_warm=-1, mm_span=16),
                   dict(lane_merge=1, hp_dedupe=1, interleaved=-1)):
        check_clip(det, x, tuning=tuning, **kw)
    recs, rel, info = det.detect_batch(np.stack([x, x[::-1].copy()]), tuning=dict(lane_merge=1), warm=0, **kw)
    recs2, rel2, _ = det.detect_batch(np.stack([x, x[::-1].copy()]), tuning=dict(lane_merge=1, interleaved=-1), warm=0, **kw)
    assert np.array_equal(bits(rel), bits(rel2))
    for a, b in zip(recs, recs2):
        assert np.array_equal(a["sample"], b["sample"]) and np.array_equal(a["channel"], b["channel"])


@pytest.mark.parametrize("tuning", [
    dict(hp_chunk=1024, hp_warm=2048, ar_chunk=2048, ar_warm=60000, mm_chunk=4096, mm_warm=50000),
    # no speculative warm-up at all: every chunk starts wrong, the repair passes must fix it
    dict(hp_chunk=20000, hp_warm=-1, ar_chunk=30000, ar_warm=-1, mm_chunk=25000, mm_warm=-1),
    dict(hp_chunk=777, hp_warm=100, ar_chunk=5000, ar_warm=1000, mm_chunk=999, mm_warm=10),
])
def test_time_parallel_passes_are_exact(det, tuning):
    x = synth.c2_drums(3.0, 4, SR, seed=5)
    info, n = check_clip(det, x, tuning=tuning, block_size=256, sr=SR)
    assert n > 10
    if tuning["hp_warm"] == -1:
        assert info["repaired"] > 0  # the fallback path really ran


@pytest.mark.parametrize("kw", [
    dict(block_size=128, hipass_freq=0, on_threshold=6.0, off_threshold=4.0, cooldown=20),
    dict(block_size=32, hipass_freq=1000.0, on_threshold=0.2, off_threshold=0.15, cooldown=0, fast_ar=(2.0, 966.0)),
    dict(block_size=512, hipass_freq=2000.0, on_threshold=6.1, off_threshold=2.3, cooldown=300, floor=-60.0),
    dict(block_size=128, hipass_freq=0, fast_ar=(0.3, 800.0), slow_ar=(8000.0, 8000.0), on_threshold=0.45,
         off_threshold=0.45, cooldown=9600),
    dict(block_size=64, backtrack=True, backtrack_buffer_size=128, backtrack_smooth_size=5),
    dict(block_size=128, backtrack=True, backtrack_buffer_size=128, backtrack_smooth_size=3, on_threshold=6.0,
         off_threshold=4.0),
])
def test_parameter_sets(det, kw):
    x = synth.drum_hits(3, 2.0, SR, seed=21, period=0.23)
    check_clip(det, x, sr=SR, **kw)


def test_batch_of_ragged_and_edge_cases(det):
    # several clips at once; a clip shorter than the warm-up; N not a multiple of B
    xs = np.stack([synth.c4_clip(i, 1.3, 4, SR) for i in range(5)])[:, :61234]
    recs, rel, _ = det.detect_batch(xs, block_size=256, sr=SR)
    for i in range(5):
        c, o, orel = oracle_records(xs[i], block_size=256, sr=SR)
        assert np.array_equal(recs[i]["channel"], c) and np.array_equal(recs[i]["sample"], o)
        assert np.array_equal(recs[i]["clip"], np.full(len(c), i, np.int32))
        assert np.array_equal(bits(rel[i]), bits(orel))
    # shorter than one block: nothing processed (detection.py:74-75)
    x = synth.c4_clip(0, 1.0, 2, SR)[:100]
    c, o, r = det.detect_onsets_amplitude(x, block_size=128, sr=SR)
    assert c == [] and o == [] and r.shape == (0, 2)
    # shorter than the warm-up window
    x = synth.drum_hits(2, 0.3, SR, seed=3, period=0.1)
    check_clip(det, x, block_size=128, sr=SR)


@pytest.mark.parametrize("k", range(len(G3_CONFIGS)))
def test_streaming_detector_blocks_match_golden_and_oracle(det, k):
    g = load_golden("g3_detector_blocks")
    for tag, x in (("a", g["x"]), ("b", g["x2"])):
        cfg = dict(G3_CONFIGS[k])
        B = cfg.pop("block_size")
        od = det.AmplitudeOnsetDetector(x.shape[1], B, sr=SR, **cfg)
        oo = oracle.OracleDetector(x.shape[1], B, sr=SR, **cfg)
        if k % 2 == 1:
            od.init_minmax_tracker(x[: int(0.05 * SR)])
            oo.init_minmax_tracker(x[: int(0.05 * SR)])
        recs = []
        for i in range(0, len(x) - B + 1, B):
            c, d, r = od(x[i:i + B])
            c2, d2, r2 = oo(x[i:i + B])
            assert np.array_equal(c, c2) and np.array_equal(d, d2), (i, c, c2, d, d2)
            assert np.array_equal(bits(r), bits(r2))
            recs += [(i // B, int(a), int(b)) for a, b in zip(c, d)]
        assert np.array_equal(np.array(recs, np.int64).reshape(-1, 3), g[f"rec_{k}{tag}"])


def test_streaming_multi_block_call_and_backtrack(det):
    import torch
    x = synth.drum_hits(4, 1.0, SR, seed=31, period=0.17)
    B = 64
    kw = dict(backtrack=True, backtrack_buffer_size=128, backtrack_smooth_size=5, cooldown=500)
    od = det.AmplitudeOnsetDetector(4, B, sr=SR, **kw)
    oo = oracle.OracleDetector(4, B, sr=SR, **kw)
    nb = len(x) // B
    xd = torch.from_numpy(x[: nb * B]).cuda()
    rel = torch.empty_like(xd)
    rec = torch.empty((4096, 16), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    od.process(xd, nb, 0, rel, rec, cnt)
    n = int(cnt.item())
    got = rec.cpu().numpy().view(det.ONSET_DTYPE).reshape(-1)[:n]
    exp = []
    rels = []
    for i in range(nb):
        c, d, r = oo(x[i * B:(i + 1) * B])
        rels.append(r)
        exp += [(int(a), i * B + int(b)) for a, b in zip(c, d)]
    assert n == len(exp) and n > 5
    assert [(int(a), int(b)) for a, b in zip(got["channel"], got["sample"])] == exp
    assert np.array_equal(bits(rel.cpu().numpy()), bits(np.concatenate(rels)))


def test_legacy_symbols_match_oracle(det):
    from onset_fingerprinting_amd import _lib
    L = _lib.lib()
    g = load_golden("g1_ar_envelope")
    x = g["x"]
    B, C = 64, x.shape[1]
    fp = ctypes.POINTER(ctypes.c_float)
    for k, (a, r) in enumerate(g["pairs"]):
        y = np.full((B, C), -70.0, np.float32)
        outs = []
        for i in range(0, len(x), B):
            xb = np.ascontiguousarray(x[i:i + B])
            L.ar_envelope(xb.ctypes.data_as(fp), y.ctypes.data_as(fp), np.float32(1 / a), np.float32(1 / r), C, B)
            outs.append(y.copy())
        assert np.array_equal(bits(np.concatenate(outs)), bits(g[f"y{k}"])), (a, r)
    g2 = load_golden("g2_minmax")
    x2, B2 = g2["x"], int(g2["B"])
    mn, mx = np.zeros(C, np.float32), np.full(C, 10, np.float32)
    for i in range(0, len(x2), B2):
        xb = np.ascontiguousarray(x2[i:i + B2])
        L.minmax_envelope(xb.ctypes.data_as(fp), mn.ctypes.data_as(fp), mx.ctypes.data_as(fp),
                          np.float32(1e-4), np.float32(1e-5), np.float32(2.0), B2, C)
        assert np.array_equal(bits(mn), bits(g2["mins"][i // B2])) and np.array_equal(bits(mx), bits(g2["maxs"][i // B2]))
    g9 = load_golden("g9_backtrack")
    lp = ctypes.POINTER(ctypes.c_long)
    for k in range(3):
        d = g9["deltas0"].copy()
        buf = np.ascontiguousarray(g9["buf"])
        ch = np.ascontiguousarray(g9["channels"])
        L.backtrack_onsets(buf.ctypes.data_as(fp), ch.ctypes.data_as(lp), d.ctypes.data_as(lp),
                           g9[f"alpha_{k}"], g9[f"tol_{k}"], buf.shape[0], len(ch), buf.shape[1], int(g9["B"]))
        assert np.array_equal(d, g9[f"deltas_{k}"])


def test_error_behaviour(det):
    with pytest.raises(ctypes.ArgumentError):
        det.AmplitudeOnsetDetector(2, 32)(np.zeros((32, 2), np.float64))
    with pytest.raises(AssertionError):
        det.AmplitudeOnsetDetector(2, 128, backtrack=True, backtrack_buffer_size=64)


def test_follower_and_filter_classes_match_reference_goldens(det):
    """AREnvelopeFollower / MinMaxEnvelopeFollower / ButterworthFilter: the reference's
    callable surface (detection.py:487-592), bit-for-bit against goldens g1, g2, g11."""
    g = load_golden("g1_ar_envelope")
    x = g["x"]
    for k, (a, r) in enumerate(g["pairs"][:2]):
        f = det.AREnvelopeFollower(np.full((64, x.shape[1]), -70.0, dtype=np.float32), a, r)
        ys = [f(np.ascontiguousarray(x[i:i + 64])).copy() for i in range(0, len(x), 64)]
        assert np.array_equal(bits(np.concatenate(ys)), bits(g[f"y{k}"]))
        assert f(np.ascontiguousarray(x[:64])) is f.y
    g2 = load_golden("g2_minmax")
    mm = det.MinMaxEnvelopeFollower(x0=np.array([[0, 10]] * 8).T, alpha_min=1e-4, alpha_max=1e-5, minmin=2)
    B2 = int(g2["B"])
    for i in range(0, len(g2["x"]), B2):
        mi, ma = mm(np.ascontiguousarray(g2["x"][i:i + B2]))
        assert np.array_equal(bits(mi), bits(g2["mins"][i // B2])) and np.array_equal(bits(ma), bits(g2["maxs"][i // B2]))
    with pytest.raises(ctypes.ArgumentError):
        mm(np.zeros((B2, 8), np.float64))
    g11 = load_golden("g11_lfilter")
    for k in range(3):
        cut, sr = g11[f"cfg{k}"]
        f = det.ButterworthFilter(cut, 3, 4, int(sr), "high")
        assert np.array_equal(f.b, g11[f"b{k}"]) and np.array_equal(f.a, g11[f"a{k}"])
        ys = [f(g11["x"][i:i + 500]) for i in range(0, 3000, 500)]
        assert np.array_equal(bits(np.concatenate(ys)), bits(g11[f"y{k}"]))
        assert np.array_equal(bits(f.zi), bits(g11[f"zi{k}"]))
    # default order 2 (detection.py:490) against scipy itself
    from scipy import signal as sig
    f2 = det.ButterworthFilter(1500.0, 3, sr=48000)
    y2 = f2(g11["x"])
    b, a = sig.butter(2, 1500.0, btype="high", fs=48000)
    ref, _ = sig.lfilter(np.float32(b), np.float32(a), g11["x"], axis=0, zi=np.zeros((2, 3), np.float32))
    assert np.array_equal(bits(y2), bits(ref))


@pytest.mark.parametrize("name", sorted(G15))
def test_init_calibration_matches_oracle_and_reference_golden(det, name, capsys):
    """AmplitudeOnsetDetector.init (detection.py:842-888) on the device: thresholds and the detector's
    outputs on the audio that follows equal the oracle's bit for bit (same arithmetic canon), and
    the reference's (g15) within float32 rounding of the thresholds, onsets identical; the message
    is the reference's."""
    g, cfg = load_golden("g15_init"), G15[name]
    d, _, ch, de, blk, rel = run_init_case(
        lambda C, B, sr, kw: det.AmplitudeOnsetDetector(C, B, sr=sr, **kw), cfg)
    printed = capsys.readouterr().out
    o, _, och, ode, oblk, orel = run_init_case(
        lambda C, B, sr, kw: oracle.OracleDetector(C, B, sr=sr, **kw), cfg)
    for a, b in ((d.on_threshold, o.on_threshold), (d.off_threshold, o.off_threshold), (d.mins, o.mins),
                 (d.maxs, o.maxs), (d.noise_max, o.noise_max)):
        assert np.asarray(a).dtype == np.float32 and np.array_equal(bits(a), bits(b))
    assert np.array_equal(ch, och) and np.array_equal(de, ode) and np.array_equal(blk, oblk)
    assert np.array_equal(bits(rel), bits(orel))
    np.testing.assert_allclose(d.on_threshold, g[f"{name}/on"], rtol=2e-6)
    np.testing.assert_allclose(d.off_threshold, g[f"{name}/off"], rtol=2e-6)
    np.testing.assert_allclose(d.noise_max, g[f"{name}/noise_max"], rtol=2e-6)
    assert np.array_equal(ch, g[f"{name}/ch"]) and np.array_equal(de, g[f"{name}/delta"])
    assert np.array_equal(blk, g[f"{name}/block"])
    np.testing.assert_allclose(rel[::31], g[f"{name}/rel_stride"], rtol=2e-5, atol=1e-7)
    assert printed.startswith("Approx. relative noise thresholds at [") and printed.rstrip().endswith("]!")
    if name == "manual_48k_64":
        assert len(ch) > 0


def test_init_is_refused_where_the_reference_reads_past_its_buffers(det):
    d = det.AmplitudeOnsetDetector(2, 256, sr=48000)  # 48000 is not a multiple of 256
    with pytest.raises(ValueError, match="multiples of block_size"):
        d.init(np.zeros((48000 * 2 // 256 * 256, 2), np.float32))
    with pytest.raises(ctypes.ArgumentError):
        det.AmplitudeOnsetDetector(2, 128, sr=48000).init(np.zeros((96000, 2), np.float64))


def test_g17_realtime_sets_on_the_gpu(det):
    """The realtime arguments (realtime/audio.py:39-52: no high-pass, slow follower 8000/8000, thresholds
    0.45/0.45, block 128 @ 96 kHz, 3 channels) at fast attacks >= 1 sample: GPU == the reference's golden
    indices, offline and per block."""
    from tests.golden.make_golden_r2_cfg import G17_CASES, RT
    from tests.test_oracle_golden import _stream_records
    g = load_golden("g17_realtime_sets")
    for name, (kw, sr, B) in G17_CASES.items():
        x = synth.drum_hits(3, 4.0, sr, seed=170 + len(name), period=0.37)
        recs, rel, _ = det.detect_batch(x[None], block_size=B, sr=sr, **RT, **kw)
        assert np.array_equal(recs[0]["channel"], g[f"{name}_ch"]) and np.array_equal(recs[0]["sample"], g[f"{name}_on"])
        np.testing.assert_allclose(rel[0][::211], g[f"{name}_rel"], rtol=2e-5, atol=1e-6)
    kw, sr, B = G17_CASES["a3"]
    x = synth.drum_hits(3, 1.5, sr, seed=177, period=0.21)
    got = _stream_records(lambda: det.AmplitudeOnsetDetector(3, B, sr=sr, **RT, **kw), x, B, int(0.1 * sr))
    assert np.array_equal(got, g["blk_records"])


@pytest.mark.parametrize("tuning", [None, dict(lane_merge=1, hp_dedupe=1, hp_early=1, sm_segments=1)])
def test_g20_wide_and_dense_shapes_on_the_gpu(det, tuning):
    """g20 (64 channels at block 512, no cooldown at block 32, long cooldown with absolute thresholds, odd sizes):
    GPU == the reference's golden indices, in the latency layout and in the throughput one."""
    from tests.golden.make_golden_r2_cfg import G20_CASES
    g = load_golden("g20_wide_dense")
    for name, (kw, C, secs, sr, B, rk) in G20_CASES.items():
        x = synth.drum_hits(C, secs, sr, **rk)
        recs, rel, _ = det.detect_batch(x[None], tuning=tuning, block_size=B, sr=sr, **kw)
        assert np.array_equal(recs[0]["channel"], g[f"{name}_ch"]) and np.array_equal(recs[0]["sample"], g[f"{name}_on"]), name
        np.testing.assert_allclose(rel[0][::97], g[f"{name}_rel"], rtol=2e-5, atol=1e-6)


def test_realtime_set_against_the_reference_golden_on_the_gpu(det):
    """fast_ar = (0.3, 800) on the g4 input: from the first second on the GPU's records equal the
    reference's golden rt_ch / rt_on; below it the documented, COUNTED deviation (RT_CHAOTIC)."""
    from tests.golden.make_golden_r2_cfg import RT_CHAOTIC
    from tests.test_oracle_golden import realtime_deviation
    g = load_golden("g4_end_to_end")
    x2 = synth.c2_drums(10.0, 8, SR, seed=1)
    recs, _, _ = det.detect_batch(x2[None, :, :3].copy(), block_size=128, sr=SR, hipass_freq=0, fast_ar=(0.3, 800.0),
                                  slow_ar=(8000.0, 8000.0), on_threshold=0.45, off_threshold=0.45, cooldown=9600)
    assert realtime_deviation(recs[0]["channel"].astype(np.int64), recs[0]["sample"], g, SR) == RT_CHAOTIC


def test_g18_python_backtracking_on_the_gpu(det):
    """backtrack=True per block through the streaming kernels: the records equal what the reference's
    Python loop gives on the ring-buffer stand-in (parity of the loop bound: unpinned, see the oracle test)."""
    from tests.golden.make_golden_r2_cfg import G18_CASES
    from tests.test_oracle_golden import _stream_records
    g = load_golden("g18_backtrack_py")
    for name, (kw, C, B) in G18_CASES.items():
        x = synth.drum_hits(C, 2.0, SR, seed=180 + C + B, period=0.19)
        got = _stream_records(lambda: det.AmplitudeOnsetDetector(C, B, sr=SR, **kw), x, B, 4800)
        assert np.array_equal(got, g[f"{name}_records"]), name


@pytest.mark.parametrize("host_verify", [0, 1])
def test_hundreds_of_iir_verification_rounds_stay_exact(det, host_verify):
    """One candidate per chunk, no warm-up, tiny chunks: every chunk of the IIR stage is a break and a chain
    advances one chunk per round: hundreds of rounds inside the chain-local kernel, or (host_verify) more rounds
    than the call has counter slots, so slots are recycled.
    (Found by the 1 500-case sweep of round 2: a recycled, re-zeroed slot was read as the previous round's
    "nothing left" flag and the stage stopped early.)"""
    x = synth.drum_hits(2, 6.5, SR, seed=77, period=0.31)
    info, n = check_clip(det, x, tuning=dict(hp_chunk=1024, hp_warm=-1, hp_candidates=1, host_verify=host_verify),
                         block_size=128, sr=SR)
    assert info["hp_passes"] > 260 and n > 20


@pytest.mark.parametrize("tuning", [None, dict(lane_merge=1, hp_dedupe=1, hp_early=1), dict(host_verify=-2), dict(host_verify=-3),
                                    dict(host_verify=-4, lane_merge=1)])
def test_the_whole_call_is_one_hipgraph_and_replays_on_other_clips(det, tuning):
    """ofp_detect_offline_enqueue issues no synchronisation and reads nothing back on the host, so ONE hipGraph
    captures the complete call (reference loop: detection.py:73-82); replayed on the contents of a second and a
    third clip in the same buffers it gives the oracle's records and relative envelope bit for bit."""
    import torch
    clips = [synth.c2_drums(6.0, 8, SR, seed=s) for s in (3, 4)] + [synth.drum_hits(8, 6.0, SR, seed=9, period=0.37)]
    bd = det.BatchDetector(8, 256, sr=SR)
    if tuning:
        bd.set_tuning(**tuning)
    x = torch.from_numpy(clips[0][None]).cuda().contiguous()
    out = bd.detect(x, cap_per_clip=2048)   # buffers and work space exist before the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        bd.enqueue(x, out=out, cap_per_clip=2048)
    for src in (clips[1], clips[2], clips[0]):
        x.copy_(torch.from_numpy(src[None]))
        out["rel"].zero_()
        out["counts"].zero_()
        out["records"].zero_()
        g.replay()
        torch.cuda.synchronize()
        bd.complete(x, out)
        if tuning and tuning.get("host_verify", 0) <= -2:   # the completion's fall-back, forced: the call again from the
            # IIR stage (-2), the follower stage (-3) or the tracker stage (-4) on, host-verified
            assert bd.last_info["repeated_host_verified"] == 1
        else:
            assert bd.last_info["stage_ms"]["total"] == 0 and bd.last_info["repeated_host_verified"] == 0
        recs = det.BatchDetector.records_to_numpy(out)[0]
        c, o, orel = oracle_records(src, block_size=256, sr=SR)
        assert np.array_equal(recs["channel"], c) and np.array_equal(recs["sample"], o)
        assert np.array_equal(bits(out["rel"][0].cpu().numpy()), bits(orel))
        assert len(c) > 50


THROUGHPUT = dict(lane_merge=1, hp_dedupe=1, hp_early=1, sm_segments=1)


@pytest.mark.parametrize("B,sr,C,N", [(100, 44100, 3, 133333), (250, 48000, 2, 100001), (441, 44100, 5, 88200),
                                      (30, 22050, 1, 50003), (512, 96000, 7, 200000)])
@pytest.mark.parametrize("tuning", [None, THROUGHPUT])
def test_sizes_that_are_multiples_of_nothing(det, B, sr, C, N, tuning):
    """Block sizes, rates and lengths that break every alignment the fast paths rely on (16-byte groups, block-aligned
    warm-up, whole blocks): the 16-byte stores of the IIR writers, the fused dB / follower-sum pass and the staged
    candidates must fall back or cope; in both layouts the result is the oracle's."""
    x = synth.drum_hits(C, N / sr + 0.01, sr, seed=B + C, period=0.19)[:N]
    check_clip(det, x, tuning=tuning, block_size=B, sr=sr)
    check_clip(det, x, tuning=tuning, block_size=B, sr=sr, hipass_freq=0.0, cooldown=0)
