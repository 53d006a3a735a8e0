"""Differential sweep: random detector configurations, inputs and time-parallel tunings, GPU against
the oracle, onset records and every bit of the relative envelope.  The tunings force the rarely
taken paths (chain breaks and wrong guesses in the IIR stage, repair cascades in the followers and
the tracker, runs spanning several chunks, big-batch layouts)."""
import os

import numpy as np
import pytest

import oracle
from onset_fingerprinting_amd import synth

pytestmark = pytest.mark.gpu
SR = 48000


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_case(rng):
    C = int(rng.choice([1, 2, 3, 4, 8, 16]))
    B = int(rng.choice([32, 64, 128, 256, 512]))
    secs = float(rng.uniform(1.5, 5.0))
    kind = rng.integers(0, 5)
    if kind == 0:
        x = synth.drum_hits(C, secs, SR, seed=int(rng.integers(1 << 30)), period=float(rng.uniform(0.08, 0.9)))
    elif kind == 1:
        x = synth.drum_hits(C, secs, SR, seed=int(rng.integers(1 << 30)), poisson_rate=float(rng.uniform(0.5, 8)),
                            amp_log_uniform=(0.02, 0.9))
    elif kind == 2:  # long quiet stretches: the tracker's max is not reset for seconds
        x = synth.drum_hits(C, secs, SR, seed=int(rng.integers(1 << 30)), period=float(rng.uniform(1.5, 3.0)))
    elif kind == 3:  # tone + hits: the IIR candidates have little to coalesce at
        x = synth.drum_hits(C, secs, SR, seed=int(rng.integers(1 << 30)), period=0.7)
        x += (0.05 * np.sin(2 * np.pi * rng.uniform(2500, 9000) * np.arange(len(x)) / SR))[:, None].astype(np.float32)
    else:
        x = (10 ** rng.uniform(-4, -1) * rng.standard_normal((int(secs * SR), C))).astype(np.float32)
    kw = dict(block_size=B, sr=SR, cooldown=int(rng.choice([0, 20, 1323, 9600])),
              hipass_freq=float(rng.choice([0.0, 500.0, 2000.0, 6000.0])))
    if rng.random() < 0.4:
        kw.update(on_threshold=float(rng.uniform(3, 9)), off_threshold=float(rng.uniform(1.5, 3)))
    else:
        kw.update(on_threshold=float(rng.uniform(0.2, 0.7)), off_threshold=float(rng.uniform(0.05, 0.2)))
    if rng.random() < 0.3:
        kw.update(fast_ar=(2.0, 966.0), slow_ar=(8000.0, 8000.0))
    elif rng.random() < 0.3:
        kw.update(slow_ar=(1500.0, 3000.0))  # asymmetric slow follower: the sequential guess path
    if rng.random() < 0.25:
        kw.update(backtrack=True, backtrack_buffer_size=int(max(B, rng.choice([128, 512, 1024]))),
                  backtrack_smooth_size=int(rng.choice([3, 5, 9])))
    tuning = {}
    if rng.random() < 0.7:
        tuning["hp_chunk"] = int(rng.choice([1024, 2048, 4096, 8192]))
        tuning["hp_warm"] = int(rng.choice([0, 2048, 8192, 40960])) or -1
        tuning["hp_candidates"] = int(rng.choice([1, 2, 4, 8, 16]))
        tuning["hp_span"] = int(rng.choice([1, 2, 4]))
        tuning["hp_candidate_offset"] = int(rng.choice([-1, 1, 8, 1021]))
        tuning["hp_dedupe"] = int(rng.choice([-1, 0, 1, 1]))  # staged candidates (the batch setting) on small inputs too
        tuning["hp_early"] = int(rng.choice([-1, 0, 1, 1]))   # early stop of whole re-runs
    if rng.random() < 0.5:
        tuning["ar_chunk"] = int(rng.choice([512, 1024, 4096]))
        tuning["ar_warm"] = int(rng.choice([1024, 8192, 24576]))
        tuning["ar_guess"] = int(rng.choice([0, 1]))
    if rng.random() < 0.5:
        tuning["mm_chunk"] = int(rng.choice([1024, 4096, 8192]))
        tuning["mm_warm"] = int(rng.choice([0, 4096, 49152])) or -1
    if rng.random() < 0.4:
        tuning["lane_merge"] = int(rng.choice([-1, 1]))
    if rng.random() < 0.5:
        tuning["sm_segments"] = int(rng.choice([1, 1, 2]))   # (clips this short take the sequential machine by default)
    if rng.random() < 0.3:
        tuning["scan_skip"] = -1   # the crossing pass reads every block (default: blocks that cannot matter are skipped)
    if rng.random() < 0.25:
        tuning["host_verify"] = 1   # the host-verified pass groups of rounds 1-2 (default: chain-local kernels)
    if rng.random() < 0.3:
        tuning["interleaved"] = int(rng.choice([-1, 2, 3]))  # planar copies throughout / the IIR stage on the interleaved audio too
    if rng.random() < 0.3:
        tuning["line_stores"] = -1  # lane-private 16-byte stores in the output walks (default in the merged layout: complete lines)
    if rng.random() < 0.3:
        tuning["walk_through"] = -1  # every chunk through the chunk pass (default: walk-through chunks are their own pass 0)
    return x, kw, tuning


def test_random_configurations_match_the_oracle(capsys):
    from onset_fingerprinting_amd import detection
    # OFP_FUZZ_CASES / OFP_FUZZ_SEED: longer sweeps with other seeds (tools/README.md)
    n_cases = int(os.environ.get("OFP_FUZZ_CASES", "120"))
    rng = np.random.default_rng(int(os.environ.get("OFP_FUZZ_SEED", "20261004")))
    n_onsets = 0
    for case in range(n_cases):
        x, kw, tuning = random_case(rng)
        recs, rel, info = detection.detect_batch(x[None], tuning=tuning or None, **kw)
        c, o, orel = oracle.detect_onsets_amplitude(x, **kw)
        msg = (case, x.shape, kw, tuning)
        assert np.array_equal(recs[0]["channel"], np.array(c, np.int64)), msg
        assert np.array_equal(recs[0]["sample"], np.array(o, np.int64)), msg
        assert np.array_equal(bits(rel[0]), bits(orel)), msg
        n_onsets += len(c)
    assert n_onsets > 8 * n_cases


def test_random_batches_of_clips():
    """Several clips per call (the big-batch layout heuristics) against the oracle clip by clip."""
    from onset_fingerprinting_amd import detection
    rng = np.random.default_rng(77)
    for n_clips, C, secs in ((7, 4, 2.0), (40, 2, 1.5), (3, 16, 3.0)):
        x = np.stack([synth.drum_hits(C, secs, SR, seed=int(rng.integers(1 << 30)), period=float(rng.uniform(0.1, 0.6)))
                      for _ in range(n_clips)])
        recs, rel, _ = detection.detect_batch(x, block_size=128, sr=SR)
        for i in range(n_clips):
            c, o, orel = oracle.detect_onsets_amplitude(x[i], block_size=128, sr=SR)
            assert np.array_equal(recs[i]["channel"], np.array(c, np.int64)) and np.array_equal(recs[i]["sample"], np.array(o, np.int64))
            assert np.array_equal(bits(rel[i]), bits(orel))


def test_random_groups_and_fix_onsets():
    """find_onset_groups / fix_onsets with random parameters against the oracle (exact)."""
    from onset_fingerprinting_amd import detection
    rng = np.random.default_rng(4242)
    for case in range(40):
        n, C = int(rng.integers(1, 400)), int(rng.integers(1, 12))
        on = np.sort(rng.integers(0, 200000, n)) if rng.random() < 0.8 else rng.integers(0, 200000, n)
        ch = rng.integers(0, C, n)
        md, mc = int(rng.choice([0, 10, 300, 1000, 10 ** 7])), int(rng.integers(1, C + 2))
        cc = None if rng.random() < 0.5 else int(rng.integers(0, int(ch.max()) + 1))
        got = detection.find_onset_groups(on.tolist(), ch.tolist(), md, mc, cc)
        want = oracle.find_onset_groups(on, ch, md, mc, cc)
        assert (got is None) == (want is None), case
        if want is not None:
            assert np.array_equal(got, want), case
    with np.errstate(all="ignore"):
        for case in range(12):
            C = int(rng.integers(2, 9))
            audio, on = synth.sensor_hits(int(rng.integers(1 << 30)), n_channels=C, n=30000, hits=int(rng.integers(3, 14)),
                                          jitter=int(rng.integers(0, 25)))
            kw = dict(filter_size=int(rng.choice([1, 3, 5, 7])), d=int(rng.choice([0, 1, 2])),
                      onset_direction=[None, "up", "down"][int(rng.integers(0, 3))], take_abs=bool(rng.integers(0, 2)),
                      zero_left=bool(rng.integers(0, 2)), normalization_cutoff=int(rng.choice([5, 10, 25])),
                      onset_tolerance=int(rng.choice([10, 30, 45])), shift_onsets=int(rng.choice([0, 3, -2])))
            assert np.array_equal(detection.fix_onsets(audio, on, **kw), oracle.fix_onsets(audio, on, **kw)), (case, kw)
