"""CPU-only: the fp64-evaluated log10/exp10 canon (include/ofp_math.h) against an
independent evaluation (libm double, rounded once to float), plus properties of
the follower / detector (hypothesis)."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

import oracle


def ulp_diff(a, b):
    a = a.view(np.int32).astype(np.int64)
    b = b.view(np.int32).astype(np.int64)
    return np.abs(a - b)


def test_log10_canon_is_correctly_rounded_on_a_dense_sample():
    rng = np.random.default_rng(0)
    bits = rng.integers(1, 0x7F800000, size=2_000_000, dtype=np.int64).astype(np.uint32)
    x = np.concatenate([bits.view(np.float32),
                        np.float32([1.0, 0.99999994, 1.0000001, 1e-10, 1e-45, 3.4e38, 0.70710677, 1.4142135])])
    got = oracle.log10f(x)
    want = np.log10(x.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, want)
    # libm's double log10 is itself within 1 ulp(double): disagreement is possible only at
    # near-ties, i.e. a handful per 10^8 inputs at most
    assert d.max() <= 1 and (d > 0).sum() <= 2
    with np.errstate(all="ignore"):
        assert np.isneginf(oracle.log10f(np.float32([0.0]))[0]) and np.isnan(oracle.log10f(np.float32([np.nan]))[0])
        assert np.isposinf(oracle.log10f(np.float32([np.inf]))[0])


def test_exp10_canon_is_correctly_rounded_on_a_dense_sample():
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.uniform(-46, 39, 1_000_000), rng.uniform(-4, 4, 1_000_000),
                        [0.0, -0.0, 1.0, 2.0, -45.5, 38.5]]).astype(np.float32)
    got = oracle.exp10f(v)
    with np.errstate(over="ignore", under="ignore"):
        want = np.power(10.0, v.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, want)
    assert d.max() <= 1 and (d > 0).sum() <= 2
    assert oracle.exp10f(np.float32([0.0]))[0] == 1.0 and oracle.exp10f(np.float32([3.0]))[0] == 1000.0
    assert np.isposinf(oracle.exp10f(np.float32([39.0]))[0]) and oracle.exp10f(np.float32([-50.0]))[0] == 0.0


def test_rect_db_and_rel_linear_follow_the_reference_formulas():
    x = np.float32([0.0, -1e-10, 1e-10, 0.5, -0.5, 1e-5, 3.0])
    db = oracle.rect_db(x, -70.0)
    with np.errstate(divide="ignore"):
        ref = np.clip(20 * np.log10(np.abs((x + np.float32(1e-10)).astype(np.float64))), -70, None)
    assert np.allclose(db, ref, rtol=1e-6, atol=1e-5)
    d = np.float32([-200, -70, -1, 0, 1, 36.9, 37, 200])
    lin = oracle.rel_linear(d, -70.0)
    ref = np.clip(10 ** (d.astype(np.float64) / 20) - 1e-10, 0, 70)
    assert np.allclose(lin, ref, rtol=1e-6)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 5), st.integers(1, 300), st.integers(0, 2 ** 31 - 1),
       st.sampled_from([(3.0, 383.0), (2205.0, 2205.0), (0.3, 800.0)]))
def test_ar_envelope_block_size_invariance(C, n, seed, ar):
    """The state hand-off through the last row (envelope_follower.c:13-14) makes the
    follower independent of how the stream is cut into calls."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-70, 0, (n, C)).astype(np.float32)
    a, r = np.float32(1 / ar[0]), np.float32(1 / ar[1])
    y = np.full((n, C), -70.0, np.float32)
    oracle.ar_envelope(x, y, a, r)
    cut = int(rng.integers(0, n + 1))
    last = np.full(C, -70.0, np.float32)
    parts = []
    for seg in (x[:cut], x[cut:]):
        if len(seg) == 0:
            continue
        yy = np.empty_like(seg)
        yy[-1] = last
        oracle.ar_envelope(np.ascontiguousarray(seg), yy, a, r)
        last = yy[-1].copy()
        parts.append(yy)
    assert np.array_equal(np.concatenate(parts).view(np.uint32), y.view(np.uint32))


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_detector_channel_permutation_equivariance_without_coupling(seed):
    """Permuting channels permutes the result as long as no two channels fire in
    the same block (the only cross-channel term is on_indices.max(), detection.py:790)."""
    from onset_fingerprinting_amd import synth
    rng = np.random.default_rng(seed)
    x = synth.drum_hits(3, 0.4, 48000, seed=int(rng.integers(1 << 30)), period=0.09)
    perm = rng.permutation(3)
    kw = dict(block_size=64, sr=48000, hipass_freq=0, on_threshold=6.0, off_threshold=4.0, cooldown=0)
    c1, o1, r1 = oracle.detect_onsets_amplitude(x, **kw)
    c2, o2, r2 = oracle.detect_onsets_amplitude(np.ascontiguousarray(x[:, perm]), **kw)
    assert np.array_equal(r1[:, perm].view(np.uint32), r2.view(np.uint32))
    blocks = np.array(o1) // 64
    if len(set(blocks.tolist())) == len(blocks):  # no block holds two onsets: no coupling possible
        inv = np.argsort(perm)
        a = sorted(zip(inv[np.array(c1, dtype=int)].tolist(), [int(v) for v in o1]))
        b = sorted(zip([int(c) for c in c2], [int(v) for v in o2]))
        assert a == b
