"""Configurations of the g15 (tests/golden/make_golden_init.py) and g16
(tests/golden/make_golden_fastframes.py) golden sets."""
import numpy as np

G15 = {
    "relative_48k_128": dict(sr=48000, B=128, C=3, seed=151, seconds=3.0, follow_blocks=300, kw=dict()),
    "relative_96k_256_nohp": dict(sr=96000, B=256, C=2, seed=152, seconds=2.0, follow_blocks=200,
                                  kw=dict(hipass_freq=0, on_threshold=0.4, off_threshold=0.2, cooldown=2000)),
    # manual mode (scalar on_threshold > 1 at construction, detection.py:687): the thresholds init derives
    # (dB differences) are then compared with the linear envelope directly, and hits do cross them
    "manual_48k_64": dict(sr=48000, B=64, C=3, seed=154, seconds=2.0, follow_blocks=600, amp=0.03,
                          kw=dict(on_threshold=1.01, off_threshold=0.5, cooldown=300)),
    "fast_ar_32": dict(sr=48000, B=32, C=4, seed=153, seconds=2.5, follow_blocks=900,
                       kw=dict(fast_ar=(2.0, 966.0), hipass_freq=1000.0, cooldown=500)),
}

G16 = {
    "2d_min_onset": dict(seed=161, n=6000, C=4, O=12, two_d_onsets=True, kw=dict(frame_length=64, pre_samples=8)),
    "2d_add_pre": dict(seed=162, n=5000, C=3, O=9, two_d_onsets=True,
                       kw=dict(frame_length=48, pre_samples=16, add_pre_samples=True)),
    "1d_audio": dict(seed=163, n=4000, C=0, O=10, two_d_onsets=False, kw=dict(frame_length=32, pre_samples=4)),
    "2d_audio_1d_onsets": dict(seed=164, n=4000, C=2, O=7, two_d_onsets=False, kw=dict(frame_length=40, pre_samples=0)),
    "2d_shift": dict(seed=165, n=6000, C=4, O=11, two_d_onsets=True,
                     kw=dict(frame_length=64, pre_samples=8, max_shift=5)),
    "1d_shift_add_pre": dict(seed=166, n=3000, C=0, O=6, two_d_onsets=False,
                             kw=dict(frame_length=30, pre_samples=10, max_shift=9, add_pre_samples=True)),
}


def g16_inputs(cfg):
    """-> (audio float32 [n] or [n, C], onsets int64 [O] or [O, C]) regenerated from cfg["seed"]."""
    rng = np.random.default_rng(cfg["seed"])
    shape = (cfg["n"],) if cfg["C"] == 0 else (cfg["n"], cfg["C"])
    audio = rng.standard_normal(shape).astype(np.float32)
    base = np.sort(rng.integers(200, cfg["n"] - 400, cfg["O"]))
    if cfg["two_d_onsets"]:
        onsets = base[:, None] + rng.integers(0, 60, (cfg["O"], max(cfg["C"], 1)))
    else:
        onsets = base
    return audio, onsets.astype(np.int64)
