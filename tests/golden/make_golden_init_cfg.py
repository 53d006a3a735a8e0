"""Configurations of the g15 golden set (tests/golden/make_golden_init.py)."""
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
