"""Detector configurations of golden set g3 (shared by make_golden.py and the tests)."""
G3_CONFIGS = [
    dict(block_size=128, hipass_freq=2000.0, on_threshold=0.5, off_threshold=0.1, cooldown=1323),
    dict(block_size=128, hipass_freq=0, on_threshold=0.5, off_threshold=0.1, cooldown=1323),
    dict(block_size=32, hipass_freq=2000.0, on_threshold=6.0, off_threshold=4.0, cooldown=0),
    dict(block_size=256, hipass_freq=2000.0, on_threshold=6.0, off_threshold=4.0, cooldown=20),
    dict(block_size=256, hipass_freq=0, on_threshold=0.45, off_threshold=0.45, cooldown=9600,
         fast_ar=(0.3, 800.0), slow_ar=(8000.0, 8000.0)),
    dict(block_size=512, hipass_freq=1000.0, on_threshold=0.5, off_threshold=0.1, cooldown=0),
    dict(block_size=32, hipass_freq=0, on_threshold=0.2, off_threshold=0.15, cooldown=20,
         fast_ar=(2.0, 966.0)),
    dict(block_size=128, hipass_freq=2000.0, on_threshold=6.1, off_threshold=2.3, cooldown=300,
         floor=-60.0),
]
