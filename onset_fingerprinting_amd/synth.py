"""Deterministic synthetic inputs for the BASELINE.json configurations
(recipes: SURVEY.md section 8d).  All fp32, interleaved [N, C] as the reference
expects (detection.py:36 "signal to analyse (NxC)"), 48 kHz.
"""
import numpy as np

SR = 48000


def c1_sine_clicks(seconds=10.0, sr=SR, seed=0):
    """C1: 1 ch, 0.01*sin(440 Hz) + 1e-4 noise + clicks every 0.5 s from t=1 s."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    x = 0.01 * np.sin(2 * np.pi * 440 * t) + 1e-4 * rng.standard_normal(n)
    k = np.arange(200)
    for s in np.arange(1.0, seconds - 0.01, 0.5):
        i = int(s * sr)
        m = min(200, n - i)
        x[i:i + m] += (0.8 * np.exp(-k / 30) * rng.standard_normal(200))[:m]
    return x.astype(np.float32)[:, None]


def drum_hits(n_channels, seconds, sr=SR, seed=1, period=0.5, amp=0.8,
              amp_log_uniform=None, poisson_rate=None, gain=1.0):
    """C2/C3/C4 family: noise floor 1e-3 + per-channel hits
    amp*exp(-k/40)*N(0,1) (k<300) + 0.3*exp(-k/2000)*sin(2*pi*200*k/sr) body,
    every `period` s offset by 37*c samples (or Poisson times at `poisson_rate`/s);
    amplitudes log-uniform in `amp_log_uniform` if given."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    x = 1e-3 * rng.standard_normal((n, n_channels))
    k = np.arange(300)
    kb = np.arange(8000)
    body = 0.3 * np.exp(-kb / 2000) * np.sin(2 * np.pi * 200 * kb / sr)
    if poisson_rate is None:
        base = (np.arange(period, seconds - 0.2, period) * sr).astype(np.int64)
    for c in range(n_channels):
        if poisson_rate is None:
            starts = base + 37 * c
        else:
            m = rng.poisson(poisson_rate * seconds)
            starts = np.sort(rng.integers(int(0.6 * sr), n - 9000, size=m))
        for i in starts:
            a = amp
            if amp_log_uniform is not None:
                lo, hi = amp_log_uniform
                a = float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
            m1 = min(300, n - i)
            x[i:i + m1, c] += (a * np.exp(-k / 40) * rng.standard_normal(300))[:m1]
            m2 = min(8000, n - i)
            x[i:i + m2, c] += (a / 0.8) * body[:m2]
    return (gain * x).astype(np.float32)


def c2_drums(seconds=60.0, n_channels=8, sr=SR, seed=1):
    """C2: 8 ch x 60 s drum hits (the configuration the metric is quoted on)."""
    return drum_hits(n_channels, seconds, sr, seed)


def c3_stream(seconds=600.0, n_channels=64, sr=SR, seed=2):
    """C3: 64 ch x 10 min, hit amplitudes log-uniform in [0.05, 0.9]."""
    return drum_hits(n_channels, seconds, sr, seed, amp_log_uniform=(0.05, 0.9))


def c4_clip(clip, seconds=10.0, n_channels=4, sr=SR, seed=3):
    """C4: clip `clip` of the 512-clip batch: Poisson hit times (4/s), clip gain."""
    rng = np.random.default_rng(seed + clip)
    gain = float(np.exp(rng.uniform(np.log(0.3), np.log(1.0))))
    return drum_hits(n_channels, seconds, sr, seed + clip, poisson_rate=4.0, gain=gain)


def sensor_hits(seed, n_channels=4, n=24000, hits=12, jitter=12):
    """Calibration-hit style input for fix_onsets (detection.py:373-451): every hit reaches the
    channels with a lag below 60 samples; returns (audio [n, C] float32, onsets [hits, C] int64)
    where the onsets are the true arrival times disturbed by +-`jitter` samples."""
    rng = np.random.default_rng(seed)
    audio = (0.01 * rng.standard_normal((n, n_channels))).astype(np.float32)
    t = np.arange(300)
    onsets = []
    for h in range(hits):
        t0 = 1500 + h * ((n - 3000) // hits) + int(rng.integers(0, 200))
        row = []
        for c in range(n_channels):
            lag = int(rng.integers(0, 60))
            s = np.exp(-t / 60.0) * np.sin(2 * np.pi * t / 23.0) * (0.5 + 0.5 * rng.random())
            audio[t0 + lag:t0 + lag + 300, c] += s.astype(np.float32)
            row.append(t0 + lag + int(rng.integers(-jitter, jitter + 1)))
        onsets.append(row)
    return audio, np.asarray(onsets, np.int64)


def n_frames(n_samples, n_fft, hop):
    """frames per channel of the dense metric: 1 + (N - F)//B."""
    return 0 if n_samples < n_fft else 1 + (n_samples - n_fft) // hop


def init_clip(seed, n_channels, sr, seconds, n_follow, amp=1.0):
    """Calibration recording for AmplitudeOnsetDetector.init (detection.py:842-888): half a second of
    noise floor, then hits about as loud as it gets; plus `n_follow` samples of further audio for the
    detector to run on afterwards.  -> (x [seconds*sr, C], y [n_follow, C]) float32."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)

    def make(n, first_hit):
        a = (1e-3 * rng.standard_normal((n, n_channels))).astype(np.float32)
        t = np.arange(int(0.05 * sr))
        pos = first_hit
        while pos + len(t) < n:
            for c in range(n_channels):
                a_hit = amp * (0.3 + 0.6 * rng.random())
                burst = a_hit * np.exp(-t / (0.004 * sr)) * rng.standard_normal(len(t))
                o = pos + 13 * c
                if o + len(t) < n:
                    a[o:o + len(t), c] += burst.astype(np.float32)
            pos += int(sr * (0.2 + 0.2 * rng.random()))
        return a

    return make(n, int(0.7 * sr)), make(n_follow, int(0.05 * sr))
