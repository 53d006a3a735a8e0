"""CPU oracle, TEST INFRASTRUCTURE ONLY: the spectral-flux onset detector.

Restates ``detect_onsets_spectral`` (reference detection.py:89-128).  PARITY UNPINNED: the
reference's arithmetic here is three librosa calls (stft, A_weighting, util.peak_pick) and librosa
is absent and not installable; they are restated from librosa's published definitions, the rest
(weighting, flux, percentile normalisation, hop scaling) follows the reference line by line.
"""
import numpy as np
import scipy.signal


def librosa_stft_mag(x, n_fft, hop):
    """|librosa.stft(x, hop_length=hop, n_fft=n_fft)|: center=True with zero padding of n_fft//2,
    periodic Hann window of n_fft, frames t*hop, T = 1 + len(x)//hop; computed in double and
    rounded into complex64 as librosa does for float32 input."""
    x = np.asarray(x, np.float32)
    xp = np.pad(x, n_fft // 2)
    T = 1 + len(x) // hop
    w = scipy.signal.get_window("hann", n_fft, fftbins=True)
    fr = np.lib.stride_tricks.sliding_window_view(xp, n_fft)[::hop][:T]
    S = np.fft.rfft(w * fr.astype(np.float64), axis=-1).astype(np.complex64)
    return np.abs(S).T  # [bins, T] float32


def a_weighting(frequencies, min_db=-80.0):
    f_sq = np.asanyarray(frequencies, dtype=np.float64) ** 2.0
    const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
    with np.errstate(divide="ignore"):
        w = 2.0 + 20.0 * (np.log10(const[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + const[0])
                          - np.log10(f_sq + const[1]) - 0.5 * np.log10(f_sq + const[2])
                          - 0.5 * np.log10(f_sq + const[3]))
    return w if min_db is None else np.maximum(min_db, w)


def peak_pick(x, pre_max, post_max, pre_avg, post_avg, delta, wait):
    """librosa.util.peak_pick: x[n] is a peak iff x[n] == max(x[n-pre_max : n+post_max]),
    x[n] >= mean(x[n-pre_avg : n+post_avg]) + delta and n - previous_n > wait."""
    x = np.asarray(x)
    pre_max, post_max, pre_avg, post_avg, wait = (int(v) for v in (pre_max, post_max, pre_avg, post_avg, wait))
    peaks, last = [], -np.inf
    for n in range(len(x)):
        if x[n] <= 0:
            continue
        if x[n] != x[max(0, n - pre_max):n + post_max].max():
            continue
        if np.float64(x[n]) < np.mean(x[max(0, n - pre_avg):n + post_avg], dtype=np.float64) + delta:
            continue
        if n > last + wait:
            peaks.append(n)
            last = n
    return np.asarray(peaks, dtype=np.int64)


def detect_onsets_spectral(x, n_fft=256, hop=32, sr=96000, return_oe=False):
    """detection.py:96-128."""
    D = librosa_stft_mag(x, n_fft, hop)
    freq = np.fft.fftfreq(n_fft, 1 / sr)[:len(D)]
    aw = a_weighting(freq)[:, None]
    D *= (aw - aw.min()) / np.abs(aw.min())
    oe = D[:, 1:] - D[:, :-1]
    oe = np.maximum(0.0, oe)
    oe = oe.mean(0)
    oe /= np.percentile(oe, 99.9)
    peaks = peak_pick(oe, pre_max=0.12 * sr // hop, post_max=0.01 * sr // hop, pre_avg=0.12 * sr // hop,
                      post_avg=0.01 * sr // hop + 1, delta=0.1, wait=sr * 0.07 // hop)
    peaks = peaks * hop
    return (peaks, oe) if return_oe else peaks


class EMAMinMax:
    """ASSUMED arithmetic of loopmate's EMA_MinMaxTracker (absent; realtime/recording.py:251-256): the update of
    envelope_follower.c:27-57 with a single alpha, optional lower bounds for the two tracked values.  PARITY
    UNPINNED: nothing in the reference or its tests fixes this class."""

    def __init__(self, min0=0.0, max0=1.0, alpha=0.001, minmin=None, minmax=None):
        self.min_val, self.max_val, self.alpha = np.float32(min0), np.float32(max0), np.float32(alpha)
        self.minmin, self.minmax = minmin, minmax

    def add_sample(self, x):
        x, a = np.float32(x), self.alpha
        self.max_val = x if x > self.max_val else (np.float32(1) - a) * self.max_val + a * x
        self.min_val = x if x < self.min_val else (np.float32(1) - a) * self.min_val + a * x
        if self.minmax is not None:
            self.max_val = max(self.max_val, np.float32(self.minmax))
        if self.minmin is not None:
            self.min_val = max(self.min_val, np.float32(self.minmin))

    def normalize_sample(self, x):
        return (np.float32(x) - self.min_val) / (self.max_val - self.min_val)


class HopStrength:
    """RecAnalysis.fft + onset_strength per hop (realtime/recording.py:273-311) on a plain history array.
    max_length / avg_length stand for config.MAX_LENGTH / AVG_LENGTH, which realtime/config.py does not define."""

    def __init__(self, n_fft, n_channels, max_length, avg_length, ring, tg_win_length=None):
        from scipy.signal.windows import hann
        self.tg_win_length = tg_win_length
        self.tg_window = hann(tg_win_length).astype(np.float32) if tg_win_length else None   # :250
        self.n_fft = n_fft
        self.window = hann(n_fft).astype(np.float32)                       # :249
        self.audio = np.zeros((n_fft, n_channels), np.float32)             # audio[-n_fft:] of the ring buffer
        self.prev = np.zeros(n_fft // 2 + 1, np.float32)
        self.logspec = EMAMinMax(max0=10.0, minmax=0.0, alpha=0.0005)      # :254-256 (min side unused)
        self.oe = EMAMinMax(min0=0.0, minmin=0.0, max0=1.0, alpha=0.001)   # :251-253
        self.env = np.zeros(ring, np.float32)
        self.max_length, self.avg_length = max_length, avg_length

    def __call__(self, hop):
        self.audio = np.concatenate([self.audio, hop])[-self.n_fft:]
        X = np.fft.rfft((self.window * self.audio.mean(-1)).astype(np.float64))   # :276
        mag = (X.real ** 2 + X.imag ** 2).astype(np.float32)
        s = 10.0 * np.log10(np.maximum(1e-10, mag))                        # :290
        self.logspec.add_sample(s.max())
        floor = self.logspec.max_val - 80
        s = np.maximum(s, floor)
        sm1 = np.maximum(10.0 * np.log10(np.maximum(1e-10, self.prev)), floor)
        onset_env = np.maximum(0.0, s - sm1).mean()                        # :296
        self.prev = mag
        self.oe.add_sample(onset_env)
        norm = self.oe.normalize_sample(onset_env)
        self.env = np.concatenate([self.env[1:], [norm]]).astype(np.float32)
        return np.array([onset_env, norm, self.env[-self.max_length:].max(), self.env[-self.avg_length:].mean()])

    def tempogram(self):
        """realtime/recording.py:313-327 on the envelope history as it stands after __call__: the reference's own
        expression (transform length tg_pad = 2 W - 1, config.py:56), evaluated in float64."""
        W = self.tg_win_length
        P = 2 * W - 1
        X = np.fft.rfft((self.tg_window * self.env[-W:]).astype(np.float64), n=P)
        tg = np.fft.irfft(X.real ** 2 + X.imag ** 2, n=P)[:W]
        return (tg / (tg.max() + 1e-10)).astype(np.float32)
