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
