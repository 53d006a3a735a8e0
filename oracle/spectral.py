"""numpy restatement of the reference's framing / STFT / fingerprint functions
(data.py) -- TEST INFRASTRUCTURE ONLY.

Parity status:
* frame_extract, stft_frame, stft, window_contribution_weights: PINNED against
  golden vectors captured from the reference (tests/golden/make_golden.py).
* mel_filterbank, power_to_db, cspec_to_mfcc: PARITY UNPINNED vs librosa.  The
  reference's ``cspec_to_mfcc`` (data.py:657-680) is three calls into librosa,
  which is unpinned (pyproject.toml:22), not vendored and not installable here;
  the functions below restate librosa's published definitions (Slaney mel scale
  and area normalisation, power_to_db ref=1 amin=1e-10 top_db=80, DCT-II ortho).
  They are pinned only by self-consistency known-answer tests.
"""
import numpy as np
import scipy.fft
import scipy.signal


def hann_periodic(n):
    """data.py:627: librosa.filters.get_window("hann", n, fftbins=True) is
    scipy.signal.get_window (float64, periodic)."""
    return scipy.signal.get_window("hann", n, fftbins=True)


def pad_center(data, size, axis=-1):
    """librosa.util.pad_center: zero-pad symmetrically (extra sample on the right)."""
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths)


def window_contribution_weights(window, hop_length, hop_edge_padding=False):
    """data.py:562-578."""
    w = []
    start_idx = len(window) // 2 if not hop_edge_padding else hop_length
    for i in range(start_idx, len(window) + hop_length, hop_length):
        w.append(np.trapezoid(window[:i]))
    w += w[-2::-1]
    return np.array(w) / max(w)


def stft_frame(x, n_fft, window):
    """data.py:581-590: float64 window * float32 frame -> double rFFT."""
    if n_fft > x.shape[-1]:
        x = pad_center(x, n_fft)
    return np.fft.rfft(window * x)


def stft(audio, onset, frame_length=256, hop_length=64, n_fft=512,
         hop_edge_padding=False, method="zerozero"):
    """data.py:593-654: audio is [(C,) N]; returns complex64 [(C,) n_fft/2+1, n_frames]."""
    y = audio[..., onset: onset + frame_length]
    pad_length = frame_length - hop_length if hop_edge_padding else frame_length // 2
    dim0 = 1 if y.ndim == 1 else y.shape[0]
    pad = np.zeros((dim0, pad_length), dtype=np.float32).squeeze()
    pre = audio[..., onset - pad_length: onset]
    window = hann_periodic(frame_length)
    if n_fft > frame_length:
        window = pad_center(window, n_fft)
    if method == "zerozero":
        y = np.concatenate((pad, y, pad), axis=-1)
    elif method == "prezero":
        y = np.concatenate((pre, y, pad), axis=-1)
    elif method == "pre":
        y = np.concatenate((pre, y), axis=-1)
    n_frames = 1 + (y.shape[-1] - frame_length) // hop_length
    S = np.empty((dim0, n_fft // 2 + 1, n_frames), dtype=np.complex64).squeeze()
    for i in range(n_frames):
        S[..., i] = stft_frame(y[..., hop_length * i: hop_length * i + frame_length],
                               n_fft, window)
    return S


def frame_extract(audio, onsets, frame_length, pre_samples, add_pre_samples=False,
                  use_min_onset=True):
    """data.py:55-120 with max_shift == 0: audio [N(,C)], onsets [O(,C)] -> [O(,C),W]."""
    if add_pre_samples:
        frame_length += pre_samples
    view = np.lib.stride_tricks.sliding_window_view(audio, window_shape=frame_length, axis=0)
    if audio.ndim == 2:
        if use_min_onset:
            return view[onsets.min(axis=1) - pre_samples]
        return np.stack([view[onsets[:, i] - pre_samples, i, :]
                         for i in range(audio.shape[1])], axis=1)
    return view[onsets - pre_samples]


def dense_power_frames(audio, n_fft, hop):
    """The metric's dense workload (SURVEY.md 8a row a9 "dense equivalent"):
    for audio [N, C], every hop h of every channel c:
    P[c, h, :] = |rfft(hann(n_fft) * audio[h*hop : h*hop+n_fft, c])|^2,
    fp64 arithmetic as data.py:588-590, returned as float64 [C, H, n_fft/2+1]."""
    N, C = audio.shape
    H = 1 + (N - n_fft) // hop if N >= n_fft else 0
    w = hann_periodic(n_fft)
    fr = np.lib.stride_tricks.sliding_window_view(audio, n_fft, axis=0)[::hop][:H]  # [H, C, F]
    X = np.fft.rfft(w * fr.astype(np.float64), axis=-1)
    P = X.real ** 2 + X.imag ** 2
    return np.ascontiguousarray(P.transpose(1, 0, 2))


# ---- librosa restatement (PARITY UNPINNED) --------------------------------

def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz,
                    min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels=40, fmin=0.0, fmax=None):
    """librosa.filters.mel(htk=False, norm="slaney") -> float32 [n_mels, n_fft/2+1]."""
    if fmax is None:
        fmax = float(sr) / 2
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.power_to_db(ref=1.0): 10*log10(max(amin,S)), floored at max-top_db
    where the max is taken over the WHOLE array passed in."""
    S = np.asarray(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, 1.0))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def cspec_to_mfcc(S, sr, fmin=0, fmax=None, n_mels=40, n_mfcc=14):
    """data.py:657-680: S complex [(C,) bins, T] -> MFCC [(C,) n_mfcc, T]."""
    P = np.abs(S) ** 2
    n_fft = 2 * (P.shape[-2] - 1)
    fb = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    mels = np.einsum("...ft,mf->...mt", P, fb)
    db = power_to_db(mels)
    return scipy.fft.dct(db, axis=-2, type=2, norm="ortho")[..., :n_mfcc, :]


def fourier_resample(x, num):
    """scipy.signal.resample (real input, time domain, no window) restated with explicit DFT sums in fp64:
    keep the bins up to the Nyquist of the shorter length (doubling / halving that bin when the length is
    even, as scipy does), synthesise `num` samples, scale by num / len(x).  x [Nx] -> [num]."""
    x = np.asarray(x, dtype=np.float64)
    Nx = len(x)
    N = min(num, Nx)
    K = N // 2
    n = np.arange(Nx)
    Y = np.array([np.sum(x * np.exp(-2j * np.pi * k * n / Nx)) for k in range(K + 1)])
    if N % 2 == 0:
        if num < Nx:
            Y[K] *= 2.0
        elif Nx < num:
            Y[K] *= 0.5
    m = np.arange(num)
    y = np.full(num, Y[0].real)
    for k in range(1, K + 1):
        e = np.exp(2j * np.pi * k * m / num)
        y += (Y[k] * e).real * (1.0 if 2 * k == num else 2.0)
    return y / num * (num / Nx)


def stretch_frames(audio, onsets, shifts, frame_length, pre_samples):
    """StretchFrameExtractor.__call__ (data.py:207-223) for given shifts: -> [O, C, L] (or [O, L] for 1-D audio)."""
    audio = np.asarray(audio)
    onsets = np.asarray(onsets)
    st = (onsets.min(axis=1) if audio.ndim == 2 else onsets) - pre_samples
    out = np.empty(onsets.shape + (frame_length,), dtype=np.float32)
    for i, (o, sh) in enumerate(zip(st, shifts)):
        w = audio[o:o + frame_length + sh]
        if audio.ndim == 2:
            out[i] = np.stack([fourier_resample(w[:, c], frame_length) for c in range(audio.shape[1])])
        else:
            out[i] = fourier_resample(w, frame_length)
    return out
