"""CPU oracle, TEST INFRASTRUCTURE ONLY: cross-correlation lag and onset fixing.

Restates ``cross_correlation_lag`` (reference detection.py:195-268), ``adjust_onset`` (:299-352)
and ``fix_onsets`` (:373-451).  Parity: lags and fixed onsets pinned index-for-index by
tests/golden/g12_xcorr.npz and g13_fix.npz (captured from the reference).  The dot products
follow the canon stated in oracle/ofp_oracle.c (fp64 accumulation, one rounding to fp32);
np.correlate's own float32 summation order is BLAS-dependent, so the reference's cc VALUES are
not a fixed target -- its argmax is, away from ties.
"""
import ctypes

import numpy as np
from scipy.ndimage import median_filter

from .detector import lib

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
lib.oracle_xcorr_slice.argtypes = [_f32p, _f32p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _f32p]


def py_slice(start, stop, length):
    """What cc[start:stop] selects on an array of `length` (negative indices wrap once, then clamp)."""
    s, e, _ = slice(start, stop).indices(length)
    return s, max(e, s)


def lag_window(n, onsets=None, legal_lags=None, onset_tolerance=50):
    """detection.py:256-264 -> (lo, hi, max_adjust): the slice of the full cc that is searched."""
    if legal_lags is not None:
        lo, hi = py_slice(n - legal_lags[1], n - legal_lags[0], 2 * n - 1)
        return lo, hi, legal_lags[1]
    if onsets is not None:
        current_lag = onsets[1] - onsets[0]
        lag_center = n - current_lag
        lo, hi = py_slice(lag_center - onset_tolerance, lag_center + onset_tolerance, 2 * n - 1)
        return lo, hi, current_lag + onset_tolerance
    raise ValueError("cross_correlation_lag needs onsets or legal_lags (the reference raises NameError)")


def xcorr_slice(x, y, cutoff, lo, hi):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    cc = np.empty(max(hi - lo, 0), np.float32)
    lib.oracle_xcorr_slice(x, y, len(x), cutoff, lo, hi, cc)
    return cc


def cross_correlation_lag(x, y, onsets=None, legal_lags=None, d=0, normalization_cutoff=10, onset_tolerance=50,
                          take_abs=False, return_cc=False):
    """detection.py:238-268."""
    x = np.diff(np.asarray(x, np.float32), d)
    y = np.diff(np.asarray(y, np.float32), d)
    if take_abs:
        x, y = np.abs(x), np.abs(y)
    n = len(x)
    lo, hi, max_adjust = lag_window(n, onsets, legal_lags, onset_tolerance)
    cc = xcorr_slice(x, y, normalization_cutoff, lo, hi)
    if len(cc) == 0:
        return (None, cc) if return_cc else None
    lag = -(int(np.argmax(cc)) - max_adjust)
    return (lag, cc) if return_cc else lag


def adjust_onset(onsets, x, y, new_lag):
    """detection.py:310-352 (the weights are np.exp of np.linspace, the sums np.sum in fp64)."""
    oa, ob = int(onsets[0]), int(onsets[1])
    lag_diff = (ob - oa) - new_lag
    exp = np.exp(np.linspace(0, -np.e, abs(lag_diff)))
    n = len(x)
    if lag_diff < 0:
        x_start, x_end = max(oa + lag_diff, 0), min(oa, n)
        y_start, y_end = min(ob, n), min(ob - lag_diff, n)
    else:
        x_start, x_end = oa, min(oa + lag_diff, n)
        y_start, y_end = max(ob - lag_diff, 0), min(ob, n)
    if x_end > x_start:
        da = np.sum(x[x_start:x_end] * exp[len(exp) - (x_end - x_start):]) / x.max()
    else:
        da = 0.0  # the reference's expression cannot be evaluated here (empty slice times weights)
    if y_end <= y_start:
        db = 0
    else:
        db = np.sum(y[y_start:y_end] * exp[len(exp) - (y_end - y_start):][::-1]) / y.max()
    if da > db:
        if oa + lag_diff < 0:
            return 0, -lag_diff
        return lag_diff, 0
    return 0, -lag_diff


def fix_onsets(audio, onsets, filter_size=5, d=0, onset_direction=None, take_abs=False, zero_left=False,
               normalization_cutoff=10, onset_tolerance=30, shift_onsets=0):
    """detection.py:412-451."""
    audio = np.asarray(audio, np.float32)
    lookaround = normalization_cutoff + onset_tolerance
    onsets = np.array(onsets, dtype=np.int64) + shift_onsets
    for og in onsets:
        idx = np.argsort(og, kind="stable")
        a, b = og[idx[0]], og[idx[-1]]
        assert a - lookaround >= 0 and b + lookaround <= len(audio), "onset group too close to the clip edge"
        section = np.diff(median_filter(audio[a - lookaround:b + lookaround], filter_size, axes=0), d, axis=0)
        if onset_direction == "up":
            section[section < 0] = 0
        elif onset_direction == "down":
            section[section > 0] = 0
        if take_abs:
            section = np.abs(section)
        section_og = og - (a - lookaround)
        for i in idx[1:]:
            o = [section_og[idx[0]], section_og[i]]
            x, y = section[:, idx[0]], section[:, i]
            if zero_left:
                x[:o[0]] = 0.0
                y[:o[1]] = 0.0
            new_lag = cross_correlation_lag(x, y, o, normalization_cutoff=normalization_cutoff,
                                            onset_tolerance=onset_tolerance)
            if new_lag is not None:
                ca, cb = adjust_onset(o, x, y, new_lag)
                og[idx[0]] += ca
                og[i] += cb
                section_og[idx[0]] += ca
                section_og[i] += cb
    return onsets


def adjust_onset_rel(onsets, relx, rely, new_lag):
    """detection.py:271-296."""
    oa, ob = onsets[0], onsets[1]
    lag_diff = (ob - oa) - new_lag
    da = relx[oa + lag_diff] - relx[oa]
    db = rely[ob - lag_diff] - rely[ob]
    return (oa + lag_diff, ob) if da > db else (oa, ob - lag_diff)


def filter_data(x, direction):
    """detection.py:355-370 (returns a filtered COPY; the reference works in place)."""
    x = np.array(x, copy=True)
    diff = np.diff(x, 1, axis=0, prepend=x[:1])
    if direction == "up":
        x[diff < 0] = 0
    elif direction == "down":
        x[diff > 0] = 0
    else:
        raise RuntimeError(f"Unknown onset direction {direction=}!")
    return x


def detect_onset_region(audio, detected_onset, n=256, median_filter_size=5, threshold_factor=0.5):
    """detection.py:454-484, spelled out without scipy: zero-padded running median, threshold, opening with a
    centred 5-sample structure (erosion with the outside False, then dilation), first True (0 if none)."""
    audio = np.asarray(audio)
    start = max(detected_onset - n // 2, 0)
    end = min(detected_onset + n // 2, len(audio))
    a = np.abs(audio[start:end]).astype(np.float32)
    h = median_filter_size // 2
    pad = np.concatenate([np.zeros(h, np.float32), a, np.zeros(h, np.float32)])
    f = np.array([np.sort(pad[i:i + median_filter_size])[h] for i in range(len(a))], np.float32)
    b = f > np.float32(threshold_factor) * f.max()
    bp = np.concatenate([np.zeros(2, bool), b, np.zeros(2, bool)])
    er = np.array([bp[i:i + 5].all() for i in range(len(b))])
    ep = np.concatenate([np.zeros(2, bool), er, np.zeros(2, bool)])
    op = np.array([ep[i:i + 5].any() for i in range(len(b))])
    return start + int(np.argmax(op))
