"""ctypes front end of oracle/ofp_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "libofp_oracle.so"


def _load():
    import os
    so = os.environ.get("OFP_ORACLE_SO")  # the CPU sanitizer test substitutes its ASan / UBSan build (oracle/Makefile)
    if so:
        return ctypes.CDLL(so)
    if not _SO.exists():
        subprocess.check_call(["make", "-C", str(_HERE), str(_SO)])
    return ctypes.CDLL(str(_SO))


lib = _load()

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class _Params(ctypes.Structure):
    _fields_ = [
        ("C", ctypes.c_int),
        ("B", ctypes.c_int),
        ("floor_db", ctypes.c_float),
        ("hp_on", ctypes.c_int),
        ("b", ctypes.c_float * 5),
        ("a", ctypes.c_float * 5),
        ("fast_att", ctypes.c_float),
        ("fast_rel", ctypes.c_float),
        ("slow_att", ctypes.c_float),
        ("slow_rel", ctypes.c_float),
        ("alpha_min", ctypes.c_float),
        ("alpha_max", ctypes.c_float),
        ("minmin", ctypes.c_float),
        ("manual", ctypes.c_int),
        ("cooldown", ctypes.c_long),
        ("backtrack", ctypes.c_int),
        ("bt_N", ctypes.c_long),
        ("bt_alpha", ctypes.c_float),
        ("bt_tol", ctypes.c_float),
    ]


lib.oracle_ar_envelope.argtypes = [_f32p, _f32p, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int]
lib.oracle_minmax_envelope.argtypes = [_f32p, _f32p, _f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int]
lib.oracle_backtrack_onsets.argtypes = [_f32p, _i64p, _i64p, ctypes.c_float, ctypes.c_float, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long]
lib.oracle_lfilter4.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_long, ctypes.c_int]
for _n in ("oracle_rect_db", "oracle_rel_linear"):
    getattr(lib, _n).argtypes = [_f32p, _f32p, ctypes.c_long, ctypes.c_float]
for _n in ("oracle_log10f", "oracle_exp10f"):
    getattr(lib, _n).argtypes = [_f32p, _f32p, ctypes.c_long]
lib.oracle_detector_create.restype = ctypes.c_void_p
lib.oracle_detector_create.argtypes = [ctypes.POINTER(_Params), _f64p, _f64p]
lib.oracle_detector_destroy.argtypes = [ctypes.c_void_p]
lib.oracle_detector_warmup.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_long]
lib.oracle_detector_calibrate.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _f32p]
lib.oracle_detector_set_thresholds.argtypes = [ctypes.c_void_p, _f64p, _f64p]
lib.oracle_detector_block.restype = ctypes.c_long
lib.oracle_detector_block.argtypes = [ctypes.c_void_p, _f32p, _f32p, _i64p, _i64p]
lib.oracle_detect.restype = ctypes.c_long
lib.oracle_detect.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_long, ctypes.c_long, _f32p, _i64p, _i64p, ctypes.c_long]
_MATHFN = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_long, ctypes.c_float)
lib.oracle_detector_set_math.argtypes = [ctypes.c_void_p, _MATHFN, _MATHFN]
lib.oracle_detector_get_state.argtypes = [ctypes.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _u8p, _f64p, _i64p]


def ar_envelope(x, y, attack, release):
    """envelope_follower.c:6-25; x,y [n][C] float32, y in/out. `attack`/`release`
    are the coefficients (np.float32(1/attack), detection.py:514-515)."""
    n, size = x.shape
    lib.oracle_ar_envelope(x, y, attack, release, size, n)
    return y


def minmax_envelope(x, min_val, max_val, alpha_min, alpha_max, minmin):
    """envelope_follower.c:27-57."""
    n, C = x.shape
    lib.oracle_minmax_envelope(x, min_val, max_val, alpha_min, alpha_max, minmin, n, C)
    return min_val, max_val


def backtrack_onsets(buffer, channels, deltas, alpha, tol, block_size):
    """envelope_follower.c:59-85; deltas modified in place."""
    N, C = buffer.shape
    lib.oracle_backtrack_onsets(buffer, channels, deltas, alpha, tol, N, len(channels), C, block_size)
    return deltas


def butter_hp_f32(cutoff, order, sr, btype="high"):
    """detection.py:492-496: scipy design, cast to float32."""
    from scipy import signal as sig

    b, a = sig.butter(order, cutoff, btype=btype, analog=False, output="ba", fs=sr)
    return np.float32(b), np.float32(a)


def lfilter4(x, b, a, zi):
    """detection.py:499-501 for order 4: returns y, updates zi [4][C] in place."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib.oracle_lfilter4(x, y, np.ascontiguousarray(b, np.float32), np.ascontiguousarray(a, np.float32), zi, x.shape[0], x.shape[1])
    return y


def _np_db(xp, yp, n, floor):
    """detection.py:747-748 with the HOST numpy's float32 log10 (what the
    reference itself executes on this machine)."""
    x = np.ctypeslib.as_array(xp, (n,))
    y = np.ctypeslib.as_array(yp, (n,))
    y[:] = (20 * np.log10(np.abs(x + 1e-10))).clip(np.float32(floor))


def _np_lin(xp, yp, n, floor):
    """detection.py:753-754 with the HOST numpy's float32 power."""
    x = np.ctypeslib.as_array(xp, (n,))
    y = np.ctypeslib.as_array(yp, (n,))
    y[:] = (10 ** (x / 20) - 1e-10).clip(0, np.float32(-floor))


_NP_DB, _NP_LIN = _MATHFN(_np_db), _MATHFN(_np_lin)


def host_math_probe():
    """Fixed inputs through the host numpy's float32 log10 / power; compared with
    the probe stored beside the golden vectors to decide whether this host's
    numpy reproduces the machine the goldens were captured on."""
    rng = np.random.default_rng(77)
    a = np.exp(rng.uniform(-23, 2, 4096)).astype(np.float32)
    v = rng.uniform(-3.5, 3.5, 4096).astype(np.float32)
    return a, np.log10(a), v, np.float32(10) ** v


def _ew(fn, x, *extra):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    fn(x.reshape(-1), y.reshape(-1), x.size, *extra)
    return y


def log10f(x):
    return _ew(lib.oracle_log10f, x)


def exp10f(x):
    return _ew(lib.oracle_exp10f, x)


def rect_db(x, floor):
    """detection.py:747-748."""
    return _ew(lib.oracle_rect_db, x, floor)


def rel_linear(d, floor):
    """detection.py:753-754."""
    return _ew(lib.oracle_rel_linear, d, floor)


def init_ranges(n, block_size, sr):
    """Rows [r0, r1) of the settling blocks of AmplitudeOnsetDetector.init
    (detection.py:855-860: `range(int(0.1 * sr), int(0.5 * sr), block_size)`), after checking that the
    reference itself stays inside its buffers for this (n, block_size, sr)."""
    starts = range(int(0.1 * sr), int(0.5 * sr), block_size)
    r0, r1 = starts[0], starts[-1] + block_size
    if n % block_size or sr % block_size or r1 > n or n < sr:
        raise ValueError(
            f"init: len(x) = {n} and sr = {sr} must be multiples of block_size = {block_size}, with at least "
            f"max(sr, {r1}) samples: the reference's follower calls always process block_size rows "
            "(detection.py:534-537) and read past the end of a shorter last block")
    return r0, r1


class OracleDetector:
    """AmplitudeOnsetDetector (detection.py:595-888) restated; same constructor
    arguments and ``__call__`` contract: x [B][C] float32 ->
    (channels int64[k], deltas int64[k], rel float32 [B][C])."""

    def __init__(self, n_signals, block_size=32, floor=-70.0, hipass_freq=2000.0,
                 fast_ar=(3.0, 383.0), slow_ar=(2205.0, 2205.0), on_threshold=0.5,
                 off_threshold=0.1, cooldown=1323, backtrack=False,
                 backtrack_buffer_size=80, backtrack_smooth_size=5, sr=44100,
                 host_math=False):
        p = _Params()
        p.C, p.B = n_signals, block_size
        p.floor_db = floor
        p.hp_on = int(hipass_freq != 0)
        if p.hp_on:
            b, a = butter_hp_f32(hipass_freq, 4, sr)
            p.b[:] = list(b)
            p.a[:] = list(a)
        p.fast_att, p.fast_rel = np.float32(1 / fast_ar[0]), np.float32(1 / fast_ar[1])
        p.slow_att, p.slow_rel = np.float32(1 / slow_ar[0]), np.float32(1 / slow_ar[1])
        p.alpha_min, p.alpha_max, p.minmin = 1e-4, 1e-5, 2.0
        on = np.broadcast_to(np.asarray(on_threshold, dtype=np.float64), (n_signals,)).copy()
        off = np.broadcast_to(np.asarray(off_threshold, dtype=np.float64), (n_signals,)).copy()
        p.manual = int(np.all(on > 1))
        p.cooldown = int(cooldown)
        p.backtrack = int(backtrack)
        p.bt_N = backtrack_buffer_size
        if backtrack:
            assert block_size <= backtrack_buffer_size
            b_alpha = np.float32(2 / (backtrack_smooth_size + 1))
            p.bt_alpha = b_alpha
            p.bt_tol = np.float32((1 - b_alpha) ** backtrack_buffer_size)
        self.params = p
        self.on_threshold, self.off_threshold = on_threshold, off_threshold
        self.n_signals, self.block_size, self.sr = n_signals, block_size, sr
        self._h = ctypes.c_void_p(lib.oracle_detector_create(ctypes.byref(p), on, off))
        if host_math:  # tests only: the host numpy's log10/power instead of the canon
            lib.oracle_detector_set_math(self._h, _NP_DB, _NP_LIN)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.oracle_detector_destroy(self._h)
            self._h = None

    def init_minmax_tracker(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        lib.oracle_detector_warmup(self._h, x, x.shape[0])

    def init(self, x):
        """detection.py:842-888.  Defined where the reference is: len(x) and sr multiples of the block
        size (its follower calls always process block_size rows, detection.py:534-537), the settling
        blocks inside x, at least one second of audio."""
        from scipy.ndimage import maximum_filter1d

        x = np.ascontiguousarray(x, dtype=np.float32)
        n, B, sr = len(x), self.block_size, self.sr
        r0, r1 = init_ranges(n, B, sr)
        rel = np.empty((n, self.n_signals), np.float32)
        lib.oracle_detector_calibrate(self._h, x, n, r0, r1, sr, rel)
        self.mins = np.median(rel[:sr], axis=0)
        self.maxs = np.max(rel, axis=0)
        self.on_threshold = self.maxs * self.on_threshold + self.mins      # :871-872
        self.off_threshold = self.maxs * self.off_threshold + self.mins
        self.noise_max = np.median(maximum_filter1d(rel, int(sr * 0.01), axis=0), axis=0)
        lib.oracle_detector_set_thresholds(self._h, np.asarray(self.on_threshold, np.float64),
                                           np.asarray(self.off_threshold, np.float64))
        return rel

    def __call__(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        C = self.n_signals
        rel = np.empty((self.block_size, C), np.float32)
        ch = np.empty(C, np.int64)
        de = np.empty(C, np.int64)
        k = lib.oracle_detector_block(self._h, x, rel, ch, de)
        return ch[:k].copy(), de[:k].copy(), rel

    def detect(self, x, warm):
        x = np.ascontiguousarray(x, dtype=np.float32)
        N, C = x.shape
        B = self.block_size
        nb = N // B
        rel = np.empty((nb * B, C), np.float32)
        cap = max(1, nb * C)
        ch = np.empty(cap, np.int64)
        on = np.empty(cap, np.int64)
        k = lib.oracle_detect(self._h, x, N, min(int(warm), N), rel, ch, on, cap)
        return ch[:k].copy(), on[:k].copy(), rel

    def state(self):
        C = self.n_signals
        s = dict(zi=np.empty((4, C), np.float32), yf=np.empty(C, np.float32),
                 ys=np.empty(C, np.float32), mn=np.empty(C, np.float32),
                 mx=np.empty(C, np.float32), state=np.empty(C, np.uint8),
                 prev=np.empty(C, np.float64), deb=np.empty(C, np.int64))
        lib.oracle_detector_get_state(self._h, s["zi"], s["yf"], s["ys"], s["mn"], s["mx"],
                                      s["state"], s["prev"], s["deb"])
        return s


def detect_onsets_amplitude(x, block_size=128, floor=-70.0, hipass_freq=2000.0,
                            fast_ar=(3.0, 383.0), slow_ar=(2205.0, 2205.0),
                            on_threshold=0.5, off_threshold=0.1, cooldown=1323,
                            backtrack=False, backtrack_buffer_size=128,
                            backtrack_smooth_size=5, sr=96000, host_math=False):
    """detection.py:19-86: returns (channels list, onsets list, rel [N'][C])."""
    od = OracleDetector(x.shape[1], block_size, floor=floor, hipass_freq=hipass_freq,
                        fast_ar=fast_ar, slow_ar=slow_ar, on_threshold=on_threshold,
                        off_threshold=off_threshold, cooldown=cooldown, sr=sr,
                        backtrack=backtrack, backtrack_buffer_size=backtrack_buffer_size,
                        backtrack_smooth_size=backtrack_smooth_size, host_math=host_math)
    ch, on, rel = od.detect(x, int(0.5 * sr))
    return list(ch), list(on), rel
