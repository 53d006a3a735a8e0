"""Onset detection on MI355X -- drop-in surface of the reference's
``onset_fingerprinting/detection.py`` for the amplitude detector.

Same names, arguments, defaults, return shapes and dtypes as the reference
(detection.py:12-86, :487-888); the work is done by hand-written HIP kernels in
``libonsetfp.so`` (csrc/ofp_detect.hip, ofp_stream.hip) through ctypes.  There is
no CPU path: without the library or a gfx950 GPU every call raises.

Additional, batched entry points (`BatchDetector`, `detect_batch`) keep audio,
relative envelope and onset records resident in HBM.
"""
import ctypes
from typing import Optional

import numpy as np
import torch
from scipy import signal as sig

from . import _lib
from ._lib import DetectorParams, DetectTuning, OnsetFPError, check

ONSET_DTYPE = np.dtype([("clip", np.int32), ("channel", np.int32), ("sample", np.int64)])


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev(device):
    if isinstance(device, torch.device):
        return device
    return torch.device("cuda", int(device))


class _DeviceDetector:
    """Owns the opaque ofp_detector handle built from the reference's kwargs."""

    def __init__(self, n_signals, block_size, floor, hipass_freq, fast_ar, slow_ar, on_threshold,
                 off_threshold, cooldown, backtrack, backtrack_buffer_size, backtrack_smooth_size, sr,
                 device=0):
        self.lib = _lib.lib()
        _lib.require_gpu(_dev(device).index or 0)
        self.device = _dev(device)
        torch.cuda.set_device(self.device)
        p = DetectorParams()
        p.n_channels = int(n_signals)
        p.block_size = int(block_size)
        p.floor_db = floor
        p.hp_enabled = int(hipass_freq != 0)
        if p.hp_enabled:
            # detection.py:492-496: scipy design in double, coefficients cast to float32
            b, a = sig.butter(4, hipass_freq, btype="high", analog=False, output="ba", fs=sr)
            p.hp_b[:] = [float(v) for v in np.float32(b)]
            p.hp_a[:] = [float(v) for v in np.float32(a)]
        # detection.py:514-515: the follower coefficients are np.float32(1 / x)
        p.fast_attack, p.fast_release = np.float32(1 / fast_ar[0]), np.float32(1 / fast_ar[1])
        p.slow_attack, p.slow_release = np.float32(1 / slow_ar[0]), np.float32(1 / slow_ar[1])
        # detection.py:703-708
        p.alpha_min, p.alpha_max, p.minmin = 1e-4, 1e-5, 2.0
        p.min0, p.max0 = 0.0, 10.0
        on = np.ascontiguousarray(np.broadcast_to(np.asarray(on_threshold, dtype=np.float64), (n_signals,)))
        off = np.ascontiguousarray(np.broadcast_to(np.asarray(off_threshold, dtype=np.float64), (n_signals,)))
        p.manual = int(bool(np.all(np.asarray(on_threshold) > 1)))  # detection.py:687
        p.cooldown = int(cooldown)
        p.backtrack = int(bool(backtrack))
        p.backtrack_buffer_size = int(backtrack_buffer_size)
        if backtrack:
            assert block_size <= backtrack_buffer_size, \
                "backtrack_buffer_size should be at least block_size!"  # detection.py:716-718
            b_alpha = np.float32(2 / (backtrack_smooth_size + 1))
            p.backtrack_alpha = b_alpha
            p.backtrack_tol = np.float32((1 - b_alpha) ** backtrack_buffer_size)
        self.params = p
        self._on, self._off = on, off
        h = ctypes.c_void_p()
        check(self.lib.ofp_detector_create(ctypes.byref(p), on.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                           off.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                           ctypes.byref(h)), "ofp_detector_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ofp_detector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, **kw):
        t = DetectTuning()
        for k, v in kw.items():
            setattr(t, k, int(v))
        check(self.lib.ofp_detector_set_tuning(self.handle, ctypes.byref(t)), "ofp_detector_set_tuning")


class BatchDetector:
    """Offline detection of a batch of independent clips, resident in HBM.

    One instance == the reference's ``detect_onsets_amplitude`` configuration
    (detection.py:19-86) applied to every clip of ``x [n_clips, N, C]``.
    """

    def __init__(self, n_signals, block_size=128, floor=-70.0, hipass_freq=2000.0,
                 fast_ar=(3.0, 383.0), slow_ar=(2205.0, 2205.0), on_threshold=0.5, off_threshold=0.1,
                 cooldown=1323, backtrack=False, backtrack_buffer_size=128, backtrack_smooth_size=5,
                 sr=96000, device=0):
        self.d = _DeviceDetector(n_signals, block_size, floor, hipass_freq, fast_ar, slow_ar, on_threshold,
                                 off_threshold, cooldown, backtrack, backtrack_buffer_size,
                                 backtrack_smooth_size, sr, device)
        self.n_signals, self.block_size, self.sr = n_signals, block_size, sr
        self.device = self.d.device
        self._ws = None
        self.last_info = None

    def set_tuning(self, **kw):
        self.d.set_tuning(**kw)

    def workspace_bytes(self, n_clips, n_samples, warm):
        return int(self.d.lib.ofp_detect_workspace_bytes(self.d.handle, n_clips, n_samples, warm))

    def reserve(self, n_clips, n_samples, warm):
        need = self.workspace_bytes(n_clips, n_samples, warm)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def begin(self, x, warm=None):
        """Enqueue only the head of `detect` (input transpose + the IIR candidate launch) on the
        current stream and return; follow with ``detect(..., begun=True)`` and the same x/warm."""
        if x.dim() == 2:
            x = x.unsqueeze(0)
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        n_clips, N, C = x.shape
        warm = int(0.5 * self.sr) if warm is None else int(warm)
        ws = self.reserve(n_clips, N, warm)
        check(self.d.lib.ofp_detect_offline_begin(self.d.handle, x.data_ptr(), n_clips, N, warm, ws.data_ptr(),
                                                  ws.numel(), _stream_ptr(x.device)), "ofp_detect_offline_begin")

    def begin_input(self, x, warm=None):
        """`begin` in two calls: this one enqueues the planar copy of the input only (after it,
        `planar_input` is valid on this stream), `begin_iir` the IIR candidate launch."""
        self._part("ofp_detect_offline_begin_input", x, warm)

    def begin_iir(self, x, warm=None):
        self._part("ofp_detect_offline_begin_iir", x, warm)

    def _part(self, name, x, warm):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        n_clips, N, C = x.shape
        warm = int(0.5 * self.sr) if warm is None else int(warm)
        ws = self.reserve(n_clips, N, warm)
        check(getattr(self.d.lib, name)(self.d.handle, x.data_ptr(), n_clips, N, warm, ws.data_ptr(), ws.numel(),
                                        _stream_ptr(x.device)), name)

    def planar_input(self, x, warm=None):
        """(device address, stride in floats) of the planar copy of `x` (one series per clip and
        channel, `stride` floats apart) that `begin` (or `detect`) left in the work space; valid
        until the next begin/detect on this detector.  None when the detector's layout reads `x` itself."""
        if x.dim() == 2:
            x = x.unsqueeze(0)
        n_clips, N, C = x.shape
        warm = int(0.5 * self.sr) if warm is None else int(warm)
        ws = self.reserve(n_clips, N, warm)
        stride = int(self.d.lib.ofp_detect_planar_stride(self.d.handle, n_clips, N, warm))
        if stride == 0:  # the detector works on the interleaved array itself (tuning `interleaved`): there is no planar copy
            return None
        return (self.d.lib.ofp_detect_planar_input(self.d.handle, n_clips, N, warm, ws.data_ptr()), stride)

    def _call_args(self, x, warm, want_rel, cap_per_clip, out):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        n_clips, N, C = x.shape
        assert C == self.n_signals
        B = self.block_size
        warm = int(0.5 * self.sr) if warm is None else int(warm)
        nb = N // B
        cap = int(cap_per_clip) if cap_per_clip is not None else max(1, min(nb * C, 1 << 20))
        ws = self.reserve(n_clips, N, warm)
        if out is None:
            out = {
                "records": torch.empty((n_clips, cap, 16), dtype=torch.uint8, device=x.device),
                "counts": torch.zeros(n_clips, dtype=torch.int64, device=x.device),
                "rel": torch.empty((n_clips, nb * B, C), dtype=torch.float32, device=x.device) if want_rel else None,
            }
        rel = out["rel"]
        out["cap"] = cap
        return out, (self.d.handle, x.data_ptr(), n_clips, N, warm, rel.data_ptr() if rel is not None else None,
                     out["records"].data_ptr(), cap, out["counts"].data_ptr(), ws.data_ptr(), ws.numel())

    def _set_info(self, info):
        self.last_info = dict(
            hp_passes=info[0], ar_passes=info[1], mm_passes=info[2], repaired=info[3],
            # stage durations, HIP events on the launch stream (milliseconds; zero for a call captured in a graph)
            stage_ms=dict(hp=info[4] / 1e6, db=info[5] / 1e6, ar=info[6] / 1e6, rel=info[7] / 1e6,
                          mm=info[8] / 1e6, logic=info[9] / 1e6, total=info[10] / 1e6,
                          hp_candidates=info[11] / 1e6),
            hp_candidate_steps=info[12],
            # staged candidates only (0 otherwise): distinct runs that walked a chunk
            hp_chunk_runs=info[13],
            sequential_machine_decided=bool(info[14]),
            # non-zero: the call was repeated with host-verified passes (include/onsetfp.h, info 15)
            repeated_host_verified=int(info[15]))

    def detect(self, x, warm=None, want_rel=True, cap_per_clip=None, out=None, begun=False):
        """x: float32 CUDA tensor [n_clips, N, C] (or [N, C]).  Returns a dict of
        device tensors: ``records`` (uint8 view of ofp_onset [n_clips, cap]),
        ``counts`` int64 [n_clips], ``rel`` float32 [n_clips, N', C] or None.
        One stream synchronisation, at the end of the call."""
        out, args = self._call_args(x, warm, want_rel, cap_per_clip, out)
        info = (ctypes.c_int64 * 16)()
        name = "ofp_detect_offline_finish" if begun else "ofp_detect_offline"  # (begun: the head was enqueued by begin())
        check(getattr(self.d.lib, name)(*args, info, _stream_ptr(x.device)), name)
        self._set_info(info)
        return out

    def enqueue(self, x, warm=None, want_rel=True, cap_per_clip=None, out=None, begun=False):
        """`detect` without its synchronisation: the whole call is only ENQUEUED on the current stream (which may be
        capturing a hipGraph: allocate `out` and call `reserve` before the capture).  Synchronise (or replay the
        graph and synchronise), then call ``complete`` with the same arguments."""
        out, args = self._call_args(x, warm, want_rel, cap_per_clip, out)
        name = "ofp_detect_offline_finish_enqueue" if begun else "ofp_detect_offline_enqueue"
        check(getattr(self.d.lib, name)(*args, _stream_ptr(x.device)), name)
        return out

    def complete(self, x, out, warm=None):
        """After the stream of ``enqueue`` (or a replay of the graph that captured it) has been synchronised."""
        out, args = self._call_args(x, warm, out["rel"] is not None, out["cap"], out)
        info = (ctypes.c_int64 * 16)()
        check(self.d.lib.ofp_detect_offline_complete(*args, info, _stream_ptr(x.device)), "ofp_detect_offline_complete")
        self._set_info(info)
        return out

    @staticmethod
    def records_to_numpy(out):
        """-> list (one per clip) of structured arrays (clip, channel, sample), in the
        reference's order (block, then channel)."""
        counts = out["counts"].cpu().numpy()
        cap = out["cap"]
        if counts.max(initial=0) > cap:
            raise OnsetFPError(f"{counts.max()} onsets in a clip exceed the record capacity {cap}")
        recs = out["records"].cpu().numpy().view(ONSET_DTYPE).reshape(len(counts), cap)
        return [recs[i, :counts[i]].copy() for i in range(len(counts))]


def detect_batch(x, block_size=128, sr=96000, device=0, warm=None, want_rel=True, tuning=None, **kw):
    """Convenience: numpy/tensor [n_clips, N, C] -> (list of record arrays, rel numpy or None)."""
    xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)) if not torch.is_tensor(x) else x
    xt = xt.to(_dev(device))
    if xt.dim() == 2:
        xt = xt.unsqueeze(0)
    bd = BatchDetector(xt.shape[-1], block_size, sr=sr, device=device, **kw)
    if tuning:
        bd.set_tuning(**tuning)
    out = bd.detect(xt.contiguous(), warm=warm, want_rel=want_rel)
    recs = BatchDetector.records_to_numpy(out)
    rel = out["rel"].cpu().numpy() if want_rel else None
    return recs, rel, bd.last_info


def group_onsets_device(out, n_channels, max_distance=1000, min_channels=3, close_channel=None, cap_groups=None):
    """find_onset_groups (detection.py:131-189) for every clip of a `BatchDetector.detect`
    result, on the device.  Returns ``groups`` int64 [n_clips, cap_groups, n_channels] and
    ``n_groups`` int64 [n_clips] (device tensors; no host synchronisation)."""
    L = _lib.lib()
    recs, counts = out["records"], out["counts"]
    n_clips, cap = recs.shape[0], recs.shape[1]
    cap_groups = int(cap_groups) if cap_groups is not None else cap
    ws = torch.empty(int(L.ofp_group_workspace_bytes(n_clips, cap)), dtype=torch.uint8, device=recs.device)
    groups = torch.empty((n_clips, cap_groups, n_channels), dtype=torch.int64, device=recs.device)
    n_groups = torch.zeros(n_clips, dtype=torch.int64, device=recs.device)
    check(L.ofp_group_onsets(recs.data_ptr(), cap, counts.data_ptr(), n_clips, n_channels, int(max_distance),
                             int(min_channels), -1 if close_channel is None else int(close_channel),
                             groups.data_ptr(), cap_groups, n_groups.data_ptr(), ws.data_ptr(), ws.numel(),
                             _stream_ptr(recs.device)), "ofp_group_onsets")
    return groups, n_groups


def group_windows_device(x, groups, n_groups, frame_length, pre_samples, use_min_onset=True, cap_total=None):
    """FrameExtractor (data.py:90-120, max_shift 0) over device group rows: x float32 CUDA
    [n_clips, N, C] -> ``windows`` [cap_total, C, frame_length] (rows of all clips back to back)
    and ``offsets`` int64 [n_clips + 1] (first row of each clip; last entry = rows in use)."""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    n_clips, N, C = x.shape
    cap_groups = groups.shape[1]
    assert groups.shape == (n_clips, cap_groups, C) and groups.dtype == torch.int64 and groups.is_contiguous()
    cap_total = int(cap_total) if cap_total is not None else n_clips * cap_groups
    windows = torch.empty((cap_total, C, frame_length), dtype=torch.float32, device=x.device)
    offsets = torch.empty(n_clips + 1, dtype=torch.int64, device=x.device)
    check(_lib.lib().ofp_group_windows(x.data_ptr(), n_clips, N, C, groups.data_ptr(), cap_groups, n_groups.data_ptr(),
                                  int(pre_samples), int(bool(use_min_onset)), int(frame_length), windows.data_ptr(),
                                  cap_total, offsets.data_ptr(), _stream_ptr(x.device)), "ofp_group_windows")
    return windows, offsets


def find_onset_groups(onsets, channels, max_distance: int = 1000, min_channels: int = 3,
                      close_channel: Optional[int] = None, device=0):
    """detection.py:131-189: same arguments, same return (int array [G, max(channels)+1] or
    None).  The lists go to the GPU as one clip's onset records and are grouped by
    ``ofp_group_onsets``; channels must be non-negative."""
    ch = np.asarray(channels, dtype=np.int64)
    on = np.asarray(onsets, dtype=np.int64)
    n = min(len(ch), len(on))  # zip() semantics, detection.py:160
    width = int(max(channels)) + 1  # raises ValueError on an empty list as the reference does (:158)
    if n and ch[:n].min() < 0:
        raise ValueError("find_onset_groups: negative channel index")
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    rec = np.zeros(max(n, 1), dtype=ONSET_DTYPE)
    rec["channel"][:n], rec["sample"][:n] = ch[:n], on[:n]
    out = {"records": torch.from_numpy(rec.view(np.uint8).reshape(1, -1, 16)).to(dev),
           "counts": torch.tensor([n], dtype=torch.int64, device=dev)}
    groups, n_groups = group_onsets_device(out, width, max_distance, min_channels, close_channel)
    g = int(n_groups.cpu()[0])
    return groups[0, :g].cpu().numpy().astype(int) if g else None


def _py_slice(start, stop, length):
    s, e, _ = slice(start, stop).indices(length)
    return s, max(e, s)


def xcorr_lags_device(x, y, lo, hi, d=0, take_abs=False, normalization_cutoff=10, want_cc=False):
    """Batched core of cross_correlation_lag: x, y float32 CUDA [P, n_in]; lo, hi int32 CUDA [P]
    (slice of the normalised full correlation of the d-times differenced rows).  Returns argmax
    int32 [P] (-1: empty slice) and, with want_cc, the slice values [P, max(hi-lo)]."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and y.shape == x.shape and y.is_contiguous()
    P, n_in = x.shape
    am = torch.empty(P, dtype=torch.int32, device=x.device)
    cc = None
    if want_cc:
        width = max(int((hi - lo).max().item()), 1) if P else 1
        cc = torch.zeros((P, width), dtype=torch.float32, device=x.device)
    check(_lib.lib().ofp_xcorr_lag(x.data_ptr(), y.data_ptr(), P, n_in, int(d), int(bool(take_abs)),
                                   int(normalization_cutoff), lo.data_ptr(), hi.data_ptr(), am.data_ptr(),
                                   cc.data_ptr() if cc is not None else None, cc.shape[1] if cc is not None else 0,
                                   _stream_ptr(x.device)), "ofp_xcorr_lag")
    return (am, cc) if want_cc else am


def cross_correlation_lag(x: np.ndarray, y: np.ndarray, onsets=None, legal_lags=None, d: int = 0,
                          normalization_cutoff: int = 10, onset_tolerance: int = 50, take_abs: bool = False,
                          device=0):
    """detection.py:195-268: same arguments and return (int lag, or None for an empty window)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    n = len(x) - d
    if legal_lags is not None:  # detection.py:256-258
        lo, hi = _py_slice(n - legal_lags[1], n - legal_lags[0], 2 * n - 1)
        max_adjust = legal_lags[1]
    elif onsets is not None:  # detection.py:259-264
        current_lag = onsets[1] - onsets[0]
        lo, hi = _py_slice(n - current_lag - onset_tolerance, n - current_lag + onset_tolerance, 2 * n - 1)
        max_adjust = current_lag + onset_tolerance
    else:
        raise ValueError("cross_correlation_lag: give onsets or legal_lags")
    if hi <= lo:
        return None
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    am = xcorr_lags_device(torch.from_numpy(x[None]).to(dev), torch.from_numpy(y[None]).to(dev),
                           torch.tensor([lo], dtype=torch.int32, device=dev),
                           torch.tensor([hi], dtype=torch.int32, device=dev), d, take_abs, normalization_cutoff)
    return -(int(am.cpu()[0]) - int(max_adjust))


_DIRECTIONS = {None: 0, "up": 1, "down": 2}


def fix_onsets_device(audio, onsets, filter_size=5, d=0, onset_direction=None, take_abs=False, zero_left=False,
                      normalization_cutoff=10, onset_tolerance=30, shift_onsets=0, max_section=None, n_groups=None):
    """fix_onsets on device tensors: audio float32 CUDA [N, C] with onsets int64 CUDA [G, C], or a
    batch [n_clips, N, C] with [n_clips, cap_groups, C] (the layout `group_onsets_device` writes)
    and `n_groups` int64 [n_clips].  Onsets are updated IN PLACE.  Returns status int32, one per
    row: 0 fixed, 1 shifted only (group outside the clip, with a missing channel, or longer than
    `max_section` samples, default 4096), 2 row not in use."""
    assert audio.is_cuda and audio.dtype == torch.float32 and audio.is_contiguous()
    assert onsets.is_cuda and onsets.dtype == torch.int64 and onsets.is_contiguous()
    if onset_direction not in _DIRECTIONS:
        raise RuntimeError(f"Unknown onset direction {onset_direction=}!")
    L = _lib.lib()
    if audio.dim() == 2:
        audio, onsets_b = audio.unsqueeze(0), onsets.unsqueeze(0)
    else:
        onsets_b = onsets
    n_clips, N, C = audio.shape
    G = onsets_b.shape[1]
    assert onsets_b.shape == (n_clips, G, C)
    max_section = 4096 if max_section is None else int(max_section)
    status = torch.zeros(onsets.shape[:-1], dtype=torch.int32, device=audio.device)
    ws = torch.empty(max(int(L.ofp_fix_onsets_workspace_bytes(n_clips * G, C, max_section)), 16), dtype=torch.uint8,
                     device=audio.device)
    check(L.ofp_fix_onsets(audio.data_ptr(), n_clips, N, C, onsets.data_ptr(), G,
                           n_groups.data_ptr() if n_groups is not None else None, int(filter_size), int(d),
                           _DIRECTIONS[onset_direction], int(bool(take_abs)), int(bool(zero_left)),
                           int(normalization_cutoff), int(onset_tolerance), int(shift_onsets), max_section,
                           status.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(audio.device)),
          "ofp_fix_onsets")
    return status


def fix_onsets(audio: np.ndarray, onsets: np.ndarray, filter_size: int = 5, d: int = 0, onset_direction=None,
               take_abs: bool = False, zero_left: bool = False, normalization_cutoff: int = 10,
               onset_tolerance: int = 30, shift_onsets: int = 0, device=0):
    """detection.py:373-451: same arguments, returns the fixed copy of `onsets` [O, C].  A group
    whose section would leave the clip (the reference's slice would wrap or shrink there) raises."""
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    a = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(dev)
    o = torch.from_numpy(np.ascontiguousarray(onsets, dtype=np.int64).copy()).to(dev)
    spread = int((np.max(onsets, axis=1) - np.min(onsets, axis=1)).max(initial=0))
    status = fix_onsets_device(a, o, filter_size, d, onset_direction, take_abs, zero_left, normalization_cutoff,
                               onset_tolerance, shift_onsets,
                               max_section=max(spread + 2 * (normalization_cutoff + onset_tolerance), 1))
    if int(status.max().item() if status.numel() else 0) != 0:
        raise IndexError("fix_onsets: an onset group lies within normalization_cutoff + onset_tolerance samples of "
                         "the clip edge (or spans more than 4096 samples)")
    return o.cpu().numpy().astype(np.asarray(onsets).dtype, copy=False)


def adjust_onset_rel(onsets, relx: np.ndarray, rely: np.ndarray, new_lag: int):
    """detection.py:271-296: adjust one onset of a pair to a target lag by comparing four samples of the
    two relative envelopes -- index logic on the caller's arrays, no sample pass (host, like
    `window_contribution_weights`)."""
    oa, ob = onsets[0], onsets[1]
    lag_diff = (ob - oa) - new_lag
    da = relx[oa + lag_diff] - relx[oa]
    db = rely[ob - lag_diff] - rely[ob]
    if da > db:
        oa += lag_diff
    else:
        ob -= lag_diff
    return oa, ob


def adjust_onsets_device(x, y, onsets, new_lag):
    """Batched `adjust_onset`: x, y float32 CUDA [P, n]; onsets int32 CUDA [P, 2]; new_lag int32 CUDA [P]
    -> moves int32 [P, 2] (the amounts to add to the two onsets), one wave per pair."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and y.shape == x.shape and y.is_contiguous()
    P, n = x.shape
    moves = torch.empty((P, 2), dtype=torch.int32, device=x.device)
    check(_lib.lib().ofp_adjust_onset(x.data_ptr(), y.data_ptr(), P, n, onsets.data_ptr(), new_lag.data_ptr(),
                                      moves.data_ptr(), _stream_ptr(x.device)), "ofp_adjust_onset")
    return moves


def adjust_onset(onsets, x: np.ndarray, y: np.ndarray, new_lag: int, device=0):
    """detection.py:299-352: same arguments, returns the pair (move of onset x, move of onset y)."""
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)[None]).to(dev)
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)[None]).to(dev)
    m = adjust_onsets_device(xd, yd, torch.tensor([[int(onsets[0]), int(onsets[1])]], dtype=torch.int32, device=dev),
                             torch.tensor([int(new_lag)], dtype=torch.int32, device=dev)).cpu().numpy()[0]
    return int(m[0]), int(m[1])


def filter_data(x: np.ndarray, direction: str, device=0) -> np.ndarray:
    """detection.py:355-370: nulls, IN PLACE like the reference, every sample whose first difference along
    axis 0 is negative ("up") / positive ("down"); returns x."""
    if direction not in ("up", "down"):
        raise RuntimeError(f"Unknown onset direction {direction=}!")
    if not (isinstance(x, np.ndarray) and x.dtype == np.float32):
        # the reference keeps the caller's dtype and takes the difference in it; a float64 array rounded to float32 can
        # lose the sign of a difference, so anything but float32 is refused rather than answered approximately
        raise TypeError(f"filter_data: float32 array expected (the device kernel computes in float32), got "
                        f"{getattr(x, 'dtype', type(x))}")
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    a = np.ascontiguousarray(x)
    n = a.shape[0] if a.ndim else 0
    cols = int(a.size // max(n, 1)) if n else 1
    xd = torch.from_numpy(a.reshape(n, cols)).to(dev)
    yd = torch.empty_like(xd)
    check(_lib.lib().ofp_filter_direction(xd.data_ptr(), n, cols, 1 if direction == "up" else 2, yd.data_ptr(),
                                          _stream_ptr(dev)), "ofp_filter_direction")
    x[...] = yd.cpu().numpy().reshape(a.shape)
    return x


def detect_onset_regions_device(audio, onsets, n=256, median_filter_size=5, threshold_factor=0.5):
    """Batched `detect_onset_region`: audio float32 CUDA [N] (1-D), onsets int64 CUDA [K] -> int64 [K]."""
    assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 1 and audio.is_contiguous()
    out = torch.empty(onsets.shape[0], dtype=torch.int64, device=audio.device)
    check(_lib.lib().ofp_onset_region(audio.data_ptr(), audio.shape[0], onsets.data_ptr(), onsets.shape[0], int(n),
                                      int(median_filter_size), float(threshold_factor), out.data_ptr(),
                                      _stream_ptr(audio.device)), "ofp_onset_region")
    return out


def detect_onset_region(audio, detected_onset, n=256, median_filter_size=5, threshold_factor=0.5, device=0):
    """detection.py:454-484: in the |audio| around the onset, the start of the loud part (median filter,
    threshold at `threshold_factor` of its maximum, binary opening, first True)."""
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    a = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32).reshape(-1)).to(dev)
    o = torch.tensor([int(detected_onset)], dtype=torch.int64, device=dev)
    return int(detect_onset_regions_device(a, o, n, median_filter_size, threshold_factor).cpu()[0])


def detect_onsets(x: np.ndarray, sr: int = 96000, method="amp"):
    """detection.py:12-16."""
    if method == "amp":
        return detect_onsets_amplitude(x, sr=sr)
    return detect_onsets_spectral(x, sr=sr)


def a_weighting(frequencies, min_db=-80.0):
    """librosa.A_weighting as published (IEC 61672 constants), used at detection.py:105."""
    f_sq = np.asanyarray(frequencies, dtype=np.float64) ** 2.0
    const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
    with np.errstate(divide="ignore"):
        w = 2.0 + 20.0 * (np.log10(const[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + const[0])
                          - np.log10(f_sq + const[1]) - 0.5 * np.log10(f_sq + const[2])
                          - 0.5 * np.log10(f_sq + const[3]))
    return w if min_db is None else np.maximum(min_db, w)


def detect_onsets_spectral(x: np.ndarray, n_fft: int = 256, hop: int = 32, sr: int = 96000,
                           return_oe: bool = False, device=0):
    """detection.py:89-128: A-weighted positive spectral flux, normalised by its 99.9th
    percentile, peak-picked; returns peak sample positions (and the onset envelope).

    Every array step runs on the GPU (centred STFT power, flux, order statistics, peak picking).
    PARITY UNPINNED: the reference calls librosa (stft, A_weighting, util.peak_pick), which is not
    installable here; those three are restated from librosa's published definitions
    (stft: center=True with zero padding, periodic Hann, n_fft-long window)."""
    from .data import stft_power_dense
    L = _lib.lib()
    dev = _dev(device)
    _lib.require_gpu(dev.index or 0)
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    n = len(x)
    xp = torch.zeros(n + 2 * (n_fft // 2), dtype=torch.float32, device=dev)
    xp[n_fft // 2:n_fft // 2 + n] = torch.from_numpy(x).to(dev)
    power = stft_power_dense(xp.reshape(1, -1, 1), n_fft, hop)[0, 0]      # [T, bins], T = 1 + n // hop
    T, bins = power.shape
    freq = np.fft.fftfreq(n_fft, 1 / sr)[:bins]                           # :97 (last bin is -sr/2)
    aw = a_weighting(freq)
    w = torch.from_numpy(((aw - aw.min()) / np.abs(aw.min())).astype(np.float32)).to(dev)  # :105-106
    st = _stream_ptr(dev)
    m = max(T - 1, 0)
    oe = torch.zeros(max(m, 1), dtype=torch.float32, device=dev)
    if m == 0:
        return (np.zeros(0, np.int64), np.zeros(0, np.float32)) if return_oe else np.zeros(0, np.int64)
    check(L.ofp_spectral_flux(power.data_ptr(), T, bins, w.data_ptr(), oe.data_ptr(), st), "ofp_spectral_flux")
    # np.percentile(oe, 99.9), linear interpolation between two order statistics (:111)
    pos = 0.999 * (m - 1)
    lo = int(np.floor(pos))
    hi = min(lo + 1, m - 1)
    v = torch.empty(2, dtype=torch.float32, device=dev)
    check(L.ofp_select_rank(oe.data_ptr(), m, lo, v.data_ptr(), st), "ofp_select_rank")
    check(L.ofp_select_rank(oe.data_ptr(), m, hi, v.data_ptr() + 4, st), "ofp_select_rank")
    a, b = (float(t) for t in v.cpu())
    t = pos - lo
    p = a + (b - a) * t if t < 0.5 else b - (b - a) * (1 - t)
    scale = torch.tensor([p], dtype=torch.float32, device=dev)
    check(L.ofp_scale_inverse(oe.data_ptr(), m, scale.data_ptr(), st), "ofp_scale_inverse")
    # librosa.util.peak_pick arguments of :113-121
    pre_max, post_max = int(0.12 * sr // hop), int(0.01 * sr // hop)
    pre_avg, post_avg = int(0.12 * sr // hop), int(0.01 * sr // hop + 1)
    wait = int(sr * 0.07 // hop)
    peaks = torch.empty(m, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    flags = torch.empty(m, dtype=torch.uint8, device=dev)
    check(L.ofp_peak_pick(oe.data_ptr(), m, pre_max, max(post_max, 1), pre_avg, max(post_avg, 1), 0.1, wait,
                          peaks.data_ptr(), m, count.data_ptr(), flags.data_ptr(), st), "ofp_peak_pick")
    k = int(count.cpu()[0])
    out = peaks[:k].cpu().numpy() * hop  # :124
    return (out, oe.cpu().numpy()) if return_oe else out


def detect_onsets_amplitude(
    x: np.ndarray,
    block_size: int = 128,
    floor: float = -70.0,
    hipass_freq: float = 2000.0,
    fast_ar: tuple = (3.0, 383.0),
    slow_ar: tuple = (2205.0, 2205.0),
    on_threshold=0.5,
    off_threshold=0.1,
    cooldown: int = 1323,
    backtrack: bool = False,
    backtrack_buffer_size: int = 128,
    backtrack_smooth_size: int = 5,
    sr: int = 96000,
    device=0,
):
    """Detects onsets using amplitude followers (detection.py:19-86): x is NxC
    float32; returns ``(channels_flat, onsets_flat, rel)`` with ``rel`` of shape
    ``[floor(N/block_size)*block_size, C]``."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    recs, rel, _ = detect_batch(
        x[None], block_size=block_size, sr=sr, device=device, floor=floor, hipass_freq=hipass_freq,
        fast_ar=fast_ar, slow_ar=slow_ar, on_threshold=on_threshold, off_threshold=off_threshold,
        cooldown=cooldown, backtrack=backtrack, backtrack_buffer_size=backtrack_buffer_size,
        backtrack_smooth_size=backtrack_smooth_size)
    r = recs[0]
    return [np.int64(c) for c in r["channel"]], [np.int64(s) for s in r["sample"]], rel[0]


class AmplitudeOnsetDetector:
    """Multi-channel amplitude/time-domain onset detector (detection.py:595-888),
    streaming form: the detector state lives in HBM and each ``__call__`` is one
    kernel launch (csrc/ofp_stream.hip)."""

    def __init__(self, n_signals: int, block_size: int = 32, floor: float = -70.0,
                 hipass_freq: float = 2000.0, fast_ar=(3.0, 383.0), slow_ar=(2205.0, 2205.0),
                 on_threshold: float = 0.5, off_threshold: float = 0.1, cooldown: int = 1323,
                 backtrack: bool = False, backtrack_buffer_size: int = 80,
                 backtrack_smooth_size: int = 5, sr: int = 44100, device=0):
        self.n_signals = n_signals
        self.block_size = block_size
        self.floor = floor
        self.on_threshold = on_threshold
        self.manual = True if np.all(np.asarray(on_threshold) > 1) else False
        self.off_threshold = off_threshold
        self.cooldown = cooldown
        self.sr = sr
        self.backtrack = backtrack
        self.d = _DeviceDetector(n_signals, block_size, floor, hipass_freq, fast_ar, slow_ar, on_threshold,
                                 off_threshold, cooldown, backtrack, backtrack_buffer_size,
                                 backtrack_smooth_size, sr, device)
        dev = self.d.device
        nbytes = int(self.d.lib.ofp_stream_state_bytes(self.d.handle))
        self._state = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        check(self.d.lib.ofp_stream_state_init(self.d.handle, self._state.data_ptr(), _stream_ptr(dev)),
              "ofp_stream_state_init")
        self._xbuf = torch.empty((block_size, n_signals), dtype=torch.float32, device=dev)
        self._rel = torch.empty((block_size, n_signals), dtype=torch.float32, device=dev)
        self._rec = torch.empty((max(1, n_signals), 16), dtype=torch.uint8, device=dev)
        self._cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def _check_block(self, x):
        x = np.ascontiguousarray(x)
        if x.dtype != np.float32:
            # the reference's ctypes ndpointer rejects non-float32 input (detection.py:521-526)
            raise ctypes.ArgumentError(f"array must have data type float32, got {x.dtype}")
        return x

    def process(self, x_dev, n_blocks, sample_base=0, rel=None, records=None, count=None):
        """Device-resident form: x_dev float32 CUDA [n_blocks*B, C]; appends to
        `records`/`count` (device).  No host synchronisation."""
        cap = records.shape[0] if records is not None else 0
        check(self.d.lib.ofp_stream_process(
            self.d.handle, self._state.data_ptr(), x_dev.data_ptr(), n_blocks, 0, 0, sample_base,
            rel.data_ptr() if rel is not None else None,
            records.data_ptr() if records is not None else None, cap,
            count.data_ptr() if count is not None else None, _stream_ptr(x_dev.device)), "ofp_stream_process")

    def __call__(self, x: np.ndarray):
        """x: [block_size, n_signals] float32 -> (channels, deltas, relative_envelope)
        (detection.py:727-798)."""
        x = self._check_block(x)
        if x.shape != (self.block_size, self.n_signals):
            raise ValueError(f"expected block of shape {(self.block_size, self.n_signals)}, got {x.shape}")
        self._xbuf.copy_(torch.from_numpy(x))
        self._cnt.zero_()
        self.process(self._xbuf, 1, 0, self._rel, self._rec, self._cnt)
        k = int(self._cnt.item())
        recs = self._rec.cpu().numpy().view(ONSET_DTYPE).reshape(-1)[:k]
        return recs["channel"].astype(np.int64), recs["sample"].astype(np.int64), self._rel.cpu().numpy()

    def init_minmax_tracker(self, x):
        """detection.py:827-840."""
        x = self._check_block(x)
        if len(x) == 0:
            return
        xd = torch.from_numpy(x).to(self.d.device)
        check(self.d.lib.ofp_stream_process(
            self.d.handle, self._state.data_ptr(), xd.data_ptr(), 0, x.shape[0], 1, 0, None, None, 0, None,
            _stream_ptr(xd.device)), "ofp_stream_process(warmup)")
        torch.cuda.current_stream(xd.device).synchronize()


    def init(self, x):
        """Initialize the detector with data containing stretches of silence as well as stretches of
        audio approximately as loud as it will get during performance (detection.py:842-888): sets
        per-channel `on_threshold` / `off_threshold` (and `mins`, `maxs`, `noise_max`), leaves the
        filter and follower state where the reference leaves it, prints the reference's message.

        Defined where the reference is: ``len(x)`` and ``sr`` multiples of ``block_size`` (its
        follower calls always process ``block_size`` rows, detection.py:534-537, and read past the end
        of a shorter last block), at least one second of audio."""
        from scipy.ndimage import maximum_filter1d

        x = self._check_block(x)
        if x.ndim != 2 or x.shape[1] != self.n_signals:
            raise ValueError(f"expected audio of shape (n, {self.n_signals}), got {x.shape}")
        n, B, sr = len(x), self.block_size, self.sr
        starts = range(int(0.1 * sr), int(0.5 * sr), B)
        r0, r1 = starts[0], starts[-1] + B
        if n % B or sr % B or r1 > n or n < sr:
            raise ValueError(
                f"init: len(x) = {n} and sr = {sr} must be multiples of block_size = {B}, with at least "
                f"max(sr, {r1}) samples: the reference's follower calls always process block_size rows "
                "(detection.py:534-537) and read past the end of a shorter last block")
        dev = self.d.device
        xd = torch.from_numpy(x).to(dev)
        scratch = torch.empty_like(xd)
        rel_d = torch.empty_like(xd)
        check(self.d.lib.ofp_stream_calibrate(self.d.handle, self._state.data_ptr(), xd.data_ptr(), n, r0, r1, sr,
                                              scratch.data_ptr(), rel_d.data_ptr(), _stream_ptr(dev)),
              "ofp_stream_calibrate")
        rel = rel_d.cpu().numpy()
        self.mins = np.median(rel[:sr], axis=0)                              # :869-872
        self.maxs = np.max(rel, axis=0)
        self.on_threshold = self.maxs * self.on_threshold + self.mins
        self.off_threshold = self.maxs * self.off_threshold + self.mins
        self.noise_max = np.median(maximum_filter1d(rel[::], int(sr * 0.01), axis=0), axis=0)
        noise_thresh = (self.noise_max - self.mins) / self.maxs
        print("Approx. relative noise thresholds at "
              f"{[np.round(x, 3) for x in noise_thresh]}!")
        on = np.ascontiguousarray(np.broadcast_to(np.asarray(self.on_threshold, np.float64), (self.n_signals,)))
        off = np.ascontiguousarray(np.broadcast_to(np.asarray(self.off_threshold, np.float64), (self.n_signals,)))
        dp = ctypes.POINTER(ctypes.c_double)
        check(self.d.lib.ofp_detector_set_thresholds(self.d.handle, on.ctypes.data_as(dp), off.ctypes.data_as(dp)),
              "ofp_detector_set_thresholds")


_FP = ctypes.POINTER(ctypes.c_float)


def _f32_block(x, what):
    """The reference's ctypes ``ndpointer(float32, ndim=2, C_CONTIGUOUS)`` check
    (detection.py:520-531, 563-578): wrong dtype / layout raises ctypes.ArgumentError."""
    if not isinstance(x, np.ndarray) or x.dtype != np.float32 or x.ndim != 2 or not x.flags["C_CONTIGUOUS"]:
        raise ctypes.ArgumentError(f"{what}: array must be a C-contiguous 2-D float32 ndarray")
    return x


class ButterworthFilter:
    """Butterworth filter applied to multiple signals in parallel (detection.py:487-501):
    scipy design, float32 coefficients and state, ``lfilter`` semantics; the filtering runs
    on the GPU (``ofp_lfilter``, direct form II transposed in fp32, bit-identical to scipy)."""

    def __init__(self, cutoff, n, order=2, sr=44100, btype="high"):
        if order > 8:
            raise ValueError("order > 8 is not supported by the GPU filter")
        self.b, self.a = sig.butter(order, cutoff, btype=btype, analog=False, output="ba", fs=sr)
        self.b, self.a = np.float32(self.b), np.float32(self.a)
        self.zi = np.zeros((order, n), dtype=np.float32)
        self._lib = _lib.lib()
        _lib.require_gpu(0)

    def __call__(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        check(self._lib.ofp_lfilter(x.ctypes.data_as(_FP), y.ctypes.data_as(_FP), self.b.ctypes.data_as(_FP),
                                    self.a.ctypes.data_as(_FP), len(self.b) - 1, self.zi.ctypes.data_as(_FP),
                                    x.shape[0], x.shape[1]), "ofp_lfilter")
        return y


class AREnvelopeFollower:
    """Attack-release envelope follower for several signals at once (detection.py:504-538),
    through the drop-in ``ar_envelope`` symbol of libonsetfp.so."""

    def __init__(self, x0: np.ndarray, attack=3, release=383):
        self.attack = np.float32(1 / attack)
        self.release = np.float32(1 / release)
        self.y = np.ascontiguousarray(x0, dtype=np.float32).copy()
        self.c_ar_env = _lib.lib()
        _lib.require_gpu(0)
        self.n, self.size = np.int32(x0.shape)

    def __call__(self, x):
        _f32_block(x, "ar_envelope")
        self.c_ar_env.ar_envelope(x.ctypes.data_as(_FP), self.y.ctypes.data_as(_FP), self.attack, self.release,
                                  int(self.size), int(self.n))
        return self.y  # the same state array object every call (detection.py:538)


class MinMaxEnvelopeFollower:
    """EMA min/max tracker for multi-channel signals (detection.py:541-592), through the
    drop-in ``minmax_envelope`` symbol of libonsetfp.so."""

    def __init__(self, x0: np.ndarray, alpha_min=1e-5, alpha_max=1e-5, minmin=0.0):
        self.alpha_min = np.float32(alpha_min)
        self.alpha_max = np.float32(alpha_max)
        self.minmin = np.float32(minmin)
        self.min_val = np.float32(np.min(x0, axis=0))
        self.max_val = np.float32(np.max(x0, axis=0))
        self.c_tracker = _lib.lib()
        _lib.require_gpu(0)
        self.n_samples, self.n_channels = np.int32(x0.shape)

    def __call__(self, x):
        _f32_block(x, "minmax_envelope")
        self.c_tracker.minmax_envelope(x.ctypes.data_as(_FP), self.min_val.ctypes.data_as(_FP),
                                       self.max_val.ctypes.data_as(_FP), self.alpha_min, self.alpha_max,
                                       self.minmin, len(x), int(self.n_channels))
        return self.min_val, self.max_val
