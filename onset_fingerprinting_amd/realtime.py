"""Per-hop streaming on MI355X (BASELINE config 5): the data path of the reference's
realtime callback with everything between "a hop of samples arrives" and "onsets +
classifier outputs are on the host" in ONE captured hipGraph on device-resident state.

Reference call pattern:
  realtime/audio.py:96-97     copy the hop, write it into the 60 s ring buffer
  realtime/audio.py:62-74     ``self.od(audio)`` -> AmplitudeOnsetDetector.__call__ on the hop,
                              onset = current_index + delta
  multilateration.py:555-557  ``FCNN.call_np`` -> calibration.py:552-560
  realtime/recording.py:273-280  one rFFT frame of ``audio[-n_fft:]`` per hop

The sound-card I/O, VST effects, geometry solver and shared-memory plumbing of
``realtime/`` are out of scope; this module is the hot path
they call.  No CPU path: without libonsetfp.so or a gfx950 GPU every call raises.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import HopConfig, check
from .data import mel_filterbank
from .detection import ONSET_DTYPE, _DeviceDetector

# the detector arguments of the reference's realtime setup (realtime/audio.py:39-52)
REALTIME_DETECTOR_KWARGS = dict(hipass_freq=0, fast_ar=(0.3, 800), slow_ar=(8000, 8000), on_threshold=0.45,
                                off_threshold=0.45, cooldown=1323, backtrack=False)


def _band_csr(dense):
    lo, ln, off, w = [], [], [], []
    for b in range(dense.shape[0]):
        nz = np.nonzero(dense[b])[0]
        if len(nz) == 0:
            lo.append(0), ln.append(0), off.append(len(w))
            continue
        lo.append(int(nz[0])), ln.append(int(nz[-1] - nz[0] + 1)), off.append(len(w))
        w.extend(dense[b, nz[0]:nz[-1] + 1].tolist())
    return (np.asarray(lo, np.int32), np.asarray(ln, np.int32), np.asarray(off, np.int32),
            np.asarray(w if w else [0.0], np.float32))


class HopSession:
    """One stream: ring buffer + detector + per-hop spectral fingerprint + classifier.

    ``session(hop)`` takes ``[block_size, n_signals]`` float32 and returns a dict with
    ``channels`` / ``onsets`` (absolute sample indices, as ``detect_hits`` forms them,
    audio.py:65), ``logits`` ``[n_signals, n_out]`` (None without a classifier), ``mel``
    ``[n_signals, n_mels]`` and, with ``want_rel``, the hop's relative envelope.
    """

    def __init__(self, n_signals, block_size=128, sr=96000, n_fft=2048, n_mels=40, classifier=None,
                 ring_seconds=60.0, want_rel=False, device=0, floor=-70.0, hipass_freq=2000.0,
                 fast_ar=(3.0, 383.0), slow_ar=(2205.0, 2205.0), on_threshold=0.5, off_threshold=0.1,
                 cooldown=1323, backtrack=False, backtrack_buffer_size=None, backtrack_smooth_size=5,
                 onset_strength=None):
        L = _lib.lib()
        self.n_signals, self.block_size, self.sr, self.n_fft, self.n_mels = n_signals, block_size, sr, n_fft, n_mels
        if backtrack_buffer_size is None:
            backtrack_buffer_size = 2 * block_size  # audio.py:50
        self.d = _DeviceDetector(n_signals, block_size, floor, hipass_freq, fast_ar, slow_ar, on_threshold,
                                 off_threshold, cooldown, backtrack, backtrack_buffer_size, backtrack_smooth_size,
                                 sr, device)
        self.device = self.d.device
        lo, ln, off, w = _band_csr(mel_filterbank(sr, n_fft, n_mels))
        self._keep = (lo, ln, off, w)
        self.classifier = classifier
        mlp = classifier.device_mlp(self.device) if classifier is not None else None
        self.n_out = mlp.n_out if mlp is not None else 0
        cfg = HopConfig()
        cfg.n_fft = n_fft
        cfg.ring_samples = max(int(round(ring_seconds * sr)), n_fft, block_size)  # realtime/config.py:45,59
        cfg.n_mels = n_mels
        cfg.fb_lo, cfg.fb_len, cfg.fb_off, cfg.fb_w = (a.ctypes.data for a in (lo, ln, off, w))
        cfg.fb_nnz = len(w)
        cfg.mlp = mlp.handle if mlp is not None else None
        cfg.want_rel = int(bool(want_rel))
        self.want_rel = bool(want_rel)
        # per-hop onset strength of the channel mean (realtime/recording.py:273-311; PARITY UNPINNED: its two
        # trackers are loopmate.EMA_MinMaxTracker objects, loopmate is absent; see include/onsetfp.h).
        # onset_strength: None, or a dict with max_length / avg_length (the reference's undefined
        # config.MAX_LENGTH / AVG_LENGTH) and optionally ring, tg_win_length, ls_* / oe_* tracker constants
        self.onset_strength = None
        if onset_strength is not None:
            o = dict(ring=int(np.ceil(max(int(round(ring_seconds * sr)), n_fft) / block_size)), ls_max0=10.0,
                     ls_minmax=0.0, ls_alpha=0.0005, oe_min0=0.0, oe_minmin=0.0, oe_max0=1.0, oe_alpha=0.001)
            o.update(onset_strength)
            cfg.strength = 1
            cfg.strength_ring, cfg.max_length, cfg.avg_length = int(o["ring"]), int(o["max_length"]), int(o["avg_length"])
            cfg.ls_max0, cfg.ls_minmax, cfg.ls_alpha = o["ls_max0"], o["ls_minmax"], o["ls_alpha"]
            cfg.oe_min0, cfg.oe_minmin, cfg.oe_max0, cfg.oe_alpha = o["oe_min0"], o["oe_minmin"], o["oe_max0"], o["oe_alpha"]
            # tg_win_length (config.TG_WIN_LENGTH = 1024 upstream): the tempogram frame per hop as well
            # (recording.py:313-327); needs tg_win_length <= ring
            cfg.tg_win_length = int(o.get("tg_win_length") or 0)
            self.onset_strength = o
        self.ring_samples = int(cfg.ring_samples)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(L.ofp_hop_create(self.d.handle, ctypes.byref(cfg), ctypes.byref(h)), "ofp_hop_create")
        self.handle = h
        self._L = L
        self._n = ctypes.c_int64()
        self._rec = np.zeros(max(n_signals, 1), dtype=ONSET_DTYPE)
        self._logits = np.zeros((n_signals, max(self.n_out, 1)), dtype=np.float32)
        self._mel = np.zeros((n_signals, n_mels), dtype=np.float32)
        self._rel = np.zeros((block_size, n_signals), dtype=np.float32)
        self._sg = np.zeros(4 + (int(cfg.tg_win_length) if onset_strength is not None else 0), dtype=np.float32)
        self.current_index = 0  # audio.py:120

    def close(self):
        if getattr(self, "handle", None):
            self._L.ofp_hop_destroy(self.handle)
            self.handle = None
        self.d.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self._L.ofp_hop_reset(self.handle), "ofp_hop_reset")
        self.current_index = 0

    def init_minmax_tracker(self, x):
        """AmplitudeOnsetDetector.init_minmax_tracker (detection.py:827-840)."""
        x = self._f32(x, None)
        if len(x):
            check(self._L.ofp_hop_warmup(self.handle, x.ctypes.data, len(x)), "ofp_hop_warmup")

    def _f32(self, x, rows):
        x = np.ascontiguousarray(x)
        if x.dtype != np.float32:
            # the reference's ctypes ndpointer rejects non-float32 input (detection.py:521-526)
            raise ctypes.ArgumentError(f"array must have data type float32, got {x.dtype}")
        if x.ndim != 2 or x.shape[1] != self.n_signals or (rows is not None and x.shape[0] != rows):
            raise ValueError(f"expected shape ({rows if rows is not None else 'n'}, {self.n_signals}), got {x.shape}")
        return x

    def submit(self, hop):
        hop = self._f32(hop, self.block_size)
        check(self._L.ofp_hop_submit(self.handle, hop.ctypes.data), "ofp_hop_submit")

    def collect(self):
        check(self._L.ofp_hop_collect(self.handle, ctypes.byref(self._n), self._rec.ctypes.data,
                                      self._logits.ctypes.data if self.n_out else None, self._mel.ctypes.data,
                                      self._rel.ctypes.data if self.want_rel else None,
                                      self._sg.ctypes.data if self.onset_strength else None), "ofp_hop_collect")
        k = min(int(self._n.value), self.n_signals)
        self.current_index += self.block_size
        return dict(channels=self._rec["channel"][:k].astype(np.int64), onsets=self._rec["sample"][:k].copy(),
                    logits=self._logits.copy() if self.n_out else None, mel=self._mel.copy(),
                    rel=self._rel.copy() if self.want_rel else None,
                    # {flux, normalised, moving max, moving mean} of recording.py:296-311
                    strength=self._sg[:4].copy() if self.onset_strength else None,
                    # recording.py:313-327 (None unless onset_strength has tg_win_length)
                    tempogram=self._sg[4:].copy() if self.onset_strength and len(self._sg) > 4 else None)

    def __call__(self, hop):
        self.submit(hop)
        return self.collect()

    def push_raw(self, hop):
        """`__call__` without building the result dict (latency measurements): returns the number
        of onsets; the outputs stay in the session's host arrays."""
        check(self._L.ofp_hop_push(self.handle, hop.ctypes.data, ctypes.byref(self._n), self._rec.ctypes.data,
                                   self._logits.ctypes.data if self.n_out else None, self._mel.ctypes.data,
                                   self._rel.ctypes.data if self.want_rel else None,
                                   self._sg.ctypes.data if self.onset_strength else None), "ofp_hop_push")
        self.current_index += self.block_size
        return int(self._n.value)

    def audio(self, n):
        """``rec_audio[-n:]`` (the last n rows of the device ring buffer, oldest first)."""
        out = np.empty((int(n), self.n_signals), dtype=np.float32)
        check(self._L.ofp_hop_ring_read(self.handle, int(n), out.ctypes.data), "ofp_hop_ring_read")
        return out
