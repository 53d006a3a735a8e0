"""Framing / STFT / fingerprint extraction on MI355X -- drop-in surface of the
hot-path functions of the reference's ``onset_fingerprinting/data.py``
(FrameExtractor :55-120, window_contribution_weights :562-578, stft_frame
:581-590, stft :593-654, cspec_to_mfcc :657-680) plus the dense, HBM-resident
batched forms the pipeline uses.  All arithmetic on sample data runs in the HIP
kernels of ``csrc/ofp_spectral.hip``; there is no CPU path.

Datasets (POSD/MCPOSD), augmentation and json/wav parsing are out of scope
(SURVEY.md section 2).
"""
import ctypes

import numpy as np
import torch
from scipy.signal import get_window

from . import _lib
from ._lib import check

SUPPORTED_NFFT = (256, 512, 1024, 2048, 4096)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _device(device=0):
    if isinstance(device, str):
        device = torch.device(device)
    if isinstance(device, torch.device):
        if device.type != "cuda":
            raise _lib.OnsetFPError(f"this library computes on an MI355X only (device {device} requested)")
        device = device.index or 0
    _lib.require_gpu(int(device))
    return torch.device("cuda", int(device))


def _pad_center(data, size):
    """librosa.util.pad_center along the last axis (zero padding, extra sample right)."""
    n = data.shape[-1]
    lpad = int((size - n) // 2)
    pads = [(0, 0)] * (data.ndim - 1) + [(lpad, int(size - n - lpad))]
    return np.pad(data, pads)


def window_contribution_weights(window: np.ndarray, hop_length: int, hop_edge_padding: bool = False):
    """data.py:562-578.  A handful of scalars derived from the window only (no
    sample data), computed on the host."""
    w = []
    start_idx = len(window) // 2 if not hop_edge_padding else hop_length
    for i in range(start_idx, len(window) + hop_length, hop_length):
        seg = np.asarray(window[:i], dtype=np.float64)
        w.append(float(np.sum((seg[1:] + seg[:-1]) / 2.0)) if len(seg) > 1 else 0.0)
    w += w[-2::-1]
    return np.array(w) / max(w)


# ---- device-resident building blocks ------------------------------------------

def stft_frames_device(x, clip, channel, start, valid_lo, valid_hi, frame_length, n_fft, window):
    """x: float32 CUDA [n_clips, N, C]; per-frame int arrays (host or device).
    Returns complex64 CUDA [n_frames, n_fft/2+1]."""
    if n_fft not in SUPPORTED_NFFT:
        raise ValueError(f"n_fft must be one of {SUPPORTED_NFFT}, got {n_fft}")
    L = _lib.lib()
    dev = x.device
    n_clips, N, C = x.shape
    as_dev = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(dev) if not torch.is_tensor(a) else a.to(dev, dt)
    clip, channel = as_dev(clip, torch.int32), as_dev(channel, torch.int32)
    start, lo, hi = as_dev(start, torch.int64), as_dev(valid_lo, torch.int64), as_dev(valid_hi, torch.int64)
    n = int(start.numel())
    win = torch.as_tensor(np.asarray(window, dtype=np.float32)).to(dev)
    assert win.numel() == n_fft
    spec = torch.empty((n, n_fft // 2 + 1), dtype=torch.complex64, device=dev)
    check(L.ofp_stft_frames(x.data_ptr(), n_clips, N, C, clip.data_ptr(), channel.data_ptr(), start.data_ptr(),
                            lo.data_ptr(), hi.data_ptr(), n, frame_length, n_fft, win.data_ptr(),
                            spec.data_ptr(), _stream(dev)), "ofp_stft_frames")
    return spec


def stft_power_dense(x, n_fft, hop, out=None):
    """Dense power spectra: x float32 CUDA [n_clips, N, C] ->
    float32 CUDA [n_clips, C, H, n_fft/2+1], H = 1 + (N - n_fft)//hop; frame h of
    channel c is |rfft(hann_periodic(n_fft) * x[h*hop : h*hop+n_fft, c])|^2."""
    if n_fft not in SUPPORTED_NFFT:
        raise ValueError(f"n_fft must be one of {SUPPORTED_NFFT}, got {n_fft}")
    L = _lib.lib()
    n_clips, N, C = x.shape
    H = 0 if N < n_fft else 1 + (N - n_fft) // hop
    if out is None:
        out = torch.empty((n_clips, C, H, n_fft // 2 + 1), dtype=torch.float32, device=x.device)
    if H:
        check(L.ofp_stft_power(x.data_ptr(), n_clips, N, C, n_fft, hop, out.data_ptr(), _stream(x.device)),
              "ofp_stft_power")
    return out


def stft_power_mel_dense(x, n_fft, hop, melbank, out_power=None, out_mel=None, want_power=True, planar=None):
    """`stft_power_dense` and `melbank(power)` in one kernel: the filterbank is applied while each
    frame's power spectrum is still in LDS.  Returns (power or None, mel [n_clips, C, H, n_mels])."""
    if n_fft not in SUPPORTED_NFFT:
        raise ValueError(f"n_fft must be one of {SUPPORTED_NFFT}, got {n_fft}")
    L = _lib.lib()
    n_clips, N, C = x.shape
    H = 0 if N < n_fft else 1 + (N - n_fft) // hop
    if want_power and out_power is None:
        out_power = torch.empty((n_clips, C, H, n_fft // 2 + 1), dtype=torch.float32, device=x.device)
    if out_mel is None:
        out_mel = torch.empty((n_clips, C, H, melbank.n_mels), dtype=torch.float32, device=x.device)
    if H:
        # planar: (device address, stride) of the same audio as one series per clip and channel
        # (BatchDetector.planar_input)
        check(L.ofp_stft_power_mel(planar[0] if planar else x.data_ptr(), n_clips, N, C, n_fft, hop,
                                   out_power.data_ptr() if want_power else None, melbank.n_mels,
                                   melbank.lo.data_ptr(), melbank.len.data_ptr(), melbank.off.data_ptr(),
                                   melbank.w.data_ptr(), melbank.w.numel(), out_mel.data_ptr(),
                                   planar[1] if planar else 0, _stream(x.device)),
              "ofp_stft_power_mel")
    return (out_power if want_power else None), out_mel


def stft_power_mel_mlp_dense(x, n_fft, hop, melbank, mlp, out_power=None, out_mel=None, out_logits=None,
                             want_power=False, want_mel=True, planar=None):
    """`stft_power_mel_dense` with the classifier in the same kernel (``ofp_stft_power_mel_mlp``): the
    band sums of 16 frames at a time go through the whole FCNN while still in LDS.  `mlp` is a
    `calibration.DeviceMLP` (``FCNN.device_mlp()``).  Returns (power or None, mel or None,
    logits [n_clips, C, H, n_out])."""
    if n_fft not in SUPPORTED_NFFT:
        raise ValueError(f"n_fft must be one of {SUPPORTED_NFFT}, got {n_fft}")
    L = _lib.lib()
    n_clips, N, C = x.shape
    H = 0 if N < n_fft else 1 + (N - n_fft) // hop
    if want_power and out_power is None:
        out_power = torch.empty((n_clips, C, H, n_fft // 2 + 1), dtype=torch.float32, device=x.device)
    if want_mel and out_mel is None:
        out_mel = torch.empty((n_clips, C, H, melbank.n_mels), dtype=torch.float32, device=x.device)
    if out_logits is None:
        out_logits = torch.empty((n_clips, C, H, mlp.n_out), dtype=torch.float32, device=x.device)
    if H:
        check(L.ofp_stft_power_mel_mlp(planar[0] if planar else x.data_ptr(), n_clips, N, C, n_fft, hop,
                                       out_power.data_ptr() if want_power else None, melbank.n_mels,
                                       melbank.lo.data_ptr(), melbank.len.data_ptr(), melbank.off.data_ptr(),
                                       melbank.w.data_ptr(), melbank.w.numel(),
                                       out_mel.data_ptr() if want_mel else None, planar[1] if planar else 0,
                                       mlp.handle, out_logits.data_ptr(), _stream(x.device)),
              "ofp_stft_power_mel_mlp")
    return (out_power if want_power else None), (out_mel if want_mel else None), out_logits


def batch_cc(a: torch.Tensor, b: torch.Tensor):
    """data.py:226-230: full cross-correlation of row i of `a` with row i of `b`,
    [n, length] x [n, length] -> [n, 2*length - 1] (what the grouped F.conv1d there computes)."""
    dev = a.device if a.is_cuda else _device(0)
    to = lambda t: t.detach().to(dev, torch.float32).contiguous()
    ad, bd = to(a), to(b)
    n, length = ad.shape
    out = torch.empty((n, 2 * length - 1), dtype=torch.float32, device=dev)
    check(_lib.lib().ofp_xcorr_full(ad.data_ptr(), bd.data_ptr(), n, length, length, length, 1, out.data_ptr(),
                                    _stream(dev)), "ofp_xcorr_full")
    return out if a.is_cuda else out.cpu()


# ---- mel filterbank (librosa's published definition; PARITY UNPINNED) --

def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    logstep = np.log(6.4) / 27.0
    lin = f / f_sp
    log = min_log_hz / f_sp + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=40, fmin=0.0, fmax=None):
    """Slaney-scale, Slaney-normalised triangular filterbank, float32 [n_mels, n_fft/2+1]
    (what librosa.feature.melspectrogram builds for data.py:674-676)."""
    fmax = float(sr) / 2 if fmax is None else fmax
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    fb = np.zeros((n_mels, len(freqs)), dtype=np.float32)
    for b in range(n_mels):
        up = (freqs - edges[b]) / width[b]
        down = (edges[b + 2] - freqs) / width[b + 1]
        fb[b] = np.clip(np.minimum(up, down), 0, None)
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None].astype(np.float32)
    return fb


class MelBank:
    """Sparse (band-wise CSR) device copy of a mel filterbank for ofp_mel."""

    def __init__(self, sr, n_fft, n_mels=40, fmin=0.0, fmax=None, device=0):
        self.dense = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        self.n_mels, self.n_bins = self.dense.shape
        lo, ln, off, w = [], [], [], []
        for b in range(self.n_mels):
            nz = np.nonzero(self.dense[b])[0]
            if len(nz) == 0:
                lo.append(0), ln.append(0), off.append(len(w))
                continue
            lo.append(int(nz[0])), ln.append(int(nz[-1] - nz[0] + 1)), off.append(len(w))
            w.extend(self.dense[b, nz[0]:nz[-1] + 1].tolist())
        dev = _device(device)
        self.lo = torch.tensor(lo, dtype=torch.int32, device=dev)
        self.len = torch.tensor(ln, dtype=torch.int32, device=dev)
        self.off = torch.tensor(off, dtype=torch.int32, device=dev)
        self.w = torch.tensor(w if w else [0.0], dtype=torch.float32, device=dev)

    def __call__(self, power, out=None):
        """power float32 CUDA [..., n_bins] -> mel [..., n_mels]."""
        L = _lib.lib()
        rows = power.numel() // self.n_bins
        if out is None:
            out = torch.empty(power.shape[:-1] + (self.n_mels,), dtype=torch.float32, device=power.device)
        check(L.ofp_mel(power.data_ptr(), rows, self.n_bins, self.n_mels, self.lo.data_ptr(), self.len.data_ptr(),
                        self.off.data_ptr(), self.w.data_ptr(), out.data_ptr(), _stream(power.device)), "ofp_mel")
        return out


def dct_ortho(n_mfcc, n_mels):
    """DCT-II, norm="ortho", as a float32 [n_mfcc, n_mels] matrix (scipy.fft.dct definition)."""
    n = np.arange(n_mels)
    k = np.arange(n_mfcc)[:, None]
    M = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_mels)) * np.sqrt(2.0 / n_mels)
    M[0] *= np.sqrt(0.5)
    return M.astype(np.float32)


def mfcc_device(mel, n_mfcc=14, amin=1e-10, top_db=80.0):
    """power_to_db (ref 1) + DCT-II ortho over the last axis: mel CUDA [..., n_mels] -> [..., n_mfcc]."""
    L = _lib.lib()
    n_mels = mel.shape[-1]
    rows = mel.numel() // n_mels
    dct = torch.from_numpy(dct_ortho(n_mfcc, n_mels)).to(mel.device)
    out = torch.empty(mel.shape[:-1] + (n_mfcc,), dtype=torch.float32, device=mel.device)
    scratch = torch.zeros(4, dtype=torch.float32, device=mel.device)
    check(L.ofp_mfcc(mel.data_ptr(), rows, n_mels, n_mfcc, amin, -1.0 if top_db is None else top_db,
                     dct.data_ptr(), out.data_ptr(), scratch.data_ptr(), _stream(mel.device)), "ofp_mfcc")
    return out


# ---- the reference's call surface ----------------------------------------------

def stft_frame(x: np.ndarray, n_fft: int, window: np.ndarray, device=0):
    """Single STFT frame (data.py:581-590): centre-pads x to n_fft, returns
    rfft(window * x) -- complex64 here (the reference stores into complex64, data.py:645)."""
    dev = _device(device)
    x = np.ascontiguousarray(x, dtype=np.float32)
    lead = x.shape[:-1]
    flat = x.reshape(-1, x.shape[-1])
    n, Lx = flat.shape
    xt = torch.from_numpy(np.ascontiguousarray(flat.T))[None].to(dev)  # [1, Lx, n] interleaved
    z = np.zeros(n, np.int64)
    spec = stft_frames_device(xt.contiguous(), np.zeros(n, np.int32), np.arange(n, dtype=np.int32), z, z,
                              z + Lx, Lx, n_fft, np.asarray(window))
    return spec.cpu().numpy().reshape(lead + (n_fft // 2 + 1,))


def stft(audio: np.ndarray, onset: int, frame_length: int = 256, hop_length: int = 64, n_fft: int = 512,
         hop_edge_padding: bool = False, method="zerozero", device=0):
    """STFT around one onset (data.py:593-654).  audio is [(C,) N] float32; returns
    complex64 [(C,) n_fft/2+1, n_frames]."""
    dev = _device(device)
    a = np.ascontiguousarray(audio, dtype=np.float32)
    two_d = a.ndim == 2
    a2 = a if two_d else a[None]
    Cn, N = a2.shape
    P = frame_length - hop_length if hop_edge_padding else frame_length // 2
    seg = min(frame_length, max(0, N - onset))  # len(audio[onset:onset+L])
    if method == "zerozero":
        lo, total = onset, P + seg + P
    elif method == "prezero":
        lo, total = onset - P, P + seg + P
    elif method == "pre":
        lo, total = onset - P, P + seg
    else:
        raise ValueError(f"unknown method {method!r}")
    if method != "zerozero" and onset - P < 0:
        raise ValueError("onset must be at least the padding length for method 'prezero'/'pre'")
    n_frames = 1 + (total - frame_length) // hop_length
    window = get_window("hann", frame_length, fftbins=True)
    if n_fft > frame_length:
        window = _pad_center(window, n_fft)
    xt = torch.from_numpy(np.ascontiguousarray(a2.T))[None].to(dev).contiguous()  # [1, N, C]
    f = np.arange(n_frames, dtype=np.int64)
    starts = np.tile(onset - P + hop_length * f, Cn)
    chans = np.repeat(np.arange(Cn, dtype=np.int32), n_frames)
    n = len(starts)
    spec = stft_frames_device(xt, np.zeros(n, np.int32), chans, starts, np.full(n, lo, np.int64),
                              np.full(n, onset + seg, np.int64), frame_length, n_fft, window)
    S = spec.cpu().numpy().reshape(Cn, n_frames, n_fft // 2 + 1).transpose(0, 2, 1)
    return np.ascontiguousarray(S).squeeze()  # data.py:645-647 squeezes every unit axis


def cspec_to_mfcc(S: np.ndarray, sr: int, fmin: int = 0, fmax=None, n_mels: int = 40, n_mfcc: int = 14,
                  device=0):
    """MFCCs from a complex spectrogram (data.py:657-680): S [(C,) bins, T] ->
    [(C,) n_mfcc, T].  Restates librosa's melspectrogram / power_to_db / mfcc
    (PARITY UNPINNED: librosa is not available offline)."""
    dev = _device(device)
    P = (np.abs(np.asarray(S)) ** 2).astype(np.float32)  # data.py:675
    bins = P.shape[-2]
    rows = np.ascontiguousarray(np.moveaxis(P, -2, -1))  # [(C,) T, bins]
    bank = MelBank(sr, 2 * (bins - 1), n_mels, fmin, fmax, device)
    mel = bank(torch.from_numpy(rows).to(dev))
    out = mfcc_device(mel, n_mfcc)
    return np.ascontiguousarray(np.moveaxis(out.cpu().numpy(), -1, -2))


class FrameExtractor:
    """data.py:55-120: gather one fixed-length window per onset (GPU gather)."""

    def __init__(self, frame_length: int, pre_samples: int, max_shift: int = 0, add_pre_samples: bool = False,
                 use_min_onset: bool = True, device=0):
        self.frame_length = frame_length
        self.pre_samples = pre_samples
        if add_pre_samples:
            self.frame_length += self.pre_samples
        self.max_shift = max_shift
        self.use_min_onset = use_min_onset
        self.device = device

    def starts(self, audio_ndim, onsets):
        onsets = np.asarray(onsets, dtype=np.int64)
        offset = self.pre_samples
        if self.max_shift:
            shifts = np.random.randint(-self.max_shift, self.max_shift + 1, len(onsets))
            offset = offset - shifts
            if audio_ndim == 2 and not self.use_min_onset:
                offset = offset[:, None]
        if audio_ndim == 2:
            if self.use_min_onset:
                s = onsets.min(axis=1) - offset
                return np.repeat(s[:, None], onsets.shape[1], axis=1)
            return onsets - offset
        return (onsets - offset)[:, None]

    def extract_device(self, x, starts):
        """x float32 CUDA [N, C]; starts int64 [O, C] -> CUDA [O, C, W]."""
        L = _lib.lib()
        N, C = x.shape
        st = torch.as_tensor(np.ascontiguousarray(starts, dtype=np.int64)).to(x.device)
        O = st.shape[0]
        out = torch.empty((O, C, self.frame_length), dtype=torch.float32, device=x.device)
        check(L.ofp_extract_frames(x.data_ptr(), N, C, st.data_ptr(), O, self.frame_length, out.data_ptr(),
                                   _stream(x.device)), "ofp_extract_frames")
        return out

    def __call__(self, audio: np.ndarray, onsets: np.ndarray) -> np.ndarray:
        dev = _device(self.device)
        a = np.ascontiguousarray(audio, dtype=np.float32)
        st = self.starts(a.ndim, onsets)
        N = a.shape[0]
        if st.min(initial=0) < 0 or st.max(initial=0) + self.frame_length > N:
            # numpy's strided view would wrap or raise (data.py:106-120); refuse loudly
            raise IndexError("frame outside the audio array")
        x = torch.from_numpy(a if a.ndim == 2 else a[:, None]).to(dev).contiguous()
        out = self.extract_device(x, st).cpu().numpy()
        return out if a.ndim == 2 else out[:, 0, :]


class StretchFrameExtractor(FrameExtractor):
    """data.py:195-223: windows of frame_length + shift samples (random shift per onset, drawn with the
    reference's own np.random calls, so a seeded run draws the same shifts) resampled to frame_length with
    scipy.signal.resample's Fourier method -- on the GPU (``ofp_resample_windows``)."""

    def __init__(self, frame_length: int, pre_samples: int, max_stretch: float = 0.03, use_min_onset=True, device=0):
        super().__init__(frame_length, pre_samples, device=device)
        if not use_min_onset:
            raise NotImplementedError("use_min_onset=False not supported yet!")
        self.max_shift = int(self.frame_length * max_stretch)

    def __call__(self, audio, onsets):
        onsets = np.asarray(onsets)
        shifts = np.random.randint(1, self.max_shift, len(onsets))       # data.py:208-209
        shifts *= np.random.choice((-1, 1), size=len(shifts))
        a = np.ascontiguousarray(audio, dtype=np.float32)
        two_d = a.ndim == 2
        starts = (onsets.min(axis=1) if two_d else onsets).astype(np.int64) - self.pre_samples
        nx = (self.frame_length + shifts).astype(np.int32)
        if len(starts) and (starts.min() < 0 or (starts + nx).max() > len(a)):
            raise ValueError("StretchFrameExtractor: a window leaves the audio (the reference's slice would be "
                             "shorter than frame_length + shift there)")
        dev = _device(self.device)
        a2 = a if two_d else a[:, None]
        C = a2.shape[1]
        out = torch.empty((len(starts), C, self.frame_length), dtype=torch.float32, device=dev)
        if len(starts):
            xd = torch.from_numpy(a2).to(dev).contiguous()
            st_d, nx_d = torch.from_numpy(starts).to(dev), torch.from_numpy(nx).to(dev)  # (kept alive across the launch)
            check(_lib.lib().ofp_resample_windows(xd.data_ptr(), a2.shape[0], C, st_d.data_ptr(), nx_d.data_ptr(),
                                                  len(starts), int(nx.max()), self.frame_length, out.data_ptr(),
                                                  _stream(dev)), "ofp_resample_windows")
        res = out.cpu().numpy()
        return res if two_d else res[:, 0, :]


class FastFrameExtractor:
    """data.py:123-192: holds the audio (in HBM) and the onsets, always starts a frame at the minimum
    onset of its group, and returns a torch tensor ``[O, C, W]`` (``[O, W]`` for 1-D audio) from
    ``__call__()``.  ``device=None`` returns CPU tensors and draws the random shifts from torch's CPU
    generator, exactly as the reference does; any other device keeps the frames on that GPU.
    Frames that would start before the audio or end after it raise ``IndexError`` (the reference's
    ``audio.unfold(...)[index]`` wraps negative indices around silently)."""

    def __init__(self, audio: np.ndarray, onsets: np.ndarray, frame_length: int, pre_samples: int,
                 max_shift: int = 0, add_pre_samples: bool = False, device=None):
        self.device = device
        self.frame_length = frame_length
        self.pre_samples = pre_samples
        if add_pre_samples:
            self.frame_length += self.pre_samples
        self.max_shift = max_shift
        self._gpu = _device(0 if device is None else device)
        onsets = np.asarray(onsets, dtype=np.int64)
        self._onsets = onsets.min(1) if onsets.ndim == 2 else onsets
        a = np.ascontiguousarray(audio, dtype=np.float32)
        self._one_d = a.ndim == 1
        self._x = torch.from_numpy(a[:, None] if self._one_d else a).to(self._gpu).contiguous()
        self._gather = FrameExtractor(self.frame_length, 0, device=self._gpu)
        if self.max_shift > 0:
            self.onsets = torch.as_tensor(self._onsets, device=device)
        else:
            self.frames = self._frames(self._onsets - self.pre_samples)

    def _frames(self, starts):
        starts = np.asarray(starts, dtype=np.int64)
        N, C = self._x.shape
        if starts.min(initial=0) < 0 or starts.max(initial=0) + self.frame_length > N:
            raise IndexError("frame outside the audio array")
        out = self._gather.extract_device(self._x, np.repeat(starts[:, None], C, axis=1))
        out = out[:, 0, :] if self._one_d else out
        return out.cpu() if self.device is None else out

    def __call__(self):
        if self.max_shift:
            shifts = torch.randint(-self.max_shift, self.max_shift + 1, (len(self._onsets),), device=self.device)
            return self._frames(self._onsets - (self.pre_samples - shifts.cpu().numpy()))
        return self.frames
