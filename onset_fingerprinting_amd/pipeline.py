"""The whole hot path, HBM-resident: detect -> rFFT power -> mel fingerprint ->
classify, for a batch of clips [n_clips, N, C] on one GPU.

This is the composition BASELINE.json's metric names ("detect+FFT+classify").
The reference never composes these steps in one function (its callers are
notebooks and the realtime callback); each step here is
the drop-in of the corresponding reference function:

  detect    detection.detect_onsets_amplitude   (detection.py:19-86)
  rFFT      data.stft_frame on every hop         (data.py:581-590; "dense
            equivalent" of SURVEY.md 8a row a9)
  mel       the mel step of data.cspec_to_mfcc   (data.py:674-676)
  classify  calibration.FCNN.forward             (calibration.py:520-527)
"""
import numpy as np
import torch

from . import _lib
from .calibration import FCNN
from .data import MelBank, stft_power_mel_dense, stft_power_mel_mlp_dense
from .detection import BatchDetector


def seeded_fcnn(n_in=40, n_out=8, seed=1234):
    """FCNN(n_in -> [10,10,10] -> n_out) with deterministic synthetic weights and
    non-trivial BatchNorm statistics (there is no trained checkpoint in the reference)."""
    g = torch.Generator().manual_seed(seed)
    m = FCNN(n_in, n_out)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Linear):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) / np.sqrt(mod.in_features))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(0.2 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
    return m.eval()


class FingerprintPipeline:
    def __init__(self, n_channels, n_fft=1024, hop=256, sr=48000, n_mels=40, classifier=None, device=0,
                 want_power=True, cap_per_clip=None, **detector_kwargs):
        self.device = torch.device("cuda", int(device))
        _lib.require_gpu(int(device))
        self.n_channels, self.n_fft, self.hop, self.sr, self.n_mels = n_channels, n_fft, hop, sr, n_mels
        self.detector = BatchDetector(n_channels, block_size=hop, sr=sr, device=device, **detector_kwargs)
        self.mel = MelBank(sr, n_fft, n_mels, device=device)
        self.classifier = classifier if classifier is not None else seeded_fcnn(n_mels, 8)
        # an FCNN runs inside the STFT kernel's epilogue (ofp_stft_power_mel_mlp); any other callable
        # classifier gets the mel bands afterwards
        self._mlp = self.classifier.device_mlp(self.device) if hasattr(self.classifier, "device_mlp") else None
        if self._mlp is not None and not self._mlp.fits:
            self._mlp = None
        self.want_power = bool(want_power)  # False: |X|^2 never leaves the chip (BASELINE config 3)
        self.cap_per_clip = cap_per_clip
        self._bufs = None
        self._side = None

    def n_frames(self, n_samples):
        return 0 if n_samples < self.n_fft else 1 + (n_samples - self.n_fft) // self.hop

    def _buffers(self, n_clips, N):
        key = (n_clips, N)
        if self._bufs is None or self._bufs["key"] != key:
            C, H, bins = self.n_channels, self.n_frames(N), self.n_fft // 2 + 1
            nb = N // self.hop
            dev = self.device
            cap = int(self.cap_per_clip) if self.cap_per_clip else max(1, min(nb * C, 1 << 20))
            n_out = self._mlp.n_out if self._mlp is not None else 0
            self._bufs = dict(
                key=key,
                det=dict(records=torch.empty((n_clips, cap, 16), dtype=torch.uint8, device=dev),
                         counts=torch.zeros(n_clips, dtype=torch.int64, device=dev),
                         rel=torch.empty((n_clips, nb * self.hop, C), dtype=torch.float32, device=dev)),
                power=torch.empty((n_clips, C, H, bins), dtype=torch.float32, device=dev) if self.want_power else None,
                mel=torch.empty((n_clips, C, H, self.n_mels), dtype=torch.float32, device=dev),
                logits=torch.empty((n_clips, C, H, n_out), dtype=torch.float32, device=dev) if n_out else None,
            )
            self.detector.reserve(n_clips, N, int(0.5 * self.sr))
        return self._bufs

    def run(self, x, timed=False):
        """x float32 CUDA [n_clips, N, C] -> dict(records, counts, cap, rel, power, mel, logits).

        The spectral branch (rFFT -> mel -> classifier) does not depend on the detector, and the
        detector is a latency-bound recurrence that occupies a small part of the chip, so the two
        branches run concurrently on two HIP streams and join at the end."""
        n_clips, N, C = x.shape
        b = self._buffers(n_clips, N)
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            self._head = torch.cuda.Event()
        side = self._side
        # head of the detector: transpose + the IIR candidate launch, then its verification rounds
        # (a handful of waves per launch and a host round trip each: anything else on the chip slows
        # them several times over)
        self.detector.begin_input(x)
        self._head.record(main)
        self.detector.begin_iir(x)
        if getattr(self, "skip_spectral", False):  # (measurement only: what the detector costs without the spectral branch)
            det = self.detector.detect(x, out=b["det"], cap_per_clip=b["det"]["records"].shape[1], begun=True)
            out = dict(records=det["records"], counts=det["counts"], cap=det["cap"], rel=det["rel"], power=None, mel=None,
                       logits=None, info=self.detector.last_info)
            if timed:
                out["spectral_ms"] = dict(stft_mel=0.0, mlp=0.0)
            return out
        # the spectral branch needs the planar copy only: it starts beside the candidate launch
        side.wait_event(self._head)
        with torch.cuda.stream(side):
            self._ev[0].record(side)
            # |X|^2, the mel bands and the classifier of every frame in ONE kernel: the filterbank is
            # applied while the frame's power spectrum is still in LDS, the FCNN while the band sums
            # of 16 frames are (it reads the transposed copy of x the detector's head just made:
            # coalesced loads)
            planar = self.detector.planar_input(x)
            if self._mlp is not None:
                power, mel, logits = stft_power_mel_mlp_dense(
                    x, self.n_fft, self.hop, self.mel, self._mlp, out_power=b["power"], out_mel=b["mel"],
                    out_logits=b["logits"], want_power=self.want_power, want_mel=True, planar=planar)
                self._ev[1].record(side)
                self._ev[2].record(side)
            else:
                power, mel = stft_power_mel_dense(x, self.n_fft, self.hop, self.mel, out_power=b["power"],
                                                  out_mel=b["mel"], want_power=self.want_power, planar=planar)
                self._ev[1].record(side)
                self._ev[2].record(side)
                logits = self.classifier(mel.reshape(-1, self.n_mels))
            self._ev[3].record(side)
        det = self.detector.detect(x, out=b["det"], cap_per_clip=b["det"]["records"].shape[1], begun=True)
        main.wait_stream(side)
        out = dict(records=det["records"], counts=det["counts"], cap=det["cap"], rel=det["rel"], power=power,
                   mel=mel, logits=logits.reshape(n_clips, C, -1, logits.shape[-1]),
                   info=self.detector.last_info)
        if timed:
            main.synchronize()  # this pipeline's streams only: other pipelines may be in flight
            e = self._ev
            out["spectral_ms"] = dict(stft_mel=e[0].elapsed_time(e[1]), mlp=e[2].elapsed_time(e[3]))
        return out
