"""``calibration.FCNN`` (calibration.py:463-560) with its forward pass on MI355X.

The module builds the same ``network`` Sequential as the reference so that a
reference ``state_dict`` (``network.{k}.weight`` ...) loads unchanged
(realtime/config.py:105-107); ``forward`` is inference-only and runs the WHOLE
network as one HIP kernel (``ofp_mlp_forward``: every layer a chain of fp32 MFMA
tiles + bias + folded eval-mode BatchNorm1d + activation, the activations staying
in LDS, csrc/ofp_mlp.h); a network too large for the LDS runs layer by layer
through ``ofp_dense`` -- bit-identical either way.  Training utilities, TDoA
calibration and the scipy optimisers of the reference file are out of scope.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import check

ACT_CODES = {nn.Identity: 0, nn.ReLU: 1, nn.SiLU: 2, nn.LeakyReLU: 3, nn.ELU: 4, nn.Tanh: 5}


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def dense_forward(x, weight, bias, scale, shift, act_code, out=None):
    """One fused layer on the GPU: act((x @ W^T + b) * scale + shift)."""
    L = _lib.lib()
    n, fin = x.shape
    fout = weight.shape[0]
    if out is None:
        out = torch.empty((n, fout), dtype=torch.float32, device=x.device)
    p = lambda t: t.data_ptr() if t is not None else None
    check(L.ofp_dense(x.data_ptr(), n, fin, fout, weight.data_ptr(), p(bias), p(scale), p(shift), act_code,
                      out.data_ptr(), _stream(x.device)), "ofp_dense")
    return out


class DeviceMLP:
    """Owner of an ``ofp_mlp`` handle (include/onsetfp.h): the folded layers of one network,
    resident on the current device."""

    def __init__(self, plan):
        """plan: list of (W [out, in], b, scale, shift, act_code) with CPU float32 tensors / None."""
        L = _lib.lib()
        n = len(plan)
        dims = [int(plan[0][0].shape[1])] + [int(w.shape[0]) for (w, *_r) in plan]
        self.n_in, self.n_out = dims[0], dims[-1]
        keep = []

        def arr(ts):
            out = (ctypes.c_void_p * n)()
            for i, t in enumerate(ts):
                if t is not None:
                    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
                    keep.append(a)
                    out[i] = a.ctypes.data
            return out

        h = ctypes.c_void_p()
        check(L.ofp_mlp_create(n, (ctypes.c_int32 * (n + 1))(*dims), (ctypes.c_int32 * n)(*[pl[4] for pl in plan]),
                               arr([pl[0] for pl in plan]), arr([pl[1] for pl in plan]), arr([pl[2] for pl in plan]),
                               arr([pl[3] for pl in plan]), ctypes.byref(h)), "ofp_mlp_create")
        self.handle = h
        self.lds_bytes = int(L.ofp_mlp_lds_bytes(h))
        self.fits = self.lds_bytes <= 160 * 1024

    def forward(self, x, out=None):
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, self.n_out), dtype=torch.float32, device=x.device)
        check(_lib.lib().ofp_mlp_forward(self.handle, x.data_ptr(), n, out.data_ptr(), _stream(x.device)),
              "ofp_mlp_forward")
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().ofp_mlp_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class FCNN(nn.Module):
    def __init__(self, input_size: int, output_size: int, hidden_layers=[10, 10, 10], activation=nn.ReLU,
                 dropout: float = 0.0, batch_norm: bool = True, l2_reg: float = 0.0, eye_init=False,
                 eye_noise_floor=0.01, bias=True) -> None:
        super().__init__()
        if activation not in ACT_CODES:
            raise ValueError(f"activation {activation} has no HIP implementation "
                             f"(supported: {[a.__name__ for a in ACT_CODES]})")
        self.l2_reg = l2_reg
        self._act_code = ACT_CODES[activation]
        sizes = [input_size] + list(hidden_layers)
        mods = []
        for a, b in zip(sizes[:-1], sizes[1:]):
            lin = nn.Linear(a, b, bias=bias)
            if eye_init:
                self.init_eye_weights(lin, eye_noise_floor)
            mods.append(lin)
            if batch_norm:
                mods.append(nn.BatchNorm1d(b))
            mods.append(activation())
            if dropout > 0:
                mods.append(nn.Dropout(p=dropout))
        last = nn.Linear(sizes[-1], output_size, bias=bias)
        if eye_init:
            self.init_eye_weights(last, eye_noise_floor)
        mods.append(last)
        self.network = nn.Sequential(*mods)
        self._plan = None
        self._plan_key = None
        self._mlp = None

    def init_eye_weights(self, layer, noise_floor=0.001):
        noise = torch.randn(layer.out_features, layer.in_features) * noise_floor
        layer.weight.data = torch.eye(layer.out_features, layer.in_features) + noise

    def _build_plan(self, device):
        """(W, b, scale, shift, act) per Linear, BatchNorm folded with running stats."""
        plan = []
        mods = list(self.network)
        i = 0
        while i < len(mods):
            lin = mods[i]
            assert isinstance(lin, nn.Linear)
            scale = shift = None
            act = 0
            j = i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                bn = mods[j]
                inv = (bn.running_var.double() + bn.eps).rsqrt()
                g = bn.weight.double() if bn.affine else torch.ones_like(inv)
                be = bn.bias.double() if bn.affine else torch.zeros_like(inv)
                scale = (g * inv).float()
                shift = (be - bn.running_mean.double() * g * inv).float()
                j += 1
            if j < len(mods) and type(mods[j]) in ACT_CODES:
                act = ACT_CODES[type(mods[j])]
                j += 1
            if j < len(mods) and isinstance(mods[j], nn.Dropout):
                j += 1  # identity in eval mode
            to = lambda t: None if t is None else t.detach().to(device, torch.float32).contiguous()
            plan.append((to(lin.weight), to(lin.bias), to(scale), to(shift), act))
            i = j
        return plan

    def _versions(self, dev):
        # the folded device copy is rebuilt whenever any parameter or buffer is replaced or changed
        # in place (load_state_dict on any submodule, init_eye_weights, optimiser steps ...)
        ts = list(self.parameters()) + list(self.buffers())
        return (str(dev),) + tuple((id(t), t.data_ptr(), t._version) for t in ts)

    def device_mlp(self, device=0):
        """The network as an ``ofp_mlp`` handle on `device` (rebuilt when the parameters changed);
        also what the fused STFT->mel->classifier kernel and the per-hop session take."""
        dev = device if isinstance(device, torch.device) else torch.device("cuda", int(device))
        key = self._versions(dev)
        if self._plan is None or self._plan_key != key:
            with torch.cuda.device(dev):
                self._plan = self._build_plan(dev)
                self._mlp = DeviceMLP(self._build_plan(torch.device("cpu")))
            self._plan_key = key
        return self._mlp

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [batch, input_size] -> [batch, output_size].  Inference only: BatchNorm uses its running
        statistics and Dropout is the identity, so a module in training mode that has either is
        refused (the reference would use batch statistics there, calibration.py:520-527)."""
        if self.training and any(isinstance(m, (nn.BatchNorm1d, nn.Dropout)) for m in self.network):
            raise RuntimeError("FCNN.forward on the GPU is inference-only: call .eval() first (BatchNorm1d / Dropout "
                               "in training mode are not implemented)")
        dev = x.device if x.is_cuda else torch.device("cuda", 0)
        _lib.require_gpu(dev.index or 0)
        mlp = self.device_mlp(dev)
        h = x.detach().to(dev, torch.float32).contiguous()
        if mlp.fits:
            h = mlp.forward(h)
        else:  # too large for the LDS: the same arithmetic, one launch per layer
            for (w, b, sc, sh, act) in self._plan:
                h = dense_forward(h, w, b, sc, sh, act)
        return h if x.is_cuda else h.cpu()

    def forward_layerwise(self, x: torch.Tensor) -> torch.Tensor:
        """The same forward pass as a chain of ``ofp_dense`` launches (what round 1 shipped): kept as
        the bit-for-bit cross-check of the fused kernel."""
        dev = x.device if x.is_cuda else torch.device("cuda", 0)
        self.device_mlp(dev)
        h = x.detach().to(dev, torch.float32).contiguous()
        for (w, b, sc, sh, act) in self._plan:
            h = dense_forward(h, w, b, sc, sh, act)
        return h if x.is_cuda else h.cpu()

    def l2_loss(self) -> torch.Tensor:
        if self.l2_reg == 0.0:
            return torch.tensor(0.0)
        return self.l2_reg * sum(torch.sum(p ** 2) for p in self.parameters())

    def call_np(self, lags) -> np.ndarray:
        """calibration.py:552-560: one sample in, one numpy row out."""
        with torch.no_grad():
            return self(torch.tensor([lags], dtype=torch.float32)).numpy()[0]
