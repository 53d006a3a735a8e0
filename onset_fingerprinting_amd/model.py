"""``model.CNN`` (model.py:52-120) with its forward pass on MI355X.

Same constructor arguments and parameter names (``conv_layers.conv{i}``, ``fc``)
as the reference so its ``state_dict`` loads unchanged; ``forward`` is
inference-only: Conv1d + bias + activation per layer and the final Linear run
as HIP kernels (csrc/ofp_nn.hip).  Lightning training steps, optimisers and the
RNN/CCCNN families are out of scope for this round.
"""
import ctypes

import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from ._lib import check
from .calibration import ACT_CODES, dense_forward


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def conv1d_forward(x, weight, bias, padding, dilation, act_code, groups=1, bn=None, pool=False, stride=1):
    """Conv1d + bias + activation (+ eval-mode BatchNorm1d `bn` + MaxPool1d(2, 2)): the layer of
    model.py:91-107 in one kernel."""
    L = _lib.lib()
    n, cin, w = x.shape
    cout, _, k = weight.shape
    wout = (w + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    if pool:
        wout //= 2
    scale = shift = None
    if bn is not None:  # y = (v - mean) / sqrt(var + eps) * gamma + beta, folded in double
        inv = (bn.running_var.double() + bn.eps).rsqrt()
        g = bn.weight.double() if bn.weight is not None else torch.ones_like(inv)
        b = bn.bias.double() if bn.bias is not None else torch.zeros_like(inv)
        scale = (g * inv).to(x.device, torch.float32).contiguous()
        shift = (b - bn.running_mean.double() * g * inv).to(x.device, torch.float32).contiguous()
    out = torch.empty((n, cout, wout), dtype=torch.float32, device=x.device)
    check(L.ofp_conv1d(x.data_ptr(), n, cin, w, weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                       cout, k, padding, dilation, groups, stride, act_code,
                       scale.data_ptr() if scale is not None else None,
                       shift.data_ptr() if shift is not None else None, int(bool(pool)), out.data_ptr(),
                       _stream(x.device)), "ofp_conv1d")
    return out


def groupnorm1_forward(x, gn, pool=False):
    """nn.GroupNorm(1, K) (+ MaxPool1d(2, 2)) on x float32 CUDA [n, K, V]."""
    assert gn.num_groups == 1
    L = _lib.lib()
    n, K, V = x.shape
    out = torch.empty((n, K, V // 2 if pool else V), dtype=torch.float32, device=x.device)
    g = gn.weight.detach().to(x.device, torch.float32).contiguous() if gn.weight is not None else None
    b = gn.bias.detach().to(x.device, torch.float32).contiguous() if gn.bias is not None else None
    check(L.ofp_groupnorm1(x.data_ptr(), n, K, V, g.data_ptr() if g is not None else None,
                           b.data_ptr() if b is not None else None, float(gn.eps), int(bool(pool)), out.data_ptr(),
                           _stream(x.device)), "ofp_groupnorm1")
    return out


def _run_conv_stack(layers, h, to, padding, dilation, act_code, groups):
    mods = list(layers)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv1d):
            bn, gn, pool, j = None, None, False, i + 1
            while j < len(mods) and not isinstance(mods[j], nn.Conv1d):
                if isinstance(mods[j], nn.BatchNorm1d):
                    bn = mods[j]
                elif isinstance(mods[j], nn.GroupNorm):
                    gn = mods[j]
                elif isinstance(mods[j], nn.MaxPool1d):
                    pool = True
                j += 1
            # conv + bias + activation (+ folded BatchNorm + pool) in one kernel; a GroupNorm needs the
            # whole item first, so it (and the pool after it) runs as a second kernel
            h = conv1d_forward(h, to(m.weight), to(m.bias) if m.bias is not None else None, padding, dilation,
                               act_code, groups=groups, bn=bn, pool=pool and gn is None, stride=m.stride[0])
            if gn is not None:
                h = groupnorm1_forward(h, gn, pool=pool)
            i = j
        else:
            i += 1
    return h


class CNN(nn.Module):
    def __init__(self, input_size: int, output_size: int, channels: int = 3, layer_sizes=[8, 16],
                 kernel_size: int = 3, dropout_rate: float = 0.5, loss=F.l1_loss, batch_norm=False, pool=False,
                 padding=1, dilation=1, groups=1, lr=1e-3, activation=nn.SiLU) -> None:
        super().__init__()
        if activation not in ACT_CODES:
            raise ValueError(f"activation {activation} has no HIP implementation")
        self._act_code = ACT_CODES[activation]
        self._padding, self._dilation, self._groups = padding, dilation, groups
        self.conv_layers = nn.Sequential()
        cur, width = channels, input_size
        for i, size in enumerate(layer_sizes):
            self.conv_layers.add_module(
                f"conv{i+1}", nn.Conv1d(cur, size, kernel_size, padding=padding, dilation=dilation, groups=groups))
            self.conv_layers.add_module(f"act{i+1}", activation())
            width = width + 2 * padding - dilation * (kernel_size - 1)
            if batch_norm:
                self.conv_layers.add_module(f"bn{i+1}", nn.BatchNorm1d(size))
            if pool:
                self.conv_layers.add_module(f"pool{i+1}", nn.MaxPool1d(kernel_size=2, stride=2))
                width //= 2
            cur = size
        self.dropout = nn.Dropout(dropout_rate)
        self.fc = nn.Linear(cur * width, output_size)
        self.loss = loss
        self.lr = lr

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [batch, channels, input_size] -> [batch, output_size] (eval-mode semantics)."""
        dev = x.device if x.is_cuda else torch.device("cuda", 0)
        _lib.require_gpu(dev.index or 0)
        to = lambda t: t.detach().to(dev, torch.float32).contiguous()
        h = _run_conv_stack(self.conv_layers, to(x), to, self._padding, self._dilation, self._act_code, self._groups)
        h = h.reshape(h.shape[0], -1)
        h = dense_forward(h, to(self.fc.weight), to(self.fc.bias), None, None, 0)
        return h if x.is_cuda else h.cpu()


def paired_xcorr(x: torch.Tensor, C: int, K: int) -> torch.Tensor:
    """model.py:12-45: cross-correlate every adjacent channel pair (1&2, 2&3, ...) in each feature
    map and average over the maps: (B, C*K, V) -> (B, C-1, 2V-1)."""
    B, CK, V = x.shape
    assert CK == C * K
    dev = x.device if x.is_cuda else torch.device("cuda", 0)
    _lib.require_gpu(dev.index or 0)
    xd = x.detach().to(dev, torch.float32).contiguous().view(B, C, K, V)
    out = torch.empty((B, C - 1, 2 * V - 1), dtype=torch.float32, device=dev)
    L = _lib.lib()
    for bi in range(B):  # rows (c, k) of a sample are contiguous: a = channels 0..C-2, b = channels 1..C-1
        a, b = xd[bi, :-1], xd[bi, 1:]
        check(L.ofp_xcorr_full(a.data_ptr(), b.data_ptr(), (C - 1) * K, V, V, V, K, out[bi].data_ptr(),
                               _stream(dev)), "ofp_xcorr_full")
    return out if x.is_cuda else out.cpu()


def autocorr_softmax(feat):
    """feat float32 CUDA [n, K, V] -> [n, 2V-1]: the correlation head of CCCNN (model.py:524-534)."""
    L = _lib.lib()
    n, K, V = feat.shape
    out = torch.empty((n, 2 * V - 1), dtype=torch.float32, device=feat.device)
    check(L.ofp_autocorr_softmax(feat.data_ptr(), n, K, V, out.data_ptr(), _stream(feat.device)),
          "ofp_autocorr_softmax")
    return out


class CCCNN(nn.Module):
    """``model.CCCNN`` (model.py:443-538): a conv stack applied to every sensor channel (shared, or one
    private stack per channel with ``group=True``), optional MaxPool, the auto-correlation of every feature map summed
    over the maps, a softmax over the lags, and a Linear on the flattened result.  Same
    constructor arguments and parameter names (``conv_layers.conv{i}``, ``fc``) as the
    reference; ``forward`` is inference-only and runs as HIP kernels."""

    def __init__(self, input_size: int, output_size: int, channels: int = 3, layer_sizes=[8, 16],
                 kernel_sizes=3, strides=1, dropout_rate: float = 0.5, batch_norm=False, pool=False, padding=1,
                 dilation=1, group: bool = False, activation=nn.SiLU) -> None:
        super().__init__()
        if isinstance(kernel_sizes, int):
            kernel_sizes = [kernel_sizes] * len(layer_sizes)
        if isinstance(strides, int):
            strides = [strides] * len(layer_sizes)
        if activation not in ACT_CODES:
            raise ValueError(f"activation {activation} has no HIP implementation")
        self.group, self.channels = group, channels
        self._act_code, self._padding, self._dilation = ACT_CODES[activation], padding, dilation
        self.conv_layers = nn.Sequential()
        g = channels if group else 1  # model.py:466,484: one private stack per sensor channel when grouped
        cur, width = g, input_size
        for i, (size, k, stride) in enumerate(zip(layer_sizes, kernel_sizes, strides)):
            self.conv_layers.add_module(
                f"conv{i+1}", nn.Conv1d(cur, size * g, k, padding=padding, dilation=dilation, stride=stride, groups=g))
            self.conv_layers.add_module(f"act{i+1}", activation())
            width = (width + 2 * padding - dilation * (k - 1) - 1) // stride + 1
            if batch_norm:  # model.py:497-501: a GroupNorm with one group, despite the argument's name
                self.conv_layers.add_module(f"bn{i+1}", nn.GroupNorm(1, size * g))
            if pool:
                self.conv_layers.add_module(f"pool{i+1}", nn.MaxPool1d(kernel_size=2, stride=2))
                width //= 2
            cur = size * g
        self.dropout = nn.Dropout(dropout_rate)
        self.fc = nn.Linear(channels * (2 * width - 1), output_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [batch, channels, input_size] -> [batch, output_size] (eval-mode semantics)."""
        dev = x.device if x.is_cuda else torch.device("cuda", 0)
        _lib.require_gpu(dev.index or 0)
        to = lambda t: t.detach().to(dev, torch.float32).contiguous()
        B, C, W = x.shape
        if self.group:  # grouped conv: [B, C*K, V], channel-major, i.e. already [B*C, K, V]
            h = _run_conv_stack(self.conv_layers, to(x), to, self._padding, self._dilation, self._act_code, C)
            h = h.reshape(B * C, h.shape[1] // C, h.shape[2])
        else:  # the shared stack sees every sensor channel as its own item
            h = _run_conv_stack(self.conv_layers, to(x).reshape(B * C, 1, W), to, self._padding, self._dilation,
                                self._act_code, 1)
        probs = autocorr_softmax(h)  # [B*C, 2V-1]
        out = dense_forward(probs.reshape(B, -1), to(self.fc.weight), to(self.fc.bias), None, None, 0)
        return out if x.is_cuda else out.cpu()


class LCCCNN(nn.Module):
    """``model.LCCCNN`` (model.py:541-580): the training wrapper around CCCNN; inference only here.
    Parameter names (``model.conv_layers.conv{i}``, ``model.fc``) follow the reference."""

    def __init__(self, input_size: int, output_size: int, channels: int = 3, layer_sizes=[8, 16], kernel_sizes=3,
                 strides=1, dropout_rate: float = 0.5, batch_norm=False, pool=False, padding=1, dilation=1,
                 group: bool = False, activation=nn.SiLU, loss=F.l1_loss, lr=1e-3) -> None:
        super().__init__()
        self.model = CCCNN(input_size, output_size, channels, layer_sizes, kernel_sizes, strides, dropout_rate,
                           batch_norm, pool, padding, dilation, group, activation)
        self.lr = lr
        self.loss = loss

    def forward(self, x):
        return self.model(x)
