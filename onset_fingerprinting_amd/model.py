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


def conv1d_forward(x, weight, bias, padding, dilation, act_code):
    L = _lib.lib()
    n, cin, w = x.shape
    cout, _, k = weight.shape
    wout = w + 2 * padding - dilation * (k - 1)
    out = torch.empty((n, cout, wout), dtype=torch.float32, device=x.device)
    check(L.ofp_conv1d(x.data_ptr(), n, cin, w, weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                       cout, k, padding, dilation, act_code, out.data_ptr(), _stream(x.device)), "ofp_conv1d")
    return out


class CNN(nn.Module):
    def __init__(self, input_size: int, output_size: int, channels: int = 3, layer_sizes=[8, 16],
                 kernel_size: int = 3, dropout_rate: float = 0.5, loss=F.l1_loss, batch_norm=False, pool=False,
                 padding=1, dilation=1, groups=1, lr=1e-3, activation=nn.SiLU) -> None:
        super().__init__()
        if batch_norm or pool or groups != 1:
            raise NotImplementedError("batch_norm / pool / groups != 1 are not on the accelerated path yet")
        if activation not in ACT_CODES:
            raise ValueError(f"activation {activation} has no HIP implementation")
        self._act_code = ACT_CODES[activation]
        self._padding, self._dilation = padding, dilation
        self.conv_layers = nn.Sequential()
        cur, width = channels, input_size
        for i, size in enumerate(layer_sizes):
            self.conv_layers.add_module(
                f"conv{i+1}", nn.Conv1d(cur, size, kernel_size, padding=padding, dilation=dilation, groups=groups))
            self.conv_layers.add_module(f"act{i+1}", activation())
            width = width + 2 * padding - dilation * (kernel_size - 1)
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
        h = to(x)
        for m in self.conv_layers:
            if isinstance(m, nn.Conv1d):
                h = conv1d_forward(h, to(m.weight), to(m.bias) if m.bias is not None else None,
                                   self._padding, self._dilation, self._act_code)
        h = h.reshape(h.shape[0], -1)
        h = dense_forward(h, to(self.fc.weight), to(self.fc.bias), None, None, 0)
        return h if x.is_cuda else h.cpu()


def autocorr_softmax(feat):
    """feat float32 CUDA [n, K, V] -> [n, 2V-1]: the correlation head of CCCNN (model.py:524-534)."""
    L = _lib.lib()
    n, K, V = feat.shape
    out = torch.empty((n, 2 * V - 1), dtype=torch.float32, device=feat.device)
    check(L.ofp_autocorr_softmax(feat.data_ptr(), n, K, V, out.data_ptr(), _stream(feat.device)),
          "ofp_autocorr_softmax")
    return out


class CCCNN(nn.Module):
    """``model.CCCNN`` (model.py:443-538), non-grouped form: a shared single-input-channel conv
    stack applied to every sensor channel, the auto-correlation of every feature map summed
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
        if group or batch_norm or pool or any(s != 1 for s in strides):
            raise NotImplementedError("group / batch_norm / pool / stride != 1 are not on the accelerated path yet")
        if activation not in ACT_CODES:
            raise ValueError(f"activation {activation} has no HIP implementation")
        self.group, self.channels = group, channels
        self._act_code, self._padding, self._dilation = ACT_CODES[activation], padding, dilation
        self.conv_layers = nn.Sequential()
        cur, width = 1, input_size
        for i, (size, k) in enumerate(zip(layer_sizes, kernel_sizes)):
            self.conv_layers.add_module(f"conv{i+1}", nn.Conv1d(cur, size, k, padding=padding, dilation=dilation))
            self.conv_layers.add_module(f"act{i+1}", activation())
            width = width + 2 * padding - dilation * (k - 1)
            cur = size
        self.dropout = nn.Dropout(dropout_rate)
        self.fc = nn.Linear(channels * (2 * width - 1), output_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [batch, channels, input_size] -> [batch, output_size] (eval-mode semantics)."""
        dev = x.device if x.is_cuda else torch.device("cuda", 0)
        _lib.require_gpu(dev.index or 0)
        to = lambda t: t.detach().to(dev, torch.float32).contiguous()
        B, C, W = x.shape
        h = to(x).reshape(B * C, 1, W)  # the shared stack sees every sensor channel as its own item
        for m in self.conv_layers:
            if isinstance(m, nn.Conv1d):
                h = conv1d_forward(h, to(m.weight), to(m.bias) if m.bias is not None else None,
                                   self._padding, self._dilation, self._act_code)
        probs = autocorr_softmax(h)  # [B*C, 2V-1]
        out = dense_forward(probs.reshape(B, -1), to(self.fc.weight), to(self.fc.bias), None, None, 0)
        return out if x.is_cuda else out.cpu()
