"""numpy restatement of the reference's classifier forward passes (eval mode)
-- TEST INFRASTRUCTURE ONLY.  PINNED against golden vectors captured from the
reference's own torch modules (tests/golden/make_golden.py).

Weights are given as a torch-style ``state_dict`` of numpy arrays with the
reference's parameter names (``network.{k}.weight`` for calibration.FCNN,
``conv_layers.conv{i}.weight`` / ``fc.weight`` for model.CNN).
"""
import numpy as np

_ACT = {
    "relu": lambda x: np.maximum(x, 0.0),
    "silu": lambda x: x / (1.0 + np.exp(-x)),
    "leakyrelu": lambda x: np.where(x >= 0, x, 0.01 * x),
    "elu": lambda x: np.where(x > 0, x, np.expm1(np.minimum(x, 0.0))),
    "tanh": np.tanh,
    "identity": lambda x: x,
}


def fcnn_forward(sd, x, activation="relu", eps=1e-5):
    """calibration.py:463-527 in eval mode: [Linear -> (BatchNorm1d running
    stats) -> act] per hidden layer, then Linear.  fp64 arithmetic."""
    x = np.asarray(x, dtype=np.float64)
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("network.")})
    lin = [i for i in idx if sd[f"network.{i}.weight"].ndim == 2]
    act = _ACT[activation]
    for n, i in enumerate(lin):
        W = sd[f"network.{i}.weight"].astype(np.float64)
        x = x @ W.T
        if f"network.{i}.bias" in sd:
            x = x + sd[f"network.{i}.bias"].astype(np.float64)
        if n == len(lin) - 1:
            break
        j = i + 1
        if f"network.{j}.running_mean" in sd:
            mu = sd[f"network.{j}.running_mean"].astype(np.float64)
            var = sd[f"network.{j}.running_var"].astype(np.float64)
            g = sd[f"network.{j}.weight"].astype(np.float64)
            b = sd[f"network.{j}.bias"].astype(np.float64)
            x = (x - mu) / np.sqrt(var + eps) * g + b
        x = act(x)
    return x


def cnn_forward(sd, x, padding=1, dilation=1, activation="silu"):
    """model.py:52-120 in eval mode with batch_norm=False, pool=False, groups=1:
    [Conv1d(k, padding, dilation) -> act] per layer, flatten, Linear.
    x [B, C, W] -> [B, out].  fp64 arithmetic."""
    x = np.asarray(x, dtype=np.float64)
    act = _ACT[activation]
    i = 1
    while f"conv_layers.conv{i}.weight" in sd:
        W = sd[f"conv_layers.conv{i}.weight"].astype(np.float64)  # [O, I, K]
        b = sd[f"conv_layers.conv{i}.bias"].astype(np.float64)
        O, I, K = W.shape
        Bn, _, L = x.shape
        xp = np.pad(x, ((0, 0), (0, 0), (padding, padding)))
        Lout = L + 2 * padding - dilation * (K - 1)
        y = np.zeros((Bn, O, Lout))
        for k in range(K):
            seg = xp[:, :, k * dilation: k * dilation + Lout]  # [B, I, Lout]
            y += np.einsum("bil,oi->bol", seg, W[:, :, k])
        x = act(y + b[None, :, None])
        i += 1
    x = x.reshape(x.shape[0], -1)
    return x @ sd["fc.weight"].astype(np.float64).T + sd["fc.bias"].astype(np.float64)


def cccnn_forward(sd, x, padding=1, dilation=1, activation="silu"):
    """model.py:443-538 (group=False, no batch_norm / pool, stride 1) in eval mode:
    shared 1-input-channel conv stack per sensor channel -> full auto-correlation of every
    feature map (F.conv1d(x, x, groups, padding=V-1)) summed over the maps -> softmax over
    the lags -> flatten -> Linear.  x [B, C, W] -> [B, out].  fp64 arithmetic."""
    x = np.asarray(x, dtype=np.float64)
    B, C, W = x.shape
    act = _ACT[activation]
    h = x.reshape(B * C, 1, W)
    i = 1
    while f"conv_layers.conv{i}.weight" in sd:
        Wt = sd[f"conv_layers.conv{i}.weight"].astype(np.float64)  # [O, I, K]
        b = sd[f"conv_layers.conv{i}.bias"].astype(np.float64)
        O, I, K = Wt.shape
        n, _, L = h.shape
        hp = np.pad(h, ((0, 0), (0, 0), (padding, padding)))
        Lout = L + 2 * padding - dilation * (K - 1)
        y = np.zeros((n, O, Lout))
        for k in range(K):
            y += np.einsum("bil,oi->bol", hp[:, :, k * dilation: k * dilation + Lout], Wt[:, :, k])
        h = act(y + b[None, :, None])
        i += 1
    n, K, V = h.shape
    cc = np.zeros((n, 2 * V - 1))
    for j in range(2 * V - 1):
        sh = j - (V - 1)
        lo, hi = max(0, -sh), min(V, V - sh)
        cc[:, j] = (h[:, :, lo + sh: hi + sh] * h[:, :, lo:hi]).sum(axis=(1, 2))
    e = np.exp(cc - cc.max(axis=1, keepdims=True))
    probs = (e / e.sum(axis=1, keepdims=True)).reshape(B, -1)
    return probs @ sd["fc.weight"].astype(np.float64).T + sd["fc.bias"].astype(np.float64)
