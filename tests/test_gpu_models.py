"""GPU parity of the classifier forwards (fp32 MFMA dense layers, Conv1d) against
the reference's golden outputs and the numpy oracle; tolerance 1e-4 relative."""
import numpy as np
import pytest
import torch

import oracle
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

ACTS = {"relu": torch.nn.ReLU, "silu": torch.nn.SiLU, "elu": torch.nn.ELU}


def rel_err(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-12)


def test_fcnn_matches_reference_golden():
    from onset_fingerprinting_amd.calibration import FCNN
    g = load_golden("g8_models")
    cfgs = {
        "fc_a": dict(input_size=40, output_size=8),
        "fc_b": dict(input_size=2, output_size=2, hidden_layers=[16, 12], activation=torch.nn.SiLU, batch_norm=False),
        "fc_c": dict(input_size=14, output_size=3, hidden_layers=[32], activation=torch.nn.ELU, bias=False),
    }
    for name, kw in cfgs.items():
        m = FCNN(**kw)
        sd = {k.split("/", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + "/network")}
        m.load_state_dict(sd)  # the reference's own state_dict keys load unchanged
        m.eval()
        x = g[f"{name}/x"]
        y = m(torch.from_numpy(x)).numpy()
        assert y.shape == g[f"{name}/y"].shape
        assert rel_err(y, g[f"{name}/y"]) < 1e-4, name
        ynp = {k: v.numpy() for k, v in sd.items()}
        assert rel_err(y, oracle.fcnn_forward(ynp, x, activation=str(g[f"{name}/act"]))) < 1e-4
        # call_np: one sample (calibration.py:552-560)
        one = m.call_np(tuple(float(v) for v in x[0]))
        assert one.shape == (kw["output_size"],) and rel_err(one, g[f"{name}/y"][0]) < 1e-4


def test_fcnn_large_batch_and_ragged_rows():
    from onset_fingerprinting_amd.calibration import FCNN
    torch.manual_seed(0)
    m = FCNN(40, 8).eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 1.5)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    for n in (1, 15, 17, 100003):
        x = torch.randn(n, 40)
        y = m(x.cuda()).cpu().numpy()
        assert rel_err(y, oracle.fcnn_forward(sd, x.numpy())) < 1e-4


def test_cnn_matches_reference_golden():
    from onset_fingerprinting_amd.model import CNN
    g = load_golden("g8_models")
    for name, kw in (("cnn_a", dict(input_size=256, output_size=2, channels=4)),
                     ("cnn_b", dict(input_size=64, output_size=3, channels=3, layer_sizes=[4, 6, 8],
                                    kernel_size=5, padding=2))):
        m = CNN(**kw).eval()
        sd = {k.split("/", 1)[1]: torch.from_numpy(g[k]) for k in g.files
              if k.startswith(name + "/") and k.split("/")[1] not in ("x", "y")}
        m.load_state_dict(sd)
        y = m(torch.from_numpy(g[f"{name}/x"])).numpy()
        assert rel_err(y, g[f"{name}/y"]) < 1e-4, name


def test_cnn_cccnn_constructor_options_match_reference_golden():
    """batch_norm (eval), MaxPool, groups, dilation for CNN; group / pool / GroupNorm / strides for CCCNN and
    the LCCCNN configuration of the reference's train.py:79-90 (model.py:62-67, 451-456, 541-580)."""
    from onset_fingerprinting_amd import model
    from tests.golden.make_golden_next_cfg import G14
    g = load_golden("g14_model_variants")
    for name, (cls, kw) in G14.items():
        m = getattr(model, cls)(**kw).eval()
        sd = {k.split("/", 1)[1]: torch.from_numpy(g[k]) for k in g.files
              if k.startswith(name + "/") and k.split("/")[1] not in ("x", "y")}
        m.load_state_dict(sd)  # the reference's own keys, BatchNorm buffers included
        y = m(torch.from_numpy(g[f"{name}/x"])).numpy()
        assert y.shape == g[f"{name}/y"].shape and rel_err(y, g[f"{name}/y"]) < 1e-4, name


def test_batch_cc_and_paired_xcorr_against_torch():
    """data.batch_cc (data.py:226-230) and model.paired_xcorr (model.py:12-45): the same grouped
    F.conv1d expressions evaluated by torch on the CPU are the fp32 reference."""
    import torch.nn.functional as F
    from onset_fingerprinting_amd import data, model
    torch.manual_seed(3)
    for n, length in ((1, 1), (7, 33), (64, 256), (3, 1000)):
        a, b = torch.randn(n, length), torch.randn(n, length)
        ref = F.conv1d(a.reshape(1, n, length), b[:, None, :], padding=length - 1, groups=n)[0]
        got = data.batch_cc(a, b)
        assert got.shape == ref.shape and rel_err(got.numpy(), ref.numpy()) < 1e-4
    for B, C, K, V in ((2, 3, 4, 50), (1, 4, 5, 128), (3, 2, 1, 17)):
        x = torch.randn(B, C * K, V)
        xv = x.view(B, C, K, V)
        a, b = xv[:, :-1].reshape(B, (C - 1) * K, V), xv[:, 1:].reshape(B, (C - 1) * K, V)
        M = B * (C - 1) * K
        ref = F.conv1d(F.pad(a, (V - 1, V - 1)).view(1, M, 3 * V - 2), b.reshape(M, 1, V), groups=M)
        ref = ref.view(B, C - 1, K, 2 * V - 1).mean(dim=2)
        got = model.paired_xcorr(x, C, K)
        assert got.shape == ref.shape and rel_err(got.numpy(), ref.numpy()) < 1e-4
