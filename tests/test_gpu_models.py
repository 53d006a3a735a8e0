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


def _bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("kw", [
    dict(input_size=40, output_size=8),
    dict(input_size=2, output_size=2, hidden_layers=[16, 12], activation=torch.nn.SiLU, batch_norm=False),
    dict(input_size=14, output_size=3, hidden_layers=[32], activation=torch.nn.ELU, bias=False),
    dict(input_size=37, output_size=19, hidden_layers=[40, 33, 17, 5], activation=torch.nn.LeakyReLU),
    dict(input_size=5, output_size=70, hidden_layers=[], activation=torch.nn.Tanh),
])
def test_fused_fcnn_is_bit_identical_to_the_layer_chain(kw):
    """ofp_mlp_forward (one launch, activations in LDS) against the ofp_dense chain of round 1: the
    same MFMA chain per output element, hence the same bits -- also for ragged row counts, widths
    that are no multiple of 4 or 16, several column tiles and networks without hidden layers."""
    from onset_fingerprinting_amd.calibration import FCNN
    torch.manual_seed(3)
    m = FCNN(**kw).eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 1.5)
    assert m.device_mlp(0).fits
    for n in (1, 16, 33, 4099):
        x = torch.randn(n, kw["input_size"]).cuda()
        assert np.array_equal(_bits(m(x)), _bits(m.forward_layerwise(x))), (kw, n)


def test_fcnn_plan_follows_parameter_updates_and_refuses_training_mode():
    """the folded device copy must not go stale after in-place updates, a submodule
    load_state_dict or init_eye_weights; training-mode BatchNorm / Dropout is refused."""
    from onset_fingerprinting_amd.calibration import FCNN
    torch.manual_seed(5)
    m = FCNN(6, 3, hidden_layers=[7]).eval()
    x = torch.randn(9, 6)
    y0 = m(x).numpy()
    with torch.no_grad():
        m.network[0].weight.mul_(2.0)                      # in place
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    y1 = m(x).numpy()
    assert not np.allclose(y0, y1) and rel_err(y1, oracle.fcnn_forward(sd, x.numpy())) < 1e-4
    m.network[0].load_state_dict({"weight": torch.randn(7, 6), "bias": torch.zeros(7)})   # submodule
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    assert rel_err(m(x).numpy(), oracle.fcnn_forward(sd, x.numpy())) < 1e-4
    m.init_eye_weights(m.network[0])                       # replaces .data
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    assert rel_err(m(x).numpy(), oracle.fcnn_forward(sd, x.numpy())) < 1e-4
    m.network[1].running_mean.add_(0.25)                   # buffer in place
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    assert rel_err(m(x).numpy(), oracle.fcnn_forward(sd, x.numpy())) < 1e-4
    m.train()
    with pytest.raises(RuntimeError):
        m(x)
    FCNN(6, 3, hidden_layers=[7], batch_norm=False).train()(x)  # nothing mode-dependent: allowed


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
