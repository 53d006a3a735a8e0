#!/usr/bin/env python3
"""Captures golden input/output vectors from the upstream reference.

Run in the build container only (needs /root/reference and
`make -C oracle ref`):   python tests/golden/make_golden.py
Writes small .npz files next to this script.  Only DATA is stored: inputs
(or the seed of the generator in onset_fingerprinting_amd/synth.py plus a
checksum) and the reference's outputs.  No reference source is copied.

Vector sets:
  g0_hostmath         numpy float32 log10/power of this host on a fixed probe
  g1_ar_envelope      reference .so via detection.AREnvelopeFollower
  g2_minmax           reference .so via detection.MinMaxEnvelopeFollower
  g3_detector_blocks  detection.AmplitudeOnsetDetector.__call__, per block
  g4_end_to_end       detection.detect_onsets_amplitude on C1 and a C2 slice
  g6_stft             data.stft / data.stft_frame
  g7_frames           data.FrameExtractor
  g8_models           calibration.FCNN / model.CNN forward (eval)
  g9_backtrack        reference .so backtrack_onsets (ctypes, unbound upstream)
  g10_wcw             data.window_contribution_weights
  g11_lfilter         scipy.signal.lfilter float32 path used by ButterworthFilter
"""
import ctypes
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from _refload import REF_SO_DIR, load_reference  # noqa: E402

from onset_fingerprinting_amd import synth  # noqa: E402

warnings.filterwarnings("ignore")
ref = load_reference()
det, data = ref.detection, ref.data


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}: {path.stat().st_size / 1024:.1f} KiB")


AR_PAIRS = [(3.0, 383.0), (2205.0, 2205.0), (0.3, 800.0), (8000.0, 8000.0), (2.0, 966.0)]


def g1_g2():
    rng = np.random.default_rng(101)
    nblk, B, C = 12, 64, 8
    # dB-range input with steps, floor plateaus and spikes
    x = rng.uniform(-70, -20, (nblk * B, C)).astype(np.float32)
    x[100:140] = -70.0
    x[300:310, 2] = 0.0
    x[500:, 5] = np.linspace(-70, -5, nblk * B - 500, dtype=np.float32)
    out = {"x": x, "pairs": np.array(AR_PAIRS)}
    for k, (a, r) in enumerate(AR_PAIRS):
        f = det.AREnvelopeFollower(np.full((B, C), -70.0, dtype=np.float32), a, r)
        ys = []
        for i in range(nblk):
            ys.append(f(np.ascontiguousarray(x[i * B:(i + 1) * B])).copy())
        out[f"y{k}"] = np.concatenate(ys)
    save("g1_ar_envelope", **out)

    xr = rng.uniform(0.0, 6.0, (nblk * B, C)).astype(np.float32)
    xr[200:230, 1] = 25.0
    xr[400:402, 3] = 69.5
    xr[::37, 4] = 1.5  # values < minmin
    mm = det.MinMaxEnvelopeFollower(x0=np.array([[0, 10]] * C).T, alpha_min=1e-4,
                                    alpha_max=1e-5, minmin=2)
    mins, maxs = [], []
    for i in range(nblk):
        mi, ma = mm(np.ascontiguousarray(xr[i * B:(i + 1) * B]))
        mins.append(mi.copy())
        maxs.append(ma.copy())
    # a second tracker with larger alphas / minmin 0 to exercise the EMA branches
    mm2 = det.MinMaxEnvelopeFollower(x0=np.array([[1, 3]] * C).T, alpha_min=1e-2,
                                     alpha_max=3e-3, minmin=0.0)
    mins2, maxs2 = [], []
    for i in range(nblk):
        mi, ma = mm2(np.ascontiguousarray(xr[i * B:(i + 1) * B]))
        mins2.append(mi.copy())
        maxs2.append(ma.copy())
    save("g2_minmax", x=xr, B=B, mins=np.array(mins), maxs=np.array(maxs),
         mins2=np.array(mins2), maxs2=np.array(maxs2))


from make_golden_cfg import G3_CONFIGS  # noqa: E402


def _run_blocks(od, x, B):
    recs, rels = [], []
    for i in range(0, len(x) - B + 1, B):
        c, d, r = od(np.ascontiguousarray(x[i:i + B]))
        rels.append(r.copy())
        for cc, dd in zip(c, d):
            recs.append((i // B, int(cc), int(dd)))
    rel = np.concatenate(rels) if rels else np.zeros((0, x.shape[1]), np.float32)
    return np.array(recs, dtype=np.int64).reshape(-1, 3), rel


def g3():
    sr = 48000
    x = synth.drum_hits(3, 0.5, sr, seed=11, period=0.11)
    # second input: staggered hits inside one block to trigger the cross-channel
    # off-threshold mask (detection.py:790)
    x2 = 1e-3 * np.random.default_rng(12).standard_normal((8192, 2))
    k = np.arange(60)
    for start, c, a in [(2100, 0, 0.9), (2100 + 90, 1, 0.9), (4200, 1, 0.5), (4200 + 200, 0, 0.7),
                        (6000, 0, 0.8), (6003, 1, 0.8)]:
        x2[start:start + 60, c] += a * np.exp(-k / 6.0) * np.sign(np.sin(k))
    x2 = x2.astype(np.float32)
    out = {"x": x, "x2": x2, "sr": sr, "n_configs": len(G3_CONFIGS)}
    for k, cfg in enumerate(G3_CONFIGS):
        for tag, xin in (("a", x), ("b", x2)):
            cfg2 = dict(cfg)
            B = cfg2.pop("block_size")
            od = det.AmplitudeOnsetDetector(xin.shape[1], B, sr=sr, **cfg2)
            if k % 2 == 1:  # odd configs: with the warm-up pass
                od.init_minmax_tracker(xin[: int(0.05 * sr)])
            recs, rel = _run_blocks(od, xin, B)
            out[f"rec_{k}{tag}"] = recs
            out[f"rel_{k}{tag}"] = rel[::5].copy()
            out[f"relsum_{k}{tag}"] = rel.astype(np.float64).sum(axis=0)
            out[f"state_{k}{tag}"] = np.concatenate([
                od.state.astype(np.float64), od.prev_values, od.debounce_count.astype(np.float64),
                od.minmax_tracker.min_val.astype(np.float64),
                od.minmax_tracker.max_val.astype(np.float64)])
        out[f"cfg_{k}"] = np.array(repr(cfg))
    save("g3_detector_blocks", **out)


def g4():
    sr = 48000
    out = {}
    x1 = synth.c1_sine_clicks(10.0, sr, seed=0)
    for B in (128, 256):
        c, o, rel = det.detect_onsets_amplitude(x1, block_size=B, sr=sr)
        out[f"c1_B{B}_ch"] = np.array(c, dtype=np.int64)
        out[f"c1_B{B}_on"] = np.array(o, dtype=np.int64)
        out[f"c1_B{B}_rel"] = rel[::97].copy()
        out[f"c1_B{B}_relsum"] = rel.astype(np.float64).sum(axis=0)
    out["c1_xsum"] = x1.astype(np.float64).sum()
    x2 = synth.c2_drums(10.0, 8, sr, seed=1)
    out["c2_xsum"] = x2.astype(np.float64).sum()
    c, o, rel = det.detect_onsets_amplitude(x2, block_size=256, sr=sr)
    out["c2_ch"] = np.array(c, dtype=np.int64)
    out["c2_on"] = np.array(o, dtype=np.int64)
    out["c2_rel"] = rel[::997].copy()
    out["c2_relsum"] = rel.astype(np.float64).sum(axis=0)
    # realtime parameter set (realtime/audio.py:39-52) on the same slice
    c, o, rel = det.detect_onsets_amplitude(
        x2[:, :3].copy(), block_size=128, hipass_freq=0, fast_ar=(0.3, 800.0),
        slow_ar=(8000.0, 8000.0), on_threshold=0.45, off_threshold=0.45, cooldown=9600, sr=sr)
    out["rt_ch"] = np.array(c, dtype=np.int64)
    out["rt_on"] = np.array(o, dtype=np.int64)
    out["rt_relsum"] = rel.astype(np.float64).sum(axis=0)
    # C4-style clip (Poisson hits, 4 ch)
    x4 = synth.c4_clip(7, 4.0, 4, sr)
    out["c4_xsum"] = x4.astype(np.float64).sum()
    c, o, rel = det.detect_onsets_amplitude(x4, block_size=256, sr=sr)
    out["c4_ch"] = np.array(c, dtype=np.int64)
    out["c4_on"] = np.array(o, dtype=np.int64)
    out["c4_relsum"] = rel.astype(np.float64).sum(axis=0)
    save("g4_end_to_end", **out)


def g6_g7_g10():
    rng = np.random.default_rng(61)
    a1 = rng.standard_normal(6000).astype(np.float32)
    a2 = rng.standard_normal((3, 6000)).astype(np.float32)
    out = {"a1": a1, "a2": a2}
    cases = []
    k = 0
    for audio_name, audio in (("a1", a1), ("a2", a2)):
        for method in ("zerozero", "prezero", "pre"):
            for (L, hop, nfft) in ((256, 64, 256), (256, 64, 512), (1024, 256, 1024)):
                for hep in (False, True):
                    onset = 2500
                    S = data.stft(audio, onset, L, hop, nfft, hep, method)
                    out[f"S{k}"] = S
                    cases.append((audio_name, method, L, hop, nfft, int(hep), onset))
                    k += 1
    out["cases"] = np.array([repr(c) for c in cases])
    from scipy.signal import get_window
    w = get_window("hann", 256, fftbins=True)
    fr = rng.standard_normal(256).astype(np.float32)
    out["frame_x"] = fr
    out["frame_S"] = data.stft_frame(fr, 256, w)
    save("g6_stft", **out)

    audio = rng.standard_normal((5000, 4)).astype(np.float32)
    onsets = np.array([[300, 310, 305, 299], [1200, 1190, 1210, 1205], [4000, 4010, 3990, 4005]])
    fe1 = data.FrameExtractor(256, 16)
    fe2 = data.FrameExtractor(256, 16, use_min_onset=False)
    fe3 = data.FrameExtractor(128, 32, add_pre_samples=True)
    save("g7_frames", audio=audio, onsets=onsets, f1=fe1(audio, onsets), f2=fe2(audio, onsets),
         f3=fe3(audio, onsets), f1d=data.FrameExtractor(64, 8)(audio[:, 0].copy(), onsets[:, 0]))

    np.trapz = getattr(np, "trapz", np.trapezoid)
    w = get_window("hann", 1024, fftbins=True)
    save("g10_wcw", w256_64=data.window_contribution_weights(get_window("hann", 256, fftbins=True), 64),
         w1024_256=data.window_contribution_weights(w, 256),
         w1024_256_hep=data.window_contribution_weights(w, 256, True))


def g8():
    import torch
    torch.manual_seed(8)
    out = {}
    # FCNN(40 -> [10,10,10] -> 8), the C3 classifier; BatchNorm with non-trivial stats
    for name, kw, act in (("fc_a", dict(input_size=40, output_size=8), "relu"),
                          ("fc_b", dict(input_size=2, output_size=2, hidden_layers=[16, 12],
                                        activation=torch.nn.SiLU, batch_norm=False), "silu"),
                          ("fc_c", dict(input_size=14, output_size=3, hidden_layers=[32],
                                        activation=torch.nn.ELU, bias=False), "elu")):
        m = ref.calibration.FCNN(**kw)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.5)
                mod.running_var.uniform_(0.5, 2.0)
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.normal_(0, 0.2)
        m.eval()
        x = torch.randn(33, kw["input_size"])
        with torch.no_grad():
            y = m(x)
        for k, v in m.state_dict().items():
            out[f"{name}/{k}"] = v.numpy()
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/act"] = x.numpy(), y.numpy(), np.array(act)
    for name, kw in (("cnn_a", dict(input_size=256, output_size=2, channels=4)),
                     ("cnn_b", dict(input_size=64, output_size=3, channels=3, layer_sizes=[4, 6, 8],
                                    kernel_size=5, padding=2))):
        m = ref.model.CNN(**kw)
        m.eval()
        x = torch.randn(5, kw["channels"], kw["input_size"])
        with torch.no_grad():
            y = m(x)
        for k, v in m.state_dict().items():
            out[f"{name}/{k}"] = v.numpy()
        out[f"{name}/x"], out[f"{name}/y"] = x.numpy(), y.numpy()
    # CCCNN (model.py:443-538), non-grouped: shared conv stack per channel + autocorrelation head
    for name, kw in (("cccnn_a", dict(input_size=64, output_size=2, channels=3, layer_sizes=[4, 6],
                                      kernel_sizes=[3, 5], padding=1)),
                     ("cccnn_b", dict(input_size=96, output_size=3, channels=4, layer_sizes=[5],
                                      kernel_sizes=7, padding=3, activation=torch.nn.ReLU))):
        m = ref.model.CCCNN(**kw)
        m.eval()
        x = torch.randn(4, kw["channels"], kw["input_size"])
        with torch.no_grad():
            y = m(x)
        for k, v in m.state_dict().items():
            out[f"{name}/{k}"] = v.numpy()
        out[f"{name}/x"], out[f"{name}/y"] = x.numpy(), y.numpy()
    save("g8_models", **out)


def g9():
    so = ctypes.CDLL(str(REF_SO_DIR / "envelope_follower.so"))
    f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    so.backtrack_onsets.argtypes = [f32p, i64p, i64p, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long]
    rng = np.random.default_rng(9)
    N, C, B = 160, 4, 128
    t = np.arange(N)[:, None]
    buf = (np.exp((t - 150) / 9.0) * (1 + 0.2 * rng.standard_normal((N, C))) +
           0.01 * rng.random((N, C))).astype(np.float32)
    buf[:, 2] = np.maximum.accumulate(buf[:, 2])
    channels = np.array([0, 1, 2, 3, 0, 2], dtype=np.int64)
    deltas0 = np.array([120, 100, 127, 64, 5, 90], dtype=np.int64)
    outs = {}
    for k, smooth in enumerate((5, 2, 11)):
        alpha = np.float32(2 / (smooth + 1))
        tol = np.float32((1 - alpha) ** N)
        d = deltas0.copy()
        so.backtrack_onsets(buf, channels, d, alpha, tol, N, len(channels), C, B)
        outs[f"deltas_{k}"] = d
        outs[f"alpha_{k}"] = alpha
        outs[f"tol_{k}"] = tol
    save("g9_backtrack", buf=buf, channels=channels, deltas0=deltas0, B=B, **outs)


def g11():
    rng = np.random.default_rng(111)
    x = (0.1 * rng.standard_normal((3000, 3))).astype(np.float32)
    x[1000:1010] += 0.9
    out = {"x": x}
    for k, (cut, sr) in enumerate(((2000.0, 48000), (2000.0, 96000), (1000.0, 44100))):
        f = det.ButterworthFilter(cut, 3, 4, sr, "high")
        ys = []
        for i in range(0, 3000, 500):
            ys.append(f(x[i:i + 500]))
        out[f"y{k}"] = np.concatenate(ys)
        out[f"zi{k}"] = f.zi
        out[f"b{k}"], out[f"a{k}"] = f.b, f.a
        out[f"cfg{k}"] = np.array([cut, sr])
    save("g11_lfilter", **out)


def g0():
    import oracle
    a, la, v, pv = oracle.host_math_probe()
    save("g0_hostmath", log10=la, pow10=pv)


if __name__ == "__main__":
    g0()
    g1_g2()
    g3()
    g4()
    g6_g7_g10()
    g8()
    g9()
    g11()
