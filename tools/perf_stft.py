"""Stand-alone timing of the STFT kernel variants (nothing else on the GPU): frames/s and the share of the
HBM peak for the bytes each variant must move.   python tools/perf_stft.py [n_fft=1024] [hop=256]"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from onset_fingerprinting_amd.data import (MelBank, stft_power_dense, stft_power_mel_dense,  # noqa: E402
                                           stft_power_mel_mlp_dense)
from onset_fingerprinting_amd.pipeline import seeded_fcnn  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    hop = int(sys.argv[2]) if len(sys.argv) > 2 else F // 4
    n_clips, C, N = 64, 8, 480000
    torch.manual_seed(0)
    x = torch.randn((n_clips, N, C), device="cuda")
    xp = x.permute(0, 2, 1).contiguous()  # planar [clip][C][N]
    planar = (xp.data_ptr(), N)
    mb = MelBank(48000, F, 40)
    mlp = seeded_fcnn(40, 8).device_mlp(0)
    H = 1 + (N - F) // hop
    frames = n_clips * C * H
    bins = F // 2 + 1
    P = torch.empty((n_clips, C, H, bins), device="cuda")
    M = torch.empty((n_clips, C, H, 40), device="cuda")
    L = torch.empty((n_clips, C, H, 8), device="cuda")
    variants = {
        "power (interleaved in)": (lambda: stft_power_dense(x, F, hop, out=P), 4 * hop + 4 * bins),
        "power+mel (planar in)": (lambda: stft_power_mel_dense(x, F, hop, mb, out_power=P, out_mel=M, planar=planar), 4 * hop + 4 * bins + 160),
        "mel only (planar in)": (lambda: stft_power_mel_dense(x, F, hop, mb, out_mel=M, want_power=False, planar=planar), 4 * hop + 160),
        "power+mel+mlp (interleaved in)": (lambda: stft_power_mel_mlp_dense(x, F, hop, mb, mlp, out_power=P, out_mel=M, out_logits=L, want_power=True), 4 * hop + 4 * bins + 192),
        "power+mel+mlp (planar in)": (lambda: stft_power_mel_mlp_dense(x, F, hop, mb, mlp, out_power=P, out_mel=M, out_logits=L, want_power=True, planar=planar), 4 * hop + 4 * bins + 192),
        "mel+mlp, no power (planar in)": (lambda: stft_power_mel_mlp_dense(x, F, hop, mb, mlp, out_mel=M, out_logits=L, want_power=False, planar=planar), 4 * hop + 192),
        "logits only (planar in)": (lambda: stft_power_mel_mlp_dense(x, F, hop, mb, mlp, out_logits=L, want_power=False, want_mel=False, planar=planar), 4 * hop + 32),
    }
    for name, (fn, bpf) in variants.items():
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(json.dumps({"variant": name, "n_fft": F, "hop": hop, "frames": frames, "ms": round(t * 1e3, 3),
                          "Mframes_per_s": round(frames / t / 1e6, 1), "algorithmic_GBps": round(bpf * frames / t / 1e9, 1),
                          "hbm_frac": round(bpf * frames / t / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
