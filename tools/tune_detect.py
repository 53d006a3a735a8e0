"""Detector-only timing of a batch under several ofp_detect_tuning settings (data generated once).

    python tools/tune_detect.py c4|c2x8|c3s '<json list of tuning dicts>'

c4: 512 clips x 4 ch x 10 s; c2x8: 8 clips x 8 ch x 60 s; c3s: 1 clip x 64 ch x 60 s (a tenth of C3); c3: all of it.
Every setting gives the same bytes (checked against the first one); prints ms per call and the stages."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.argv[0] = str(Path(__file__).resolve().parents[1] / "bench.py")
import bench  # noqa: E402  (synth_batch: parallel host synthesis)
from onset_fingerprinting_amd import detection, synth  # noqa: E402


def main():
    kind = sys.argv[1]
    tunings = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [{}]
    if kind == "c4":
        xs = bench.synth_batch("c4", range(512), 10.0, 4, 16)
        B = 256
    elif kind == "c2x8":
        xs = bench.synth_batch("c2", [1 + 7919 * i for i in range(8)], 60.0, 8, 8)
        B = 256
    elif kind == "c3":
        xs = [synth.c3_stream(600.0, 64, 48000, seed=2)]   # BASELINE config C3 at full size
        B = 512
    else:
        xs = [synth.c3_stream(60.0, 64, 48000, seed=2)]
        B = 512
    x = torch.from_numpy(np.stack(xs)).cuda().contiguous()
    C = x.shape[2]
    ref = None
    for t in tunings:
        bd = detection.BatchDetector(C, B, sr=48000)
        if t:
            bd.set_tuning(**t)
        out = bd.detect(x, cap_per_clip=4096)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = bd.detect(x, out=out, cap_per_clip=4096)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        n = out["counts"].cpu().numpy()
        sig = (int(n.sum()), int(out["rel"].view(torch.int32).sum(dtype=torch.int64).item()),
               int(out["records"].view(torch.int64)[..., 1].sum().item() if False else 0))
        if ref is None:
            ref = (sig, out["rel"].clone(), out["counts"].clone())
        same = bool(torch.equal(out["rel"], ref[1]) and torch.equal(out["counts"], ref[2]))
        i = bd.last_info
        print(json.dumps(dict(tuning=t, ms=round(min(ts) * 1e3, 2), same_bytes=same,
                              stage_ms={k: round(v, 2) for k, v in i["stage_ms"].items()},
                              passes=[i["hp_passes"], i["ar_passes"], i["mm_passes"], i["repaired"]])), flush=True)
        del bd


if __name__ == "__main__":  # (the synthesis pool spawns: the module must be importable without side effects)
    main()
