#!/usr/bin/env python3
"""Golden vectors for data.FastFrameExtractor (data.py:123-192), captured from the reference.

Run in the build container only:   python tests/golden/make_golden_fastframes.py
  g16_fastframes   frames for 1-D / 2-D audio and onsets, add_pre_samples both ways, and random
                   shifts (torch.manual_seed before construction AND before each call; device=None,
                   i.e. torch's CPU generator, which the mirror draws from in the same way).
Inputs are regenerated from the seed; only the reference's outputs are stored.
"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from _refload import load_reference  # noqa: E402
from make_golden_init_cfg import G16, g16_inputs  # noqa: E402

warnings.filterwarnings("ignore")
data = load_reference().data
out = {}
for name, cfg in G16.items():
    audio, onsets = g16_inputs(cfg)
    torch.manual_seed(cfg["seed"])
    fe = data.FastFrameExtractor(audio, onsets, **cfg["kw"])
    for call in range(2):
        torch.manual_seed(cfg["seed"] + 1 + call)
        out[f"{name}/call{call}"] = fe().numpy().copy()
    print(name, out[f"{name}/call0"].shape)
path = HERE / "g16_fastframes.npz"
np.savez_compressed(path, **out)
print(f"g16_fastframes: {path.stat().st_size / 1024:.1f} KiB")
