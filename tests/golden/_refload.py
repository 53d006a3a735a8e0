"""Loader for the upstream reference, used ONLY by make_golden.py in the build
container (the reference cannot travel to the GPU box; tests never import this).

The reference's hot-path modules import packages that are not installed
(librosa, loopmate, lightning, audiomentations, soundfile, seaborn).  Placeholder
modules are registered in ``sys.modules`` so the imports succeed; nothing is
fetched.  Two thin delegations make ``data.stft`` runnable:
``librosa.filters.get_window -> scipy.signal.get_window`` and
``librosa.util.pad_center -> numpy.pad``.

The reference's ctypes followers look for ``envelope_follower.so`` next to
``detection.py`` (detection.py:517-519); /root/reference is read-only, so the
library is built from the reference's own C file by ``oracle/Makefile`` into
``$OFP_REF_DIR`` (default ``/tmp/ofp_ref``: outside the repository, so nothing built
from the reference's sources ever travels to the GPU box) and ``detection.__file__``
is pointed there after import.
"""
import os
import sys
import types
from pathlib import Path

import numpy as np
import scipy.signal

REPO = Path(__file__).resolve().parents[2]
REF_ROOT = Path("/root/reference")
REF_SO_DIR = Path(os.environ.get("OFP_REF_DIR", "/tmp/ofp_ref"))


def _pad_center(data, *, size, axis=-1, **kwargs):
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, **kwargs)


def _install_placeholders():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "librosa" not in sys.modules:
        librosa = mod("librosa")
        librosa.filters = mod("librosa.filters")
        librosa.util = mod("librosa.util")
        librosa.feature = mod("librosa.feature")
        librosa.filters.get_window = (
            lambda window, Nx, fftbins=True: scipy.signal.get_window(
                window, Nx, fftbins=fftbins
            )
        )
        librosa.util.pad_center = _pad_center
    if "loopmate" not in sys.modules:
        loopmate = mod("loopmate")
        ca = mod("loopmate.circular_array")

        class CircularArray:  # placeholder: backtrack=True path is not run
            def __init__(self, *a, **k):
                raise RuntimeError("loopmate is not available")

        ca.CircularArray = CircularArray
        loopmate.circular_array = ca
    class _Anything(types.ModuleType):
        # module-level constants in data.py instantiate augmentation objects
        def __getattr__(self, attr):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return lambda *a, **k: None

    for name in ("audiomentations", "soundfile", "seaborn"):
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    if "lightning" not in sys.modules:
        import torch

        L = mod("lightning")
        L.LightningModule = torch.nn.Module


def load_reference():
    """Returns the reference package modules (detection, data, model, calibration)."""
    if not (REF_SO_DIR / "envelope_follower.so").exists():
        raise RuntimeError("run `make -C oracle ref` first")
    _install_placeholders()
    if str(REF_ROOT) not in sys.path:
        sys.path.insert(0, str(REF_ROOT))
    sys.dont_write_bytecode = True
    from onset_fingerprinting import calibration, data, detection, model

    detection.__file__ = str(REF_SO_DIR / "detection.py")
    return types.SimpleNamespace(
        detection=detection, data=data, model=model, calibration=calibration
    )
