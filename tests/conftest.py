import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a BASELINE configuration at its full stated size (minutes of host-side synthesis and oracle time; part of -m gpu, deselect with -m 'gpu and not slow')")


def load_golden(name):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
