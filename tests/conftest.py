import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the reference itself (tests/golden/make_golden.py)."""
    with np.load(REPO / "tests" / "golden" / "codec_golden.npz") as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def fullgrad_golden():
    """Full-shape (B=2, T=375) gradient vectors produced by the reference itself (tests/golden/make_golden_fullgrad.py)."""
    with np.load(REPO / "tests" / "golden" / "fullgrad_golden.npz") as z:
        return {k: z[k] for k in z.files}
