import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases(fname):
    """npz with keys '<case>__<field>' -> {case: {field: array}}."""
    z = np.load(os.path.join(GOLDEN, fname))
    out = {}
    for k in z.files:
        case, field = k.split("__")
        out.setdefault(case, {})[field] = z[k]
    return out


@pytest.fixture(scope="session")
def reference_module():
    """The reference's own CPU ROIAlign / NMS sources compiled where they lie (oracle/build_ref.py) — in the build
    container only; None on the GPU box, where the committed vectors under tests/golden/ are the pin.  The built
    files are removed when the session ends: no compiled reference code stays in the tree (or travels with it)."""
    from oracle.build_ref import build, clean, load_prebuilt

    if not os.path.isdir("/root/reference"):
        yield None
        return
    m = load_prebuilt() or build()
    yield m
    clean()


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
