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
    """The reference's own CPU ROIAlign build (oracle/_ref), or None when not present."""
    from oracle.build_ref import build, load_prebuilt

    m = load_prebuilt()
    if m is None and os.path.isdir("/root/reference"):
        m = build()
    return m


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
