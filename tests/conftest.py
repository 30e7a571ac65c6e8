import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run via gpurun)")


def load_cases(name):
    """Load a golden .npz written by tests/golden/make_golden.py as a list of dicts."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    n = int(z["n_cases"])
    cases = [dict() for _ in range(n)]
    for key in z.files:
        if key == "n_cases":
            continue
        idx, field = key.split(":", 1)
        cases[int(idx[1:])][field] = z[key]
    return cases


def scalar(x):
    return x.item() if hasattr(x, "item") else x


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _deterministic_torch_rng(request):
    """Every test starts from the same torch generator state (CPU and, when present, GPU): a few tests draw inputs
    or module initialisations from the global generator, and a tolerance check must not depend on the draw of the day."""
    import zlib

    import torch

    torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF)
    yield
