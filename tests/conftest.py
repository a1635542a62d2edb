import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_inputs():
    g = np.load(os.path.join(GOLDEN, "golden_inputs.npz"))
    return {k: g[k] for k in g.files}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))["out"]
