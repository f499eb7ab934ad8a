import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "yolo-compression-and-deployment-in-fpga_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    g = {}
    for f in ("prep_layers.npz", "e2e.npz", "guard.npz", "r3.npz"):
        with np.load(os.path.join(ROOT, "tests", "golden", f)) as z:
            for k in z.files:
                g[k] = z[k]
    return g
