import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    from oracle import c_oracle
    c_oracle.build_for_host()     # before any test initialises the GPU


@pytest.fixture(scope="session")
def topologies():
    out = {}
    for name in ("cora", "citeseer", "pubmed"):
        z = np.load(os.path.join(GOLDEN, f"{name}_csr.npz"), allow_pickle=False)
        out[name] = (z["rowptr"], z["col"])
    return out
