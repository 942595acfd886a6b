import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run under gpurun)")


@pytest.fixture(scope="session")
def orc():
    """The C oracle (oracle/pvn3d_oracle.c), built on demand with gcc."""
    from oracle import native
    native.build()
    return native


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
