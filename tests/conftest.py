import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def seeded_rand(shape, seed, lo=0.0, hi=1.0):
    """Input recipe shared with oracle/gen_golden.py::rand (pinned by sha256 digests in the fixtures)."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.rand(tuple(int(s) for s in shape), generator=g, dtype=torch.float32) * (hi - lo) + lo


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")
