import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    """Device handle for -m gpu tests; fails (not skips) when the HIP library
    or the GPU is missing, so a silent fallback can never pass."""
    import torch
    from modest_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
