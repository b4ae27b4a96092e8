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
def gpu(request):
    """Device handle for -m gpu tests.  Selected with `-m gpu` (the GPU box) a missing HIP library or
    GPU FAILS the test, so a silent fallback can never pass; in a plain `pytest tests` run on a
    machine without a GPU the gpu-marked tests are skipped instead of aborting the session."""
    import torch
    from modest_amd import _lib
    markexpr = request.config.getoption("-m") or ""
    required = ("gpu" in markexpr and "not gpu" not in markexpr) or os.environ.get("MODEST_REQUIRE_GPU") == "1"
    if not torch.cuda.is_available() and not required:
        pytest.skip("no GPU on this host (run with -m gpu on an MI355X)")
    _lib.load()
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
