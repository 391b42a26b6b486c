import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# The oracles work on many small tensors; OpenMP fork/join dominates them with the default thread count (measured in the
# build container: one 256x192 oracle frame 0.8 s single-threaded, 5-30 s with 4-8 threads).
torch.set_num_threads(int(os.environ.get("ADK_TEST_THREADS", "1")))

import artdeco_amd  # noqa: E402

artdeco_amd.install_dropins()

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand in the CPU container; prebuilt on the GPU box)."""
    from artdeco_amd import _lib, build
    try:
        build.build()  # incremental: a no-op when the prebuilt .so is newer than its sources
    except RuntimeError:
        if not os.path.exists(_lib.LIB_PATH):
            raise
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
