import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    """Initialised HIP device; GPU tests fail loudly (no skip, no fallback) when the library or device is missing."""
    from hyrise_amd import abi
    try:   # tests that hand torch tensors to the library: PyTorch brings its own HIP runtime, which wants to find the device first
        import torch
        torch.cuda.init()
    except Exception:
        pass
    lib = abi.load_library()
    abi.check(lib.hy_init(0))
    yield lib
    lib.hy_shutdown()
