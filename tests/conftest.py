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


class _Options:
    """The library's options (hy_set_option) for the duration of one test: tests force a path, then compare it with the oracle."""

    def __init__(self):
        self.saved = {}

    def set(self, option_id, value):
        from hyrise_amd import abi
        lib = abi.load_library()
        if option_id not in self.saved:
            before = C.c_int64(0)
            abi.check(lib.hy_get_option(option_id, C.byref(before)))
            self.saved[option_id] = before.value
        abi.check(lib.hy_set_option(option_id, value))

    def reset(self, option_id=None):
        from hyrise_amd import abi
        lib = abi.load_library()
        for key in ([option_id] if option_id is not None else list(self.saved)):
            if key in self.saved:
                abi.check(lib.hy_set_option(key, self.saved.pop(key)))


@pytest.fixture
def options():
    o = _Options()
    yield o
    o.reset()
