import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def emu_backend():
    """Host-emulated build of the HIP kernels (tests/emu). CPU tests only; never on the product path."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from transfuser_amd import _lib
    path = build_emu.build()
    _lib._install_test_backend(ctypes.CDLL(path))
    return _lib
