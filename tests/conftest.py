import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """Opt-in (TF_TEST_PARALLEL=1) for a CPU box: the host-emulated kernel / model tests are single-threaded fibers, so the suite can be spread
    over a few xdist workers (16 min -> ~5).  Off by default: a pinning test that shares a worker with interleaved emulator tests has failed
    under it (test_point_pillars_match_reference_golden), and a serial run is the reference behaviour.  Never on a GPU box."""
    import torch
    if os.environ.get("TF_TEST_PARALLEL") != "1" or torch.cuda.is_available() or os.environ.get("PYTEST_XDIST_WORKER"):
        return
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) or getattr(config.option, "collectonly", False):
        return
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()                      # once, before the workers start (they would race on the object files)
    config.option.numprocesses = min(4, max(1, (os.cpu_count() or 2) // 2))
    config.option.dist = "loadgroup"     # see pytest_collection_modifyitems: only the emulator files are spread test by test


def pytest_collection_modifyitems(config, items):
    """xdist grouping: the pinning / dataprep / distributed files import the reference through sys.path / sys.modules stand-ins and rely on
    their in-file order, so each of them stays on ONE worker; the host-emulated kernel and model tests are independent."""
    for item in items:
        fname = os.path.basename(str(item.fspath))
        if fname not in ("test_model_emu.py", "test_kernels_emu.py"):
            item.add_marker(pytest.mark.xdist_group(fname))


@pytest.fixture(scope="session")
def emu_backend():
    """Host-emulated build of the HIP kernels (tests/emu). CPU tests only; never on the product path."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from transfuser_amd import _lib
    # TF_EMU_ASAN=1: the AddressSanitizer build of the emulated kernels (run pytest under LD_PRELOAD=<clang's libclang_rt.asan-x86_64.so> with
    # ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0): out-of-bounds reads / writes of the kernels show up on the CPU
    path = build_emu.build(asan=os.environ.get("TF_EMU_ASAN") == "1")
    _lib._install_test_backend(ctypes.CDLL(path))
    if os.environ.get("PYTEST_XDIST_WORKER"):
        import torch
        torch.set_num_threads(2)           # several workers share the host
    return _lib
