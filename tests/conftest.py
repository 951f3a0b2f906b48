"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """No GPU visible (the CPU run of the suite: kernels on the single-threaded SIMT emulator, ~19 minutes in one process): spread
    the tests over the cores with pytest-xdist, as if `-n <cores>` had been given -- 4 to 5 minutes on 8 cores.  Not on a GPU
    box (one process there, so that the library the tests load is the one the driver sees), not when the caller chose `-n`
    itself, not inside an xdist worker; PPSCI_TEST_SERIAL=1 switches it off."""
    opt = config.option
    if (os.environ.get("PPSCI_TEST_SERIAL") == "1" or os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput")
            or not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", None) is not None
            or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False) or _has_gpu()):
        return None
    n = min(8, os.cpu_count() or 1)
    if n > 1:
        opt.numprocesses = n       # (xdist's own pytest_cmdline_main turns this into dist = "load", tx = n x popen;
        opt.dist = "load"          #  set here as well in case its hook has already run)
        opt.tx = ["popen"] * n
    return None


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip_gpu)


@pytest.fixture(autouse=True)
def _poison_free_device_memory(request):
    """GPU tests: fill the caching allocator's free blocks with NaN before every test, so that any kernel that reads a
    buffer it was supposed to have written first (torch.empty hands out recycled memory) fails loudly instead of passing
    on leftover zeros of a fresh process."""
    if "gpu" in request.keywords and _has_gpu():
        import torch

        # large pool (>= 1 MB blocks) and small pool (< 1 MB): both are recycled by torch.empty
        blocks = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(4)]  # 4 x 256 MB
        blocks += [torch.full((n,), float("nan"), device="cuda") for n in (1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10, 256)
                   for _ in range(64)]
        del blocks
    yield
