import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()      # (does not initialise the GPU)
    except Exception:
        return 0


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """On a GPU box the suite runs on a few worker processes (pytest-xdist):
    its time is host set-up of the problems (numpy / scipy / the OpenMP
    helpers), which parallelises across tests, while the kernels of several
    processes share the one GPU.  `python -m pytest tests -m gpu` stays the
    command; `-n 0`, `-p no:xdist` or FENAPACK_AMD_TEST_WORKERS=0 give the
    serial run.  Without xdist, without a GPU, or with an explicit `-n`
    nothing changes."""
    opt = config.option
    if not hasattr(opt, "numprocesses") or opt.numprocesses is not None:
        return None
    if os.environ.get("PYTEST_XDIST_WORKER") or _gpus() < 1:
        return None
    want = os.environ.get("FENAPACK_AMD_TEST_WORKERS")
    n = int(want) if want is not None else min(4, max(1, (os.cpu_count() or 1) // 8))
    if n > 1:
        opt.numprocesses = n
        # a worker's OpenMP helpers take their share of the cores
        os.environ.setdefault("FENAPACK_AMD_HOST_THREADS",
                              str(max(2, min(32, (os.cpu_count() or 8) // (2 * n)))))
    return None


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests must exercise it, never a fallback."""
    from fenapack_amd._cabi import hip_library
    return hip_library()
