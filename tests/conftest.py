import os, sys
import pytest

# the CPU oracle's OpenMP loops are tiny: cap its threads (must be set before libgomp starts)
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]
    return load


@pytest.fixture()
def emul():
    """Bind the ctypes layer to oracle/libsluamd_emul.so (the library's HOST sources + a serial CPU restatement of the
    kernels: test infrastructure, built by `make -C oracle`) for the duration of one test."""
    import ctypes, subprocess
    from superlu_dist_amd import _lib
    so = os.path.join(ROOT, "oracle", "libsluamd_emul.so")
    if not getattr(emul, "_made", False):
        # once per session, and not only when the file is missing: a stale CPU test build would test yesterday's host sources (make is a
        # no-op when the library is newer than every source it is built from)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libsluamd_emul.so"])
        emul._made = True
    saved = _lib._lib
    _lib._lib = _lib.bind(ctypes.CDLL(so))
    yield _lib._lib
    _lib._lib = saved
