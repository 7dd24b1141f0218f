"""The C-ABI library loads on a CPU-only box and exports every symbol include/superlu_dist_amd.h declares."""
import ctypes, os, re
import pytest
from superlu_dist_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "superlu_dist_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sluamd_[A-Za-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol():
    L = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in the header but not exported"
    for s in _lib.EXPORTS:
        assert s in syms


def test_no_cpu_fallback_without_device():
    import numpy as np
    L = _lib.load()
    if L.sluamd_device_count() > 0:
        pytest.skip("GPU present")
    from superlu_dist_amd import matgen, driver
    n, rp, ci, v = matgen.poisson3d(3)
    with pytest.raises(RuntimeError, match="no HIP device|failed"):
        driver.pdgssvx3d(n, rp, ci, v, np.ones(n))
