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


def test_replaced_allocation_operators_stay_inside_the_library():
    """sluamd_alloc.cpp replaces operator new / delete for the library's OWN host tables (2 MB-aligned, MADV_HUGEPAGE blocks); the link's version script
    (csrc/sluamd.map) must keep them local -- a drop-in library that exported them would replace the host application's allocator."""
    import shutil, subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    lib = os.path.join(ROOT, "superlu_dist_amd", "libsluamd.so")
    if not os.path.exists(lib) or not (shutil.which("nm") or os.path.exists(nm)):
        pytest.skip("library or nm not present")
    dyn = subprocess.run([nm, "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    leaked = [ln for ln in dyn.splitlines() if re.search(r"\b_Z(nw|na|dl|da)[mP]", ln)]
    assert not leaked, leaked
    local = subprocess.run([nm, lib], capture_output=True, text=True, check=True).stdout
    assert re.search(r" t _Znwm$", local, flags=re.M), "the library's own operator new is missing (sluamd_alloc.cpp not linked?)"
