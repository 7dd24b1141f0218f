#!/usr/bin/env python3
"""Debug: phase breakdown of k_diag_lu2<256> over one factorisation of the bench workload (serial mode)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import _lib, driver, matgen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
L = _lib.load()
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
h = driver.LUHandle.from_symbolic(symb, v)
out = (C.c_ulonglong * 8)()
os.environ["SLUAMD_NO_LOOKAHEAD"] = "1"
h.pdgstrf3d(0.0); h.reset_values()
L.sluamd_debug_diag_profile(out, 1)
h.pdgstrf3d(0.0)
L.sluamd_debug_diag_profile(out, 0)
tot = sum(out[:5]) or 1
names = ["A col-panel update", "B head LU", "C head inverses", "D L21 + store", "E row panel"]
nblk = sum(1 for w in np.diff(symb.xsup()) if w > 128)
print(f"supernodes wider than 128: {nblk}; ticks per such block (100 MHz clock -> x10 ns):")
for nm, t in zip(names, out[:5]):
    print(f"  {nm:22s} {t:12d} ticks  {100.0 * t / tot:5.1f} %   {t / max(nblk, 1) / 100.0:8.1f} us/block")
