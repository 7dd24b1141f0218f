#!/usr/bin/env python3
"""Development aid: host wall time of the three calls of a bench step (reset_values / pdgstrf3d / pdgstrs3d) next to the HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import _lib, driver, matgen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = _lib.load()
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
h = driver.LUHandle.from_symbolic(symb, v)
xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
thresh = driver.pivot_thresh(n, rp, ci, np.abs(v))
h.pdgstrf3d(thresh); h.pdgstrs3d(xp)
if len(sys.argv) > 2:
    h.reset_values(); h.pdgstrf3d(thresh); h.pdgstrs3d(xp)
for it in range(4):
    L.sluamd_device_synchronize()
    t0 = time.perf_counter(); h.reset_values(); t1 = time.perf_counter()
    L.sluamd_device_synchronize(); t1s = time.perf_counter()
    h.pdgstrf3d(thresh); t2 = time.perf_counter()
    y = h.pdgstrs3d(xp); t3 = time.perf_counter()
    st = h.stats()
    print("reset call %.2f ms (+sync %.2f) | factor call %.2f ms (events %.2f) | solve call %.2f ms (events %.2f) | total %.2f" % (
        1e3 * (t1 - t0), 1e3 * (t1s - t1), 1e3 * (t2 - t1s), st["t_factor_ms"], 1e3 * (t3 - t2), st["t_solve_ms"], 1e3 * (t3 - t0)))
