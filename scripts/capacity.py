"""Capacity planning from the symbolic structure alone (no GPU): nnz(L+U), flops and the per-rank storage of the factors on
process grids -- with the ancestor panels replicated along Z as the 3D algorithm keeps them (dinit3DLUstructForest,
pd3dcomm.c:334-800) -- for the N^3 7-point Poisson family of BASELINE.json.  usage: capacity.py N [N ...]  > profiles/..."""
import os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import matgen, driver

HBM = 288e9
for N in [int(a) for a in sys.argv[1:]]:
    t0 = time.time()
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    t1 = time.time()
    s = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
    t2 = time.time()
    print(f"N={N} n={n} nnz(A)={len(v)} nsupers={s.nsupers} nnz(L+U)={s.nnzL + s.nnzU} = {(s.nnzL + s.nnzU) * 8 / 1e9:.1f} GB  flops={s.flops:.4e}  "
          f"matgen {t1 - t0:.0f} s, symbolic {t2 - t1:.0f} s, host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB", flush=True)
    for g in [(1, 1, 1), (1, 1, 2), (1, 2, 2), (2, 2, 2), (2, 2, 4), (2, 4, 1)]:
        t = s.partition(g[2]) if g[2] > 1 else None
        vals, rep, idx = s.grid_footprint(*g, t)
        worst = int(vals.max())
        print(f"  grid {g[0]}x{g[1]}x{g[2]}: values per rank max {worst * 8 / 1e9:7.1f} GB  mean {vals.mean() * 8 / 1e9:7.1f} GB  (replicated along Z: max {rep.max() * 8 / 1e9:6.1f} GB, "
              f"total {rep.sum() * 8 / 1e9:7.1f} GB = {100.0 * rep.sum() / max(1, vals.sum() - rep.sum()):4.1f} % of the factors)  index entries per rank max {idx.max():.3e} "
              f"({'fits' if worst * 8 * 1.12 < HBM else 'DOES NOT FIT'} 288 GB with 12 % for scratch / tables)", flush=True)
    s.free()
