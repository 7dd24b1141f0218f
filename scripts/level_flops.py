"""Development aid: per-DAG-level Schur flop tallies of the bench workload (exact and as executed by 128/64 tiles), to be
joined with the SLUAMD_PROFILE_DUMP table of a profiled factorisation.  usage: level_flops.py N out.json"""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import driver, matgen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
fs = symb.flat_store(values=False)
xs = fs.xsup; ns_ = symb.nsupers
level = np.zeros(ns_, dtype=np.int64)
rec = {}
for k in range(ns_):
    w = xs[k + 1] - xs[k]
    li = fs.Lrowind[fs.Lrowind_off[k]:fs.Lrowind_off[k + 1]]
    ui = fs.Ufstnz[fs.Ufstnz_off[k]:fs.Ufstnz_off[k + 1]]
    rows, cols, ldu, succ = [], [], 0, []
    if len(li) >= 2:
        p = 2
        for b in range(li[0]):
            g, nr = li[p], li[p + 1]
            if g != k: rows.append(nr); succ.append(g)
            p += 2 + nr
    if len(ui) >= 3:
        p = 3
        for b in range(ui[0]):
            jb = ui[p]; wj = xs[jb + 1] - xs[jb]
            seg = xs[k + 1] - ui[p + 2:p + 2 + wj]
            cols.append(int((seg > 0).sum())); ldu = max(ldu, int(seg.max())); succ.append(jb)
            p += 2 + wj
    for j in succ: level[j] = max(level[j], level[k] + 1)
    if not rows or not cols: continue
    rows = np.array(rows); cols = np.array(cols)
    cells = float(rows.sum()) * float(cols.sum())
    t128r = np.ceil(rows / 128).sum(); t128c = np.ceil(cols / 128).sum()
    big = w >= 96 and cells / (t128r * t128c * 16384.0) >= 0.5
    tm = 128 if big else 64
    K16 = np.ceil(ldu / 16) * 16
    r = rec.setdefault(int(level[k]), dict(exact=0.0, exec_tiles=0.0, tiles=0, nodes=0, big=0, dest_bytes=0.0))
    r["exact"] += 2.0 * ldu * cells
    r["exec_tiles"] += 2.0 * K16 * (np.ceil(rows / tm).sum() * tm) * (np.ceil(cols / tm).sum() * tm)
    r["tiles"] += int(np.ceil(rows / tm).sum() * np.ceil(cols / tm).sum()); r["nodes"] += 1; r["big"] += int(big)
    r["dest_bytes"] += 16.0 * cells
json.dump(rec, open(sys.argv[2] if len(sys.argv) > 2 else "/dev/stdout", "w"), indent=0)
