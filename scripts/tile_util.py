"""Host-side estimate of Schur tile utilisation: useful flops vs flops of the (128- or 64-) tiles launched."""
import sys, numpy as np
sys.path.insert(0, '.')
from superlu_dist_amd import matgen, driver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
s = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
fs = s.flat_store(values=False)
xs = fs.xsup; sz = np.diff(xs)
useful = 0.0; tiled = 0.0; tiles128 = 0; tiles64 = 0; kpad = 0.0
hist = {}
for k in range(fs.nsupers):
    li = fs.Lrowind[fs.Lrowind_off[k]:fs.Lrowind_off[k + 1]]
    nb = li[0]; p = 2; rows = []
    for b in range(nb):
        if li[p] != k: rows.append(li[p + 1])
        p += 2 + li[p + 1]
    rows = np.array(rows)
    if rows.size == 0: continue
    cols = rows  # symmetric pattern
    ns = sz[k]
    R = rows.sum()
    t128 = np.ceil(rows / 128).sum()
    util128 = R * R / (t128 * t128 * 128.0 * 128.0)
    big = ns >= 96 and util128 >= 0.5
    tm = 128 if big else 64
    nt = np.ceil(rows / tm).sum()
    useful += 2.0 * ns * R * R
    kp = np.ceil(ns / 16) * 16
    tiled += 2.0 * kp * (nt * tm) ** 2
    if big: tiles128 += nt * nt
    else: tiles64 += nt * nt
    key = (tm, int(ns) // 32 * 32)
    h = hist.setdefault(key, [0.0, 0.0]); h[0] += 2.0 * ns * R * R; h[1] += 2.0 * kp * (nt * tm) ** 2
print(f"N={N} nsupers={fs.nsupers} useful {useful:.3e} tiled {tiled:.3e} util {useful/tiled:.3f} tiles128 {tiles128:.3e} tiles64 {tiles64:.3e}")
for key in sorted(hist):
    h = hist[key]
    print(key, f"useful {h[0]:.3e} tiled {h[1]:.3e} util {h[0]/h[1]:.3f}")
print("supernode size histogram (>=128):", np.bincount(sz[sz >= 128] // 16 * 16)[128:].nonzero()[0] + 128, np.bincount(sz[sz>=128]//16*16)[128:][np.bincount(sz[sz>=128]//16*16)[128:].nonzero()[0]])
