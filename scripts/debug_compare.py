"""Debug helper: factor on the GPU and with the oracle, report per-supernode mismatches (diag / L / U)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
from superlu_dist_amd import matgen, driver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
leaf, relax, maxsup = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 64, 256)
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
symb.distribute_host(v)
fs = symb.flat_store()
o = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz, fs.Unzval_off, fs.Unzval)
h = driver.LUHandle.from_store(fs)
info = h.pdgstrf3d(0.0)
h.copy_to_host()
orc.dfactor(o)
print("info", info, "nsupers", fs.nsupers)
bad = 0
for k in range(fs.nsupers):
    ns = fs.xsup[k + 1] - fs.xsup[k]
    li = fs.Lrowind[fs.Lrowind_off[k]:fs.Lrowind_off[k + 1]]
    nsupr = li[1]
    a = fs.Lnzval[fs.Lnzval_off[k]:fs.Lnzval_off[k + 1]].reshape((nsupr, ns), order="F")
    b = o.Lnzval[o.Lnzval_off[k]:o.Lnzval_off[k + 1]].reshape((nsupr, ns), order="F")
    with np.errstate(invalid="ignore"):
        ed = np.nanmax(np.abs(a[:ns] - b[:ns])) if not np.isnan(a[:ns]).any() else np.inf
        el = (np.nanmax(np.abs(a[ns:] - b[ns:])) if nsupr > ns else 0.0) if not np.isnan(a[ns:]).any() else np.inf
        ua = fs.Unzval[fs.Unzval_off[k]:fs.Unzval_off[k + 1]]; ub = o.Unzval[o.Unzval_off[k]:o.Unzval_off[k + 1]]
        eu = (np.abs(ua - ub).max() if len(ua) else 0.0) if not np.isnan(ua).any() else np.inf
    if max(ed, el, eu) > 1e-10:
        print(f"k={k} fst={fs.xsup[k]} ns={ns} nsupr={nsupr} nblk={li[0]} err diag={ed:.2e} L={el:.2e} U={eu:.2e}")
        if ed > 1e-10 and ed != np.inf:
            d = np.abs(a[:ns] - b[:ns]); ij = np.unravel_index(np.argmax(d), d.shape); print("   first bad diag entries: argmax", ij, "cols with err", np.where(d.max(axis=0) > 1e-10)[0][:10], "rows", np.where(d.max(axis=1) > 1e-10)[0][:10])
        bad += 1
        if bad > 8:
            break
print("bad supernodes:", bad)
