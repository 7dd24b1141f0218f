"""Per-level traffic / time table of one pdgstrs3d of the bench workload: joins the level structure (bytes of L, U and the inverses per DAG
level, from the symbolic structure) with the kernel durations of scripts/solve_timeline.py's output (profiles/rNN_solve_timeline_final.txt).
Launch pattern of the 1 x 1 layer sweeps (solve_fwd_links / solve_bwd_links): forward  sweep(l) = diagonal solves of level l + far updates of
level l-1, update(l) = near updates of level l;  backward: update(l) = U(k,:) x of level l, sweep(l) = diagonal solves of level l + far chunks of l-1.
usage: solve_levels.py N profiles/r03_solve_timeline_final.txt > profiles/r03_solve_levels.txt"""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import driver, matgen
N = int(sys.argv[1]); tlf = sys.argv[2]
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
fs = symb.flat_store(values=False)
xs = fs.xsup; ns_ = symb.nsupers
level = np.zeros(ns_, dtype=np.int64)
Lb, Ub, Db, cnt, wmax = {}, {}, {}, {}, {}
for k in range(ns_):
    w = int(xs[k + 1] - xs[k])
    li = fs.Lrowind[fs.Lrowind_off[k]:fs.Lrowind_off[k + 1]]
    ui = fs.Ufstnz[fs.Ufstnz_off[k]:fs.Ufstnz_off[k + 1]]
    succ = []; rows = 0; useg = 0
    if len(li) >= 2:
        p = 2
        for b in range(li[0]):
            g, nr = int(li[p]), int(li[p + 1])
            if g != k: rows += nr; succ.append(g)
            p += 2 + nr
    if len(ui) >= 3:
        p = 3
        for b in range(ui[0]):
            jb = int(ui[p]); wj = int(xs[jb + 1] - xs[jb])
            seg = xs[k + 1] - ui[p + 2:p + 2 + wj]
            useg += int(seg.sum()); succ.append(jb)
            p += 2 + wj
    for j in succ: level[j] = max(level[j], level[k] + 1)
    l = int(level[k])
    Lb[l] = Lb.get(l, 0) + rows * w * 8; Ub[l] = Ub.get(l, 0) + useg * 8; Db[l] = Db.get(l, 0) + w * w * 8
    cnt[l] = cnt.get(l, 0) + 1; wmax[l] = max(wmax.get(l, 0), w)
nl = max(Lb) + 1
tl = [ln.split() for ln in open(tlf) if re.match(r"^\d+\s", ln)]
fw = tl[:2 * nl - 1]; bw = tl[2 * nl - 1:]
print(f"# {N}^3: {nl} levels, {len(tl)} launches; L {sum(Lb.values()) / 1e9:.2f} GB, U {sum(Ub.values()) / 1e9:.2f} GB, inverses 2 x {sum(Db.values()) / 1e9:.2f} GB")
print("# forward sweep.  Row l = the launches between the diagonal solves of level l and those of level l+1: update(l) [near updates of level l] + sweep(l+1)")
print("# [far updates of level l + Linv GEMVs of level l+1];  bytes = L(l) + Linv(l+1)")
print("#   level nodes max_width |   L MB  Linv(l+1) MB | update us  sweep(l+1) us |  GB/s")
ts0 = float(fw[0][2])
print(f"F  -1 {'':>6} {'':>4} | {0.0:8.1f} {Db[0] / 1e6:7.1f} | {0.0:6.1f} {ts0:7.1f} | {Db[0] / 1e3 / ts0:7.0f}   (Linv GEMVs of level 0)")
tf = ts0
for l in range(nl):
    tu = float(fw[2 * l + 1][2]) if 2 * l + 1 < len(fw) else 0.0
    ts = float(fw[2 * l + 2][2]) if 2 * l + 2 < len(fw) else 0.0
    by = Lb[l] + (Db[l + 1] if l + 1 < nl else 0)
    tf += tu + ts
    if tu + ts > 0: print(f"F {l:3d} {cnt[l]:6d} {wmax[l]:4d} | {Lb[l] / 1e6:8.1f} {(Db[l + 1] if l + 1 < nl else 0) / 1e6:7.1f} | {tu:6.1f} {ts:7.1f} | {by / 1e3 / (tu + ts):7.0f}")
print(f"# forward total {tf:.0f} us")
print("# backward sweep (top level first).  Row l: sweep(l+1) [Uinv GEMVs of level l+1 + far chunks of U(l)] + update(l) [near chunks of U(l)];  bytes = U(l) + Uinv(l+1)")
print("#   level nodes max_width |   U MB  Uinv(l+1) MB | sweep(l+1) us  update us |  GB/s")
tb = 0.0
seq = [float(r[2]) for r in bw]     # sweep(top), update(top-1), sweep(top-1), update(top-2), ...
for idx, l in enumerate(range(nl - 2, -1, -1)):
    ts = seq[2 * idx] if 2 * idx < len(seq) else 0.0
    tu = seq[2 * idx + 1] if 2 * idx + 1 < len(seq) else 0.0
    by = Ub[l] + Db[l + 1]
    tb += ts + tu
    print(f"B {l:3d} {cnt[l]:6d} {wmax[l]:4d} | {Ub[l] / 1e6:8.1f} {Db[l + 1] / 1e6:7.1f} | {ts:7.1f} {tu:6.1f} | {by / 1e3 / max(ts + tu, 1e-9):7.0f}")
tlast = seq[-1] if len(seq) % 2 == 1 else 0.0
print(f"B  -1 {'':>6} {'':>4} | {0.0:8.1f} {Db[0] / 1e6:7.1f} | {tlast:7.1f} {0.0:6.1f} | {Db[0] / 1e3 / max(tlast, 1e-9):7.0f}   (Uinv GEMVs of level 0)")
print(f"# backward total {tb + tlast:.0f} us")
