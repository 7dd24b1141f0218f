"""Per-level traffic / time table of one pdgstrs3d of the bench workload with the JOINED links of round 4 (solve_fwd_join / solve_bwd_join): joins the level
structure (bytes of L, U and the inverses per DAG level, from the symbolic structure) with the kernel durations of scripts/solve_timeline.py's output.
Launch pattern: forward -- level 0's diagonal blocks, then per level l the launch(es) that apply the panels of level l and solve the diagonal blocks of level
l + 1: ONE (k_sweep_join) when level l + 1 holds <= JOIN_MAX supernodes, else two (k_fwd_update: near updates, k_sweep: strips + far updates); backward the
mirror image, top level first.  usage: solve_levels_join.py N profiles/r04_solve_timeline_final.txt [JOIN_MAX=32] > profiles/r04_solve_levels.txt"""
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
JM = int(sys.argv[3]) if len(sys.argv) > 3 else 32
tl = [ln.split() for ln in open(tlf) if re.match(r"^\d+\s", ln)]
dur = [float(r[2]) for r in tl]
names = [" ".join(r[4:]) for r in tl]
joined = lambda m: cnt[m] <= JM
print(f"# {N}^3: {nl} levels, {len(tl)} launches; L {sum(Lb.values()) / 1e9:.2f} GB, U {sum(Ub.values()) / 1e9:.2f} GB, inverses 2 x {sum(Db.values()) / 1e9:.2f} GB; levels of <= {JM} supernodes joined")
print("# forward sweep.  Row l = the launch(es) that apply the panels of level l and solve the diagonal blocks of level l + 1;  bytes = L(l) + Linv(l+1)")
print("#   level nodes max_width |   L MB  Linv(l+1) MB | launches      us |  GB/s")
i = 0
print(f"F  -1 {'':>6} {'':>4} | {0.0:8.1f} {Db[0] / 1e6:7.1f} | {1:8d} {dur[0]:7.1f} | {Db[0] / 1e3 / dur[0]:7.0f}   (diagonal blocks of level 0)")
i = 1; tf = dur[0]
for l in range(nl):
    n = 1 if (l + 1 == nl or joined(l + 1)) else 2
    if i >= len(dur): break
    if l + 1 == nl and Lb[l] == 0: n = 0      # the top level of a single forest has no panel rows: no launch
    t = sum(dur[i:i + n]); i += n; tf += t
    by = Lb[l] + (Db[l + 1] if l + 1 < nl else 0)
    if t > 0: print(f"F {l:3d} {cnt[l]:6d} {wmax[l]:4d} | {Lb[l] / 1e6:8.1f} {(Db[l + 1] if l + 1 < nl else 0) / 1e6:7.1f} | {n:8d} {t:7.1f} | {by / 1e3 / t:7.0f}")
print(f"# forward total {tf:.0f} us, {i} launches")
print("# backward sweep (top level first).  Row l = the launch(es) that solve the diagonal blocks of level l (joined: one; else k_bwd_update + k_sweep) and apply the far chunks of U(l-1);  bytes = U(l) near part is inside; listed: U(l-1) + Uinv(l)")
print("#   level nodes max_width | U(l-1) MB  Uinv(l) MB | launches      us |  GB/s")
tb = 0.0
for l in range(nl - 1, -1, -1):
    n = 1 if joined(l) else 2
    if i >= len(dur): break
    t = sum(dur[i:i + n]); i += n; tb += t
    by = (Ub[l - 1] if l > 0 else 0) + Db[l]
    print(f"B {l:3d} {cnt[l]:6d} {wmax[l]:4d} | {(Ub[l - 1] if l > 0 else 0) / 1e6:8.1f} {Db[l] / 1e6:7.1f} | {n:8d} {t:7.1f} | {by / 1e3 / max(t, 1e-9):7.0f}")
print(f"# backward total {tb:.0f} us; launches consumed {i} of {len(dur)}")
