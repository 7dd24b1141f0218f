"""One-GPU proxy for the work inflation of a process grid: the library's own pipeline on a Pr x Pc x Pz grid whose ranks are THREADS
sharing one GPU (in-process transport).  Prints the wall time of pdgstrf3d / pdgstrs3d (max over ranks) and, from one extra PROFILED
factorisation (serial schedule, HIP events per kernel family on every rank's own stream), per rank: Schur ms, panel ms (incl. the exchange
phases), exchange ms, K-fused pairs -- and their sums over the ranks.  Not a scaling measurement (the ranks compete for the same device,
and a rank's event intervals stretch while another rank's kernels hold the CUs: the sums are UPPER bounds of the work), but the sum of
the ranks' kernel time against the 1 x 1 x 1 figure is what an XY layer costs beyond the single-GPU schedule.
usage: xy_bench.py N Pr Pc Pz [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import driver, grid3d, matgen
N, Pr, Pc, Pz = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
sn_tree = symb.partition(Pz) if Pz > 1 else None
P = Pr * Pc * Pz
comms = grid3d.local_comms(Pr, Pc, Pz)
xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
import threading
bar = threading.Barrier(P)
tf = [[0.0] * reps for _ in range(P)]; ts = [[0.0] * reps for _ in range(P)]

def rank_body(rank):
    h = grid3d.GridHandle.from_symbolic(symb, v, comms[rank], sn_tree)
    y = None
    for it in range(reps):
        if it: h.reset_values()
        bar.wait(); t0 = time.perf_counter()
        info = h.pdgstrf3d(0.0)
        bar.wait(); t1 = time.perf_counter()
        y = h.pdgstrs3d(xp)
        bar.wait(); t2 = time.perf_counter()
        tf[rank][it] = t1 - t0; ts[rank][it] = t2 - t1
        assert info == 0
    h.set_profile(True)                  # collective: every rank runs the serial schedule
    h.reset_values(); bar.wait()
    h.pdgstrf3d(0.0)
    st = h.stats()
    prof[rank] = (st["t_schur_ms"], st["t_panel_ms"], st["t_exchange_ms"], st["t_reduce_ms"], st["t_factor_ms"], st["reserved_i"], st["bytes_device"])
    h.set_profile(False)
    h.destroy()
    return y

prof = [None] * P
out = grid3d.run_ranks(P, rank_body)
x = out[0][symb.perm_c, :]
res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b)
print("grid %dx%dx%d N %d: factor ms %s  solve ms %s  residual %.2e" % (Pr, Pc, Pz, N, ["%.1f" % (max(tf[r][i] for r in range(P)) * 1e3) for i in range(reps)],
      ["%.1f" % (max(ts[r][i] for r in range(P)) * 1e3) for i in range(reps)], res))
for r in range(P):
    print("  rank %d: schur %.1f ms  panel %.1f ms (exchange %.1f)  reduce %.1f  profiled factor %.1f ms  fused pairs %d  bytes_device %.2f GB" % ((r,) + prof[r][:6] + (prof[r][6] / 1e9,)))
print("  sum over ranks: schur %.1f ms  panel %.1f ms  exchange %.1f ms  schur + panel - exchange %.1f ms" % (
    sum(p[0] for p in prof), sum(p[1] for p in prof), sum(p[2] for p in prof), sum(p[0] + p[1] - p[2] for p in prof)))
