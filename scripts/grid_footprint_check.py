"""How much device memory does a rank of a process grid really take, against the stored factor values the symbolic structure
predicts (sluamd_symb_grid_footprint)?  Ranks = threads sharing this box's GPU over the in-process transport; prints per rank
bytes_device / predicted value bytes -- the allowance bench.py's pre-flight check adds for exchange scratch, index images and
inverse blocks.  usage: grid_footprint_check.py N Pr Pc Pz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import driver, grid3d, matgen

N, Pr, Pc, Pz = (int(a) for a in sys.argv[1:5])
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
tree = symb.partition(Pz) if Pz > 1 else None
vals, rep, idx = symb.grid_footprint(Pr, Pc, Pz, tree)
comms = grid3d.local_comms(Pr, Pc, Pz)
xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b


def body(rank):
    h = grid3d.GridHandle.from_symbolic(symb, v, comms[rank], tree)
    st = h.stats()
    info = h.pdgstrf3d(0.0)
    y = h.pdgstrs3d(xp)
    st2 = h.stats()
    h.destroy()
    return st["bytes_device"], st["nnz_L"] + st["nnz_U"], info, y, st2["t_factor_ms"]


out = grid3d.run_ranks(Pr * Pc * Pz, body)
x = out[0][3][symb.perm_c, :]
print(f"N={N} grid {Pr}x{Pc}x{Pz}: residual {np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b):.1e}")
for r, (bd, nnz, info, _, tf) in enumerate(out):
    print(f"  rank {r}: predicted values {vals[r] * 8 / 1e9:7.3f} GB (own nnz reported {nnz * 8 / 1e9:7.3f} GB)  bytes_device {bd / 1e9:7.3f} GB  ratio {bd / (vals[r] * 8):.3f}  factor {tf:.0f} ms info {info}")
