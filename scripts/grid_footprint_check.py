"""How much device memory does a rank of a process grid really take, against the stored factor values the symbolic structure
predicts (sluamd_symb_grid_footprint)?  Prints per rank bytes_device / predicted value bytes -- the allowance bench.py's
pre-flight check adds for exchange scratch, index images and inverse blocks.
usage: grid_footprint_check.py N Pr Pc Pz [--create-only]
  default        ranks = threads sharing this box's GPU over the in-process transport; factor + solve run (residual printed)
  --create-only  the handles of the ranks are created ONE AFTER THE OTHER and destroyed again (the own-pipeline creation path derives
                 every table from the replicated symbolic structure: no exchange): bytes_device of every rank without running anything --
                 also on a box whose memory holds one rank at a time, and on CPU through SLUAMD_LIB=oracle/libsluamd_emul.so (the
                 planner is host code; what it would allocate on the device is what it allocates there)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superlu_dist_amd import driver, grid3d, matgen

args = [a for a in sys.argv[1:] if not a.startswith("--")]
create_only = "--create-only" in sys.argv
N, Pr, Pc, Pz = (int(a) for a in args[:4])
n, rp, ci, v = matgen.poisson3d(N)
perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
tree = symb.partition(Pz) if Pz > 1 else None
vals, rep, idx = symb.grid_footprint(Pr, Pc, Pz, tree)
comms = grid3d.local_comms(Pr, Pc, Pz)
P = Pr * Pc * Pz
tag = " ".join(f"{k}={os.environ[k]}" for k in ("SLUAMD_NO_LEVEL_SPLIT", "SLUAMD_NO_FUSE", "SLUAMD_NO_TILE_MAPS") if k in os.environ)
if create_only:
    print(f"N={N} grid {Pr}x{Pc}x{Pz} (handles created one at a time, nothing run) {tag}")
    worst = 0.0
    for r in range(P):
        h = grid3d.GridHandle.from_symbolic(symb, v, comms[r], tree)
        st = h.stats()
        h.destroy()
        ratio = st["bytes_device"] / (vals[r] * 8)
        worst = max(worst, ratio)
        print(f"  rank {r}: predicted values {vals[r] * 8 / 1e9:7.3f} GB (Z replicas {rep[r] * 8 / 1e9:6.3f} GB)  bytes_device {st['bytes_device'] / 1e9:7.3f} GB  ratio {ratio:.3f}  "
              f"levels {st['num_levels']} fused pairs {st['reserved_i']} planned Schur tile executions {st['schur_tiles']}")
        tiles = tiles + st["schur_tiles"] if r else st["schur_tiles"]
    print(f"  worst allocated / values = {worst:.3f}; planned Schur tile executions, sum over ranks = {tiles}")
    sys.exit(0)
xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b


def body(rank):
    h = grid3d.GridHandle.from_symbolic(symb, v, comms[rank], tree)
    st = h.stats()
    info = h.pdgstrf3d(0.0)
    y = h.pdgstrs3d(xp)
    st2 = h.stats()
    h.destroy()
    return st["bytes_device"], st["nnz_L"] + st["nnz_U"], info, y, st2["t_factor_ms"], st2["bytes_device"]


out = grid3d.run_ranks(P, body)
x = out[0][3][symb.perm_c, :]
print(f"N={N} grid {Pr}x{Pc}x{Pz}: residual {np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b):.1e} {tag}")
for r, (bd, nnz, info, _, tf, bd2) in enumerate(out):
    print(f"  rank {r}: predicted values {vals[r] * 8 / 1e9:7.3f} GB (own nnz reported {nnz * 8 / 1e9:7.3f} GB)  bytes_device {bd / 1e9:7.3f} GB  ratio {bd / (vals[r] * 8):.3f}  "
          f"(with tile records {bd2 / 1e9:7.3f} GB, ratio {bd2 / (vals[r] * 8):.3f})  factor {tf:.0f} ms info {info}")
