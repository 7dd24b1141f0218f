"""One rank of a multi-process grid run (launched by torch.distributed.run from the tests): own symbolic factorisation,
grid handle over the callback transport (gloo), factor + solve, residual check.  --engine emul binds the CPU test build
of the library's host sources (CPU tests), --engine hip the product library (GPU tests: ranks share the box's GPU)."""
import argparse, ctypes as C, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="hip")
    ap.add_argument("--grid", type=int, nargs=3, default=[1, 1, 2])
    ap.add_argument("--side", dest="n", type=int, default=8)
    ap.add_argument("--transport", default="callbacks", choices=["callbacks", "rccl"])
    ap.add_argument("--one-gpu-per-rank", action="store_true")
    a = ap.parse_args()
    import torch.distributed as dist
    from superlu_dist_amd import _lib, driver, grid3d, matgen
    if a.engine == "emul":
        _lib._lib = _lib.bind(C.CDLL(os.path.join(ROOT, "oracle", "libsluamd_emul.so")))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    Pr, Pc, Pz = a.grid
    assert Pr * Pc * Pz == world
    N = a.n
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(N)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    symb = driver.Symbolic(n, rp, ci, perm, relax=16, maxsup=64)
    sn_tree = symb.partition(Pz) if Pz > 1 else None
    if a.transport == "rccl":
        comm = grid3d.rccl_comm(dist, Pr, Pc, Pz, int(os.environ.get("LOCAL_RANK", "0")) if a.one_gpu_per_rank else 0)
        print("RCCL_COMM_READY", flush=True)     # past ncclCommInitRank: from here on a failure is a transport bug, not "one GPU"
        _lib.check(_lib.load().sluamd_comm_selftest(comm, 1 << 20), "sluamd_comm_selftest")
        tc = None
    else:
        tc = grid3d.TorchComm(dist, Pr, Pc, Pz)
        comm = tc.handle
    h = grid3d.GridHandle.from_symbolic(symb, v, comm, sn_tree)
    info = h.pdgstrf3d(0.0)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    y = h.pdgstrs3d(xp)
    x = y[symb.perm_c, :]
    res = float(np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b))
    assert info == 0 and res < 1e-10, (info, res)
    assert np.abs(x - xt).max() < 1e-9
    h.destroy()
    dist.barrier()
    if rank == 0:
        print(f"GRID_WORKER_OK grid {Pr}x{Pc}x{Pz} residual {res:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
