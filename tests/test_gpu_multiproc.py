"""Cooperative Z-sharded factorisation with REAL processes on one GPU: world_size ranks share device 0, each with its
own HIP context and sluamd handle, exchanging through torch.distributed/gloo with host staging (the box has a single
GPU, so RCCL itself cannot be exercised here; what runs is the product's group/communicator logic, the cooperative
entry points and the solve, end to end, across process boundaries)."""
import os, socket, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, N, nrhs, coop, out_path):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from superlu_dist_amd import driver, grid3d, matgen
    torch.cuda.set_device(0)
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(9)
    v = v * (1.0 + 0.2 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    symb = driver.Symbolic(n, rp, ci, perm, relax=16, maxsup=128)
    layer = grid3d.GpuLayer(symb, v, world, rank, device=0)
    comm = grid3d.DistComm(dist, npdep=world, host_staging=True)
    info = (grid3d.pdgstrf3d_coop if coop else grid3d.pdgstrf3d)(layer, comm, rank, world, 0.0)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    xp = np.zeros((nrhs, n)); xp[:, symb.perm_c] = b.T
    x = grid3d.init_rhs(layer, rank, world, torch.from_numpy(xp).to(layer.device))
    grid3d.pdgstrs3d(layer, comm, rank, world, x)
    sol = x.cpu().numpy()[:, symb.perm_c].T
    res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, sol)) / np.linalg.norm(b)
    np.savez(out_path + f".{rank}.npz", info=info, res=res, err=np.abs(sol - xt).max())
    layer.destroy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,coop", [(2, 14, True), (4, 16, True), (4, 16, False)])
def test_multiprocess_single_gpu(world, N, coop, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(world, _free_port(), N, 2, coop, out), nprocs=world, join=True)
    for rank in range(world):
        r = np.load(out + f".{rank}.npz")
        assert int(r["info"]) == 0
        assert float(r["res"]) < 1e-10
        assert float(r["err"]) < 1e-8
