"""N>1 path on CPU: the Z-sharded orchestration (superlu_dist_amd/grid3d.py) over torch.distributed/gloo with the CPU
oracle as the per-layer engine, world sizes 2 and 4; checks the distributed solution against the single-layer one."""
import os, socket, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, N, nrhs, out_path, coop=False):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from superlu_dist_amd import driver, grid3d, matgen
    from zbackend_oracle import OracleLayer
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(3)
    v = v * (1.0 + 0.2 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=16)
    symb = driver.Symbolic(n, rp, ci, perm, relax=8, maxsup=32)
    layer = OracleLayer(symb, v, world, rank)
    comm = grid3d.DistComm(dist, npdep=world if coop else 1)
    info = (grid3d.pdgstrf3d_coop if coop else grid3d.pdgstrf3d)(layer, comm, rank, world, 0.0)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    xp = np.zeros((nrhs, n)); xp[:, symb.perm_c] = b.T                       # (Pc b)^T
    x = grid3d.init_rhs(layer, rank, world, torch.from_numpy(xp))
    grid3d.pdgstrs3d(layer, comm, rank, world, x)
    sol = x.numpy()[:, symb.perm_c].T                                         # Pc^T y
    res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, sol)) / np.linalg.norm(b)
    if rank == 0:
        np.savez(out_path, info=info, res=res, sol=sol, xt=xt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,nrhs", [(2, 8, 1), (4, 10, 2)])
def test_z_sharded_factor_and_solve_gloo(world, N, nrhs, tmp_path):
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, nrhs, out), nprocs=world, join=True)
    r = np.load(out)
    assert int(r["info"]) == 0
    assert float(r["res"]) < 1e-12
    assert np.abs(r["sol"] - r["xt"]).max() < 1e-10


@pytest.mark.parametrize("world,N,nrhs", [(2, 8, 1), (4, 10, 2), (8, 12, 1)])
def test_z_sharded_cooperative_ancestors_gloo(world, N, nrhs, tmp_path):
    """Same system, ancestor forests factored cooperatively by the layers that share them (grid3d.pdgstrf3d_coop)."""
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, nrhs, out, True), nprocs=world, join=True)
    r = np.load(out)
    assert int(r["info"]) == 0
    assert float(r["res"]) < 1e-12
    assert np.abs(r["sol"] - r["xt"]).max() < 1e-10


def test_partition_properties():
    sys.path.insert(0, HERE)
    from superlu_dist_amd import driver, grid3d, matgen
    N = 12
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=16)
    symb = driver.Symbolic(n, rp, ci, perm, relax=8, maxsup=64)
    fs = symb.flat_store(values=False)
    for npdep in (1, 2, 4, 8):
        t = symb.partition(npdep)
        nf = 2 * npdep - 1
        assert t.min() >= 0 and t.max() < nf
        # every block of supernode k lives in k's own tree or in one of its heap ancestors
        for k in range(fs.nsupers):
            li = fs.Lrowind[fs.Lrowind_off[k]:fs.Lrowind_off[k + 1]]
            p = 2
            for b in range(li[0]):
                g = li[p]; anc = t[k]; ok = False
                while True:
                    if t[g] == anc: ok = True; break
                    if anc == 0: break
                    anc = (anc - 1) // 2
                assert ok, (npdep, k, g, t[k], t[g])
                p += 2 + li[p + 1]
        if npdep > 1:   # leaf forests are reasonably balanced (greedy split of the two sub-domains)
            sizes = [np.diff(fs.xsup)[t == (npdep - 1 + z)].sum() for z in range(npdep)]
            assert min(sizes) > 0.4 * max(sizes)
        for z in range(npdep):
            assert grid3d.path_trees(npdep, z)[-1] == 0
