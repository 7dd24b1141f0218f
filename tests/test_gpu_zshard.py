"""Z-sharded GPU path on ONE device: Pz layers emulated by threads (each with its own sluamd handle holding only its
sub-forest + ancestors) and an in-process communicator, so the per-level factor/solve entry points, the local arenas and
the ancestor-slice reduction are exercised on real hardware even though the box has a single GPU."""
import queue, threading
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class ThreadComm:
    def __init__(self, world):
        self.q = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.lock = threading.Lock()
        self.tls = threading.local()
        self.gbar = {}

    def bind(self, z):
        self.tls.z = z

    def send(self, t, dst):
        self.q[(self.tls.z, dst)].put(t.clone())

    def recv(self, t, src):
        t.copy_(self.q[(src, self.tls.z)].get(timeout=120))

    def group(self, ilvl, z0):
        G = 1 << ilvl
        with self.lock:
            if (z0, G) not in self.gbar:
                self.gbar[(z0, G)] = threading.Barrier(G)
        return (z0, G)

    def allreduce_sum(self, t, group=None):
        import torch
        z0, G = group if group is not None else (0, len(self.slots))
        bar = self.gbar[(z0, G)] if group is not None else self.bar
        torch.cuda.synchronize()
        self.slots[self.tls.z] = t.clone()
        torch.cuda.synchronize()
        bar.wait()
        tot = self.slots[z0].clone()
        for zz in range(z0 + 1, z0 + G):
            tot += self.slots[zz]
        torch.cuda.synchronize()
        bar.wait()
        t.copy_(tot)
        torch.cuda.synchronize()

    def reduce_sum(self, t, dst, group=None):
        self.allreduce_sum(t, group)                 # emulation: a reduce is an all-reduce whose other copies are unused

    def broadcast(self, t, src, group=None):
        import torch
        z0, G = group if group is not None else (0, len(self.slots))
        bar = self.gbar[(z0, G)] if group is not None else self.bar
        torch.cuda.synchronize()
        if self.tls.z == src:
            self.slots[src] = t
        bar.wait()
        if self.tls.z != src:
            t.copy_(self.slots[src])
        torch.cuda.synchronize()
        bar.wait()

    def allreduce_min_int(self, v, device):
        self.slots[self.tls.z] = v
        self.bar.wait()
        m = min(self.slots)
        self.bar.wait()
        return m


@pytest.mark.parametrize("npdep,N,nrhs,coop", [(2, 12, 1, False), (4, 14, 2, False), (2, 12, 1, True), (4, 14, 2, True),
                                                 (8, 16, 1, True)])
def test_z_sharded_on_one_gpu(npdep, N, nrhs, coop):
    import torch
    from superlu_dist_amd import driver, grid3d, matgen
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(5)
    v = v * (1.0 + 0.2 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    symb = driver.Symbolic(n, rp, ci, perm, relax=16, maxsup=128)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    xp = np.zeros((nrhs, n)); xp[:, symb.perm_c] = b.T
    comm = ThreadComm(npdep)
    results, errors = [None] * npdep, []

    def run(z):
        try:
            comm.bind(z)
            layer = grid3d.GpuLayer(symb, v, npdep, z)
            info = (grid3d.pdgstrf3d_coop if coop else grid3d.pdgstrf3d)(layer, comm, z, npdep, 0.0)
            x = grid3d.init_rhs(layer, z, npdep, torch.from_numpy(xp).to(layer.device))
            grid3d.pdgstrs3d(layer, comm, z, npdep, x)
            results[z] = (info, x.cpu().numpy(), layer.stats()["nnz_L"])
            layer.destroy()
        except Exception as e:                      # surface worker failures in the main thread
            errors.append(e)
            for bar in [comm.bar] + list(comm.gbar.values()):
                try:
                    bar.abort()
                except Exception:
                    pass

    ths = [threading.Thread(target=run, args=(z,)) for z in range(npdep)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errors, errors
    full_nnzL = symb.nnzL
    for z in range(npdep):
        info, x, nnzl = results[z]
        assert info == 0
        sol = x[:, symb.perm_c].T
        res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, sol)) / np.linalg.norm(b)
        assert res < 1e-10
        assert nnzl < full_nnzL                      # a layer stores only its sub-forest + ancestors
