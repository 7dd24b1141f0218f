"""Single-GPU estimate of the Z-sharded factorisation's COMPUTE critical path on Pz GPUs.

All Pz layers live on one device and are stepped one after the other; every library call is timed on its own
(device-synchronised wall time, so launch overheads are included) and the critical path is the sum over steps of the
maximum over the layers that work concurrently in a real run.  Exchanges are executed (so the numerics are the real
ones and the result is checked) but not timed: their volume is reported so that a link model can be added
(xGMI ~ 50-100 GB/s effective per direction for ring all-reduce).  Usage: python scripts/zshard_critical_path.py N Pz"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superlu_dist_amd import driver, grid3d, matgen   # noqa: E402


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t)


def allreduce(tensors):
    tot = tensors[0].clone()
    for t in tensors[1:]:
        tot += t
    for t in tensors:
        t.copy_(tot)
    return tot.numel() * 8


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
    anorm = float(np.max(np.add.reduceat(np.abs(v), rp[:-1])))
    thresh = float(np.finfo(np.float32).eps) * anorm
    layers = [grid3d.GpuLayer(symb, v, P, z) for z in range(P)]
    maxlvl = grid3d.max_level(P)
    F = symb.flops
    for mode in ("coop", "reference-style"):
        for L in layers:
            L.reset_values()
        cp, comm_bytes, detail = 0.0, 0, []
        t0 = [timed(lambda L=L: L.factor_level(0, thresh)) for L in layers]
        cp += max(t0); detail.append(("leaf", max(t0), min(t0)))
        for ilvl in range(1, maxlvl):
            G = 1 << ilvl
            if mode == "coop":
                lvl_t = 0.0
                for z0 in range(0, P, G):
                    grp = layers[z0:z0 + G]
                    sl = [L.value_slices(ilvl, ilvl + 1) for L in grp]
                    for i in range(len(sl[0])):
                        b = allreduce([s[i] for s in sl])
                        if z0 == 0:
                            comm_bytes += b
                    nlev, mx = grp[0].coop_info(ilvl)
                    for L in grp:
                        L.coop_info(ilvl)
                    stages = [torch.empty(max(mx, 1), dtype=torch.float64, device="cuda") for _ in grp]
                    gt = 0.0
                    for l in range(nlev):
                        nodes = grp[0].coop_level_nodes(ilvl, l)
                        if len(nodes) == 1:
                            k = int(nodes[0])
                            tp = [timed(lambda L=L, g=g: L.coop_panel(ilvl, l, G, g, thresh, None)) for g, L in enumerate(grp)]
                            src = grp[k % G].panel_tensors(k)
                            for g, L in enumerate(grp):
                                if g != k % G:
                                    for a, b2 in zip(L.panel_tensors(k), src):
                                        a.copy_(b2)
                            if z0 == 0:
                                comm_bytes += sum(t.numel() for t in src) * 8
                            tu = [timed(lambda L=L, g=g: L.coop_update(ilvl, l, G, g, None)) for g, L in enumerate(grp)]
                        else:
                            tp = [timed(lambda L=L, g=g: L.coop_panel(ilvl, l, G, g, thresh, stages[g])) for g, L in enumerate(grp)]
                            sz = grp[0]._stage_sz[l]
                            b = allreduce([s[:sz] for s in stages])
                            if z0 == 0:
                                comm_bytes += b
                            tu = [timed(lambda L=L, g=g: L.coop_update(ilvl, l, G, g, stages[g])) for g, L in enumerate(grp)]
                        gt += max(tp) + max(tu)
                    for g, L in enumerate(grp):
                        L.coop_mask_u(ilvl, G, g)
                    us = [L.u_slices(ilvl) for L in grp]
                    for i in range(len(us[0])):
                        b = allreduce([u[i] for u in us])
                        if z0 == 0:
                            comm_bytes += b
                    lvl_t = max(lvl_t, gt)
                cp += lvl_t; detail.append((f"level {ilvl} (G={G}, {nlev} DAG levels)", lvl_t, 0.0))
            else:
                step = 1 << (ilvl - 1)
                for z in range(0, P, 2 * step):
                    a, b2 = layers[z].value_slices(ilvl), layers[z + step].value_slices(ilvl)
                    for x, y in zip(a, b2):
                        x += y
                        if z == 0:
                            comm_bytes += x.numel() * 8
                tl = [timed(lambda L=layers[z]: L.factor_level(ilvl, thresh)) for z in range(0, P, G)]
                cp += max(tl); detail.append((f"level {ilvl}", max(tl), min(tl)))
        infos = [L.info()[0] for L in layers]
        print(f"[{mode}] N={N} Pz={P}: compute critical path {cp:.1f} ms -> {F / cp / 1e9:.2f} TFLOP/s aggregate "
              f"(comm volume on rank 0's path {comm_bytes / 1e9:.2f} GB, not timed), info={infos}")
        for d in detail:
            print("    %-40s max %.1f ms  min %.1f ms" % d)
    # correctness of the last (reference-style) and the coop result is covered by tests/; here just the leader's info


if __name__ == "__main__":
    main()
