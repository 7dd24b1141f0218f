"""SURVEY 8(f)-4 pinned to the reference: sluamd_dsymbfact_unsym -- the exact unsymmetric symbolic factorisation with the
reference's supernode rules (SRC/prec-independent/symbfact.c:83-200: relax_snode, the T2 / subset / maxsup boundary test of column_dfs, full-
height segments of relaxed supernodes) -- reproduces the structure the REAL symbfact + pddistribute3d built for every one-rank golden fixture
(tests/golden/*.npz were recorded from the reference by tests/golden/make_golden.py; its defaults: relax = 30, maxsup = 256, util.c:230-231):
the supernode partition xsup, the row set of every L block, every first-nonzero entry of Ufstnz, hence nnz(L) and nnz(U) exactly -- on
symmetric (Poisson ND), unsymmetric (unsym300, g20) and complex-valued patterns alike.  Then the structure is USED: host distribution +
the CPU oracle factor and solve on it (the -m gpu twin runs the device on it, tests/test_gpu_parity.py)."""
import glob, os
import numpy as np
import pytest
import oracle as orc
from superlu_dist_amd import driver, matgen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ONE_RANK = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "*.npz")) if int(np.load(f)["nranks"][0]) == 1)


def _blocks(xsup, Loff, Lidx, Uoff, Uidx):
    L, U = [], []
    for k in range(len(xsup) - 1):
        li = Lidx[Loff[k]:Loff[k + 1]]
        blk, p = {}, 2
        for _ in range(li[0]):
            gid, nb = int(li[p]), int(li[p + 1])
            blk[gid] = sorted(li[p + 2:p + 2 + nb].tolist())          # the reference keeps the rows of a block in discovery order
            p += 2 + nb
        L.append(blk)
        ui = Uidx[Uoff[k]:Uoff[k + 1]]
        ub = {}
        if len(ui) >= 3:
            p = 3
            for _ in range(ui[0]):
                jb = int(ui[p]); w = int(xsup[jb + 1] - xsup[jb])
                ub[jb] = ui[p + 2:p + 2 + w].tolist()
                p += 2 + w
        U.append(ub)
    return L, U


@pytest.mark.parametrize("name", ONE_RANK)
def test_structure_equals_the_reference_symbfact(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = int(g["r0__n"][0])
    rp, ci = g["r0__A_rowptr"].astype(np.int32), g["r0__A_colind"].astype(np.int32)
    pr, pc = g["r0__perm_r"], g["r0__perm_c"].astype(np.int32)
    # the matrix symbfact saw: rows permuted by perm_r (LargeDiag_MC64 / NOROWPERM), then Pc applied symmetrically
    order = np.argsort(pr, kind="stable")                       # new row q = old row order[q]
    cnt = np.diff(rp)[order]
    rp2 = np.concatenate(([0], np.cumsum(cnt))).astype(np.int32)
    ci2 = np.concatenate([ci[rp[i]:rp[i + 1]] for i in order]).astype(np.int32) if n else ci
    s = driver.Symbolic(n, rp2, ci2, pc, relax=30, maxsup=256, unsym=True)
    assert np.array_equal(s.perm_c, pc)                         # the recorded perm_c is already an etree postorder: kept as is
    fs = s.flat_store(values=False)
    xs = g["r0__xsup"]
    assert s.nsupers == len(xs) - 1 and np.array_equal(fs.xsup, xs)
    Lr, Ur = _blocks(xs, g["r0__Lrowind_off"], g["r0__Lrowind"], g["r0__Ufstnz_off"], g["r0__Ufstnz"])
    Lo, Uo = _blocks(fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Ufstnz_off, fs.Ufstnz)
    assert Lo == Lr and Uo == Ur
    assert s.nnzL == int(g["r0__Lnzval_off"][-1]) and s.nnzU == int(g["r0__Unzval_off"][-1])
    s.free()


@pytest.mark.parametrize("kind", ["unsym", "stencil_unsym", "poisson_nd"])
def test_unsymmetric_structure_factors_and_solves(kind):
    """The structure is closed under the elimination: distribute A into it, factor with the CPU oracle (no entry may fall outside), solve;
    and it is never larger than the symmetrised structure of the same ordering and supernode parameters."""
    if kind == "unsym":
        n, rp, ci, v = matgen.random_unsym(260, 0.03, seed=3); perm = None
    elif kind == "stencil_unsym":
        n, rp, ci, v = matgen.stencil3d_unsym(9, drop=0.35, seed=5); perm = matgen.nd_perm_grid3d(9, 9, 9, leaf=8)
    else:
        n, rp, ci, v = matgen.poisson3d(9); perm = matgen.nd_perm_grid3d(9, 9, 9, leaf=27)
    s = driver.Symbolic(n, rp, ci, perm, relax=12, maxsup=40, unsym=True)
    s.distribute_host(v)
    fs = s.flat_store()
    assert np.diff(fs.xsup).max() <= 40
    st = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz, fs.Unzval_off, fs.Unzval)
    info, tiny, flops = orc.dfactor(st)
    assert info == 0
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    xp = np.zeros_like(b, order="F"); xp[s.perm_c, :] = b
    x = orc.dsolve(st, xp)[s.perm_c, :]
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b) < 1e-12
    tree = s.partition(4)                                      # forest partition from the etree of A + A^T: every dependency goes up the tree
    assert tree.min() >= 0 and tree.max() <= 6
    s.free()
