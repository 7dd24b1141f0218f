"""Host-side symbolic factorisation + distribution (libsluamd.so, CPU code) produce a valid reference-format
L/U store: factoring it with the oracle and solving reproduces A x = b."""
import numpy as np
import pytest
import oracle as orc
from superlu_dist_amd import matgen, driver


def _check_store_structure(fs):
    ns = fs.nsupers
    for k in range(ns):
        li = fs.Lrowind[fs.Lrowind_off[k]:fs.Lrowind_off[k + 1]]
        nb, nsupr = li[0], li[1]
        p, rows = 2, 0
        gids = []
        for b in range(nb):
            gid, nbrow = li[p], li[p + 1]
            r = li[p + 2:p + 2 + nbrow]
            assert np.all(np.diff(r) > 0)
            assert np.all((r >= fs.xsup[gid]) & (r < fs.xsup[gid + 1]))
            gids.append(gid); rows += nbrow; p += 2 + nbrow
        assert rows == nsupr and gids[0] == k and gids == sorted(gids)
        assert fs.Lnzval_off[k + 1] - fs.Lnzval_off[k] == nsupr * (fs.xsup[k + 1] - fs.xsup[k])


def _solve_with_oracle(n, rp, ci, v, perm, relax, maxsup, nrhs=2):
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    symb.distribute_host(v)
    fs = symb.flat_store()
    _check_store_structure(fs)
    st = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz,
                     fs.Unzval_off, fs.Unzval)
    info, tiny, flops = orc.dfactor(st)
    assert info == 0
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    y = orc.dsolve(st, xp)
    x = y[symb.perm_c, :]
    res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b)
    return res, symb, flops


@pytest.mark.parametrize("N,leaf,relax,maxsup", [(6, 8, 4, 16), (8, 16, 16, 64), (9, 27, 32, 256), (7, 1000, 1, 8)])
def test_poisson_nd(N, leaf, relax, maxsup):
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
    res, symb, _ = _solve_with_oracle(n, rp, ci, v, perm, relax, maxsup)
    assert res < 1e-13
    assert sorted(symb.perm_c.tolist()) == list(range(n))


def test_unsymmetric_values_natural_order():
    n, rp, ci, v = matgen.random_unsym(150, 0.03, seed=5)
    res, symb, _ = _solve_with_oracle(n, rp, ci, v, None, 8, 32)
    assert res < 1e-12


def test_flops_match_oracle_tally():
    N = 8
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=16)
    res, symb, flops = _solve_with_oracle(n, rp, ci, v, perm, 16, 64)
    # symmetric pattern -> full U segments -> padded Schur flops == 2*nsupc*r*r summed
    total = flops[0] + flops[1]
    assert abs(total - symb.flops) / symb.flops < 0.05
