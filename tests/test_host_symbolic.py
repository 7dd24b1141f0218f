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


def test_matrix_market_reader_expands_symmetric_storage(tmp_path):
    """dreadMM.c's job: symmetric / skew-symmetric coordinate files store one triangle; general files everything; duplicates are summed."""
    import numpy as np
    from superlu_dist_amd import matgen
    n, rp, ci, v = matgen.elasticity3d_like(5, drop=0.2, seed=4)
    matgen.write_matrix_market(str(tmp_path / "s.mtx"), n, rp, ci, v, symmetric=True)
    matgen.write_matrix_market(str(tmp_path / "g.mtx"), n, rp, ci, v, symmetric=False)
    for name in ("s.mtx", "g.mtx"):
        n2, rp2, ci2, v2 = matgen.read_matrix_market(str(tmp_path / name))
        assert n2 == n and np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.allclose(v2, v, rtol=0, atol=1e-15)
    (tmp_path / "p.mtx").write_text("%%MatrixMarket matrix coordinate pattern symmetric\n% c\n3 3 4\n1 1\n2 1\n3 2\n3 3\n")
    n3, rp3, ci3, v3 = matgen.read_matrix_market(str(tmp_path / "p.mtx"))
    assert n3 == 3 and list(rp3) == [0, 2, 4, 6] and list(ci3) == [0, 1, 0, 2, 1, 2]
    (tmp_path / "k.mtx").write_text("%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 3.5\n")
    n4, rp4, ci4, v4 = matgen.read_matrix_market(str(tmp_path / "k.mtx"))
    assert list(ci4) == [1, 0] and list(v4) == [-3.5, 3.5]


def test_graph_nested_dissection_ordering(emul):
    """sluamd_order_nd: a permutation, without geometry; far less fill than the natural order on a randomly renumbered mesh operator and on an
    unsymmetric-pattern matrix; disconnected components and tiny inputs are handled; the factorisation with it solves the system."""
    import numpy as np
    from superlu_dist_amd import driver, matgen
    n, rp, ci, v = matgen.elasticity3d_like(9, drop=0.1, seed=5)
    p = driver.order_nd(n, rp, ci, leaf=32)
    assert sorted(p.tolist()) == list(range(n))
    s_nd = driver.Symbolic(n, rp, ci, p, relax=16, maxsup=128)
    s_nat = driver.Symbolic(n, rp, ci, np.arange(n, dtype=np.int32), relax=16, maxsup=128)
    assert s_nd.nnzL + s_nd.nnzU < 0.5 * (s_nat.nnzL + s_nat.nnzU)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, p, relax=16, maxsup=128)
    assert info == 0 and np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b) < 1e-10
    s_nd.free(); s_nat.free()
    # two disconnected copies + isolated vertices
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    B = sp.block_diag([A, A, sp.identity(3)]).tocsr(); B.sort_indices()
    pb = driver.order_nd(B.shape[0], B.indptr.astype(np.int32), B.indices.astype(np.int32), leaf=32)
    assert sorted(pb.tolist()) == list(range(B.shape[0]))
    n2, rp2, ci2, v2 = matgen.stencil3d_unsym(10, drop=0.3, seed=6)
    p2 = driver.order_nd(n2, rp2, ci2)
    assert sorted(p2.tolist()) == list(range(n2))
    assert sorted(driver.order_nd(1, np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32)).tolist()) == [0]


def test_library_poisson_generator_equals_the_numpy_construction():
    """sluamd_poisson3d (what matgen.poisson3d calls at bench sizes) writes the same CSR arrays, entry for entry, as the numpy construction the
    small test problems use -- also on a non-cubic grid and on degenerate ones (a line, a single point)."""
    import ctypes as C
    from superlu_dist_amd import _lib
    L = _lib.load()
    P_int, P_dbl = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    for nx, ny, nz in ((7, 5, 3), (1, 1, 9), (1, 1, 1), (4, 1, 6), (12, 12, 12)):
        n, rp, ci, v = matgen.poisson3d(0, nx, ny, nz)          # numpy path (below the size threshold)
        nnz = 7 * n - 2 * (nx * ny + ny * nz + nx * nz)
        assert len(v) == nnz
        rp2 = np.empty(n + 1, dtype=np.int32); ci2 = np.empty(nnz, dtype=np.int32); v2 = np.empty(nnz)
        got = L.sluamd_poisson3d(nx, ny, nz, rp2.ctypes.data_as(P_int), ci2.ctypes.data_as(P_int), v2.ctypes.data_as(P_dbl))
        assert got == nnz and (rp2 == rp).all() and (ci2 == ci).all() and (v2 == v).all(), (nx, ny, nz)


@pytest.mark.parametrize("case", ["poisson_nd", "stencil_unsym", "natural", "poisson_nd_cut_below_relax", "random_unsym_cut_below_relax"])
def test_parallel_supernodal_structure_equals_the_serial_pass(case, monkeypatch):
    """sluamd_dsymbfact builds the row structures of disjoint etree subtrees on worker threads and finishes the supernodes that reach a subtree's root (and
    everything above the cut) serially: the structure (supernode partition, L index, U index, value offsets, final perm_c) must be the serial pass's,
    whatever the cut -- SLUAMD_SYMB_CUT forces the task path on small structures (0: serial)."""
    import os
    if case == "poisson_nd":
        n, rp, ci, v = matgen.poisson3d(18); perm = matgen.nd_perm_grid3d(18, 18, 18, leaf=27); relax, maxsup = 8, 64
    elif case == "stencil_unsym":
        n, rp, ci, v = matgen.stencil3d_unsym(12, drop=0.3, seed=4); perm = matgen.nd_perm_grid3d(12, 12, 12, leaf=27); relax, maxsup = 4, 32
    elif case == "poisson_nd_cut_below_relax":
        # a task subtree strictly INSIDE a relaxed subtree (cut < relax) used to emit units the serial pass then covered again with the relaxed unit
        n, rp, ci, v = matgen.poisson3d(14); perm = matgen.nd_perm_grid3d(14, 14, 14, leaf=27); relax, maxsup = 16, 64
    elif case == "random_unsym_cut_below_relax":
        n, rp, ci, v = matgen.random_unsym(3000, 0.002, seed=3); perm = None; relax, maxsup = 32, 64
    else:
        n, rp, ci, v = matgen.poisson3d(12); perm = None; relax, maxsup = 1, 16
    ref = None
    for cut in (["0", "3", "5", "12", "20", "150"] if case.endswith("cut_below_relax") else ["0", "1", "7", "150", "100000"]):
        monkeypatch.setenv("SLUAMD_SYMB_CUT", cut)
        s = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
        fs = s.flat_store(values=False)
        got = (fs.xsup.copy(), fs.Lrowind.copy(), fs.Ufstnz.copy(), fs.Lnzval_off.copy(), fs.Unzval_off.copy(), np.asarray(s.perm_c).copy())
        s.free()
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                assert np.array_equal(a, b), (case, cut)


def test_nested_dissection_ordering_does_not_depend_on_the_thread_count():
    """sluamd_order_nd runs the recursive bisection as independent jobs on the planner's host threads; every job carries the end of its label range (separator on
    top, far half below it, near half last), so the permutation is the one of the depth-first single-thread order whatever the number of threads."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
            "from superlu_dist_amd import driver, matgen\n"
            "out = []\n"
            "for gen, leaf in ((lambda: matgen.poisson3d(0, 120, 120, 1), 32), (lambda: matgen.elasticity3d_like(12, drop=0.05, seed=1), 64), (lambda: matgen.random_unsym(4000, 0.002, seed=3), 16)):\n"
            "    n, rp, ci, v = gen()\n"
            "    p = driver.order_nd(n, rp, ci, leaf=leaf)\n"
            "    assert np.array_equal(np.sort(p), np.arange(n))\n"
            "    out.append(hashlib.sha1(np.ascontiguousarray(p).tobytes()).hexdigest())\n"
            "print(' '.join(out))\n") % root
    res = {}
    for threads in ("1", "2", "7", "16"):
        env = dict(os.environ, SLUAMD_PLAN_THREADS=threads)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[threads] = r.stdout.strip().splitlines()[-1]
    assert len(set(res.values())) == 1, res


def test_merged_child_runs_give_the_sorted_row_structure(monkeypatch):
    """The structure pass keeps the rows a unit takes from each child unit as ascending runs and merges them (units of more than 2048 collected rows) instead of
    sorting the collection; SLUAMD_SYMB_NO_RUN_MERGE=1 restores the sort.  60^3: the pieces of the top separators collect up to ~7000 rows."""
    N = 60
    n, rp, ci, v = matgen.poisson3d(N); perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    got = {}
    for key, env in (("merge", None), ("sort", "1")):
        if env: monkeypatch.setenv("SLUAMD_SYMB_NO_RUN_MERGE", env)
        else: monkeypatch.delenv("SLUAMD_SYMB_NO_RUN_MERGE", raising=False)
        s = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
        fs = s.flat_store(values=False)
        got[key] = (fs.xsup.copy(), fs.Lrowind.copy(), fs.Ufstnz.copy(), fs.Lnzval_off.copy(), fs.Unzval_off.copy(), np.asarray(s.perm_c).copy())
        s.free()
    for a, b in zip(got["merge"], got["sort"]):
        assert np.array_equal(a, b)
