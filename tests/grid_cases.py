"""Shared bodies of the process-grid tests: the same checks run against the product library on a GPU (test_gpu_grid.py)
and against the CPU test build of the library's host sources (oracle/libsluamd_emul.so, test_grid_emul.py).
Ranks are threads of this process over the library's in-process transport (sluamd_comm_create_local)."""
import numpy as np
from superlu_dist_amd import driver, grid3d, matgen

GRID_FIXTURES = ["g20_1x1x2", "poisson8_nd_1x1x2", "g20_2x1x1", "g20_2x2x2"]
ZGRID_FIXTURES = ["z_cg20_1x1x2", "z_poisson8_nd_1x1x2",     # complex16 (pzgstrf3d / pzgstrs3d) on two Z layers
                  "z_cg20_2x1x1", "z_cg20_1x2x1", "z_cg20_2x2x2"]   # ... and on XY block-cyclic layers (round 3)


def forests_of(g, rank):
    p = f"r{rank}__"
    ml = int(g[p + "maxLvl"][0])
    return dict(maxLvl=ml, myTreeIdxs=g[p + "myTreeIdxs"], myZeroTrIdxs=g[p + "myZeroTrIdxs"],
                nodeLists=[g[p + f"forest{f}_nodeList"] if int(g[p + f"forest{f}_nNodes"][0]) > 0 else None
                           for f in range((1 << ml) - 1)])


def check_fixture_grid(g):
    """Every rank's post-pdgstrf3d L/U values against the reference's per-rank record (1e-12 * ||A||_max), info, and
    every recorded pdgstrs3d call against the reference's output (layer 0 holds the result; 1e-10 relative)."""
    P = int(g["nranks"][0])
    Pr, Pc, Pz = [int(v) for v in g["grid"]]
    comms = grid3d.local_comms(Pr, Pc, Pz)
    n = int(g["r0__n"][0])
    pr_, pc_ = g["r0__perm_r"], g["r0__perm_c"]
    scale = max(max(np.abs(g[f"r{q}__Lnzval_pre"]).max() if len(g[f"r{q}__Lnzval_pre"]) else 0.0 for q in range(P)), 1e-300)

    def rank_body(rank):
        p = f"r{rank}__"
        r, c, z = int(g[p + "myrow"][0]), int(g[p + "mycol"][0]), int(g[p + "myz"][0])
        w = (z * Pr + r) * Pc + c
        st = driver.FlatStore.from_golden(g, rank, "pre")
        h = grid3d.GridHandle.from_store(st, forests_of(g, rank), comms[w], replace_tiny=bool(g[p + "ReplaceTinyPivot"][0]))
        info = h.pdgstrf3d(float(g[p + "thresh"][0]))
        h.copy_to_host(st)
        assert info == int(g[p + "info"][0])
        eL = np.abs(st.Lnzval - g[p + "Lnzval_post"]).max() if len(st.Lnzval) else 0.0
        eU = np.abs(st.Unzval - g[p + "Unzval_post"]).max() if len(st.Unzval) else 0.0
        assert eL <= 1e-12 * scale and eU <= 1e-12 * scale, (rank, eL, eU)
        sols = []
        si = 0
        while f"r0__solve{si}_B_in" in g:
            nrhs = int(g[f"r0__solve{si}_nrhs"][0])
            B = np.zeros((n, nrhs), order="F", dtype=g[f"r0__solve{si}_B_in"].dtype)
            for q in range(Pr * Pc):     # layer 0 holds the 2-D row distribution of B
                f0 = int(g[f"r{q}__solve{si}_fst_row"][0]); ml = int(g[f"r{q}__solve{si}_m_loc"][0])
                B[f0:f0 + ml, :] = g[f"r{q}__solve{si}_B_in"].reshape((ml, nrhs), order="F")
            xp = np.zeros((n, nrhs), order="F", dtype=B.dtype); xp[pc_[pr_], :] = B
            y = h.pdgstrs3d(xp)
            if z == 0:
                f0 = int(g[p + f"solve{si}_fst_row"][0]); ml = int(g[p + f"solve{si}_m_loc"][0])
                X = g[p + f"solve{si}_B_out"].reshape((ml, nrhs), order="F")
                assert np.abs(y[f0:f0 + ml, :] - X).max() <= 1e-10 * max(1.0, np.abs(X).max())
            sols.append(y)
            si += 1
        assert si >= 1
        h.destroy()
        return sols

    sols = grid3d.run_ranks(P, rank_body)
    for q in range(1, P):               # every rank received the complete solution
        for a, b in zip(sols[0], sols[q]):
            assert np.array_equal(a, b)


def check_own_pipeline(N, grid, nrhs=1, leaf=27, relax=16, maxsup=64, unsym=False, refactor=False, make_comms=None):
    """Library's own symbolic factorisation + device-side distribution on a Pr x Pc x Pz grid: residual on the original
    system < 1e-10 and the solution equal (1e-10) to the single-rank one."""
    n, rp, ci, v = matgen.poisson3d(N)
    if unsym:
        rng = np.random.default_rng(N)
        v = v * (1.0 + 0.3 * rng.random(v.size))
        v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
    check_matrix_on_grid(n, rp, ci, v, perm, grid, nrhs=nrhs, relax=relax, maxsup=maxsup, refactor=refactor, make_comms=make_comms)


def stream_ordered_comms(Pr, Pc, Pz):
    """Emulation library only: one communicator per rank of the in-process STREAM-ORDERED transport that stands behind
    sluamd_comm_create_rccl in the CPU test build (oracle/emul/comm_norccl.cpp: RCCL's contract -- end(s) only queues -- for
    ranks that are threads)."""
    import ctypes as C
    from superlu_dist_amd import _lib
    L = _lib.load()
    idbuf = (C.c_char * 128)()
    _lib.check(L.sluamd_comm_rccl_unique_id(idbuf), "sluamd_comm_rccl_unique_id")
    out = []
    for rank in range(Pr * Pc * Pz):
        r, c, z = grid3d.grid_coords(rank, Pr, Pc, Pz)
        h = C.c_void_p()
        _lib.check(L.sluamd_comm_create_rccl(C.byref(h), idbuf, Pr, Pc, Pz, r, c, z, -1), "sluamd_comm_create_rccl")
        out.append(h)
    return out


def check_transport_selftest(comms, nbytes):
    """sluamd_comm_selftest on every rank of a world (threads): ring exchange as one stream-ordered group, host-buffer group,
    min-all-reduce.  A one-rank world sends to itself."""
    from superlu_dist_amd import _lib
    L = _lib.load()

    def body(rank):
        _lib.check(L.sluamd_comm_selftest(comms[rank], nbytes), "sluamd_comm_selftest")
        return True

    assert all(grid3d.run_ranks(len(comms), body))


def check_fixture_on_one_rank_comm(g, comm):
    """A 1 x 1 x 1 reference fixture through sluamd_dCreateLUHandleGrid over a one-rank communicator: factor and every recorded
    solve against the reference's record (the grid entry points with a transport attached, no peer to talk to)."""
    n = int(g["r0__n"][0])
    st = driver.FlatStore.from_golden(g, 0, "pre")
    h = grid3d.GridHandle.from_store(st, None, comm, replace_tiny=bool(g["r0__ReplaceTinyPivot"][0]))
    assert h.pdgstrf3d(float(g["r0__thresh"][0])) == int(g["r0__info"][0])
    h.copy_to_host(st)
    scale = max(np.abs(g["r0__Lnzval_pre"]).max(), np.abs(g["r0__Unzval_pre"]).max())
    assert np.abs(st.Lnzval - g["r0__Lnzval_post"]).max() <= 1e-12 * scale
    assert np.abs(st.Unzval - g["r0__Unzval_post"]).max() <= 1e-12 * scale
    pr_, pc_ = g["r0__perm_r"], g["r0__perm_c"]
    si = 0
    while f"r0__solve{si}_B_in" in g:
        nrhs = int(g[f"r0__solve{si}_nrhs"][0])
        B = g[f"r0__solve{si}_B_in"].reshape((n, nrhs), order="F")
        xp = np.zeros((n, nrhs), order="F", dtype=B.dtype); xp[pc_[pr_], :] = B
        y = h.pdgstrs3d(xp)
        X = g[f"r0__solve{si}_B_out"].reshape((n, nrhs), order="F")
        assert np.abs(y - X).max() <= 1e-10 * max(1.0, np.abs(X).max())
        si += 1
    assert si >= 1
    h.destroy()


def check_matrix_on_grid(n, rp, ci, v, perm, grid, nrhs=1, relax=16, maxsup=64, refactor=False, make_comms=None, stats_out=None, unsym_symb=False):
    """Any CSR matrix (unpivoted LU must be stable for it) through the own pipeline on a Pr x Pc x Pz grid.  stats_out: list that
    receives every rank's sluamd_stats_t (after the factorisation) for assertions on the plan (levels, K-fused pairs)."""
    Pr, Pc, Pz = grid
    P = Pr * Pc * Pz
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    x1, info1, _ = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=relax, maxsup=maxsup)
    assert info1 == 0
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup, unsym=unsym_symb)     # unsym_symb: the exact unsymmetric structure (reference rules)
    sn_tree = symb.partition(Pz) if Pz > 1 else None
    comms = (make_comms or grid3d.local_comms)(Pr, Pc, Pz)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b

    # the reference's boundary: B distributed by block rows over layer 0 (uneven blocks, one of them empty when the layer has >= 3 ranks)
    nl0 = Pr * Pc
    cuts = np.linspace(0, n, nl0 + 1).astype(np.int64)
    if nl0 >= 3:
        cuts[1] = cuts[0]
    cuts[1:-1] += np.arange(1, nl0) % 3

    def rank_body(rank):
        h = grid3d.GridHandle.from_symbolic(symb, v, comms[rank], sn_tree)
        info = h.pdgstrf3d(0.0)
        if stats_out is not None:
            stats_out.append(dict(h.stats(), rank=rank))
        y = h.pdgstrs3d(xp)
        # distributed form (sluamd_pdgstrs3d_dist: pdReDistribute3d_B_to_X / X_to_B inside): my rows of b in, my rows of x out
        f0, f1 = (int(cuts[rank]), int(cuts[rank + 1])) if rank < nl0 else (0, 0)
        xl = h.pdgstrs3d_dist(b[f0:f1, :], f0, symb.perm_c)
        assert xl.shape == (f1 - f0, b.shape[1])
        if f1 > f0:
            assert np.abs(xl - y[symb.perm_c, :][f0:f1, :]).max() <= 1e-12 * np.abs(y).max()
        xr = h.pdgstrs3d_dist(b[f0:f1, :], f0, symb.perm_c, perm_out=None)      # the reference's convention: rows of the PERMUTED solution
        if f1 > f0:
            assert np.abs(xr - y[f0:f1, :]).max() <= 1e-12 * np.abs(y).max()
        if nl0 >= 2:
            # B re-partitioned between two solves on the same handle: only the boundary between ranks 0 and 1 moves, so every other rank
            # (the rest of layer 0, all of the layers above) sees an unchanged (m_loc, fst_row, perm) -- the decision to rebuild the routing
            # plan must be collective all the same (a rank that skipped the rebuild left its peers waiting in the exchange)
            cuts2 = cuts.copy(); cuts2[1] = min(cuts2[1] + 5, cuts2[2])
            g0, g1 = (int(cuts2[rank]), int(cuts2[rank + 1])) if rank < nl0 else (0, 0)
            xl2 = h.pdgstrs3d_dist(b[g0:g1, :], g0, symb.perm_c)
            if g1 > g0:
                assert np.abs(xl2 - y[symb.perm_c, :][g0:g1, :]).max() <= 1e-12 * np.abs(y).max()
        if refactor:                    # device-side re-distribution + second factorisation reproduce the solution
            h.reset_values()
            assert h.pdgstrf3d(0.0) == 0
            y2 = h.pdgstrs3d(xp)
            assert np.abs(y2 - y).max() <= 1e-12 * np.abs(y).max()
        h.destroy()
        return info, y

    out = grid3d.run_ranks(P, rank_body)
    for info, y in out:
        assert info == 0
        x = y[symb.perm_c, :]
        res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b)
        assert res < 1e-10
        assert np.abs(x - x1).max() <= 1e-10 * np.abs(x1).max()
    symb.free()


def forests_from_partition(tree, Pz, z):
    """The dtrf3Dpartition_t view a 1 x 1 x Pz rank would hold, from the heap-ordered tree id of every supernode
    (Symbolic.partition): forest Pz-1+z at level 0, then its ancestors; a rank idles at level l when z % 2^l != 0."""
    ml = int(np.log2(Pz)) + 1
    idx, t = [], Pz - 1 + z
    for _ in range(ml):
        idx.append(t); t = (t - 1) // 2
    return dict(maxLvl=ml, myTreeIdxs=np.array(idx, dtype=np.int32),
                myZeroTrIdxs=np.array([1 if z % (1 << l) else 0 for l in range(ml)], dtype=np.int32),
                nodeLists=[np.flatnonzero(tree == f).astype(np.int32) if np.any(tree == f) else None
                           for f in range(2 * Pz - 1)])


def shuffle_block_rows(fs, seed):
    """Permute the rows INSIDE every off-diagonal L block (index entries and values alike): the reference's symbfact keeps the
    row subscripts of a block in discovery order, not sorted."""
    rng = np.random.default_rng(seed)
    for k in range(fs.nsupers):
        a, e = int(fs.Lrowind_off[k]), int(fs.Lrowind_off[k + 1])
        if e - a < 2:
            continue
        li = fs.Lrowind[a:e]
        nb, nsupr = int(li[0]), int(li[1])
        w = int(fs.xsup[k + 1] - fs.xsup[k])
        V = fs.Lnzval[int(fs.Lnzval_off[k]):int(fs.Lnzval_off[k + 1])].reshape((nsupr, w), order="F")
        p, r0 = 2, 0
        for b in range(nb):
            g, nr = int(li[p]), int(li[p + 1])
            if g != k and nr > 1:
                perm = rng.permutation(nr)
                li[p + 2:p + 2 + nr] = li[p + 2:p + 2 + nr][perm]
                V[r0:r0 + nr, :] = V[r0:r0 + nr, :][perm, :]
            p += 2 + nr; r0 += nr
        fs.Lnzval[int(fs.Lnzval_off[k]):int(fs.Lnzval_off[k + 1])] = V.reshape(-1, order="F")


def check_wide_supernodes(N, maxsup, Pz, orc, shuffle=False):
    """Reference-format panels with supernodes of 257..512 columns (sp_ienv(3) <= MAX_SUPER_SIZE = 512) through the view
    path on a 1 x 1 x Pz grid: every L/U value against the CPU oracle's factorisation of the same store, then a solve.
    Layer z > 0 starts with zeroed ancestor panels (pddistribute3d / zeroSetLU)."""
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(N)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=maxsup)
    symb.distribute_host(v)
    fs0 = symb.flat_store()
    if shuffle:
        shuffle_block_rows(fs0, N)
    assert np.diff(fs0.xsup).max() > 256 or maxsup <= 256
    o = orc.LUStore(fs0.n, fs0.xsup, fs0.Lrowind_off, fs0.Lrowind, fs0.Lnzval_off, fs0.Lnzval.copy(), fs0.Ufstnz_off,
                    fs0.Ufstnz, fs0.Unzval_off, fs0.Unzval.copy())
    orc.dfactor(o)
    tree = symb.partition(Pz) if Pz > 1 else np.zeros(symb.nsupers, dtype=np.int32)
    comms = grid3d.local_comms(1, 1, Pz)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 3)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    scale = np.abs(v).max()

    def rank_body(z):
        fs = symb.flat_store()
        if shuffle:
            shuffle_block_rows(fs, N)
        fs.grid, fs.coords = (1, 1, Pz), (0, 0, z)
        fs._build_view()
        fr = forests_from_partition(tree, Pz, z) if Pz > 1 else None
        mine = np.zeros(symb.nsupers, dtype=bool)
        if Pz > 1:
            for l, t in enumerate(fr["myTreeIdxs"]):
                if not fr["myZeroTrIdxs"][l]:
                    mine |= tree == t
            for k in np.flatnonzero(tree < Pz - 1) if z else []:      # ancestors start at zero on layers z > 0
                fs.Lnzval[fs.Lnzval_off[k]:fs.Lnzval_off[k + 1]] = 0.0
                fs.Unzval[fs.Unzval_off[k]:fs.Unzval_off[k + 1]] = 0.0
        else:
            mine[:] = True
        h = grid3d.GridHandle.from_store(fs, fr, comms[z])
        assert h.pdgstrf3d(0.0) == 0
        h.copy_to_host(fs)
        for k in np.flatnonzero(mine):
            a, e = fs.Lnzval_off[k], fs.Lnzval_off[k + 1]
            assert np.abs(fs.Lnzval[a:e] - o.Lnzval[a:e]).max() <= 1e-12 * scale, ("L", z, k, np.abs(fs.Lnzval[a:e] - o.Lnzval[a:e]).max())
            a, e = fs.Unzval_off[k], fs.Unzval_off[k + 1]
            if e > a:
                assert np.abs(fs.Unzval[a:e] - o.Unzval[a:e]).max() <= 1e-12 * scale, ("U", z, k)
        y = h.pdgstrs3d(xp)
        h.destroy()
        return y

    for y in grid3d.run_ranks(Pz, rank_body):
        x = y[symb.perm_c, :]
        assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b) < 1e-10
        assert np.abs(x - xt).max() <= 1e-8 * np.abs(xt).max()
    symb.free()


def check_own_pipeline_complex16(Pz, N=12, leaf=27, relax=16, maxsup=64, Pr=1, Pc=1, make_comms=None):
    """complex16 through the library's own symbolic factorisation + device-side distribution on a Pr x Pc x Pz grid: residual on the
    original system and agreement with the single-rank solution."""
    n, rp, ci, v = matgen.poisson3d(N)
    v = matgen.complex_shift(v, rp, ci, seed=2)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
    rng = np.random.default_rng(3)
    xt = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    b = np.zeros_like(xt)
    for i in range(n):
        b[i, :] = v[rp[i]:rp[i + 1]] @ xt[ci[rp[i]:rp[i + 1]], :]
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    h1 = driver.LUHandle.from_symbolic(symb, v)
    assert h1.pzgstrf3d(0.0) == 0
    x1 = h1.pzgstrs3d(xp)[symb.perm_c, :]
    h1.destroy()
    sn_tree = symb.partition(Pz) if Pz > 1 else None
    comms = (make_comms or grid3d.local_comms)(Pr, Pc, Pz)

    nl0 = Pr * Pc
    cuts = np.linspace(0, n, nl0 + 1).astype(np.int64)

    def rank_body(rank):
        h = grid3d.GridHandle.from_symbolic(symb, v, comms[rank], sn_tree)
        info = h.pdgstrf3d(0.0)
        y = h.pdgstrs3d(xp)
        f0, f1 = (int(cuts[rank]), int(cuts[rank + 1])) if rank < nl0 else (0, 0)      # pzgstrs3d's own boundary: B distributed over layer 0
        xl = h.pdgstrs3d_dist(b[f0:f1, :], f0, symb.perm_c)
        if f1 > f0:
            assert np.abs(xl - y[symb.perm_c, :][f0:f1, :]).max() <= 1e-12 * np.abs(y).max()
        h.destroy()
        return info, y

    for info, y in grid3d.run_ranks(Pr * Pc * Pz, rank_body):
        assert info == 0
        x = y[symb.perm_c, :]
        assert np.abs(x - xt).max() <= 1e-10 * np.abs(xt).max()
        assert np.abs(x - x1).max() <= 1e-10 * np.abs(x1).max()
    symb.free()
