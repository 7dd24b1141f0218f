"""Parity of the HIP hot path (through the C ABI) against the golden fixtures recorded from the real reference
and against the CPU oracle.  Tolerances: L/U values 1e-12*||A||_max (summation order differs from the CPU loop:
MFMA k-blocking + fp64 atomics), solutions 1e-10 relative (BASELINE.json north_star)."""
import os
import numpy as np
import pytest
import oracle as orc
from superlu_dist_amd import _lib, driver, matgen

pytestmark = pytest.mark.gpu

CASES_1RANK = ["g20_1x1x1", "g20_1x1x1_nrhs3", "poisson8_nd", "poisson10_nd", "unsym300", "unsym120_tiny",
               # complex16 (pzgstrf3d / pzgstrs3d): cg20.cua = BASELINE.json config 5's matrix family
               "z_cg20_1x1x1", "z_cg20_1x1x1_nrhs2", "z_poisson8_nd", "z_unsym200", "z_grid24_nd"]


def test_mfma_f64_fragment_layout():
    import ctypes as C
    L = _lib.load()
    rng = np.random.default_rng(1)
    A = rng.standard_normal((16, 4)); B = rng.standard_normal((4, 16))   # asymmetric on purpose
    D = np.zeros((16, 16))
    Ac, Bc = np.ascontiguousarray(A), np.ascontiguousarray(B)
    rc = L.sluamd_mfma_selftest(Ac.ctypes.data_as(_lib.P_dbl), Bc.ctypes.data_as(_lib.P_dbl), D.ctypes.data_as(_lib.P_dbl))
    assert rc == 0
    assert np.abs(D - A @ B).max() < 1e-13


def _factor(g, deterministic=False):
    st = driver.FlatStore.from_golden(g, 0, "pre")
    h = driver.LUHandle.from_store(st, replace_tiny=bool(g["r0__ReplaceTinyPivot"][0]), deterministic=deterministic)
    info = h.pdgstrf3d(float(g["r0__thresh"][0]))
    h.copy_to_host()
    return st, h, info


@pytest.mark.parametrize("case", CASES_1RANK)
def test_factor_matches_reference(golden, case):
    g = golden(case)
    st, h, info = _factor(g)
    assert info == int(g["r0__info"][0])
    assert h.stats()["tiny_pivots"] == int(g["r0__TinyPivots"][0])
    scale = max(np.abs(g["r0__Lnzval_pre"]).max(), np.abs(g["r0__Unzval_pre"]).max())
    assert np.abs(st.Lnzval - g["r0__Lnzval_post"]).max() <= 1e-12 * scale
    assert np.abs(st.Unzval - g["r0__Unzval_post"]).max() <= 1e-12 * scale
    # and against the CPU oracle on the same input
    o = orc.LUStore.from_golden(g, 0, "pre")
    orc.dfactor(o, g["r0__forest0_nodeList"], bool(g["r0__ReplaceTinyPivot"][0]), float(g["r0__thresh"][0]))
    assert np.abs(st.Lnzval - o.Lnzval).max() <= 1e-12 * scale
    assert np.abs(st.Unzval - o.Unzval).max() <= 1e-12 * scale
    h.destroy()


@pytest.mark.parametrize("case", ["poisson10_nd", "unsym300", "g20_1x1x1", "z_grid24_nd"])
def test_balanced_xcd_ranges_on_every_bulk_launch(golden, case, monkeypatch):
    """The bulk tile lists cut into eight XCD ranges of equal modelled cost (LevelSched::x_off, supernodes listed longest tiles first) are used from 1 024 tiles per
    launch by default -- no fixture has that many.  SLUAMD_BALANCE_MIN_TILES=1: every bulk launch, including those of fewer than eight tiles (empty ranges): the
    factors of the reference."""
    monkeypatch.setenv("SLUAMD_BALANCE_MIN_TILES", "1")
    g = golden(case)
    st, h, info = _factor(g)
    assert info == int(g["r0__info"][0])
    scale = max(np.abs(g["r0__Lnzval_pre"]).max(), np.abs(g["r0__Unzval_pre"]).max())
    assert np.abs(st.Lnzval - g["r0__Lnzval_post"]).max() <= 1e-12 * scale
    assert np.abs(st.Unzval - g["r0__Unzval_post"]).max() <= 1e-12 * scale
    h.destroy()


@pytest.mark.parametrize("case", CASES_1RANK)
def test_solve_matches_reference(golden, case):
    g = golden(case)
    st, h, info = _factor(g)
    n = st.n
    pr, pc = g["r0__perm_r"], g["r0__perm_c"]
    s = 0
    while f"r0__solve{s}_B_in" in g:
        nrhs = int(g[f"r0__solve{s}_nrhs"][0])
        B = g[f"r0__solve{s}_B_in"].reshape((n, nrhs), order="F")
        X = g[f"r0__solve{s}_B_out"].reshape((n, nrhs), order="F")
        xp = np.zeros((n, nrhs), order="F", dtype=B.dtype); xp[pc[pr], :] = B
        got = h.pdgstrs3d(xp)
        assert np.abs(got - X).max() <= 1e-10 * max(1.0, np.abs(X).max())
        s += 1
    assert s >= 1
    h.destroy()


def test_merged_chain_groups_of_the_sweeps_give_the_ungrouped_solution():
    """SLUAMD_SOLVE_GROUPS=1 (opt-in): chains of up to four separator supernodes solved as ONE node of the sweeps through the inverse of their block triangle
    (built during pdgstrf3d by batched dense products) -- same factors, fewer launches, the solution of the ungrouped sweeps to rounding; a block of right-hand
    sides too wide for the group strips' staging takes the ungrouped schedule."""
    N = 24
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    rng = np.random.default_rng(5)
    out = {}
    saved = os.environ.get("SLUAMD_SOLVE_GROUPS")
    for grouped in (False, True):
        os.environ["SLUAMD_SOLVE_GROUPS"] = "1" if grouped else "0"       # explicit both ways: the suite may be run with the variable set
        try:
            symb = driver.Symbolic(n, rp, ci, perm, relax=32, maxsup=64)
            h = driver.LUHandle.from_symbolic(symb, v)
            assert h.pdgstrf3d(0.0) == 0
            xs, launches = [], None
            for nrhs in (1, 3, 200):
                xt = np.asfortranarray(np.random.default_rng(nrhs).standard_normal((n, nrhs)))
                b = np.column_stack([matgen.csr_matvec(n, rp, ci, v, xt[:, q]) for q in range(nrhs)])
                pc = np.asarray(symb.perm_c)
                bp = np.zeros_like(b); bp[pc] = b                   # the factored system: P A P^T
                x = h.pdgstrs3d(bp)[pc]
                assert np.abs(x - xt).max() <= 1e-10 * max(1.0, np.abs(xt).max()), (grouped, nrhs, np.abs(x - xt).max())
                xs.append(x)
                if launches is None:
                    launches = h.stats()["solve_launches"]
            out[grouped] = (xs, launches)
            h.destroy(); symb.free()
        finally:
            if saved is None: os.environ.pop("SLUAMD_SOLVE_GROUPS", None)
            else: os.environ["SLUAMD_SOLVE_GROUPS"] = saved
    for a, b in zip(out[False][0], out[True][0]):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
    assert out[True][1] < out[False][1], (out[True][1], out[False][1])      # the groups were found and used


@pytest.mark.parametrize("case", ["poisson10_nd_diaginv", "unsym300_diaginv"])
def test_diagonal_inverses_match_the_reference(golden, case):
    """Linv / Uinv of every diagonal block (k_full_inv / k_full_inv64 from the 32 x 32 inverses of the diagonal kernels) against what the reference's
    pdCompute_Diag_Inv left in Llu->Linv_bc_ptr / Uinv_bc_ptr (dtrtri; recorded from a DiagInv = YES run of the reference built on LAPACK, oracle/_ref_mkl):
    1e-10 relative to the largest entry of each inverse, explicit zeros in the other triangle, and the factors of that run as usual."""
    g = golden(case)
    st, h, info = _factor(g)
    assert info == int(g["r0__info"][0])
    scale = max(np.abs(g["r0__Lnzval_pre"]).max(), np.abs(g["r0__Unzval_pre"]).max())
    assert np.abs(st.Lnzval - g["r0__Lnzval_post"]).max() <= 1e-12 * scale
    xs, off = g["r0__xsup"], g["r0__diaginv_off"]
    checked = 0
    for k in range(len(xs) - 1):
        ns = int(xs[k + 1] - xs[k])
        if off[k + 1] == off[k]:
            continue
        Lr = g["r0__Linv"][off[k]:off[k + 1]].reshape((ns, ns), order="F"); Ur = g["r0__Uinv"][off[k]:off[k + 1]].reshape((ns, ns), order="F")
        Li, Ui = h.diag_inv(k, ns)
        assert np.abs(Li - Lr).max() <= 1e-10 * max(1.0, np.abs(Lr).max()), k
        assert np.abs(Ui - Ur).max() <= 1e-10 * max(1.0, np.abs(Ur).max()), k
        assert np.all(np.triu(Li, 1) == 0) and np.all(np.tril(Ui, -1) == 0) and np.all(np.diag(Li) == 1.0)
        checked += 1
    assert checked == len(xs) - 1
    # the inverses belong to FACTORED blocks: after new values went up the call refuses until the next factorisation
    h.set_values(driver.FlatStore.from_golden(g, 0, "pre"))
    with pytest.raises(RuntimeError):
        h.diag_inv(0, int(xs[1] - xs[0]))
    h.destroy()


def test_deterministic_mode_is_bitwise_reproducible(golden):
    g = golden("poisson10_nd")
    a, h1, _ = _factor(g, deterministic=True)
    b, h2, _ = _factor(g, deterministic=True)
    assert np.array_equal(a.Lnzval, b.Lnzval) and np.array_equal(a.Unzval, b.Unzval)
    h1.destroy(); h2.destroy()


def test_refactor_same_pattern(golden):
    g = golden("unsym300")
    st, h, _ = _factor(g)
    first = st.Lnzval.copy()
    st2 = driver.FlatStore.from_golden(g, 0, "pre")
    h.set_values(st2)
    h.pdgstrf3d(float(g["r0__thresh"][0]))
    h.copy_to_host(st2)
    scale = np.abs(g["r0__Lnzval_pre"]).max()
    assert np.abs(st2.Lnzval - first).max() <= 1e-12 * scale
    h.destroy()


def test_zero_pivot_reports_info():
    # 2x2 block of zeros on the diagonal -> exact zero pivot at column 1 (1-based), pdgstrf2.c:568-571
    n = 4
    rp = np.array([0, 2, 4, 6, 8], dtype=np.int32)
    ci = np.array([0, 1, 0, 1, 2, 3, 2, 3], dtype=np.int32)
    v = np.array([0.0, 1.0, 1.0, 0.0, 2.0, 1.0, 1.0, 2.0])
    symb = driver.Symbolic(n, rp, ci, None, relax=1, maxsup=4)
    h = driver.LUHandle.from_symbolic(symb, v)
    info = h.pdgstrf3d(0.0)
    assert info == 1
    h.destroy()


@pytest.mark.parametrize("N,leaf,relax,maxsup,nrhs", [(8, 16, 16, 64, 1), (12, 27, 32, 128, 3), (16, 64, 64, 256, 1)])
def test_own_pipeline_poisson_residual(N, leaf, relax, maxsup, nrhs):
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=relax, maxsup=maxsup)
    assert info == 0
    res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b)
    assert res < 1e-10
    assert np.abs(x - xt).max() < 1e-9
    # CPU oracle on the same store: solutions agree to 1e-10
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    symb.distribute_host(v)
    fs = symb.flat_store()
    o = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz,
                    fs.Unzval_off, fs.Unzval)
    orc.dfactor(o)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    xo = orc.dsolve(o, xp)[symb.perm_c, :]
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()


def test_small_tile_configuration_agrees(golden, monkeypatch):
    """Forcing the 64x64 Schur tiles everywhere gives the same factors as the mixed 128/64 configuration."""
    N = 20
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    outs = []
    for force_small in (False, True):
        if force_small:
            monkeypatch.setenv("SLUAMD_NO_BIG_TILES", "1")
        symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
        symb.distribute_host(v)
        fs = symb.flat_store()
        h = driver.LUHandle.from_store(fs)
        assert h.pdgstrf3d(0.0) == 0
        h.copy_to_host(); h.destroy()
        outs.append(fs)
    assert np.abs(outs[0].Lnzval - outs[1].Lnzval).max() <= 1e-12 * 6.0
    assert np.abs(outs[0].Unzval - outs[1].Unzval).max() <= 1e-12 * 6.0


@pytest.mark.parametrize("N,leaf,relax,maxsup", [(24, 64, 64, 256), (20, 27, 20, 200), (22, 64, 48, 130), (28, 64, 64, 128)])
def test_large_supernodes_match_oracle(N, leaf, relax, maxsup, monkeypatch):
    """Wide supernodes: blocked diagonal LU (ns > 128), multi-block MFMA TRSMs, 64x64 Schur tiles with several
    row/column tiles per block pair; every L/U value against the CPU oracle."""
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(N)
    v = v * (1.0 + 0.3 * rng.random(v.size))          # unsymmetric values on the symmetric pattern
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    symb.distribute_host(v)
    fs = symb.flat_store()
    assert np.diff(fs.xsup).max() > 128 or maxsup <= 130
    o = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz,
                    fs.Unzval_off, fs.Unzval)
    h = driver.LUHandle.from_store(fs)
    info = h.pdgstrf3d(0.0)
    h.copy_to_host()
    orc.dfactor(o)
    scale = np.abs(v).max()
    assert info == 0
    assert np.abs(fs.Lnzval - o.Lnzval).max() <= 1e-12 * scale
    assert np.abs(fs.Unzval - o.Unzval).max() <= 1e-12 * scale
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    x = h.pdgstrs3d(xp)[symb.perm_c, :]
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b) < 1e-10
    fused = h.stats()["reserved_i"]
    h.destroy()
    if N == 28:
        # the top separator is a dense chain of >= 6 supernodes: consecutive pieces are K-fused (one scatter for two
        # supernodes' updates); the same factorisation with fusion disabled must agree to summation-order accuracy
        assert fused >= 2
        monkeypatch.setenv("SLUAMD_NO_FUSE", "1")
        symb.distribute_host(v)
        fs2 = symb.flat_store()
        h2 = driver.LUHandle.from_store(fs2)
        assert h2.stats()["reserved_i"] == 0
        assert h2.pdgstrf3d(0.0) == 0
        h2.copy_to_host()
        assert np.abs(fs2.Lnzval - fs.Lnzval).max() <= 1e-12 * scale
        assert np.abs(fs2.Unzval - fs.Unzval).max() <= 1e-12 * scale
        h2.destroy()


@pytest.mark.parametrize("shape,leaf,relax,maxsup", [((40, 40, 1), 16, 16, 64), ((12, 12, 12), 27, 32, 128)])
def test_own_pipeline_complex16(shape, leaf, relax, maxsup):
    """pzgssvx3d-equivalent on a complex grid operator (cg20-style 2-D 5-pt and a 3-D 7-pt): device-side distribution of
    complex values, pzgstrf3d (split-plane MFMA Schur kernel), pzgstrs3d; residual and solution against x_true."""
    nx, ny, nz = shape
    n, rp, ci, v = matgen.poisson3d(0, nx, ny, nz)
    v = matgen.complex_shift(v, rp, ci, seed=2)
    perm = matgen.nd_perm_grid3d(nx, ny, nz, leaf=leaf)
    rng = np.random.default_rng(0)
    xt = (rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2)))
    b = matgen.csr_matvec(n, rp, ci, v, xt)
    x, info, st = driver.pzgssvx3d(n, rp, ci, v, b, perm, relax=relax, maxsup=maxsup)
    assert info == 0
    res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b)
    assert res < 1e-10
    assert np.abs(x - xt).max() < 1e-9 * np.abs(xt).max()


@pytest.mark.parametrize("complex16", [False, True])
def test_tile_records_match_table_lookups(complex16, monkeypatch):
    """The per-tile records of the Schur tiles (written by a plan-time pass of the kernel, read by every later factorisation) against the
    same factorisation with the tiles chasing the tables (SLUAMD_NO_TILE_MAPS): same factors to summation-order accuracy, both tile
    configurations in use, a second factorisation on the same handle (records reused) included."""
    N = 24
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(3)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    if complex16:
        v = matgen.complex_shift(v, rp, ci, seed=4)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    symb = driver.Symbolic(n, rp, ci, perm, relax=32, maxsup=160)
    out = []
    xt = rng.standard_normal((n, 2)) + (1j * rng.standard_normal((n, 2)) if complex16 else 0.0)
    b = matgen.csr_matvec(n, rp, ci, v, xt)
    for no_maps in (False, True):
        if no_maps:
            monkeypatch.setenv("SLUAMD_NO_TILE_MAPS", "1")
        h = driver.LUHandle.from_symbolic(symb, v)
        assert h.pdgstrf3d(0.0) == 0
        by0 = h.stats()["bytes_device"]
        if not no_maps:
            h.reset_values()
            assert h.pdgstrf3d(0.0) == 0          # the records were built by the first factorisation
            assert h.stats()["bytes_device"] == by0
        xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
        x = h.pdgstrs3d(xp)[symb.perm_c, :]
        assert np.abs(x - xt).max() < 1e-9 * np.abs(xt).max()
        out.append((x, by0))
        h.destroy()
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * np.abs(out[1][0]).max()
    if "emul" in os.environ.get("SLUAMD_LIB", ""):
        assert out[0][1] >= out[1][1]             # CPU test build: no records (its Schur restatement reads the tables every time)
    else:
        assert out[0][1] > out[1][1]              # the records are device memory the statistics report
    symb.free()


@pytest.mark.parametrize("shape,leaf,relax,maxsup,nrhs", [((30, 30, 1), 8, 4, 6, 1), ((30, 30, 1), 16, 12, 12, 3), ((40, 40, 1), 16, 24, 24, 7),
                                                          ((40, 40, 1), 16, 48, 48, 5), ((12, 12, 12), 27, 32, 100, 9), ((14, 14, 14), 27, 64, 200, 2)])
def test_complex16_values_match_oracle(shape, leaf, relax, maxsup, nrhs):
    """Every complex16 kernel variant against the CPU oracle, value by value: supernode widths of at most 6 / 12 / 24 / 48 columns (the one-wave
    diagonal LU with 8 / 16 / 32 / 2 x 32 columns in registers), 100 and 200 (workgroup LU; Schur tiles of 32 x 64 and 64 x 128 on the real
    embedding), and the blocked diagonal solves with 1..9 right-hand sides (one wave per right-hand side, four at a time)."""
    nx, ny, nz = shape
    n, rp, ci, v = matgen.poisson3d(0, nx, ny, nz)
    v = matgen.complex_shift(v, rp, ci, seed=5)
    perm = matgen.nd_perm_grid3d(nx, ny, nz, leaf=leaf)
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    symb.distribute_host(v.real); fr = symb.flat_store()      # the distribution is linear in the values: real and imaginary parts apart
    symb.distribute_host(v.imag); fi = symb.flat_store()
    assert np.diff(fr.xsup).max() <= maxsup
    Lz = fr.Lnzval + 1j * fi.Lnzval; Uz = fr.Unzval + 1j * fi.Unzval
    o = orc.LUStore(fr.n, fr.xsup, fr.Lrowind_off, fr.Lrowind, fr.Lnzval_off, Lz, fr.Ufstnz_off, fr.Ufstnz, fr.Unzval_off, Uz)
    fs = driver.FlatStore(fr.n, fr.xsup, fr.Lrowind_off, fr.Lrowind, fr.Lnzval_off, Lz, fr.Ufstnz_off, fr.Ufstnz, fr.Unzval_off, Uz)
    h = driver.LUHandle.from_store(fs)
    assert h.z
    assert h.pdgstrf3d(0.0) == 0
    h.copy_to_host()
    info, _, _ = orc.dfactor(o)
    assert info == 0
    scale = np.abs(v).max()
    assert np.abs(fs.Lnzval - o.Lnzval).max() <= 1e-12 * scale
    assert np.abs(fs.Unzval - o.Unzval).max() <= 1e-12 * scale
    rng = np.random.default_rng(nrhs)
    xp = np.asfortranarray(rng.standard_normal((n, nrhs)) + 1j * rng.standard_normal((n, nrhs)))
    x = h.pdgstrs3d(xp)
    xo = orc.dsolve(o, xp)
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    h.destroy(); symb.free()


@pytest.mark.parametrize("nrhs", [1, 3, 60])
def test_distributed_boundary_on_a_single_rank(nrhs):
    """sluamd_pdgstrs3d_dist (pdgstrs3d's own boundary: original row order in, pdReDistribute3d_B_to_X / X_to_B inside) on a single-rank
    handle gives what the permuted-vector entry point gives; a wrong row range is an error, not a wrong answer."""
    N = 10
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, nrhs)
    symb = driver.Symbolic(n, rp, ci, perm, relax=16, maxsup=64)
    h = driver.LUHandle.from_symbolic(symb, v)
    assert h.pdgstrf3d(0.0) == 0
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    x1 = h.pdgstrs3d(xp)[symb.perm_c, :]
    x2 = h.pdgstrs3d_dist(b, symb.perm_c)
    assert np.abs(x2 - x1).max() <= 1e-13 * np.abs(x1).max()
    assert np.abs(x2 - xt).max() <= 1e-9
    with pytest.raises(RuntimeError, match="row range"):
        h.pdgstrs3d_dist(b[:-3, :], symb.perm_c)
    h.destroy(); symb.free()


def test_several_exact_zero_pivots_report_the_first_column():
    """A structurally non-singular matrix whose unpivoted elimination meets several exact zero pivots, in different supernodes and
    levels of the elimination DAG: `info` is the smallest 1-based column with a zero pivot -- what the reference's documentation of
    `info` says (pdgstrf2.c:493-497; its code keeps the one met LAST in execution order, :568-571: DESIGN.md section 1) -- the same on
    every run, whatever order the workgroups reach them in."""
    n = 96
    rows, cols, vals = [], [], []
    for i in range(n):                     # block-diagonal 2 x 2 blocks [[0, 1], [1, 0]] at three places, identity-like elsewhere
        rows.append(i); cols.append(i); vals.append(2.0)
    rp = np.arange(n + 1, dtype=np.int32); ci = np.arange(n, dtype=np.int32); v = np.array(vals)
    import scipy.sparse as sp
    A = sp.lil_matrix((n, n)); A.setdiag(2.0)
    for i in range(n - 1):
        A[i, i + 1] = -0.5; A[i + 1, i] = -0.5
    for z in (10, 41, 77):                 # exact zero pivots: a_zz = 0 and nothing earlier updates it
        A[z, z] = 0.0
        if z > 0:
            A[z, z - 1] = 0.0; A[z - 1, z] = 0.0
    A = A.tocsr(); A.sort_indices()
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    perm = np.arange(n, dtype=np.int32)
    for rep in range(3):
        symb = driver.Symbolic(n, rp, ci, perm, relax=4, maxsup=8)
        h = driver.LUHandle.from_symbolic(symb, v)
        info = h.pdgstrf3d(0.0)
        expected = min(int(symb.perm_c[z]) for z in (10, 41, 77)) + 1
        assert info == expected, (info, expected)
        h.destroy(); symb.free()
    # SLUAMD_INFO_LAST=1: the rule the reference's CODE implements -- Local_Dgstrf2 overwrites *info at every zero pivot (pdgstrf2.c:568-571), so a rank
    # keeps the one it met LAST in elimination order (here: the largest column), pdgstrf3d takes the MIN over the ranks (pdgstrf3d.c:388-392)
    os.environ["SLUAMD_INFO_LAST"] = "1"
    try:
        symb = driver.Symbolic(n, rp, ci, perm, relax=4, maxsup=8)
        h = driver.LUHandle.from_symbolic(symb, v)
        info = h.pdgstrf3d(0.0)
        assert info == max(int(symb.perm_c[z]) for z in (10, 41, 77)) + 1, info
        h.destroy(); symb.free()
    finally:
        del os.environ["SLUAMD_INFO_LAST"]
    # the same rule through the ABI (sluamd_options_t::info_rule = SLUAMD_INFO_REFERENCE: what bindings/superlu_dist/sluamd_binding.c sets)
    symb = driver.Symbolic(n, rp, ci, perm, relax=4, maxsup=8)
    h = driver.LUHandle.from_symbolic(symb, v, info_rule=1)
    info = h.pdgstrf3d(0.0)
    assert info == max(int(symb.perm_c[z]) for z in (10, 41, 77)) + 1, info
    h.destroy(); symb.free()


@pytest.mark.parametrize("kind", ["stencil_unsym", "random_unsym"])
def test_unsymmetric_symbolic_structure_on_the_device(kind):
    """sluamd_dsymbfact_unsym (the exact unsymmetric structure with the reference's supernode rules, pinned to the real symbfact by
    tests/test_symbolic_parity.py) through the device path: device-side distribution into ragged skylines, pdgstrf3d, pdgstrs3d -- every
    factor value against the CPU oracle on the same store, the solution against x_true; the merged Schur tiles (rows across the L blocks of a
    destination panel, columns across the U blocks of a destination row) are in the lists of these irregular structures."""
    if kind == "stencil_unsym":
        n, rp, ci, v = matgen.stencil3d_unsym(14, drop=0.3, seed=4); perm = matgen.nd_perm_grid3d(14, 14, 14, leaf=27)
    else:
        n, rp, ci, v = matgen.random_unsym(500, 0.02, seed=9); perm = None
    symb = driver.Symbolic(n, rp, ci, perm, relax=24, maxsup=96, unsym=True)
    symb.distribute_host(v)
    fs = symb.flat_store()
    o = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz, fs.Unzval_off, fs.Unzval)
    h = driver.LUHandle.from_symbolic(symb, v)
    assert h.pdgstrf3d(0.0) == 0
    fs2 = symb.flat_store()
    h2 = driver.LUHandle.from_store(fs2)             # the same structure through the caller's-view path
    assert h2.pdgstrf3d(0.0) == 0
    h2.copy_to_host()
    orc.dfactor(o)
    scale = np.abs(v).max()
    assert np.abs(fs2.Lnzval - o.Lnzval).max() <= 1e-12 * scale and np.abs(fs2.Unzval - o.Unzval).max() <= 1e-12 * scale
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    for hh in (h, h2):
        x = hh.pdgstrs3d(xp)[symb.perm_c, :]
        assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b) < 1e-10
        assert np.abs(x - xt).max() < 1e-8 * max(1.0, np.abs(xt).max())
    h.destroy(); h2.destroy(); symb.free()
