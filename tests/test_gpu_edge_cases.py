"""Edge cases of the hot path through the C ABI: degenerate structures, sizes at the limits, argument errors."""
import numpy as np
import pytest
import oracle as orc
from superlu_dist_amd import _lib, driver, matgen

pytestmark = pytest.mark.gpu


def _csr(dense):
    n = dense.shape[0]
    rp = [0]; ci = []; v = []
    for i in range(n):
        nz = np.nonzero(dense[i])[0]
        ci += nz.tolist(); v += dense[i, nz].tolist(); rp.append(len(ci))
    return n, np.array(rp, dtype=np.int32), np.array(ci, dtype=np.int32), np.array(v)


def _solve_and_check(n, rp, ci, v, perm=None, relax=4, maxsup=32, nrhs=1, tol=1e-10):
    rng = np.random.default_rng(1)
    xt = rng.standard_normal((n, nrhs)) + (1j * rng.standard_normal((n, nrhs)) if np.iscomplexobj(v) else 0)
    b = matgen.csr_matvec(n, rp, ci, v, xt)
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=relax, maxsup=maxsup)
    assert info == 0
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) <= tol * np.linalg.norm(b)
    return st


def test_one_by_one_matrix():
    _solve_and_check(*_csr(np.array([[4.0]])))


def test_diagonal_matrix_all_singleton_supernodes():
    st = _solve_and_check(*_csr(np.diag(np.arange(1.0, 41.0))), relax=1, maxsup=1)
    assert st["flops_schur_exact"] == 0.0          # no off-diagonal block anywhere: every Schur launch is empty


def test_dense_matrix_is_one_supernode():
    rng = np.random.default_rng(2)
    A = rng.standard_normal((96, 96)) + 96 * np.eye(96)
    _solve_and_check(*_csr(A), relax=128, maxsup=128)


def test_arrow_matrix_long_panels_tiny_supernodes():
    n = 300
    A = np.eye(n) * 10.0
    A[-1, :] = 1.0; A[:, -1] = 1.0; A[-1, -1] = 400.0
    _solve_and_check(*_csr(A), relax=1, maxsup=8)


def test_supernode_width_exactly_256_and_multi_rhs():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 300)) * 0.1 + 300 * np.eye(300)
    st = _solve_and_check(*_csr(A), relax=256, maxsup=256, nrhs=7)
    assert st["nnz_L"] > 0


@pytest.mark.parametrize("maxsup,nrhs", [(32, 300), (256, 100)])
def test_many_right_hand_sides(maxsup, nrhs):
    """x_k is staged in LDS by the sweeps: narrow supernodes x 300 right-hand sides need more than the default 64 KiB of dynamic
    LDS in the 256-thread kernels; 256-wide supernodes x 100 right-hand sides are solved in several column chunks."""
    N = 12
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    _solve_and_check(n, rp, ci, v, perm=perm, relax=16, maxsup=maxsup, nrhs=nrhs)


def test_unsymmetric_values_wide_range_of_supernode_sizes():
    n, rp, ci, v = matgen.random_unsym(700, 0.01, seed=8)
    _solve_and_check(n, rp, ci, v, relax=8, maxsup=48, nrhs=3)


def test_nrhs_zero_is_a_noop_and_negative_is_rejected():
    n, rp, ci, v = matgen.poisson3d(4)
    symb = driver.Symbolic(n, rp, ci, None, relax=4, maxsup=16)
    h = driver.LUHandle.from_symbolic(symb, v)
    assert h.pdgstrf3d(0.0) == 0
    L = _lib.load()
    x = np.ones(n)
    assert L.sluamd_pdgstrs3d(h._h, x.ctypes.data_as(_lib.P_dbl), n, 0) == 0        # nrhs = 0
    assert np.all(x == 1.0)
    assert L.sluamd_pdgstrs3d(h._h, x.ctypes.data_as(_lib.P_dbl), n, -1) != 0       # pdgstrs3d: info = -9 for nrhs < 0
    assert L.sluamd_pdgstrs3d(h._h, x.ctypes.data_as(_lib.P_dbl), n - 1, 1) != 0    # ldx < n
    assert b"solve" in L.sluamd_last_error()
    h.destroy()


def test_dense_supernode_of_300_columns_through_the_own_symbolic_path():
    """SUPERLU_MAXSUP may go up to 512 in the reference (sp_ienv.c).  One 300-column supernode (dense matrix, relax = maxsup = 300)
    through sluamd_dCreateLUHandleFromSymb: refined into two pieces internally, A distributed on the device into the pieces."""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((300, 300)) * 0.1 + 300 * np.eye(300)
    n, rp, ci, v = _csr(A)
    symb = driver.Symbolic(n, rp, ci, None, relax=300, maxsup=300)
    if np.diff(symb.xsup()).max() <= 256:
        pytest.skip("symbolic did not produce a wide supernode")
    h = driver.LUHandle.from_symbolic(symb, v)
    assert h.pdgstrf3d(0.0) == 0
    b = rng.standard_normal((n, 2))
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    x = h.pdgstrs3d(xp)[symb.perm_c, :]
    assert np.abs(A @ x - b).max() <= 1e-10 * np.abs(b).max()
    h.destroy()


def test_one_wide_supernode_on_an_xy_layer():
    """A dense 300-column matrix = one supernode of 300 columns on a 2 x 1 layer: both of its pieces live on its owners, the diagonal
    block is split between them through the refined U slot."""
    from superlu_dist_amd import grid3d
    rng = np.random.default_rng(4)
    A = rng.standard_normal((300, 300)) * 0.1 + 300 * np.eye(300)
    n, rp, ci, v = _csr(A)
    symb = driver.Symbolic(n, rp, ci, None, relax=300, maxsup=300)
    if np.diff(symb.xsup()).max() <= 256:
        pytest.skip("symbolic did not produce a wide supernode")
    b = rng.standard_normal((n, 2))
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    comms = grid3d.local_comms(2, 1, 1)

    def body(rank):
        h = grid3d.GridHandle.from_symbolic(symb, v, comms[rank], None)
        info = h.pdgstrf3d(0.0)
        y = h.pdgstrs3d(xp)
        h.destroy()
        return info, y

    for info, y in grid3d.run_ranks(2, body):
        assert info == 0
        x = y[symb.perm_c, :]
        assert np.abs(A @ x - b).max() <= 1e-10 * np.abs(b).max()


def test_dense_supernode_of_300_columns_through_the_view_path():
    """One 300-column supernode (a dense matrix with relax = maxsup = 300): refined into two pieces internally, factors
    returned in the caller's single 300 x 300 panel; against numpy's LU without pivoting (diagonally dominant)."""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((300, 300)) * 0.1 + 300 * np.eye(300)
    n, rp, ci, v = _csr(A)
    symb = driver.Symbolic(n, rp, ci, None, relax=300, maxsup=300)
    if np.diff(symb.xsup()).max() <= 256:
        pytest.skip("symbolic did not produce a wide supernode")
    symb.distribute_host(v)
    fs = symb.flat_store()
    h = driver.LUHandle.from_store(fs)
    assert h.pdgstrf3d(0.0) == 0
    b = rng.standard_normal((n, 2))
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    x = h.pdgstrs3d(xp)[symb.perm_c, :]
    assert np.abs(A @ x - b).max() <= 1e-10 * np.abs(b).max()
    h.destroy()


def test_malformed_structure_is_rejected(golden):
    g = golden("g20_1x1x1")
    st = driver.FlatStore.from_golden(g, 0, "pre")
    st.Lrowind = st.Lrowind.copy()                       # the fixture dict is cached for the whole session
    st.Lrowind[int(st.Lrowind_off[3]) + 1] += 5          # corrupt the LDA of panel 3
    st._build_view()
    with pytest.raises(RuntimeError, match="mismatch|malformed"):
        driver.LUHandle.from_store(st)


def test_factor_twice_after_device_reset_reproduces_the_solution():
    """deterministic=1 fixes the summation order of the FACTORISATION (bitwise test in test_gpu_parity); the solve
    accumulates lsum contributions with fp64 atomics, so solutions agree to rounding, not bitwise."""
    n, rp, ci, v = matgen.poisson3d(9)
    perm = matgen.nd_perm_grid3d(9, 9, 9, leaf=27)
    symb = driver.Symbolic(n, rp, ci, perm, relax=16, maxsup=64)
    h = driver.LUHandle.from_symbolic(symb, v, deterministic=True)
    b = np.ones((n, 1))
    h.pdgstrf3d(0.0); x1 = h.pdgstrs3d(b)
    h.reset_values(); h.pdgstrf3d(0.0); x2 = h.pdgstrs3d(b)
    assert np.abs(x1 - x2).max() <= 1e-13 * np.abs(x1).max()
    h.destroy()


@pytest.mark.parametrize("join_max", [1 << 30, 32, 4, 1])
def test_joined_links_of_the_sweeps_on_the_device(monkeypatch, join_max):
    """Round 4: one launch per level in the triangular sweeps of a 1 x 1 layer (k_sweep_join: the 64 x 64 blocks of the next level's diagonal inverses
    apply the adjacent level's updates to their own block of the right-hand side and ADD into zeroed rows; the regular units skip those rows / columns)
    against the two-launch links on the same matrix -- every level joined, the two forms mixed at three depths (SLUAMD_JOIN_MAX_NODES), three right-hand
    sides, supernodes of up to 256 columns (ten blocks), unsymmetric values.  Oracle-sized twin: tests/test_stream_order.py::test_joined_links_*."""
    import numpy as np
    from superlu_dist_amd import driver, matgen
    N = 28
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(17)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 3)
    monkeypatch.setenv("SLUAMD_SOLVE_JOIN", "0")
    x_ref, info, st0 = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=64, maxsup=256)
    assert info == 0
    monkeypatch.setenv("SLUAMD_SOLVE_JOIN", "1")
    monkeypatch.setenv("SLUAMD_JOIN_MAX_NODES", str(join_max))
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=64, maxsup=256)
    assert info == 0 and st["solve_launches"] < st0["solve_launches"]
    assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()
    assert np.abs(x - xt).max() <= 1e-9 * np.abs(xt).max()
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("nrhs", [2, 3, 4, 5, 9, 16])
def test_blocks_of_right_hand_sides_against_single_solves(nrhs):
    """Round 6: with nrhs >= 2 the update and diagonal-strip units of the sweeps take FOUR right-hand sides per pass over their factor entries (the reference's
    nrhs > 1 path is a GEMM, pdgstrs_lsum.c:414-960), from nrhs >= 4 on every level in the two-launch form: each column of a block solve must be what a
    solve of that column alone gives (same factors, nrhs = 1 path), on a problem with narrow, mid and 256-column supernodes and unsymmetric values."""
    import numpy as np
    from superlu_dist_amd import driver, matgen
    N = 26
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(23)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
    h = driver.LUHandle.from_symbolic(symb, v)
    assert h.pdgstrf3d(0.0) == 0
    xt = rng.standard_normal((n, nrhs))
    b = np.asfortranarray(np.column_stack([matgen.csr_matvec(n, rp, ci, v, xt[:, q:q + 1])[:, 0] for q in range(nrhs)]))
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    y = h.pdgstrs3d(xp.copy(order="F"))
    for q in range(nrhs):
        y1 = h.pdgstrs3d(np.asfortranarray(xp[:, q:q + 1].copy()))
        assert np.abs(y[:, q] - y1[:, 0]).max() <= 1e-12 * np.abs(y1).max(), q
    x = y[symb.perm_c, :]
    assert np.abs(x - xt).max() <= 1e-9 * np.abs(xt).max()
    h.destroy(); symb.free()


def test_value_arenas_of_later_handles_come_from_the_device_pool():
    """Round 6 (sluamd_devpool.cpp): the arena of a destroyed 1 x 1 x 1 handle stays with the process as physical chunks; the next handle re-maps them (no
    driver-side clearing of re-used device memory), results unchanged; sluamd_device_pool_trim returns what no handle uses and reports its size."""
    import os
    import numpy as np
    from superlu_dist_amd import driver, matgen, _lib
    if "emul" in os.path.basename(os.environ.get("SLUAMD_LIB", "")):
        pytest.skip("the CPU test build has no device memory to pool (oracle/emul/emul_rt.cpp: plain allocations)")
    L = _lib.load()
    L.sluamd_device_pool_trim.restype = __import__("ctypes").c_int64
    N = 56                     # 1.3 GB of factors: above the pool's 1 GiB chunk
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
    L.sluamd_device_pool_trim(-1)
    xs = []
    for rep in range(3):
        symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)
        h = driver.LUHandle.from_symbolic(symb, v)
        assert h.pdgstrf3d(0.0) == 0
        xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
        xs.append(h.pdgstrs3d(xp)[symb.perm_c, :])
        h.destroy(); symb.free()
        assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, xs[-1])) <= 1e-12 * np.linalg.norm(b)
    held = L.sluamd_device_pool_trim(-1)
    assert held >= (1 << 30) and held % (1 << 30) == 0, held
    assert L.sluamd_device_pool_trim(-1) == 0
