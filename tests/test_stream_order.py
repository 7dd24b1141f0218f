"""Stream-order check of the drivers (CPU): the library's host code (look-ahead schedule of pdgstrf3d on four streams, XY panel
exchanges, Z reduction, triangular sweeps) runs on the emulated HIP runtime with its ADVERSARIAL scheduler on
(oracle/emul/emul_rt.cpp): stream work is queued and executed in a seeded order that honours only stream order, event waits and
the synchronising calls.  A dependency the drivers forgot to express (a missing hipStreamWaitEvent, a host read before a
synchronisation) gives a wrong factorisation here, whatever the timing on a real device happens to be."""
import ctypes
import numpy as np
import pytest
import grid_cases
from superlu_dist_amd import driver, matgen


def _sched(lib, mode, seed=1):
    lib.sluamd_emul_sched(ctypes.c_int(mode), ctypes.c_uint(seed))


def _stats(lib):
    run, reord = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    lib.sluamd_emul_sched_stats(ctypes.byref(run), ctypes.byref(reord))
    return run.value, reord.value


@pytest.fixture()
def sched(emul):
    yield emul
    _sched(emul, 0)          # back to immediate execution for whatever test comes next


def _factor_and_solve(n, rp, ci, v, perm, b, **kw):
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, **kw)
    assert info == 0
    return x


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lookahead_schedule_single_rank(sched, mode, seed):
    """Deep elimination DAG (narrow supernodes -> many levels, K-fused pairs, two-level look-ahead on four streams)."""
    N = 10
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(5)
    v = v * (1.0 + 0.2 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=8)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    x_ref = _factor_and_solve(n, rp, ci, v, perm, b, relax=4, maxsup=16)       # immediate execution, issue order
    _sched(sched, mode, seed)
    x = _factor_and_solve(n, rp, ci, v, perm, b, relax=4, maxsup=16)
    run, reordered = _stats(sched)
    _sched(sched, 0)
    assert run > 100 and reordered > 0          # the scheduler did reorder work across streams
    assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("join_max", [1 << 30, 6, 2, 1])
@pytest.mark.parametrize("mode,seed", [(1, 2), (2, 3), (3, 7)])
def test_joined_links_of_the_triangular_sweeps(sched, monkeypatch, join_max, mode, seed):
    """Round 4: ONE launch per level in the sweeps of a 1 x 1 layer -- the 64-column-block units of the next level's diagonal inverses apply the adjacent
    level's updates to their own block of the right-hand side themselves and ADD their share of the solved block into zeroed rows, the regular units skip
    those rows / columns (LevelSched::join, k_sweep_join).  The emulated launch runs joined and regular units in one seeded order.  join_max: every level
    joined / the two forms mixed at different depths (levels of more supernodes keep the two-launch links; the forms meet in both orders in the backward
    sweep).  Three right-hand sides, unsymmetric values, supernodes of up to 200 columns (four column blocks, > 3 sources at the lower levels)."""
    monkeypatch.setenv("SLUAMD_JOIN_MAX_NODES", str(join_max))
    N = 14
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(11)
    v = v * (1.0 + 0.4 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=8)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 3)
    monkeypatch.setenv("SLUAMD_SOLVE_JOIN", "0")
    x_ref, info, st0 = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=4, maxsup=200)
    monkeypatch.setenv("SLUAMD_SOLVE_JOIN", "1")
    _sched(sched, mode, seed)
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=4, maxsup=200)
    _sched(sched, 0)
    assert info == 0
    assert st["solve_launches"] < st0["solve_launches"]
    assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("mode,seed", [(1, 1), (1, 4), (2, 2), (3, 5), (3, 9)])
def test_merged_chain_groups_of_the_sweeps(sched, monkeypatch, mode, seed):
    """SLUAMD_SOLVE_GROUPS=1 (opt-in): the inverse of a group's block triangle is built during pdgstrf3d on a stream of its OWN (gather of the members' panels and
    inverses, then up to six dependent batches of dense products) -- after the panels and inverses of the group's last member (event on the bulk stream), before the
    factorisation returns (the bulk stream waits for the group stream); the sweeps then run the contracted schedule (group strips in the two-launch form, dead rows /
    columns skipped).  Under the adversarial scheduler a missing wait on either side gives a wrong solution."""
    N = 20
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(7)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    monkeypatch.setenv("SLUAMD_SOLVE_GROUPS", "0")
    x_ref, info, st0 = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=16, maxsup=64)
    monkeypatch.setenv("SLUAMD_SOLVE_GROUPS", "1")
    _sched(sched, mode, seed)
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=16, maxsup=64)
    _sched(sched, 0)
    assert info == 0
    assert st["solve_launches"] < st0["solve_launches"]                 # groups were found
    assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()
    assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("sort_rows,pinned", [(True, None), (False, None), (True, 4096), (True, 1000)])
def test_unsorted_panel_rows_of_a_view(emul, monkeypatch, sort_rows, pinned):
    """The reference's symbfact leaves the row subscripts INSIDE an L block in discovery order.  Round 4: a handle created from such a view keeps the rows
    ascending internally (the values are permuted on their way through the staging buffer, both directions) so that the merged Schur tiles and the joined
    sweeps -- which take the rows of a block that fall into one 64-column block of the target as ONE range -- apply to reference-produced stores as well:
    same launch count as the sorted store of the same matrix, the factors copied back in the CALLER's row order equal to the oracle's factorisation of the
    shuffled store (also with a staging buffer so small that every panel crosses many flushes).  SLUAMD_SORT_BLOCK_ROWS=0: the store is taken as it is and the sweeps fall back to the two-launch links."""
    import oracle as orc
    monkeypatch.setenv("SLUAMD_SORT_BLOCK_ROWS", "1" if sort_rows else "0")
    if pinned: monkeypatch.setenv("SLUAMD_PINNED_BYTES", str(pinned))      # a staging buffer of a few hundred values: every panel is split, pieces start mid-column
    N = 12
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(3)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=8)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    symb = driver.Symbolic(n, rp, ci, perm, relax=4, maxsup=96)
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    out = {}
    for shuffled in (False, True):
        symb.distribute_host(v)
        fs = symb.flat_store()
        if shuffled:
            grid_cases.shuffle_block_rows(fs, 5)
        o = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind.copy(), fs.Lnzval_off, fs.Lnzval.copy(), fs.Ufstnz_off, fs.Ufstnz, fs.Unzval_off, fs.Unzval.copy())
        assert orc.dfactor(o)[0] == 0
        rows_before = fs.Lrowind.copy()
        h = driver.LUHandle.from_store(fs)
        assert h.pdgstrf3d(0.0) == 0
        x = h.pdgstrs3d(xp)[symb.perm_c, :]
        st = h.stats()
        h.copy_to_host()
        h.destroy()
        assert np.array_equal(fs.Lrowind, rows_before)                      # the caller's index arrays are not touched
        scale = np.abs(o.Lnzval).max()
        assert np.abs(fs.Lnzval - o.Lnzval).max() <= 1e-12 * scale and np.abs(fs.Unzval - o.Unzval).max() <= 1e-12 * scale
        assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) <= 1e-12 * np.linalg.norm(b)
        out[shuffled] = (x, st["solve_launches"], st["num_levels"])
    symb.free()
    nl = out[True][2]
    if sort_rows: assert out[True][1] == out[False][1]
    else: assert out[False][1] < out[True][1] and out[True][1] >= 4 * nl - 3       # joined: about one launch per level and sweep; fallback: two
    assert np.abs(out[False][0] - out[True][0]).max() <= 1e-11 * np.abs(xt).max()


@pytest.mark.parametrize("grid", [(1, 1, 2), (1, 1, 4)])
def test_joined_links_per_forest_on_z_layers(sched, monkeypatch, grid):
    """Joined links on 1 x 1 x Pz grids: every forest of a layer's path zeroes ITS rows of the two vectors (k_zero_nodes, not a memset: the rows of the
    ancestor forests carry the reduced right-hand side / the solved blocks of the other Z levels)."""
    monkeypatch.setenv("SLUAMD_JOIN_MAX_NODES", "3")
    _sched(sched, 2, 5)
    grid_cases.check_own_pipeline(8, grid, nrhs=2, unsym=True, refactor=True)
    _sched(sched, 0)


@pytest.mark.parametrize("mode,seed", [(1, 1), (1, 2), (2, 1), (3, 1)])
def test_wide_supernodes_big_tiles_and_fused_pairs(sched, mode, seed):
    """256-wide supernodes: the 128 x 128 tile lists, K-fused chain pairs, Crout diagonal kernel + full inverses on the chain."""
    N = 14
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
    x_ref = _factor_and_solve(n, rp, ci, v, perm, b, relax=64, maxsup=256)
    _sched(sched, mode, seed)
    x = _factor_and_solve(n, rp, ci, v, perm, b, relax=64, maxsup=256)
    _sched(sched, 0)
    assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()


@pytest.mark.parametrize("mode,seed", [(1, 1), (1, 4), (2, 2), (3, 3)])
def test_complex16_through_the_lookahead_schedule(sched, mode, seed):
    """pzgstrf3d runs the same four-stream look-ahead schedule as pdgstrf3d (urgent tile lists, panels of the next two levels beside
    the bulk): a deep complex16 elimination DAG under the adversarial stream scheduler against immediate execution."""
    N = 10
    n, rp, ci, v = matgen.poisson3d(N)
    v = matgen.complex_shift(v, rp, ci, seed=9)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=8)
    rng = np.random.default_rng(2)
    xt = rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))
    b = matgen.csr_matvec(n, rp, ci, v, xt)

    def run():
        symb = driver.Symbolic(n, rp, ci, perm, relax=4, maxsup=16)
        h = driver.LUHandle.from_symbolic(symb, v)
        assert h.pzgstrf3d(0.0) == 0
        xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
        x = h.pzgstrs3d(xp)[symb.perm_c, :]
        st = h.stats()
        h.destroy(); symb.free()
        return x, st

    x_ref, st = run()
    assert st["num_levels"] > 8
    _sched(sched, mode, seed)
    x, _ = run()
    run_count, reordered = _stats(sched)
    _sched(sched, 0)
    assert reordered > 0
    assert np.abs(x - x_ref).max() <= 1e-11 * np.abs(x_ref).max()
    assert np.abs(x - xt).max() <= 1e-9 * np.abs(xt).max()


@pytest.mark.parametrize("grid", [(2, 2, 1), (1, 1, 2), (2, 2, 2)])
@pytest.mark.parametrize("mode,seed", [(1, 1), (2, 1), (3, 2)])
def test_grid_drivers(sched, grid, mode, seed):
    """XY panel exchange (its own look-ahead rule: the received panels of a level share a scratch copy by level parity), Z ancestor
    reduction and the distributed sweeps, ranks = threads over the in-process transport."""
    _sched(sched, mode, seed)
    grid_cases.check_own_pipeline(8, grid, nrhs=2, unsym=True, refactor=(grid == (2, 2, 2)))
    _sched(sched, 0)


@pytest.mark.parametrize("grid,mode,seed,stream_ordered", [((2, 2, 1), 1, 3, False), ((2, 2, 1), 2, 1, False), ((2, 2, 1), 3, 5, True), ((2, 2, 1), 2, 1, True),
                                                           ((2, 1, 2), 2, 1, False), ((2, 1, 2), 3, 5, True)])
def test_xy_layers_with_fused_pairs_and_cut_levels(sched, monkeypatch, grid, mode, seed, stream_ordered):
    """Round 4 on XY layers: K-fused chain pairs (a deferred supernode's RECEIVED panels are read again by its partner's tiles one level
    later: three scratch copies by level modulo 3, and the exchange of level m still has to wait for the bulk of level m - 2 -- deleting
    that wait fails this test, scripts/stream_order_mutations.sh) and wide DAG levels cut into sub-levels of <= 1/4 of the largest one
    (SLUAMD_LEVEL_SPLIT_MIN lowered so that the 28^3 tree is cut), under the adversarial scheduler over both in-process transports."""
    monkeypatch.setenv("SLUAMD_LEVEL_SPLIT_MIN", "16")
    N = 28
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(N)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    st = []
    _sched(sched, mode, seed)
    grid_cases.check_matrix_on_grid(n, rp, ci, v, perm, grid, nrhs=2, relax=64, maxsup=128, refactor=(mode == 1), stats_out=st,
                                    make_comms=grid_cases.stream_ordered_comms if stream_ordered else None)
    _sched(sched, 0)
    assert max(s["reserved_i"] for s in st) >= 2           # K-fused pairs formed on the XY layer
    monkeypatch.setenv("SLUAMD_NO_LEVEL_SPLIT", "1")
    st0 = []
    grid_cases.check_matrix_on_grid(n, rp, ci, v, perm, grid, nrhs=1, relax=64, maxsup=128, stats_out=st0)
    a = {s["rank"]: s for s in st}; b = {s["rank"]: s for s in st0}
    assert all(a[r]["num_levels"] > b[r]["num_levels"] for r in a)                   # the leaf levels were cut (on every layer's forests) ...
    assert sum(a[r]["bytes_device"] for r in a) < sum(b[r]["bytes_device"] for r in a)    # ... and the exchange scratch shrank with them


def test_xy_levels_cut_by_panel_weight(sched, monkeypatch):
    """The second cutting rule: levels whose panels (upper bound from the global block graph) outweigh 1/16 of the forest's total are cut
    too -- at 150^3 the largest level by bytes is a level of 256 separator supernodes, not the leaves.  Threshold lowered so that the 28^3
    tree qualifies; count rule disabled by a high SLUAMD_LEVEL_SPLIT_MIN.  Same results, more levels, less exchange scratch."""
    N = 28
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(N + 1)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    monkeypatch.setenv("SLUAMD_LEVEL_SPLIT_MIN", "100000")
    monkeypatch.setenv("SLUAMD_LEVEL_SPLIT_WMIN", "1")
    monkeypatch.setenv("SLUAMD_LEVEL_SPLIT_WDIV", "16")
    st = []
    _sched(sched, 2, 3)
    grid_cases.check_matrix_on_grid(n, rp, ci, v, perm, (2, 2, 1), nrhs=2, relax=64, maxsup=128, refactor=True, stats_out=st)
    _sched(sched, 0)
    monkeypatch.setenv("SLUAMD_LEVEL_SPLIT_WDIV", "1")
    st0 = []
    grid_cases.check_matrix_on_grid(n, rp, ci, v, perm, (2, 2, 1), nrhs=1, relax=64, maxsup=128, stats_out=st0)
    a = {s["rank"]: s for s in st}; b = {s["rank"]: s for s in st0}
    assert all(a[r]["num_levels"] > b[r]["num_levels"] for r in a)
    assert sum(a[r]["bytes_device"] for r in a) < sum(b[r]["bytes_device"] for r in a)


@pytest.mark.parametrize("sched_env", ["1,11", "2,1", "3,4"])
def test_reference_fixtures_under_the_scheduler(sched_env):
    """The per-rank parity tests against the reference's recorded grids (1x1x2, 2x1x1, 2x2x2 golden fixtures, own pipeline on seven
    grid shapes, wide supernodes) once more, the whole process under one adversarial schedule (SLUAMD_EMUL_SCHED=mode,seed)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SLUAMD_EMUL_SCHED=sched_env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_grid_emul.py"), "-q", "-x", "-k", "not gloo"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.parametrize("grid", [(2, 2, 1), (2, 2, 2)])
@pytest.mark.parametrize("mode,seed", [(1, 2), (2, 5), (3, 1)])
@pytest.mark.parametrize("stream_ordered", [False, True])
def test_complex16_grid_drivers(sched, grid, mode, seed, stream_ordered):
    """complex16 on XY layers under the adversarial scheduler, over the host-staged transport and over the queue-only (RCCL-like) one."""
    _sched(sched, mode, seed)
    grid_cases.check_own_pipeline_complex16(grid[2], Pr=grid[0], Pc=grid[1], make_comms=grid_cases.stream_ordered_comms if stream_ordered else None)
    _sched(sched, 0)


@pytest.mark.parametrize("grid", [(2, 1, 1), (1, 2, 1), (2, 2, 1), (1, 1, 2), (1, 1, 4), (2, 2, 2), (3, 2, 1)])
@pytest.mark.parametrize("mode,seed", [(0, 1), (1, 3), (2, 1), (3, 8)])
def test_grid_drivers_over_the_stream_ordered_transport(sched, grid, mode, seed):
    """The drivers the way they run over RCCL: the transport behind sluamd_comm_create_rccl in the CPU build only QUEUES an exchange on
    the stream and returns (oracle/emul/comm_norccl.cpp), so the host runs ahead of the device through the whole factorisation and solve
    and nothing but stream order and events protects the exchange staging buffer, the scratch copies of received panels and the
    ancestor reduction.  (LocalComm / the callbacks synchronise the stream at every exchange.)"""
    _sched(sched, mode, seed)
    grid_cases.check_own_pipeline(8, grid, nrhs=2, unsym=True, refactor=(grid == (2, 2, 2)), make_comms=grid_cases.stream_ordered_comms)
    _sched(sched, 0)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SLUAMD_FUZZ_CASES", "48"))))
def test_random_structures_grids_and_schedules(sched, seed):
    """Fuzz: random unsymmetric matrices (irregular supernodes, ragged skylines), random supernode limits, grid shapes, numbers of
    right-hand sides and adversarial schedules -- the solution must agree with the single-rank one and solve the original system."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(60, 260))
    n, rp, ci, v = matgen.random_unsym(n, float(rng.uniform(0.01, 0.06)), seed=seed)
    grid = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1), (1, 1, 2), (2, 1, 2), (1, 1, 4), (3, 1, 1), (2, 2, 2)][int(rng.integers(0, 9))]
    relax = int(rng.choice([1, 4, 16, 64])); maxsup = int(rng.choice([4, 16, 48, 256]))
    _sched(sched, int(rng.integers(1, 4)), int(rng.integers(1, 1000)))
    grid_cases.check_matrix_on_grid(n, rp, ci, v, None, grid, nrhs=int(rng.integers(1, 4)), relax=relax, maxsup=maxsup,
                                    refactor=bool(rng.integers(0, 2)),
                                    make_comms=grid_cases.stream_ordered_comms if rng.integers(0, 2) else None)
    _sched(sched, 0)


@pytest.mark.parametrize("seed", range(8))
def test_random_wide_supernodes_grids_and_schedules(sched, seed):
    """Fuzz of the 257..512-column refinement: nearly dense matrices (one or two supernodes of several hundred columns) on random
    grids -- XY layers included -- under random schedules."""
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(270, 430))
    n, rp, ci, v = matgen.random_unsym(n, float(rng.uniform(0.5, 0.95)), seed=100 + seed)
    grid = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1), (1, 1, 2), (2, 2, 2), (3, 2, 1)][int(rng.integers(0, 7))]
    _sched(sched, int(rng.integers(1, 4)), int(rng.integers(1, 1000)))
    grid_cases.check_matrix_on_grid(n, rp, ci, v, None, grid, nrhs=2, relax=512, maxsup=512,
                                    make_comms=grid_cases.stream_ordered_comms if seed % 2 else None)
    _sched(sched, 0)
