"""CPU coverage of the library's multi-rank HOST logic (planning, XY panel-exchange plans, Z ancestor reduction,
distributed solve, C-level transports): the library's own host sources linked against the serial kernel restatement
(oracle/libsluamd_emul.so, built by `make -C oracle`), pinned to the per-rank records of the real reference on
1x1x2, 2x1x1 and 2x2x2 grids.  The GPU twin of this file is test_gpu_grid.py (same bodies, product library)."""
import os, subprocess, sys
import numpy as np
import pytest
import grid_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", grid_cases.GRID_FIXTURES)
def test_grid_fixture_per_rank_parity(emul, golden, case):
    grid_cases.check_fixture_grid(golden(case))


@pytest.mark.parametrize("case", ["g20_1x1x1", "poisson10_nd", "unsym300", "unsym120_tiny", "z_cg20_1x1x1", "z_poisson8_nd", "z_unsym200", "z_grid24_nd"])
def test_single_rank_fixture_parity(emul, golden, case):
    """1x1x1 through the same planner (slot model with one process row / column)."""
    from superlu_dist_amd import driver
    g = golden(case)
    st = driver.FlatStore.from_golden(g, 0, "pre")
    h = driver.LUHandle.from_store(st, replace_tiny=bool(g["r0__ReplaceTinyPivot"][0]))
    assert h.pdgstrf3d(float(g["r0__thresh"][0])) == int(g["r0__info"][0])
    h.copy_to_host()
    scale = max(np.abs(g["r0__Lnzval_pre"]).max(), np.abs(g["r0__Unzval_pre"]).max())
    assert np.abs(st.Lnzval - g["r0__Lnzval_post"]).max() <= 1e-12 * scale
    assert np.abs(st.Unzval - g["r0__Unzval_post"]).max() <= 1e-12 * scale
    h.destroy()


@pytest.mark.parametrize("case", grid_cases.ZGRID_FIXTURES)
def test_complex16_grid_fixture_per_rank_parity(emul, golden, case):
    """pzgstrf3d / pzgstrs3d against the reference's per-rank records on 1 x 1 x 2 (Z ancestor reduction and Z sweeps on pairs of doubles)
    and on 2 x 1 x 1 / 1 x 2 x 1 / 2 x 2 x 2 (XY panel exchange of complex16 panels, distributed complex solves)."""
    grid_cases.check_fixture_grid(golden(case))


@pytest.mark.parametrize("Pz", [2, 4])
def test_own_pipeline_complex16_on_z_layers(emul, Pz):
    grid_cases.check_own_pipeline_complex16(Pz)


@pytest.mark.parametrize("Pz", [1, 2])
def test_own_pipeline_complex16_with_supernodes_up_to_512_columns(emul, Pz):
    """complex16 supernodes of 257..512 columns: refined like the double ones (the pieces are ordinary supernodes to the complex
    kernels).  CPU twin of the test in test_gpu_grid.py; the reference's pzgssvx3d with SUPERLU_MAXSUP=512 runs against this engine in
    test_gpu_suite_on_emulation.py."""
    grid_cases.check_own_pipeline_complex16(Pz, N=18, leaf=64, relax=64, maxsup=512)


@pytest.mark.parametrize("N,grid,nrhs,unsym", [(8, (1, 1, 2), 1, False), (8, (2, 2, 1), 2, True), (10, (2, 2, 2), 1, True),
                                                (8, (1, 2, 4), 1, False), (8, (3, 2, 1), 1, True), (8, (1, 1, 8), 3, False)])
def test_own_pipeline_on_grids(emul, N, grid, nrhs, unsym):
    grid_cases.check_own_pipeline(N, grid, nrhs=nrhs, unsym=unsym, refactor=(grid == (2, 2, 2)))


@pytest.mark.parametrize("grid", [(1, 1, 1), (2, 2, 1)])
def test_many_right_hand_sides_are_solved_in_chunks(emul, grid):
    """nrhs beyond what the LDS-staged solve kernels take at once (the reference accepts any nrhs): 70 columns, 256-wide supernodes."""
    grid_cases.check_own_pipeline(8, grid, nrhs=70, leaf=64, relax=64, maxsup=256)


def test_grid_without_communicator_is_rejected(emul, golden):
    """A rank of a multi-rank grid cannot be factored alone: the old silent-wrong-factors path is an error now."""
    from superlu_dist_amd import driver
    g = golden("g20_1x1x2")
    st = driver.FlatStore.from_golden(g, 0, "pre")
    with pytest.raises(RuntimeError, match="communicator"):
        driver.LUHandle.from_store(st, forests=grid_cases.forests_of(g, 0))


@pytest.mark.parametrize("world,grid", [(2, (1, 1, 2)), (2, (2, 1, 1)), (4, (2, 2, 1))])
def test_gloo_processes_through_the_callback_transport(world, grid, tmp_path):
    """Real processes (torch.distributed gloo, world size > 1) driving the C orchestration through
    sluamd_comm_create_callbacks -- the same path the reference-side MPI binding uses."""
    port = 29500 + (os.getpid() % 400) + world
    script = os.path.join(ROOT, "tests", "grid_worker.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, "--engine", "emul", "--grid", *[str(v) for v in grid], "--side", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GRID_WORKER_OK" in r.stdout


@pytest.mark.parametrize("N,maxsup,Pz,shuffle", [(18, 512, 1, False), (20, 512, 1, True), (24, 384, 1, False), (18, 512, 2, True), (24, 512, 2, False)])
def test_supernodes_257_to_512_columns(emul, N, maxsup, Pz, shuffle):
    import oracle as orc
    grid_cases.check_wide_supernodes(N, maxsup, Pz, orc, shuffle)


def test_view_grid_must_match_the_communicator(emul, golden):
    """A 1x1x1 LU view handed a 1x1x2 communicator is a caller error (it used to be accepted and the Z reduction skipped)."""
    from superlu_dist_amd import driver, grid3d
    st = driver.FlatStore.from_golden(golden("g20_1x1x1"), 0, "pre")
    comms = grid3d.local_comms(1, 1, 2)
    with pytest.raises(RuntimeError, match="does not match"):
        grid3d.GridHandle.from_store(st, None, comms[0])


@pytest.mark.parametrize("grid", [(1, 1, 1), (1, 1, 2), (2, 1, 1), (2, 2, 2)])
def test_own_pipeline_with_supernodes_up_to_512_columns(emul, grid):
    """maxsup = 512 through the library's own symbolic factorisation + device-side distribution: wide supernodes are refined
    at handle creation (on XY layers the pieces stay with the owners of their supernode), A's entries are scattered straight into the pieces."""
    grid_cases.check_own_pipeline(18, grid, nrhs=2, leaf=64, relax=64, maxsup=512)


@pytest.mark.parametrize("kind,grid", [("local", (1, 1, 1)), ("local", (1, 3, 1)), ("local", (2, 2, 2)), ("stream", (1, 1, 1)), ("stream", (1, 1, 2)), ("stream", (2, 2, 1))])
@pytest.mark.parametrize("nbytes", [8, 4096 + 24])
def test_transport_selftest(emul, kind, grid, nbytes):
    """sluamd_comm_selftest over the in-process transports: a one-rank world exchanges with itself (the shape of the GPU box's
    one-rank RCCL test), larger worlds run a ring."""
    from superlu_dist_amd import grid3d
    comms = grid3d.local_comms(*grid) if kind == "local" else grid_cases.stream_ordered_comms(*grid)
    grid_cases.check_transport_selftest(comms, nbytes)


@pytest.mark.parametrize("kind", ["local", "stream"])
def test_fixture_through_the_grid_entry_points_on_a_one_rank_communicator(emul, golden, kind):
    from superlu_dist_amd import grid3d
    comm = (grid3d.local_comms(1, 1, 1) if kind == "local" else grid_cases.stream_ordered_comms(1, 1, 1))[0]
    grid_cases.check_fixture_on_one_rank_comm(golden("poisson10_nd"), comm)


def test_a_failing_rank_releases_its_peers(emul):
    """A size mismatch inside an exchange used to leave the sender waiting for ever: now the failing rank poisons the
    in-process world and every rank returns an error."""
    import ctypes as C
    from superlu_dist_amd import _lib, grid3d
    L = _lib.load()
    comms = grid3d.local_comms(1, 1, 2)

    def body(rank):
        return L.sluamd_comm_selftest(comms[rank], 64 if rank == 0 else 128)   # the two ranks disagree on the message size

    rcs = grid3d.run_ranks(2, body)
    assert all(rc != 0 for rc in rcs), rcs


@pytest.mark.parametrize("grid", [(2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 2), (3, 2, 1)])
def test_own_pipeline_complex16_on_xy_layers(emul, grid):
    """complex16 on XY block-cyclic layers (round 3; pzgstrf3d's panel exchange -- ztrfCommWrapper.c, zcommunication_aux.c -- and the
    distributed pzgstrs3d): own symbolic factorisation + device-side distribution, residual and agreement with the single-rank solution."""
    grid_cases.check_own_pipeline_complex16(grid[2], Pr=grid[0], Pc=grid[1])


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 2)])
def test_own_pipeline_complex16_with_wide_supernodes_on_xy_layers(emul, grid):
    """... with supernodes of 257..512 columns refined into pieces that stay with the owners of their supernode."""
    grid_cases.check_own_pipeline_complex16(grid[2], N=18, leaf=64, relax=64, maxsup=512, Pr=grid[0], Pc=grid[1])


@pytest.mark.parametrize("grid", [(1, 1, 1), (2, 1, 1), (2, 2, 1), (1, 1, 2), (2, 2, 2)])
def test_unsymmetric_symbolic_structure_on_grids(emul, grid):
    """sluamd_dsymbfact_unsym (exact unsymmetric structure, the reference's supernode rules) through the own pipeline on process grids:
    device-side distribution into ragged skylines, XY block-cyclic slots, forests from the etree of A + A^T; solution against the
    single-rank run on the symmetrised structure."""
    from superlu_dist_amd import matgen
    n, rp, ci, v = matgen.stencil3d_unsym(10, drop=0.35, seed=7)
    perm = matgen.nd_perm_grid3d(10, 10, 10, leaf=8)
    grid_cases.check_matrix_on_grid(n, rp, ci, v, perm, grid, nrhs=2, relax=12, maxsup=48, unsym_symb=True, refactor=True)


def test_merged_schur_tiles_cover_the_same_updates(emul, monkeypatch):
    """Merged Schur tiles (round 4): per U block the rows of all L blocks at and below its supernode -- one destination panel -- and per L block the
    columns of all U blocks to its right -- one destination U row -- are re-cut into tiles across block boundaries.  Fewer tile executions, the
    same factors (to summation order) as the one-tile-set-per-block-pair lists; the emulation engine checks every merged list entry against the block
    tables and resolves its destinations by linear searches that share nothing with the kernel's binary searches."""
    from superlu_dist_amd import driver, matgen
    N = 18
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(3)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 2)
    out = {}
    for mode in ("merged", "plain"):
        if mode == "plain":
            monkeypatch.setenv("SLUAMD_NO_MERGE_TILES", "1")
        symb = driver.Symbolic(n, rp, ci, perm, relax=20, maxsup=96)
        symb.distribute_host(v)
        fs = symb.flat_store()
        h = driver.LUHandle.from_store(fs)
        planned = h.stats()["schur_tiles"]
        assert h.pdgstrf3d(0.0) == 0
        assert h.stats()["schur_tiles"] == planned          # what ran is what was planned
        h.copy_to_host()
        xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
        x = h.pdgstrs3d(xp)[symb.perm_c, :]
        assert np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b) < 1e-12
        out[mode] = (planned, fs.Lnzval.copy(), fs.Unzval.copy())
        h.destroy(); symb.free()
    assert out["merged"][0] < 0.9 * out["plain"][0]
    scale = np.abs(v).max()
    assert np.abs(out["merged"][1] - out["plain"][1]).max() <= 1e-12 * scale
    assert np.abs(out["merged"][2] - out["plain"][2]).max() <= 1e-12 * scale


@pytest.mark.parametrize("sched_mode", [0, 2])
def test_k_fused_groups_of_three(emul, monkeypatch, sched_mode):
    """K-fused GROUPS (round 4 default on 1 x 1 layers, on levels of >= 8 supernodes: a supernode's tiles also accumulate the deferred updates of its
    two chain predecessors -- one prologue and one scatter for three sources): forced onto every level of a small tree
    (SLUAMD_FUSE_GROUP_MIN_NODES=1), plain and under an adversarial stream schedule, against the same factorisation without any fusion."""
    import ctypes
    from superlu_dist_amd import driver, matgen
    N = 28
    n, rp, ci, v = matgen.poisson3d(N)
    rng = np.random.default_rng(11)
    v = v * (1.0 + 0.3 * rng.random(v.size))
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 1.0
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=64)
    symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=128)
    out = {}
    for mode in ("groups", "none"):
        monkeypatch.setenv("SLUAMD_FUSE_GROUP_MIN_NODES", "1")
        if mode == "none":
            monkeypatch.setenv("SLUAMD_NO_FUSE", "1")
        symb.distribute_host(v)
        fs = symb.flat_store()
        h = driver.LUHandle.from_store(fs)
        fused = h.stats()["reserved_i"]
        if mode == "groups":
            emul.sluamd_emul_sched(ctypes.c_int(sched_mode), ctypes.c_uint(5))
        assert h.pdgstrf3d(0.0) == 0
        emul.sluamd_emul_sched(ctypes.c_int(0), ctypes.c_uint(1))
        h.copy_to_host(); h.destroy()
        out[mode] = (fused, fs.Lnzval.copy(), fs.Unzval.copy())
    symb.free()
    assert out["groups"][0] >= 3 and out["none"][0] == 0
    scale = np.abs(v).max()
    assert np.abs(out["groups"][1] - out["none"][1]).max() <= 1e-12 * scale
    assert np.abs(out["groups"][2] - out["none"][2]).max() <= 1e-12 * scale
