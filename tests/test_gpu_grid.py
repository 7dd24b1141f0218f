"""Process grids on the GPU through the product library: the per-rank records of the real reference on 1x1x2, 2x1x1 and
2x2x2 grids (ranks = threads sharing the box's GPU over the in-process transport), the library's own pipeline on several
grid shapes, real processes over the callback transport (gloo, host-staged), and -- when RCCL accepts two ranks on one
device -- the direct RCCL transport."""
import os, subprocess, sys
import pytest
import grid_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", grid_cases.GRID_FIXTURES)
def test_grid_fixture_per_rank_parity(golden, case):
    grid_cases.check_fixture_grid(golden(case))


@pytest.mark.parametrize("N,grid,nrhs,unsym", [(12, (1, 1, 2), 1, False), (12, (2, 2, 1), 2, True), (16, (2, 2, 2), 1, True),
                                                (12, (1, 2, 4), 1, False), (10, (3, 2, 1), 1, True), (12, (1, 1, 8), 3, False)])
def test_own_pipeline_on_grids(N, grid, nrhs, unsym):
    grid_cases.check_own_pipeline(N, grid, nrhs=nrhs, unsym=unsym, leaf=27, relax=32, maxsup=128, refactor=(grid == (2, 2, 2)))


@pytest.mark.parametrize("grid", [(1, 1, 1), (2, 2, 1)])
def test_many_right_hand_sides_are_solved_in_chunks(grid):
    """nrhs beyond what the LDS-staged solve kernels take at once (the reference accepts any nrhs): 70 columns, 256-wide supernodes."""
    grid_cases.check_own_pipeline(12, grid, nrhs=70, leaf=64, relax=64, maxsup=256)


def test_wide_supernodes_on_a_2x2x2_grid():
    """256-wide supernodes (128x128 Schur tiles, blocked diagonal LU, multi-block TRSMs) with panels received from peers."""
    grid_cases.check_own_pipeline(24, (2, 2, 2), nrhs=2, unsym=True, leaf=64, relax=64, maxsup=256)


def _launch(world, grid, extra, tmp_path, timeout=600):
    port = 29500 + (os.getpid() % 400) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "grid_worker.py"), "--engine", "hip",
           "--grid", *[str(v) for v in grid], "--side", "12"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"))


@pytest.mark.parametrize("world,grid", [(2, (1, 1, 2)), (4, (2, 2, 1))])
def test_processes_sharing_one_gpu_over_callbacks(world, grid, tmp_path):
    r = _launch(world, grid, [], tmp_path)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GRID_WORKER_OK" in r.stdout


def _rccl_comm_one_rank():
    return grid_cases.stream_ordered_comms(1, 1, 1)[0]    # product library: sluamd_comm_rccl_unique_id + ncclCommInitRank(nranks = 1)


@pytest.mark.parametrize("nbytes", [8, 1 << 20, (1 << 26) + 40])
def test_rccl_transport_on_hardware_one_rank(nbytes):
    """Every operation of the RCCL transport on the MI355X, no skip: a one-rank communicator (ncclCommInitRank with nranks = 1 is
    legal on a one-GPU box) exchanging with itself -- ncclSend / ncclRecv inside one ncclGroup queued on a non-blocking stream
    between device fills, an empty group, the device-staged host-buffer group, ncclAllReduce(min) on the stream."""
    comm = _rccl_comm_one_rank()
    grid_cases.check_transport_selftest([comm], nbytes)
    from superlu_dist_amd import _lib
    _lib.load().sluamd_comm_destroy(comm)


def test_rccl_grid_handle_on_hardware_one_rank(golden):
    """The grid entry points over the RCCL communicator: reference fixture through sluamd_dCreateLUHandleGrid (factor parity,
    every recorded solve), then the library's own pipeline + refactor."""
    comm = _rccl_comm_one_rank()
    grid_cases.check_fixture_on_one_rank_comm(golden("poisson10_nd"), comm)
    grid_cases.check_own_pipeline(10, (1, 1, 1), nrhs=2, refactor=True, make_comms=lambda *_: [comm])


@pytest.mark.parametrize("grid", [(1, 1, 1), (1, 1, 2), (2, 2, 1)])
def test_transport_selftest_in_process(grid):
    from superlu_dist_amd import grid3d
    grid_cases.check_transport_selftest(grid3d.local_comms(*grid), 1 << 16)


def test_rccl_transport_two_ranks(tmp_path):
    """Direct RCCL transport with two ranks.  One GPU per rank is RCCL's contract: on a one-GPU box ncclCommInitRank refuses the
    second rank ("Duplicate GPU detected") and ONLY that is a skip -- once the communicator exists, any failure or hang of the
    exchanges is a failure of this test.  With two or more GPUs visible the test must pass."""
    from superlu_dist_amd import _lib
    ngpu = _lib.load().sluamd_device_count()     # the library's own count: no torch needed to decide
    extra = ["--transport", "rccl"] + (["--one-gpu-per-rank"] if ngpu >= 2 else [])
    try:
        r = _launch(2, (1, 1, 2), extra, tmp_path, timeout=300)
    except subprocess.TimeoutExpired as e:
        out = ((e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")) + \
              ((e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))
        if ngpu < 2 and "RCCL_COMM_READY" not in out:
            pytest.skip("ncclCommInitRank did not return with two ranks on one device")
        raise AssertionError("RCCL exchange hung after the communicator was created:\n" + out[-2000:])
    out = r.stdout + r.stderr
    if r.returncode != 0 and ngpu < 2 and "RCCL_COMM_READY" not in out and ("ncclCommInitRank failed" in out or "Duplicate GPU" in out):
        pytest.skip("RCCL refuses two ranks on one device (ncclCommInitRank): " + " | ".join(l for l in out.splitlines() if "ncclCommInitRank" in l or "Duplicate" in l)[:400])
    if r.returncode != 0 and "the CPU test build has no RCCL transport" in out:
        pytest.skip("emulation library (test_gpu_suite_on_emulation.py): its stream-ordered stand-in connects threads of one process")
    assert r.returncode == 0, out[-3000:]
    assert "GRID_WORKER_OK" in r.stdout


@pytest.mark.parametrize("N,maxsup,Pz,shuffle", [(18, 512, 1, True), (24, 384, 1, False), (24, 512, 2, True), (32, 512, 2, False)])
def test_supernodes_257_to_512_columns(N, maxsup, Pz, shuffle):
    """Reference-format panels with supernodes of up to MAX_SUPER_SIZE = 512 columns (superlu_defs.h:154): refined into
    <= 256-column pieces at handle creation; every L/U value in the caller's layout against the CPU oracle."""
    import oracle as orc
    grid_cases.check_wide_supernodes(N, maxsup, Pz, orc, shuffle)


@pytest.mark.parametrize("grid", [(1, 1, 1), (1, 1, 2), (2, 1, 1), (2, 2, 2)])
def test_own_pipeline_with_supernodes_up_to_512_columns(grid):
    """maxsup = 512 through the library's own symbolic factorisation + device-side distribution: wide supernodes are refined
    at handle creation (on XY layers the pieces stay with the owners of their supernode), A's entries are scattered straight into the pieces."""
    grid_cases.check_own_pipeline(18, grid, nrhs=2, leaf=64, relax=64, maxsup=512)


@pytest.mark.parametrize("case", grid_cases.ZGRID_FIXTURES)
def test_complex16_grid_fixture_per_rank_parity(golden, case):
    """pzgstrf3d / pzgstrs3d against the reference's per-rank records on 1 x 1 x 2 (the Z ancestor reduction and the Z sweeps of the
    solve move complex16 values as pairs of doubles) and on 2 x 1 x 1 / 1 x 2 x 1 / 2 x 2 x 2 (XY panel exchange of complex16 panels,
    distributed complex solves)."""
    grid_cases.check_fixture_grid(golden(case))


@pytest.mark.parametrize("Pz", [2, 4])
def test_own_pipeline_complex16_on_z_layers(Pz):
    """complex16 through the library's own symbolic factorisation + device-side distribution on a 1 x 1 x Pz grid: residual on the
    original system and agreement with the single-rank solution."""
    grid_cases.check_own_pipeline_complex16(Pz)


@pytest.mark.parametrize("Pz", [1, 2])
def test_own_pipeline_complex16_with_supernodes_up_to_512_columns(Pz):
    """complex16 supernodes of 257..512 columns (468 here): refined like the double ones; the pieces are ordinary supernodes to the
    complex kernels."""
    grid_cases.check_own_pipeline_complex16(Pz, N=18, leaf=64, relax=64, maxsup=512)


@pytest.mark.parametrize("grid", [(2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 2), (3, 2, 1)])
def test_own_pipeline_complex16_on_xy_layers(grid):
    """complex16 on XY block-cyclic layers (round 3; pzgstrf3d's panel exchange -- ztrfCommWrapper.c, zcommunication_aux.c -- and the
    distributed pzgstrs3d): own symbolic factorisation + device-side distribution, residual and agreement with the single-rank solution."""
    grid_cases.check_own_pipeline_complex16(grid[2], Pr=grid[0], Pc=grid[1])


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 2)])
def test_own_pipeline_complex16_with_wide_supernodes_on_xy_layers(grid):
    """... with supernodes of 257..512 columns refined into pieces that stay with the owners of their supernode."""
    grid_cases.check_own_pipeline_complex16(grid[2], N=18, leaf=64, relax=64, maxsup=512, Pr=grid[0], Pc=grid[1])
