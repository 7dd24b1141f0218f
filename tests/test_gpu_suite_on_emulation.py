"""The bodies of the `-m gpu` parity tests once more on CPU: the same test files (fixtures of the real reference, oracle comparisons,
edge cases, grids, refinement), with the ctypes layer bound to oracle/libsluamd_emul.so through SLUAMD_LIB -- the library's own host
sources over the serial restatement of the kernels (TEST INFRASTRUCTURE, built by `make -C oracle`).  What this pins without a GPU:
the planner, the schedules, the value movement and the C ABI against every fixture; what it cannot pin is the HIP kernels themselves
(the real `-m gpu` run does that).  Second run: the whole process under an adversarial stream schedule (test_stream_order.py)."""
import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["test_gpu_parity.py", "test_gpu_edge_cases.py", "test_gpu_refine.py", "test_gpu_grid.py"]


@pytest.mark.parametrize("sched", ["", "1,7"])
def test_gpu_test_files_against_the_emulation_library(sched):
    so = os.path.join(ROOT, "oracle", "libsluamd_emul.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libsluamd_emul.so"])
    env = dict(os.environ, SLUAMD_LIB=so)
    env.pop("SLUAMD_EMUL_SCHED", None)
    if sched:
        env["SLUAMD_EMUL_SCHED"] = sched
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu"] + [os.path.join(ROOT, "tests", f) for f in FILES],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert " passed" in r.stdout


def test_reference_drop_in_against_the_emulation_library():
    """test_gpu_dropin.py on CPU: the REAL reference pipeline (oracle/_ref/slu_ref_amd: pdgssvx3d / pzgssvx3d under mpiexec on 1x1x1 ...
    2x2x2 grids, SUPERLU_MAXSUP=512, the irregular stand-in on 2x2x2) with pdgstrf3d and pdgstrs3d bound through
    oracle/ref/sluamd_binding.c -- which loads the library named by SLUAMD_LIB -- to the emulation library.  Pins the binding, the
    view path, the structure exchange and the MPI callback transport without a GPU (the n = 46 656 case is left to the GPU run)."""
    so = os.path.join(ROOT, "oracle", "libsluamd_emul.so")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "slu_ref_amd")):
        pytest.skip("prebuilt reference binaries not present (oracle/_ref is built where /root/reference exists)")
    env = dict(os.environ, SLUAMD_LIB=so)
    env.pop("SLUAMD_EMUL_SCHED", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "not at_scale", os.path.join(ROOT, "tests", "test_gpu_dropin.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1]
