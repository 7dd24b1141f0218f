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
