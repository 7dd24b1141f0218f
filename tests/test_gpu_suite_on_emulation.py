"""The bodies of the `-m gpu` parity tests once more on CPU: the same test files (fixtures of the real reference, oracle comparisons,
edge cases, grids, refinement), with the ctypes layer bound to oracle/libsluamd_emul.so through SLUAMD_LIB -- the library's own host
sources over the serial restatement of the kernels (TEST INFRASTRUCTURE, built by `make -C oracle`).  What this pins without a GPU:
the planner, the schedules, the value movement and the C ABI against every fixture; what it cannot pin is the HIP kernels themselves
(the real `-m gpu` run does that).  Second run: the whole process under an adversarial stream schedule (test_stream_order.py)."""
import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["test_gpu_parity.py", "test_gpu_edge_cases.py", "test_gpu_refine.py", "test_gpu_grid.py", "test_gpu_fuzz.py"]


def _mpirun(cmd, env, cwd):
    """mpiexec with the retry test_gpu_dropin.py uses (MPICH start-up is occasionally flaky on freshly leased boxes)."""
    for attempt in range(3):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=cwd)
        if r.returncode == 0 or not ("MPI" in r.stderr and "nit" in r.stderr):
            break
    return r


@pytest.mark.parametrize("sched", ["", "1,7"])
def test_gpu_test_files_against_the_emulation_library(sched):
    so = os.path.join(ROOT, "oracle", "libsluamd_emul.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libsluamd_emul.so"])      # no-op when up to date; never a stale build
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


@pytest.mark.parametrize("npdep", [1, 2])
def test_complex16_reference_pipeline_with_512_column_supernodes_on_the_emulation(npdep, tmp_path):
    """pzgssvx3d of the real reference with SUPERLU_MAXSUP=512 (widest supernode ~400 columns), pzgstrf3d bound to the emulation
    library: residual parity with the untouched reference.  (The device twin of the refinement itself: test_gpu_grid.py.)"""
    import re
    import numpy as np
    from superlu_dist_amd import matgen
    zamd, zref = (os.path.join(ROOT, "oracle", "_ref", b) for b in ("slu_ref_zamd", "slu_ref_zdump"))
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(zamd) and os.path.exists(zref) and os.path.exists(mpiexec)):
        pytest.skip("prebuilt reference binaries / mpiexec not present")
    N = 20
    n, rp, ci, v = matgen.poisson3d(N)
    v = matgen.complex_shift(v, rp, ci, seed=4)
    np.savetxt(tmp_path / "a.perm", matgen.nd_perm_grid3d(N, N, N, leaf=64), fmt="%d")
    matgen.write_triplet_dat(str(tmp_path / "a.dat"), n, rp, ci, v)
    env = dict(os.environ, SLUAMD_LIB=os.path.join(ROOT, "oracle", "libsluamd_emul.so"), SUPERLU_MAXSUP="512", SUPERLU_RELAX="64",
               OMP_NUM_THREADS="1", SLUAMD_BIND_DEBUG="1")
    env.pop("LD_LIBRARY_PATH", None); env.pop("SLUAMD_EMUL_SCHED", None)
    res = {}
    for name, binary in (("amd", zamd), ("ref", zref)):
        r = _mpirun([mpiexec, "-n", str(npdep), binary, "-r", "1", "-c", "1", "-d", str(npdep), "-Q", "1", "-o", "none", "-e", "0", "-p", "0",
                     "-P", str(tmp_path / "a.perm"), str(tmp_path / "a.dat")], env, str(tmp_path))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        m = re.search(r"RESIDUAL (\S+) INFO (\d+)", r.stdout)
        assert m and int(m.group(2)) == 0, r.stdout[-1500:]
        res[name] = float(m.group(1))
        if name == "amd":
            assert max(int(w) for w in re.findall(r"widest_supernode (\d+)", r.stderr)) > 256
    assert res["amd"] < 1e-10 and abs(res["amd"] - res["ref"]) < 1e-10


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLUAMD_FUZZ_CASES", "12"))))
def test_reference_pipeline_fuzz_on_the_emulation(seed, tmp_path):
    """Fuzz of the drop-in boundary on CPU: random irregular matrices through the REAL reference (defaults: equilibration, MC64, MMD,
    its symbfact and 3D partition) on a random process grid with random SUPERLU_MAXSUP / SUPERLU_RELAX, pdgstrf3d + pdgstrs3d bound
    to the emulation library over the binding's MPI transport; residual parity with the untouched reference."""
    import re
    import numpy as np
    from superlu_dist_amd import matgen
    amd, ref = (os.path.join(ROOT, "oracle", "_ref", b) for b in ("slu_ref_amd", "slu_ref_dump"))
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(amd) and os.path.exists(ref) and os.path.exists(mpiexec)):
        pytest.skip("prebuilt reference binaries / mpiexec not present")
    rng = np.random.default_rng(500 + seed)
    if rng.integers(0, 2):
        n, rp, ci, v = matgen.random_unsym(int(rng.integers(150, 700)), float(rng.uniform(0.008, 0.04)), seed=seed)
    else:
        n, rp, ci, v = matgen.stencil3d_unsym(int(rng.integers(7, 13)), drop=float(rng.uniform(0.1, 0.5)), seed=seed)
    matgen.write_triplet_dat(str(tmp_path / "a.dat"), n, rp, ci, v)
    r, c, d = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1), (1, 1, 2), (2, 2, 2), (1, 1, 4), (3, 1, 1)][int(rng.integers(0, 8))]
    env = dict(os.environ, SLUAMD_LIB=os.path.join(ROOT, "oracle", "libsluamd_emul.so"), OMP_NUM_THREADS="1",
               SUPERLU_MAXSUP=str(int(rng.choice([16, 64, 256, 512]))), SUPERLU_RELAX=str(int(rng.choice([4, 20, 60]))))
    env.pop("LD_LIBRARY_PATH", None); env.pop("SLUAMD_EMUL_SCHED", None)
    if rng.integers(0, 2):
        env["SLUAMD_EMUL_SCHED"] = "%d,%d" % (int(rng.integers(1, 4)), int(rng.integers(1, 100)))
    refine = ["-i", "0"] if rng.integers(0, 2) else []
    res = {}
    for name, binary in (("amd", amd), ("ref", ref)):
        cmd = [mpiexec, "-n", str(r * c * d), binary, "-r", str(r), "-c", str(c), "-d", str(d), "-Q", "1", "-o", "none"] + refine + [str(tmp_path / "a.dat")]
        out = _mpirun(cmd, env, str(tmp_path))
        assert out.returncode == 0, (cmd, out.stdout[-1500:] + out.stderr[-1500:])
        m = re.search(r"RESIDUAL (\S+) INFO (\d+)", out.stdout)
        assert m and int(m.group(2)) == 0, out.stdout[-1500:]
        res[name] = float(m.group(1))
    assert res["amd"] < 1e-10 and abs(res["amd"] - res["ref"]) < 1e-10, (res, (r, c, d), env["SUPERLU_MAXSUP"], env["SUPERLU_RELAX"])


@pytest.mark.parametrize("ngpu", [1, 2, 8])
def test_bench_script_flow_on_the_emulation(ngpu, tmp_path):
    """bench.py end to end on CPU (emulation library; N > 1: torch.distributed.run, gloo-staged exchanges): ONE JSON line on stdout with
    the contract's keys, the right grid for N ranks, a residual < 1e-10.  (No timing is asserted: nothing here measures the device.)"""
    import json
    env = dict(os.environ, SLUAMD_LIB=os.path.join(ROOT, "oracle", "libsluamd_emul.so"), SLUAMD_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("SLUAMD_EMUL_SCHED", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ngpu), "--grid-side", "12", "--steps", "2", "--warmup", "2",
                        "--no-cpu-baseline", "--scale-n", "14", "--scale-n2", "16", "--configs4-n", "40"], env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == ngpu and j["steps"] == 2 and j["value"] > 0 and j["residual"] < 1e-10
    sp = j["scaling_point"]                # the same job one size up, at every N
    assert "error" not in sp and sp["n"] == 14 ** 3 and sp["value"] > 0 and sp["residual"] < 1e-10
    ss = j["strong_scaling_point"]         # ... and the strong-scaling size (180^3 in the measured line: the largest cube one GPU holds)
    assert "error" not in ss and ss["n"] == 16 ** 3 and ss["value"] > 0 and ss["residual"] < 1e-10 and "setup_breakdown" in ss
    assert "problem_generation_ordering_rhs_s" in j["setup_breakdown"] and "symbolic_s" in j["setup_breakdown"]
    if ngpu > 1:
        assert set(("exchange_ms", "reduce_ms", "schur_ms", "panel_ms")) <= set(ss["phases"])
    if ngpu == 1:                          # BASELINE.json configs[4] rides in the default N = 1 line (here on a 40 x 40 member of the family)
        c4 = j["configs4"]
        assert "error" not in c4 and c4["dtype"] == "c128" and c4["n"] == 1600 and c4["value"] > 0 and c4["residual"] < 1e-10 and c4["info"] == 0
        assert c4["roofline"]["bound"] == "mfma" and c4["roofline_solve"]["bound"] == "hbm"
    if ngpu > 1:                           # per-phase times of the profiled factorisation (max over ranks)
        assert set(("exchange_ms", "reduce_ms", "schur_ms", "panel_ms")) <= set(j["phases"])
        assert j["phases"]["reduce_ms"] > 0 and (ngpu < 8 or j["phases"]["exchange_ms"] > 0)
        pr = j["predicted"]                # the scaling model's figure for this very run, from the ranks' own plan tables
        assert "error" not in pr and pr["pdgstrf3d_ms_reductions_exposed"] >= pr["pdgstrf3d_ms_reductions_hidden"] > 0 and 0 < pr["predicted_efficiency"] <= 1.0
    assert ("1x1x1" if ngpu == 1 else "1x1x2" if ngpu == 2 else "2x2x2") in j["config"]["workload"]


def test_bench_falls_back_to_staged_exchanges_when_the_rccl_communicator_cannot_be_created(tmp_path):
    """bench.py --gpus 2 with the measured transport selected (RCCL) on a library that cannot create an RCCL communicator (the CPU test build has none): every
    rank must fall back to the host-staged exchanges TOGETHER -- a measured line that names the failure, not a job hung in a barrier."""
    import json
    env = dict(os.environ, SLUAMD_LIB=os.path.join(ROOT, "oracle", "libsluamd_emul.so"), OMP_NUM_THREADS="1")
    env.pop("SLUAMD_EMUL_SCHED", None); env.pop("SLUAMD_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29583",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid-side", "12", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-scaling-point"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["residual"] < 1e-10
    assert "RCCL communicator creation FAILED" in j["config"]["parallelism"] and "host-staged" in j["config"]["parallelism"]
