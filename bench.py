#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: LU factorisation GFLOP/s (pdgstrf3d) + solve time on MI355X.

One "step" = one pass of the hot path over the synthetic workload, inputs resident in HBM:
    device-side re-distribution of A into the resident L/U store  (zero-fill + scatter; not counted as flops)
    sluamd_pdgstrf3d   (numeric factorisation, the metric)
    sluamd_pdgstrs3d   (forward/backward block solve, nrhs = 1)
Workload at N=1 GPU = BASELINE.json configs[1]: 100^3 7-point Poisson (double), 1x1x1 grid, geometric
nested-dissection perm_c (MY_PERMC), Equil=NO, RowPerm=NOROWPERM, IterRefine=NOREFINE.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and `cpu_baseline`.
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0
HBM_BYTES = 288e9              # per MI355X


def kernel_source_hash():
    """sha256 (16 hex digits) of the kernel sources: the PMC-derived traffic figure is only valid for the kernels it was measured on."""
    import hashlib
    hsh = hashlib.sha256()
    for f in ("sluamd_kernels.hip", "sluamd_zkernels.inc"):
        hsh.update(open(os.path.join(ROOT, "superlu_dist_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def build_problem(N, leaf, workload="poisson3d"):
    from superlu_dist_amd import matgen
    if workload == "zgrid2d":
        # BASELINE.json configs[4]: "pzdrive3d complex16 on cg20.cua scaled 50x": cg20.cua is a complex operator on a 20x20
        # grid (n = 400); scaled 50x per grid side = 1000 x 1000 5-point complex grid operator (n = 10^6).  N = grid side.
        n, rp, ci, v = matgen.poisson3d(0, N, N, 1)
        v = matgen.complex_shift(v, rp, ci, seed=20)
        perm = matgen.nd_perm_grid3d(N, N, 1, leaf=leaf)
        xt = np.where((np.arange(n) % 2) == 1, 1.0, -1.0)[:, None].astype(np.complex128)
        b = matgen.csr_matvec(n, rp, ci, v, xt)
        return n, rp, ci, v, perm, np.asfortranarray(xt), b
    if workload == "audikw_like":
        # BASELINE.json configs[3] stand-in (SuiteSparse audikw_1 is not available offline): SPD 3-D mesh operator with 3 unknowns per
        # node, 27-point node coupling, randomly renumbered -- N = 68: n = 943 296 (audikw_1: 943 695), ~75 entries per row (82);
        # ordered WITHOUT geometry by the library's own nested dissection (sluamd_order_nd).  --matrix file.mtx reads the real one.
        from superlu_dist_amd import driver
        n, rp, ci, v = matgen.elasticity3d_like(N, drop=0.05, seed=1)
        perm = driver.order_nd(n, rp, ci, leaf=leaf)
        xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
        return n, rp, ci, v, perm, xt, b
    if workload.endswith(".mtx"):
        from superlu_dist_amd import driver
        n, rp, ci, v = matgen.read_matrix_market(workload)
        perm = driver.order_nd(n, rp, ci, leaf=leaf)
        xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
        return n, rp, ci, v, perm, xt, b
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=leaf)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
    return n, rp, ci, v, perm, xt, b


def _mkl_reference():
    """The reference built on the image's MKL (oracle/ref/Makefile target ref_mkl: -DUSE_VENDOR_BLAS -DSLU_HAVE_LAPACK, no CBLAS objects): its path
    and a description of the BLAS, or (None, reason) when the binary or the library did not travel / is not on this box."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref_mkl", "slu_ref_dump")
    if not os.path.exists(ref_bin):
        return None, "oracle/_ref_mkl/slu_ref_dump not built (make -C oracle/ref ref_mkl where /root/reference exists)"
    lib = "/opt/conda/lib/libmkl_rt.so"
    if not os.path.exists(lib):
        return None, lib + " not on this box"
    ver = ""
    try:
        import ctypes
        m = ctypes.CDLL(lib, mode=ctypes.RTLD_LOCAL)
        buf = ctypes.create_string_buffer(200)
        m.mkl_get_version_string(buf, 200)
        ver = buf.value.decode(errors="replace").strip()
    except Exception:
        pass
    return ref_bin, (ver or "Intel MKL (libmkl_rt)")


def _run_reference(ref_bin, args, env, nproc=1, wrap_dir=None, threads=1, timeout=150):
    """One run of the reference harness (slu_ref_dump); nproc > 1 goes through mpiexec with every rank pinned to its own core range when taskset
    and a known rank variable allow it.  Returns (record or None, pinned, wall seconds, error text)."""
    import shutil
    pinned = False
    cmd = [ref_bin] + args
    if nproc > 1:
        mpiexec = os.environ.get("SLUAMD_MPIEXEC") or shutil.which("mpiexec") or ("/opt/conda/bin/mpiexec" if os.path.exists("/opt/conda/bin/mpiexec") else None)
        if not mpiexec:
            return None, False, 0.0, "mpiexec not available (SLUAMD_MPIEXEC / PATH / /opt/conda/bin)"
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
        if shutil.which("taskset") and len(allowed) >= nproc * threads and wrap_dir:
            # rank q -> the q-th run of `threads` CPUs this process may use; the rank comes from whichever variable the launcher sets (MPICH / Hydra:
            # PMI_RANK, Open MPI: OMPI_COMM_WORLD_RANK, PMIx launchers: PMIX_RANK); the wrapper REPORTS what it did, the record follows it
            wrap = os.path.join(wrap_dir, "rank.sh")
            with open(wrap, "w") as f:
                f.write("#!/bin/sh\nr=${PMI_RANK:-${OMPI_COMM_WORLD_RANK:-${PMIX_RANK:-}}}\ncase \"$r\" in\n")
                for q in range(nproc):
                    f.write(f"{q}) echo PINNED {q} >&2; exec taskset -c {','.join(str(c) for c in allowed[q * threads:(q + 1) * threads])} {ref_bin} {' '.join(args)};;\n")
                f.write(f"*) echo UNPINNED >&2; exec {ref_bin} {' '.join(args)};;\nesac\n")
            os.chmod(wrap, 0o755)
            cmd = [mpiexec, "-n", str(nproc), wrap]
        else:
            cmd = [mpiexec, "-n", str(nproc), ref_bin] + args
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return None, False, time.perf_counter() - t0, f"timeout {timeout} s"
    wall = time.perf_counter() - t0
    if nproc > 1:
        pinned = "UNPINNED" not in r.stderr and sum(1 for l in r.stderr.splitlines() if l.startswith("PINNED ")) >= nproc
    line = [l for l in r.stdout.splitlines() if l.startswith("REFTIMES")]
    if r.returncode != 0 or not line:
        return None, pinned, wall, ("rc %d: " % r.returncode) + (r.stderr or r.stdout)[-200:]
    tok = line[0].split()
    res = [l for l in r.stdout.splitlines() if l.startswith("RESIDUAL")]
    return {"factor_s": float(tok[4]), "solve_s": float(tok[7]), "ops_FACT": float(tok[10]),
            "residual": float(res[0].split()[1]) if res else None}, pinned, wall, None


def _ref_env(threads, relax, maxsup, mkl_layer=None):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores", SUPERLU_MAXSUP=str(maxsup), SUPERLU_RELAX=str(relax))
    env.pop("LD_LIBRARY_PATH", None)     # the binaries carry RUNPATH=/opt/conda/lib (MPICH, MKL)
    if mkl_layer:
        # libmkl_rt picks Intel's OpenMP runtime by default -- two OpenMP runtimes in one process with the reference's libgomp regions give WRONG
        # factors (seen here: residual O(1) at 8 threads); SEQUENTIAL = one MKL call per OpenMP thread of the reference, GNU = MKL threads on libgomp
        env["MKL_THREADING_LAYER"] = mkl_layer
    return env


def cpu_baseline_grid(N, leaf, relax, maxsup, ref_bin, host_cores, mkl=None):
    """SURVEY 8(d)'s second CPU leg: the real reference on a 2 x 2 x 2 process grid, 8 MPI ranks x host_cores/8 OpenMP threads (capped at 8:
    the reference stops scaling past ~8 threads per rank, see the 1-rank sweep), every rank pinned to its own contiguous core range.
    RowPerm stays at the reference's default (LargeDiag_MC64: the identity on this diagonally dominant matrix; v9.2.1's pdgssvx3d fails in
    symbfact with NOROWPERM on a 2 x 2 x 2 grid), everything else as the 1-rank leg.  Timer = stat.utime[FACT] of rank 0 (pdgstrf3d.c:331,395)."""
    from superlu_dist_amd import driver, matgen
    n, rp, ci, v, perm, xt, b = build_problem(N, leaf)
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    flops = symb.flops
    symb.free()
    th = max(1, min(8, host_cores // 8))     # 8 threads per rank: the best point of the 1-rank sweep (16: 69 s at 60^3 on the round-4 box, slower than 8)
    with tempfile.TemporaryDirectory() as tmp:
        mpath = os.path.join(tmp, "a.dat"); ppath = os.path.join(tmp, "a.perm")
        matgen.write_triplet_dat(mpath, n, rp, ci, v)
        np.savetxt(ppath, perm, fmt="%d")
        env = _ref_env(th, relax, maxsup, "SEQUENTIAL" if mkl else None)
        args = ["-r", "2", "-c", "2", "-d", "2", "-e", "0", "-p", "1", "-i", "0", "-Q", "1", "-P", ppath, "-o", "none", mpath]
        rec, pinned, wall, err = _run_reference(ref_bin, args, env, nproc=8, wrap_dir=tmp, threads=th, timeout=150)
    if rec is None:
        return {"error": err, "sample": f"{N}^3", "ranks": 8, "threads_per_rank": th}
    return {"kind": "reference+mkl" if mkl else "reference", "blas": mkl or "vendored f2c CBLAS (scalar dgemm)",
            "grid": "2x2x2", "ranks": 8, "threads_per_rank": th, "cores": 8 * th, "pinned": pinned,
            "sample": f"{N}^3 7-pt Poisson, same ND perm_c/relax/maxsup, 2x2x2 grid (mpiexec -n 8), nrhs=1", "flops": flops,
            "factor_s": rec["factor_s"], "solve_s": rec["solve_s"], "value": flops / rec["factor_s"] / 1e9, "unit": "GFLOP/s",
            "reference_ops_FACT": rec["ops_FACT"], "residual": rec["residual"], "wall_s": wall}


def cpu_baseline(N, leaf, relax, maxsup, want_reference=True, grid_n=0, mkl_n=60, mkl_grid_n=70):
    """Time the CPU legs on bounded samples (N^3 Poisson, same ordering / supernode parameters as the benched problem).
    kind = "reference+mkl": the real reference's pdgstrf3d built with its own vendor-BLAS switch (-DUSE_VENDOR_BLAS, CMakeLists.txt:389-407,
                            dsuperlu_blas.c:40-99) on the image's MKL -- oracle/_ref_mkl/slu_ref_dump; the headline CPU number when it is on the box
    kind = "reference":     the same on the reference's vendored f2c CBLAS (oracle/_ref/slu_ref_dump: the parity oracle's build)
    kind = "port":          oracle/slu_oracle.c (our CPU restatement, OpenMP over (L block, U block) pairs)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = min(os.cpu_count() or 1, 32)      # more threads only add OpenMP fork/join overhead on these loop sizes
    os.environ["OMP_NUM_THREADS"] = str(cores)   # before libgomp is loaded by the oracle library
    import oracle as orc                      # cpu_baseline leg: the only place bench.py touches oracle/
    from superlu_dist_amd import driver, matgen
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "slu_ref_dump")) or _mkl_reference()[0] is not None
    if not have_ref:
        # no build of the real reference on this box (round 6: it cannot be built in the build container, its generated config header is absent): the port is the
        # only CPU leg, on a sample of about ten seconds of its own work (60^3: 5.4e11 flop) instead of the 40^3 the scalar-CBLAS reference needed
        N = max(N, 60)
    n, rp, ci, v, perm, xt, b = build_problem(N, leaf)
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    flops = symb.flops
    out = {"unit": "GFLOP/s", "sample": f"{N}^3 7-pt Poisson, same ND perm_c/relax/maxsup, 1x1x1 grid, nrhs=1",
           "flops": flops}
    # ---- port (always measured: it also validates the harness) ----
    symb.distribute_host(v)
    fs = symb.flat_store()
    o = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz,
                    fs.Unzval_off, fs.Unzval)
    t0 = time.perf_counter()
    info, tiny, fl = orc.dfactor(o)
    t_port = time.perf_counter() - t0
    xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
    t0 = time.perf_counter()
    y = orc.dsolve(o, xp)
    t_port_solve = time.perf_counter() - t0
    x = y[symb.perm_c, :]
    res = float(np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b))
    out.update({"kind": "port", "value": flops / t_port / 1e9, "cores": orc.num_threads(), "factor_s": t_port,
                "solve_s": t_port_solve, "residual": res, "port_value": flops / t_port / 1e9,
                "port": "oracle/slu_oracle.c: the CPU restatement of the reference's algorithm (OpenMP over (L block, U block) pairs, its own blocked GEMM), pinned to the "
                        "real reference by tests/golden/; kind = 'port' because no build of the reference is on this box"})
    symb.free()
    host_cores = os.cpu_count() or 1
    out["host_cores"] = host_cores
    if not want_reference:
        return out

    def sweep_leg(ref_bin, Ns, layer, budget_s, thread_list):
        """thread sweep of one reference binary on the Ns^3 sample: small counts first, ends once more threads only slow it down or the budget is spent"""
        n2, rp2, ci2, v2, perm2, _, _ = build_problem(Ns, leaf)
        sy = driver.Symbolic(n2, rp2, ci2, perm2, relax=relax, maxsup=maxsup)
        fl2 = sy.flops
        sy.free()
        sweep, spent, best = [], 0.0, None
        with tempfile.TemporaryDirectory() as tmp:
            mpath = os.path.join(tmp, "a.dat"); ppath = os.path.join(tmp, "a.perm")
            matgen.write_triplet_dat(mpath, n2, rp2, ci2, v2)
            np.savetxt(ppath, perm2, fmt="%d")
            args = ["-r", "1", "-c", "1", "-d", "1", "-e", "0", "-p", "0", "-i", "0", "-Q", "1", "-P", ppath, "-o", "none", mpath]
            for th in [t for t in thread_list if t <= host_cores] or [host_cores]:
                if spent > budget_s:
                    break
                if best and th > best["threads"] and sweep and sweep[-1].get("factor_s", 0) > 1.25 * best["factor_s"]:
                    continue                 # more threads already made it slower
                rec, _, wall, err = _run_reference(ref_bin, args, _ref_env(th, relax, maxsup, layer), timeout=max(30, int(budget_s)))
                spent += wall
                if rec is None:
                    sweep.append({"threads": th, "error": err})
                    continue
                rec = dict(threads=th, gflops=fl2 / rec["factor_s"] / 1e9, **rec)
                if layer: rec["mkl_threading_layer"] = layer
                sweep.append(rec)
                if rec["residual"] is not None and rec["residual"] < 1e-10 and (best is None or rec["factor_s"] < best["factor_s"]):
                    best = rec
            if best and layer == "SEQUENTIAL" and spent + 1.5 * best["factor_s"] < budget_s:     # one more point, if it fits the budget: MKL's own threads (on libgomp) at the best thread count
                rec, _, wall, err = _run_reference(ref_bin, args, _ref_env(best["threads"], relax, maxsup, "GNU"), timeout=max(30, int(budget_s)))
                if rec is not None:
                    rec = dict(threads=best["threads"], gflops=fl2 / rec["factor_s"] / 1e9, mkl_threading_layer="GNU", **rec)
                    sweep.append(rec)
                    if rec["residual"] is not None and rec["residual"] < 1e-10 and rec["factor_s"] < best["factor_s"]:
                        best = rec
        return fl2, sweep, best

    # ---- the real reference on its vendored CBLAS (the parity oracle's build), one bounded thread sweep on the N^3 sample ----
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "slu_ref_dump")
    cblas = None
    if os.path.exists(ref_bin):
        try:
            fl2, sweep, best = sweep_leg(ref_bin, N, None, 45.0, (8, 16, 4))
            if best:
                cblas = {"kind": "reference", "blas": "vendored f2c CBLAS (scalar dgemm)", "sample": out["sample"], "flops": fl2,
                         "value": fl2 / best["factor_s"] / 1e9, "unit": "GFLOP/s", "cores": best["threads"], "factor_s": best["factor_s"],
                         "solve_s": best["solve_s"], "reference_ops_FACT": best["ops_FACT"], "residual": best["residual"], "thread_sweep": sweep}
        except Exception as e:
            out["reference_error"] = str(e)[:200]
    # ---- the real reference on MKL (its own USE_VENDOR_BLAS switch): the headline CPU leg ----
    mkl_bin, mkl_desc = _mkl_reference()
    mkl = None
    if mkl_bin:
        try:
            fl2, sweep, best = sweep_leg(mkl_bin, mkl_n, "SEQUENTIAL", 100.0, (8, 16, 32))
            if best:
                mkl = {"kind": "reference+mkl", "blas": mkl_desc, "mkl_threading_layer": best.get("mkl_threading_layer"),
                       "sample": f"{mkl_n}^3 7-pt Poisson, same ND perm_c/relax/maxsup, 1x1x1 grid, nrhs=1", "flops": fl2,
                       "value": fl2 / best["factor_s"] / 1e9, "unit": "GFLOP/s", "cores": best["threads"], "factor_s": best["factor_s"],
                       "solve_s": best["solve_s"], "reference_ops_FACT": best["ops_FACT"], "residual": best["residual"], "thread_sweep": sweep,
                       "gflops_per_core": fl2 / best["factor_s"] / 1e9 / best["threads"]}
        except Exception as e:
            out["reference_mkl_error"] = str(e)[:200]
    else:
        out["reference_mkl_unavailable"] = mkl_desc
    head = mkl or cblas
    if head:
        out.update({k: head[k] for k in ("kind", "blas", "sample", "flops", "value", "cores", "factor_s", "solve_s", "reference_ops_FACT", "residual", "thread_sweep")})
        out["gflops_per_core"] = head["value"] / head["cores"]
        if mkl:
            out["mkl_threading_layer"] = mkl["mkl_threading_layer"]
            if cblas: out["reference_cblas"] = cblas
        out["note"] = ("reference v9.2.1 pdgstrf3d (OpenMP over block pairs, OMP_PROC_BIND=close OMP_PLACES=cores) timed by stat.utime[FACT], best of the "
                       "thread sweep; " + ("BLAS = MKL through the reference's own -DUSE_VENDOR_BLAS switch (dgemm / dtrsm / dger inside its OpenMP regions), "
                                           "LAPACK on (-DSLU_HAVE_LAPACK); `reference_cblas` is the same code on its vendored scalar CBLAS. " if mkl else
                                           "BLAS = the reference's vendored f2c CBLAS (scalar dgemm, ~1 GFLOP/s per core): oracle/_ref_mkl or libmkl_rt absent on this box. ")
                       + "GFLOP/s uses our symbolic flop count of OUR supernode partition (reference_ops_FACT = the reference's own tally on the same matrix). "
                         "Measured on the bounded sample only: the reference's rate on the benched 100^3 problem is NOT measured (hours at these rates); "
                         "supernodes are wider there, so its GFLOP/s would be somewhat higher")
    if grid_n and not have_ref:
        out["grid_2x2x2"] = {"skipped": "no build of the reference on this box (oracle/_ref, oracle/_ref_mkl): the 2x2x2 CPU leg runs the reference's own pddrive3d flow"}
    elif grid_n:
        try:
            out["grid_2x2x2"] = (cpu_baseline_grid(mkl_grid_n, leaf, relax, maxsup, mkl_bin, host_cores, mkl=mkl_desc) if mkl_bin and mkl
                                 else cpu_baseline_grid(grid_n, leaf, relax, maxsup, ref_bin, host_cores))
        except Exception as e:
            out["grid_2x2x2"] = {"error": str(e)[:200]}
    return out


class ClockSampler:
    """Shader clock of the device while the timed steps run (a 0.57-vs-0.60 box difference should explain itself): a thread polls the
    current sclk level of the PCI device HIP device `dev` maps to (sysfs pp_dpm_sclk, the line marked '*') every 50 ms.  Harness only -- nothing of the hot path."""

    def __init__(self, dev=0):
        import threading
        self.samples, self._stop, self.path = [], threading.Event(), None
        bus = None
        try:
            import ctypes
            from superlu_dist_amd import _lib
            buf = ctypes.create_string_buffer(64)
            if _lib.load().sluamd_device_pci_bus_id(int(dev), buf, 64) == 0:
                bus = buf.value.decode().lower()          # "0000:c5:00.0"
        except Exception:
            bus = None
        import glob
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        if bus:
            for c in cands:
                if os.path.basename(os.path.realpath(os.path.dirname(c))).lower() == bus:
                    self.path = c
        self.cands = cands
        self._thr = threading.Thread(target=self._run, daemon=True)

    def _read(self, path):
        try:
            for ln in open(path):
                if "*" in ln:
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            if self.path:
                v = self._read(self.path)
            else:          # device not identified: the busiest visible card (the one this job drives)
                vs = [x for x in (self._read(c) for c in self.cands) if x]
                v = max(vs) if vs else None
            if v:
                self.samples.append(v)
            self._stop.wait(0.05)

    def __enter__(self):
        if not os.environ.get("SLUAMD_BENCH_NO_CLOCK"):
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thr.is_alive():
            self._thr.join(timeout=1.0)

    def summary(self):
        if not self.samples:
            return None
        x = sorted(self.samples)
        return {"sclk_mhz_min": x[0], "sclk_mhz_median": x[len(x) // 2], "sclk_mhz_max": x[-1], "samples": len(x),
                "source": (self.path or "max over /sys/class/drm/card*/device/pp_dpm_sclk") + ", 50 ms polls during the timed steps and the profiled passes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--grid-side", dest="n", type=int, default=100, help="grid side of the N^3 Poisson problem")
    ap.add_argument("--leaf", type=int, default=64)
    ap.add_argument("--relax", type=int, default=64)
    ap.add_argument("--maxsup", type=int, default=None,
                    help="supernode width cap (the reference's SUPERLU_MAXSUP / sp_ienv_dist(3)); default 256, and 64 for the complex16 "
                         "workload, whose diagonal-block kernel is one workgroup per supernode (measured on zgrid2d 1000: 64 -> 25.7 ms, "
                         "128 -> 30.0 ms, 256 -> 39.1 ms)")
    ap.add_argument("--scale-n", type=int, default=150,
                    help="grid side of the SCALING POINT reported beside the headline configuration at every N (150^3: 1.5e14 flop, "
                         "90 GB of factors -- seconds of work per step, so that exchange latency does not dominate the N > 1 runs); 0 = skip")
    ap.add_argument("--scale-n2", type=int, default=180,
                    help="grid side of the STRONG-SCALING point (180^3: 188 GB of factors, the largest cube one 288 GB GPU holds, so the 1-GPU time exists; "
                         "one timed step after one warm-up step); 0 = skip")
    ap.add_argument("--no-scaling-point", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs4", action="store_true",
                    help="skip the `configs4` block of the default line (BASELINE.json configs[4]: complex16 1000 x 1000 grid operator, 5 steps)")
    ap.add_argument("--configs4-n", type=int, default=1000, help="grid side of the configs4 block (1000 = BASELINE.json configs[4]; the CPU flow test lowers it)")
    ap.add_argument("--cpu-grid-n", type=int, default=50,
                    help="grid side of the 2x2x2 leg of the CPU baseline (8 MPI ranks x host_cores/8 threads of the real reference) on the vendored CBLAS; 0 = skip")
    ap.add_argument("--cpu-mkl-n", type=int, default=60, help="grid side of the 1-rank CPU leg when the MKL build of the reference is on the box (oracle/_ref_mkl)")
    ap.add_argument("--cpu-mkl-grid-n", type=int, default=70, help="grid side of the 2x2x2 CPU leg with the MKL build")
    ap.add_argument("--workload", default="poisson3d", choices=["poisson3d", "zgrid2d", "audikw_like"],
                    help="poisson3d = BASELINE configs[1] (default, the metric's config); zgrid2d = configs[4] family "
                         "(complex16 2-D grid operator, use --n 1000); audikw_like = configs[3] stand-in (use --n 68)")
    ap.add_argument("--matrix", default=None, help="MatrixMarket file instead of a generated workload (e.g. SuiteSparse audikw_1.mtx), ordered by sluamd_order_nd")
    args = ap.parse_args()
    if args.maxsup is None: args.maxsup = 64 if args.workload == "zgrid2d" else 256

    if args.gpus > 1 and "RANK" not in os.environ:
        # started as a plain `python bench.py --gpus N`: re-launch under torch.distributed.run, one rank per GPU
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        argv = ["--grid-side" if a == "--n" else a for a in sys.argv[1:]]   # torchrun's parser trips over the prefix "--n"
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv)

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dist_backend = os.environ.get("SLUAMD_DIST_BACKEND", "rccl")
    dist_backend_used = [dist_backend]      # "gloo" after a failed RCCL communicator creation (N > 1)
    if world > 1:
        import torch
        import torch.distributed as dist
        # torch.distributed (gloo) is only the out-of-band channel: it ships the ncclUniqueId and provides the timing
        # barrier.  Every exchange of the hot path is RCCL called directly by libsluamd.so (ncclSend / ncclRecv on its HIP
        # streams).  SLUAMD_DIST_BACKEND=gloo: debugging aid for boxes with fewer GPUs than ranks (ranks share devices, the
        # library's exchanges are staged through host memory over gloo); the measured configuration is always rccl.
        ndev = torch.cuda.device_count()     # 0 only in the CPU test of this script (SLUAMD_LIB = the emulation library)
        if dist_backend != "rccl":
            local_rank = local_rank % max(ndev, 1)
        if ndev:
            torch.cuda.set_device(local_rank)
        # gloo reports its connections ("[Gloo] Rank 0 is connected to ...") on stdout: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from superlu_dist_amd import _lib, driver, matgen
    L = _lib.load()
    if L.sluamd_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the hot path has no CPU fallback")

    def measure(N, steps, warm, workload, maxsup=None):
        """Build the N-sided problem, factor + solve `steps` times after `warm` warm-up steps; returns the measurements and keeps the
        handle alive for the caller's extra passes (profile, refinement)."""
        t_setup = time.perf_counter()
        n, rp, ci, v, perm, xt, b = build_problem(N, args.leaf, workload)
        t_problem = time.perf_counter() - t_setup
        symb = driver.Symbolic(n, rp, ci, perm, relax=args.relax, maxsup=maxsup or args.maxsup)
        t_symb = time.perf_counter() - t_setup - t_problem
        grid = (1, 1, 1)
        if world == 1:
            if (symb.nnzL + symb.nnzU) * (16 if workload == "zgrid2d" else 8) * 1.06 > HBM_BYTES:
                raise SystemExit(f"bench.py: {N}^3 does not fit one {HBM_BYTES / 1e9:.0f} GB GPU: nnz(L+U) = {symb.nnzL + symb.nnzU} = "
                                 f"{(symb.nnzL + symb.nnzU) * 8 / 1e9:.0f} GB of factor values (run it on a process grid: --gpus 8)")
            h = driver.LUHandle.from_symbolic(symb, v, device=local_rank)
        else:   # Pr x Pc x Pz process grid, one rank per GPU (8 -> 2 x 2 x 2 = BASELINE.json's grid); SLUAMD_GRID="r,c,z" overrides
            from superlu_dist_amd import grid3d
            grid = tuple(int(t) for t in os.environ["SLUAMD_GRID"].split(",")) if os.environ.get("SLUAMD_GRID") else grid3d.default_grid(world)
            assert grid[0] * grid[1] * grid[2] == world
            sn_tree = symb.partition(grid[2]) if grid[2] > 1 else None
            # pre-flight: the factors must fit the ranks' HBM -- fail loudly with the byte count, before any allocation
            vals, rep, _idx = symb.grid_footprint(*grid, sn_tree)
            # + index images / inverse blocks (~5 %) + on XY layers the double-buffered scratch for the panels received per DAG level, which
            # grows with n, not with nnz (measured on 2 x 2 x 2 at 100^3: 1.5 kB per row, profiles/r03_grid_footprint.txt)
            need = float(vals.max()) * 8 * 1.05 + (1500.0 * n if grid[0] * grid[1] > 1 else 0.0)
            if need > HBM_BYTES:
                raise SystemExit(f"bench.py: {N}^3 does not fit a {grid[0]}x{grid[1]}x{grid[2]} grid of {HBM_BYTES / 1e9:.0f} GB GPUs: the fullest rank stores "
                                 f"{vals.max() * 8 / 1e9:.1f} GB of factor values ({rep.max() * 8 / 1e9:.1f} GB of them ancestor panels replicated along Z), "
                                 f"~{need / 1e9:.0f} GB with scratch and tables; nnz(L+U) = {(symb.nnzL + symb.nnzU) * 8 / 1e9:.0f} GB in total")
            if need > 0.7 * HBM_BYTES and "SLUAMD_NO_TILE_MAPS" not in os.environ:
                # the per-tile records of the Schur kernel are an accelerator (2-4 ms at 100^3) that costs 20-29 % of the factor bytes: at sizes that
                # fill the device they stay off (bench.py --gpus 8 --n 300 picks this by itself)
                os.environ["SLUAMD_NO_TILE_MAPS"] = "1"
            if dist_backend == "rccl" and "comm" not in comm_cache:
                # the communicator is created by the library (ncclCommInitRank); if that fails on ANY rank (no peer access, a driver without dmabuf IPC ...)
                # every rank falls back to the host-staged transport together, and the line says so: a slow measured number instead of a hung job
                import torch
                err = ""
                try:
                    comm_cache["comm"] = grid3d.rccl_comm(dist, *grid, local_rank)
                except Exception as e:     # noqa: BLE001 -- reported in the JSON line
                    err = str(e)[:300]
                bad = torch.tensor([1 if err else 0])
                dist.all_reduce(bad)
                if int(bad.item()):
                    comm_cache.pop("comm", None)
                    comm_cache["rccl_error"] = err or "communicator creation failed on another rank"
                    dist_backend_used[0] = "gloo"
            if dist_backend_used[0] == "rccl":
                pass
            elif "comm" not in comm_cache:
                comm_cache["tcomm"] = grid3d.TorchComm(dist, *grid)
                comm_cache["comm"] = comm_cache["tcomm"].handle
            h = grid3d.GridHandle.from_symbolic(symb, v, comm_cache["comm"], sn_tree, device=local_rank)
        t_setup = time.perf_counter() - t_setup
        # where the pre-processing goes: synthetic input (matrix + geometric ordering + right-hand side: the harness), the symbolic
        # factorisation (sluamd_dsymbfact), and the handle's creation split by the library itself (sluamd_setup_times: planner phases, arena, uploads, A)
        setup_breakdown = {"problem_generation_ordering_rhs_s": t_problem, "symbolic_s": t_symb, "handle_create_s": t_setup - t_problem - t_symb,
                           "handle_create_phases_s": (h.setup_times() if hasattr(h, "setup_times") else {})}
        thresh = driver.pivot_thresh(n, rp, ci, np.abs(v))
        xp = np.zeros_like(b, order="F"); xp[symb.perm_c, :] = b
        # N > 1: the reference's solve boundary -- B distributed by block rows over layer 0 (sluamd_pdgstrs3d_dist)
        nl0 = grid[0] * grid[1]
        cuts = np.linspace(0, n, nl0 + 1).astype(np.int64)
        f0, f1 = (int(cuts[rank]), int(cuts[rank + 1])) if rank < nl0 else (0, 0)

        def step():
            h.reset_values()                    # device-side re-distribution of A (zero fill + scatter): part of every step
            info = h.pdgstrf3d(thresh)          # N > 1: collective -- Z-level loop, XY panel exchange, ancestor reduction, info all-reduce
            if world == 1:
                y = h.pdgstrs3d(xp)
            else:
                y = h.pdgstrs3d_dist(b[f0:f1, :], f0, symb.perm_c)     # collective: B_to_X, distributed sweeps, X_to_B; my rows of x back
            st = h.stats()
            return info, y, st["t_factor_ms"], st["t_solve_ms"]          # HIP-event times of the two phases

        t_first = None
        for w in range(warm):
            sync(); tw = time.perf_counter()
            info, y, _, _ = step()
            if w == 0:
                sync(); t_first = time.perf_counter() - tw      # the first step also writes the per-tile records of the Schur kernel and pays the runtime's one-time costs
        setup_breakdown["first_step_s"] = t_first
        sync()
        t0 = time.perf_counter()
        fact_ms, solve_ms = [], []
        for _ in range(steps):
            ts = time.perf_counter()
            info, y, fm, sm = step()
            fact_ms.append(fm); solve_ms.append(sm)
            if os.environ.get("SLUAMD_BENCH_TRACE"):
                print("step wall %.2f ms (factor %.2f solve %.2f)" % (1e3 * (time.perf_counter() - ts), fm, sm), file=sys.stderr)
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            import torch
            tt = torch.tensor([elapsed, float(np.mean(fact_ms)), float(np.mean(solve_ms))], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt[0].item())
            fact_ms, solve_ms = [float(tt[1].item())], [float(tt[2].item())]
            parts = [None] * world                     # the solution's rows from the layer-0 ranks (outside the timed region)
            dist.all_gather_object(parts, (f0, np.asarray(y)))
            x = np.zeros_like(b, order="F")
            for q0, yq in parts:
                if yq.shape[0]:
                    x[q0:q0 + yq.shape[0], :] = yq
        else:
            x = y[symb.perm_c, :]
        res = float(np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b))
        err = float(np.abs(x - xt).max())
        return dict(n=n, rp=rp, ci=ci, v=v, xt=xt, b=b, symb=symb, h=h, grid=grid, thresh=thresh, t_setup=t_setup, info=info, x=x,
                    fact_ms=fact_ms, solve_ms=solve_ms, elapsed=elapsed, res=res, err=err, steps=steps, setup_breakdown=setup_breakdown)

    def predicted_block(hh, nn, flops, measured_ms):
        """N > 1: what superlu_dist_amd/scale_model.py predicts for THIS run from the ranks' own plan tables (gathered over the out-of-band channel), beside what
        was measured -- so that the first hardware scaling run is compared against something.  predicted_efficiency = model T_1 / (N x model T_N);
        the model's one-GPU time needs the one-rank plan, which an N > 1 run does not have: it is recomputed from the SAME constants by scripts/scale_model.py and,
        for the default sizes, kept in profiles/r06_scale_model.txt -- here T_1 is the sum of the ranks' Schur and panel work at the one-GPU rates (no exchange)."""
        try:
            from superlu_dist_amd import scale_model
            tab = hh.plan_table()
            tabs = [None] * world
            dist.all_gather_object(tabs, np.asarray(tab))
            Th, Te, _rows = scale_model.predict(tabs)
            p = scale_model.DEFAULTS
            work = sum(float(t[:, 4].sum()) for t in tabs) / (p["r_big"] * 1e12) + sum(float(t[:, 6].sum()) for t in tabs) / (p["r_panel"] * 1e12)
            return {"model": "superlu_dist_amd/scale_model.py", "constants": p, "pdgstrf3d_ms_reductions_hidden": 1e3 * Th, "pdgstrf3d_ms_reductions_exposed": 1e3 * Te,
                    "measured_factor_ms": measured_ms, "measured_over_predicted": measured_ms / (1e3 * Te) if Te > 0 else None,
                    "one_gpu_work_ms_at_model_rates": 1e3 * work, "predicted_efficiency": work / (world * Te) if Te > 0 else None,
                    "note": "link constants (GB/s per peer and direction, us per exchange phase) are assumptions until this line exists on hardware"}
        except Exception as e:      # noqa: BLE001 -- reported in the line
            return {"error": str(e)[:200]}

    def sync():
        L.sluamd_device_synchronize()
        if dist is not None:
            dist.barrier()

    comm_cache = {}
    zwork = args.workload == "zgrid2d"
    if zwork and world > 1:
        raise SystemExit("bench.py: the complex16 workload is single-GPU")
    # untimed warm-up: W complete steps, each exactly what a timed step is, and never fewer than two -- the HIP runtime spreads its
    # one-time costs over the first TWO steps of a process (first step: code objects, the pinned / staging buffers of the host
    # copies; second step: another 15-25 ms inside the first host-to-device copy of the solve, SLUAMD_BENCH_TRACE=1 shows the per-step
    # wall times); "warmup" in the JSON line is the number actually run
    n_warm = max(2, args.warmup)
    if args.matrix:
        args.workload = args.matrix
    clock = ClockSampler(local_rank)
    clock.__enter__()
    M = measure(args.n, args.steps, n_warm, args.workload)
    n, rp, ci, v, xt, b, symb, h, grid, thresh = (M[k] for k in ("n", "rp", "ci", "v", "xt", "b", "symb", "h", "grid", "thresh"))
    t_setup, info, x, fact_ms, solve_ms, elapsed, res, err = (M[k] for k in ("t_setup", "info", "x", "fact_ms", "solve_ms", "elapsed", "res", "err"))

    # one extra profiled step: per-kernel-family HIP-event times on the compute stream
    # (three passes at N = 1: roofline.frac is reported from the FASTEST with the spread beside it, so that box-to-box and run-to-run differences read off the line)
    h.set_profile(True)                  # N > 1: collective like every factorisation (serial schedule on every rank)
    prof_passes = []
    for _ in range(3 if world == 1 else 1):
        h.reset_values(); h.pdgstrf3d(thresh)
        prof_passes.append(h.stats())
    stp = min(prof_passes, key=lambda q: q["t_schur_ms"])
    h.set_profile(False)
    clock.__exit__()

    # accuracy row: one untimed pass of IterRefine=SLU_DOUBLE (pdgsrfs3d on the device) on the last solution
    accuracy = None
    if world == 1 and not zwork:
        try:
            h.attach_matrix(n, rp, ci, v, symb.perm_c)
            L.sluamd_device_synchronize(); t_r = time.perf_counter()
            xr, berr, rsteps = h.pdgsrfs3d(b, x)
            L.sluamd_device_synchronize(); t_r = time.perf_counter() - t_r
            accuracy = {"iter_refine": "SLU_DOUBLE (pdgsrfs3d on the device)", "berr": float(berr.max()), "steps": int(rsteps),
                        "refine_ms": 1e3 * t_r,
                        "residual_after": float(np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, xr)) / np.linalg.norm(b)),
                        "max_abs_err_vs_xtrue_after": float(np.abs(xr - xt).max())}
        except Exception as e:
            accuracy = {"error": str(e)[:200]}

    st = h.stats()
    F = symb.flops if world > 1 else st["flops_schur_exact"] + st["flops_panel"]   # whole-matrix flop count either way (per-rank stats are local)
    ms_per_step = 1e3 * elapsed / args.steps
    value = F * args.steps / elapsed / 1e9
    schur_tf = st["flops_schur_exact"] / (stp["t_schur_ms"] * 1e-3) / 1e12 if stp["t_schur_ms"] > 0 else 0.0
    # HBM traffic of the dominant kernel from PMC counters: cannot be collected inside this process (rocprofv3 --pmc
    # needs its own passes), so the value measured on this same command line is kept under profiles/ with its provenance
    traffic, traffic_src, traffic_stale = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "r06_pmc_schur.json")
    if world == 1 and args.n == 100 and not zwork and os.path.exists(pmc_path):
        try:
            pj = json.load(open(pmc_path))
            if pj.get("kernel_source_sha16") == kernel_source_hash():
                traffic = pj["traffic_bytes_per_factorisation"] / max(1, stp["schur_launches"])   # per launch, like `achieved`
                traffic_src = "profiles/r06_pmc_schur.json: " + pj["source"]
            else:
                traffic_stale = True     # the kernels changed since the counters were collected: scripts/collect_pmc.sh regenerates them
        except Exception:
            pass
    alg_bytes_per_launch = st["schur_bytes_alg"] / max(1, stp["schur_launches"])   # 16 B per updated element
    out = {
        "metric": "LU factorization GFLOP/s (pdgstrf3d) + solve time",
        "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
        "ms_per_step": ms_per_step,
        # the headline configuration's own numbers FIRST (the driver keeps the parsed head and the tail of the line)
        "factor_ms": float(np.mean(fact_ms)), "factor_ms_min": float(np.min(fact_ms)), "factor_ms_max": float(np.max(fact_ms)),
        "solve_ms": float(np.mean(solve_ms)), "setup_s": t_setup, "residual": res,
        "higher_is_better": True,
        "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "c128" if zwork else "f64", "data": "synthetic",
        "config": {"workload": (f"pzdrive3d-equivalent on a {args.n}x{args.n} 5-point complex16 grid operator (cg20 family), 1x1x1 grid, "
                                if zwork else
                                f"pddrive3d-equivalent on the audikw_1 stand-in ({args.n}^3 nodes x 3 unknowns, 27-point node coupling, SPD, random numbering), {grid[0]}x{grid[1]}x{grid[2]} grid, graph "
                                if args.workload == "audikw_like" else
                                f"pddrive3d-equivalent on {os.path.basename(args.workload)} (MatrixMarket), {grid[0]}x{grid[1]}x{grid[2]} grid, graph "
                                if args.workload.endswith(".mtx") else
                                f"pddrive3d-equivalent on {args.n}^3 7-point Poisson (double), {grid[0]}x{grid[1]}x{grid[2]} grid, ")
                               + f"ND perm_c (leaf {args.leaf}), relax {args.relax}, maxsup {args.maxsup}, nrhs 1",
                   "n": n, "nnz_A": int(len(v)), "nnz_LU": int(st["nnz_L"] + st["nnz_U"]), "nsupers": symb.nsupers,
                   "parallelism": "single GPU" if world == 1 else
                   f"{grid[0]}x{grid[1]}x{grid[2]} process grid, one rank per GPU: XY block-cyclic panels + Z-sharded elimination forests; "
                   f"panel exchange / ancestor reduction / solve exchanges by the library's C driver over {'RCCL (ncclSend/ncclRecv)' if dist_backend_used[0] == 'rccl' else 'host-staged gloo callbacks' + (' -- RCCL communicator creation FAILED: ' + comm_cache['rccl_error'] if comm_cache.get('rccl_error') else '')}"},
        "flops_per_step": F, "flops_schur_padded": st["flops_schur_padded"],
        "factor_gflops_kernel_only": F / (np.mean(fact_ms) * 1e-3) / 1e9,
        "max_abs_err_vs_xtrue": err, "info": int(info), "accuracy": accuracy,
        "levels": st["num_levels"], "fused_level_pairs": st["reserved_i"], "launches_per_factor": st["num_launches"], "setup_breakdown": M["setup_breakdown"],
        "device_clock": clock.summary(),
        "bytes_device_this_rank": int(st["bytes_device"]),
        "roofline": {"bound": "mfma", "kernel": "k_schur<Z> (the double kernel on the real embedding of the complex update: fused gather + fp64 MFMA + scatter; 8 real flop per complex multiply-add)" if zwork else
                     "k_schur (fused gather + fp64 MFMA GEMM + scatter)",
                     "achieved": schur_tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": schur_tf / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                     "traffic_source": traffic_src, "traffic_stale": traffic_stale, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "launches": int(stp["schur_launches"]),
                     "avg_launch_ms": stp["t_schur_ms"] / max(1, stp["schur_launches"]),
                     "flops_per_launch": st["flops_schur_exact"] / max(1, stp["schur_launches"]),
                     "schur_ms": stp["t_schur_ms"], "panel_ms": stp["t_panel_ms"], "profiled_factor_ms": stp["t_factor_ms"],
                     "frac_min": min(st["flops_schur_exact"] / (q["t_schur_ms"] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS for q in prof_passes if q["t_schur_ms"] > 0) if prof_passes and prof_passes[0]["t_schur_ms"] > 0 else None,
                     "frac_max": max(st["flops_schur_exact"] / (q["t_schur_ms"] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS for q in prof_passes if q["t_schur_ms"] > 0) if prof_passes and prof_passes[0]["t_schur_ms"] > 0 else None,
                     "profiled_passes": len(prof_passes)},
    }
    # the same Schur time split by tile configuration: `roofline` above is ALL Schur launches against the MFMA peak (the number to compare across
    # rounds); the 128 x 128 instantiation -- supernodes of >= 96 columns, the MFMA-bound part by SURVEY 8(d)'s own criterion (s_k >~ 120) -- and the
    # 64 x 64 configuration of the narrow supernodes at the bottom of the tree (HBM / latency-bound: fraction of the HBM peak on its algorithmic bytes)
    tb = stp.get("t_schur_big_ms", 0.0)
    if world == 1 and tb > 0 and stp["t_schur_ms"] > tb:
        fb, bb = st["flops_schur_exact_big"], st["schur_bytes_alg_big"]
        ts = stp["t_schur_ms"] - tb
        out["roofline"]["by_configuration"] = {
            "k_schur<128,128,8>": {"bound": "mfma", "ms": tb, "flops": fb, "achieved": fb / (tb * 1e-3) / 1e12, "unit": "TFLOP/s",
                                   "frac": fb / (tb * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS},
            "k_schur<64,64,4>": {"bound": "hbm", "ms": ts, "flops": st["flops_schur_exact"] - fb, "achieved_tflops": (st["flops_schur_exact"] - fb) / (ts * 1e-3) / 1e12,
                                 "algorithmic_bytes": st["schur_bytes_alg"] - bb, "achieved": (st["schur_bytes_alg"] - bb) / (ts * 1e-3) / 1e9, "unit": "GB/s",
                                 "frac": (st["schur_bytes_alg"] - bb) / (ts * 1e-3) / 1e9 / PEAK_HBM_GBS}}
        # The epilogue of either configuration is one fp64 atomic per updated element (16 algorithmic bytes): the unit the small-tile configuration actually
        # saturates is the L2 atomic pipe, measured at 168-188 G elements/s on this part (profiles/r02_ubench_atomic_f64.txt), not HBM (NOTEBOOK.md section 5)
        ATOMIC_PEAK_GELEM_S = 187.7
        for key, nbytes, ms in (("k_schur<128,128,8>", bb, tb), ("k_schur<64,64,4>", st["schur_bytes_alg"] - bb, ts)):
            rate = nbytes / 16.0 / (ms * 1e-3) / 1e9
            out["roofline"]["by_configuration"][key]["atomics"] = {"achieved": rate, "peak": ATOMIC_PEAK_GELEM_S, "unit": "G fp64 atomics/s", "frac": rate / ATOMIC_PEAK_GELEM_S,
                                                                   "peak_source": "profiles/r02_ubench_atomic_f64.txt (cold destinations)"}
    # second roofline: the triangular solve is HBM-bound, 8 B per stored factor entry per solve (nrhs = 1)
    esz = 16 if zwork else 8
    solve_bytes = esz * float(st["nnz_L"] + st["nnz_U"])
    solve_gbs = solve_bytes / (np.mean(solve_ms) * 1e-3) / 1e9 if np.mean(solve_ms) > 0 else 0.0
    if world == 1 and not zwork and h is not None:
        # second roofline point of the sweep kernels: nrhs = 16 in ONE call -- the reference's pdgstrs3d takes any nrhs and its lsum updates are
        # GEMMs there (pdgstrs_lsum.c:414-960); the factors are still read once per solve when the right-hand sides ride along (8 B per entry per solve), so the
        # achieved bytes / s against the HBM peak is the comparable figure; `ms_per_rhs` shows what a block of right-hand sides buys
        try:
            R = 16
            xt16 = np.asfortranarray(np.where(((np.arange(n)[:, None] + np.arange(R)[None, :]) % 2) == 1, 1.0, -1.0) * (1.0 + 0.25 * np.arange(R)[None, :]))
            b16 = np.asfortranarray(np.column_stack([matgen.csr_matvec(n, rp, ci, v, xt16[:, q:q + 1])[:, 0] for q in range(R)]))
            xp16 = np.zeros_like(b16, order="F"); xp16[symb.perm_c, :] = b16
            h.pdgstrs3d(xp16.copy(order="F"))                          # untimed warm-up of this shape
            y16 = h.pdgstrs3d(xp16)
            ms16 = h.stats()["t_solve_ms"]
            x16 = y16[symb.perm_c, :]
            r16 = max(float(np.linalg.norm(b16[:, q] - matgen.csr_matvec(n, rp, ci, v, x16[:, q:q + 1])[:, 0]) / np.linalg.norm(b16[:, q])) for q in range(R))
            out["solve_nrhs16"] = {"nrhs": R, "solve_ms": ms16, "ms_per_rhs": ms16 / R, "residual_max": r16,
                                   "algorithmic_bytes": solve_bytes + 2.0 * 8.0 * n * R, "achieved": (solve_bytes + 2.0 * 8.0 * n * R) / (ms16 * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": (solve_bytes + 2.0 * 8.0 * n * R) / (ms16 * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                   "flops": 2.0 * float(st["nnz_L"] + st["nnz_U"]) * R, "tflops": 2.0 * float(st["nnz_L"] + st["nnz_U"]) * R / (ms16 * 1e-3) / 1e12}
        except Exception as e:
            out["solve_nrhs16"] = {"error": str(e)[:200]}
    if world == 1:
        out["roofline_solve"] = {"bound": "hbm", "kernel": "k_sweep_join (one launch per level: joined diagonal blocks + regular update units) + k_sweep / k_fwd_update / k_bwd_update on the levels of many supernodes",
                                 "achieved": solve_gbs, "peak": 8000.0, "unit": "GB/s", "frac": solve_gbs / 8000.0,
                                 "algorithmic_bytes_per_solve": solve_bytes,
                                 "note": "%d levels; one launch per level and sweep where a level holds <= 32 supernodes, two elsewhere (100^3: 223 launches per solve); 4-5.5 TB/s on the levels that hold the data, ~9.5 us per single-supernode level of the top separator" % st["num_levels"]}
    if world > 1:   # the dominant kernel is profiled in the N=1 run of this same command (here: this rank's share under the serial schedule)
        out["roofline"].update(achieved=None, frac=None, avg_launch_ms=None, flops_per_launch=None, algorithmic_bytes_per_launch=None,
                               note="per-kernel roofline is measured by the N=1 run (bench.py --gpus 1); N>1 lines report whole-job throughput; "
                                    "schur_ms / panel_ms / launches are rank 0's, from its extra profiled (serial-schedule) factorisation")
        # per-phase times of the profiled factorisation, MAX over ranks: where an N > 1 run spends its time beside the kernels
        import torch
        ph = torch.tensor([stp["t_exchange_ms"], stp["t_reduce_ms"], stp["t_schur_ms"], stp["t_panel_ms"], stp["t_factor_ms"]], dtype=torch.float64)
        dist.all_reduce(ph, op=dist.ReduceOp.MAX)
        out["phases"] = {"exchange_ms": float(ph[0]), "reduce_ms": float(ph[1]), "schur_ms": float(ph[2]), "panel_ms": float(ph[3]),
                         "profiled_factor_ms": float(ph[4]),
                         "note": "one extra factorisation under the serial (profiling) schedule, HIP events, max over ranks: exchange = the XY "
                                 "panel-exchange phases (dDiagFactIBCast + L / U panel broadcasts; part of panel_ms), reduce = the Z ancestor "
                                 "reduction (dreduceAllAncestors3d); the timed steps overlap the exchanges with the Schur tiles of the previous level"}
    if world > 1:
        out["predicted"] = predicted_block(h, n, symb.flops, float(np.mean(fact_ms)))
    # ---- scaling point: the same job one size up, reported beside the headline configuration at EVERY N (100^3 is a
    # 0.3 s job -- its N > 1 runs are exchange-latency-bound; 150^3 is 3 s of work and fits one GPU at 90 GB).  `value` stays the
    # headline configuration so that the driver's efficiency figure compares equal jobs; this block lets it be recomputed on 150^3.
    def scaling_block(side, steps, prof_phases):
        """the same job at another size, same ordering / supernode parameters: throughput, phase times, bytes, set-up"""
        S2 = measure(side, steps, 1, args.workload)
        st2 = S2["h"].stats()
        F2 = S2["symb"].flops
        blk = {"workload": f"{side}^3 7-point Poisson (double), {S2['grid'][0]}x{S2['grid'][1]}x{S2['grid'][2]} grid, same ordering / supernode parameters",
               "n": S2["n"], "flops_per_step": F2, "steps": S2["steps"], "warmup": 1,
               "value": F2 * S2["steps"] / S2["elapsed"] / 1e9, "unit": "GFLOP/s", "ms_per_step": 1e3 * S2["elapsed"] / S2["steps"],
               "factor_ms": float(np.mean(S2["fact_ms"])), "solve_ms": float(np.mean(S2["solve_ms"])),
               "factor_gflops_kernel_only": F2 / (np.mean(S2["fact_ms"]) * 1e-3) / 1e9,
               "residual": S2["res"], "nnz_LU_this_rank": int(st2["nnz_L"] + st2["nnz_U"]),
               "bytes_device_this_rank": int(st2["bytes_device"]), "setup_s": S2["t_setup"], "setup_breakdown": S2["setup_breakdown"]}
        if world == 1:
            blk["solve_hbm_frac"] = 8.0 * float(st2["nnz_L"] + st2["nnz_U"]) / (np.mean(S2["solve_ms"]) * 1e-3) / 1e9 / PEAK_HBM_GBS
        elif prof_phases:   # N > 1: where the time goes beside the kernels -- one extra factorisation under the serial schedule, HIP events, max over ranks
            import torch
            hh = S2["h"]
            hh.set_profile(True); hh.reset_values(); hh.pdgstrf3d(S2["thresh"]); sp = hh.stats(); hh.set_profile(False)
            ph = torch.tensor([sp["t_exchange_ms"], sp["t_reduce_ms"], sp["t_schur_ms"], sp["t_panel_ms"], sp["t_factor_ms"]], dtype=torch.float64)
            dist.all_reduce(ph, op=dist.ReduceOp.MAX)
            blk["phases"] = {"exchange_ms": float(ph[0]), "reduce_ms": float(ph[1]), "schur_ms": float(ph[2]), "panel_ms": float(ph[3]), "profiled_factor_ms": float(ph[4])}
        if S2["res"] > 1e-10:
            raise SystemExit(f"bench.py: scaling-point residual {S2['res']:.3e} exceeds 1e-10 at {side}^3")
        return blk, S2

    # ---- scaling points: the same job at larger sizes, reported beside the headline configuration at EVERY N (100^3 is a
    # 0.3 s job -- its N > 1 runs are exchange-latency-bound).  `value` stays the headline configuration so that the driver's efficiency figure
    # compares equal jobs; these blocks let it be recomputed on 150^3 (90 GB of factors, ~3 s per step) and on 180^3 (188 GB: the largest cube
    # ONE GPU holds, so T_1 exists -- the strong-scaling problem of VERDICT r4 item 6-iii; 4.5e14 flop, ~9 s per step on one GPU).
    if args.workload == "poisson3d" and not args.no_scaling_point:
        for key, side, nsteps in (("scaling_point", args.scale_n, max(1, min(args.steps, 2))), ("strong_scaling_point", args.scale_n2, 1)):
            if not side or side == args.n:
                continue
            if h is not None:
                h.destroy(); symb.free(); h, symb = None, None
            try:
                out[key], S2 = scaling_block(side, nsteps, world > 1)
                h, symb = S2["h"], S2["symb"]
            except SystemExit:
                raise
            except Exception as e:      # e.g. not enough HBM on this rank: reported, the headline line stands
                out[key] = {"error": str(e)[:300]}
                h, symb = None, None
    # ---- configs4: BASELINE.json configs[4] (pzdrive3d complex16, cg20 family scaled 50x per grid side = 1000 x 1000 5-point complex grid
    # operator, 1 GPU) measured by the SAME default command, so that the driver's record carries it:
    # 5 timed steps after 2 warm-up steps, MFMA fraction of k_schur<Z> from one extra profiled factorisation, HBM fraction of the solve
    if world == 1 and args.workload == "poisson3d" and not args.no_configs4:
        if h is not None:
            h.destroy(); symb.free(); h, symb = None, None
        try:
            Z4 = measure(args.configs4_n, 5, 2, "zgrid2d", maxsup=64)
            hz = Z4["h"]
            hz.set_profile(True); hz.reset_values(); hz.pdgstrf3d(Z4["thresh"]); stpz = hz.stats(); hz.set_profile(False)
            stz = hz.stats()
            Fz = stz["flops_schur_exact"] + stz["flops_panel"]
            z_tf = stz["flops_schur_exact"] / (stpz["t_schur_ms"] * 1e-3) / 1e12 if stpz["t_schur_ms"] > 0 else 0.0
            z_gbs = 16.0 * float(stz["nnz_L"] + stz["nnz_U"]) / (np.mean(Z4["solve_ms"]) * 1e-3) / 1e9
            out["configs4"] = {"workload": f"pzdrive3d-equivalent on a {args.configs4_n}x{args.configs4_n} 5-point complex16 grid operator (cg20 family, grid side x {args.configs4_n // 20}), 1x1x1 grid, "
                                           f"ND perm_c (leaf {args.leaf}), relax {args.relax}, maxsup 64, nrhs 1",
                               "dtype": "c128", "n": Z4["n"], "nnz_LU": int(stz["nnz_L"] + stz["nnz_U"]), "flops_per_step": Fz, "steps": Z4["steps"], "warmup": 2,
                               "value": Fz * Z4["steps"] / Z4["elapsed"] / 1e9, "unit": "GFLOP/s", "ms_per_step": 1e3 * Z4["elapsed"] / Z4["steps"],
                               "factor_ms": float(np.mean(Z4["fact_ms"])), "solve_ms": float(np.mean(Z4["solve_ms"])),
                               "factor_gflops_kernel_only": Fz / (np.mean(Z4["fact_ms"]) * 1e-3) / 1e9,
                               "residual": Z4["res"], "max_abs_err_vs_xtrue": Z4["err"], "info": int(Z4["info"]),
                               "launches_per_factor": stz["num_launches"], "levels": stz["num_levels"],
                               "roofline": {"bound": "mfma", "kernel": "k_schur<Z> (real embedding of the complex update on the fp64 MFMA kernel)",
                                            "achieved": z_tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": z_tf / PEAK_FP64_MFMA_TFLOPS,
                                            "schur_ms": stpz["t_schur_ms"], "panel_ms": stpz["t_panel_ms"], "launches": int(stpz["schur_launches"])},
                               "roofline_solve": {"bound": "hbm", "achieved": z_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": z_gbs / PEAK_HBM_GBS,
                                                  "algorithmic_bytes_per_solve": 16.0 * float(stz["nnz_L"] + stz["nnz_U"])}}
            hz.destroy(); Z4["symb"].free()
            if Z4["res"] > 1e-10:
                raise SystemExit(f"bench.py: configs4 residual {Z4['res']:.3e} exceeds 1e-10")
        except SystemExit:
            raise
        except Exception as e:
            out["configs4"] = {"error": str(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "poisson3d":
        try:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.leaf, args.relax, args.maxsup, grid_n=args.cpu_grid_n, mkl_n=args.cpu_mkl_n, mkl_grid_n=args.cpu_mkl_grid_n)
        except Exception as e:
            out["cpu_baseline"] = {"error": str(e)[:300]}
    if rank == 0:
        print(json.dumps(out))
    if h is not None:
        h.destroy(); symb.free()
    if dist is not None:
        dist.destroy_process_group()
    if res > 1e-10:
        raise SystemExit(f"bench.py: residual {res:.3e} exceeds 1e-10")


if __name__ == "__main__":
    main()
