#!/usr/bin/env python3
"""Predicted time of pdgstrf3d on a Pr x Pc x Pz grid of MI355X from the library's OWN plan: no multi-GPU box has been available in any
round, so the first hardware SCALE run should have something to be compared against.

The planner is host code: the handle of every rank is created here on the CPU build of the library (SLUAMD_LIB=oracle/libsluamd_emul.so, one rank at a time,
nothing run) and sluamd_plan_table gives, per (Z level, DAG level) and rank: Schur flops and tile executions of that rank, its diagonal-LU + panel-solve flops,
the bytes / messages of the two XY exchange phases with the busiest single peer, and the bytes of the Z reduction.

Model: superlu_dist_amd/scale_model.py (every constant is printed; the kernel constants are CALIBRATED on measured one-GPU runs of this round, the link
constants are assumptions).
usage: scale_model.py N Pr Pc Pz [--link-gbs 60] [--lat-us 20] [--t1-ms measured_one_gpu_ms]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLUAMD_LIB", os.path.join(ROOT, "oracle", "libsluamd_emul.so"))
os.environ.setdefault("SLUAMD_NO_TILE_MAPS", "1")
os.environ.setdefault("SLUAMD_EMUL_LAZY_ZERO", "1")      # the handles are only planned: their value arenas stay untouched zero pages
import numpy as np
from superlu_dist_amd import driver, grid3d, matgen, scale_model

ap = argparse.ArgumentParser()
ap.add_argument("N", type=int); ap.add_argument("Pr", type=int); ap.add_argument("Pc", type=int); ap.add_argument("Pz", type=int)
ap.add_argument("--link-gbs", type=float, default=60.0, help="achieved one-direction rate to ONE peer (xGMI: ~76 GB/s per direction and link at the 153 GB/s figure; RCCL point-to-point ~80 %% of it)")
ap.add_argument("--lat-us", type=float, default=20.0, help="fixed cost of one exchange phase (ncclGroup of sends / receives queued on the library's stream)")
ap.add_argument("--r-big", type=float, default=51.0, help="TFLOP/s of the 128 x 128 tile configuration on one MI355X (bench.py roofline.by_configuration, this round: 221.8 ms for 11.32 TFLOP)")
ap.add_argument("--r-small", type=float, default=37.5, help="TFLOP/s of the 64 x 64 tile configuration (26.1 ms for 0.98 TFLOP)")
ap.add_argument("--r-panel", type=float, default=9.0, help="TFLOP/s of the diagonal-LU / panel-solve kernels where a level holds many supernodes")
ap.add_argument("--link-wide-us", type=float, default=360.0, help="chain latency of a level whose widest supernode is > 64 columns (k_diag_lu2 250 + urgent k_panel_trsm 93 + split-K tiles ~20 us)")
ap.add_argument("--link-narrow-us", type=float, default=60.0, help="the same for levels of <= 64-column supernodes (one-wave LU + inverses + panel GEMM)")
ap.add_argument("--t1-ms", type=float, default=0.0, help="measured pdgstrf3d of this problem on ONE GPU (bench.py), printed beside the model's own one-GPU prediction")
ap.add_argument("--no-one-gpu", action="store_true", help="skip the one-GPU plan (a problem no single device or host holds: 300^3); the speed-up is then taken against the ranks' summed work at the model's one-GPU rates")
a = ap.parse_args()
P = a.Pr * a.Pc * a.Pz
t0 = time.time()
n, rp, ci, v = matgen.poisson3d(a.N)
perm = matgen.nd_perm_grid3d(a.N, a.N, a.N, leaf=64)
symb = driver.Symbolic(n, rp, ci, perm, relax=64, maxsup=256)


def tables(Pr, Pc, Pz):
    if Pr * Pc * Pz == 1:
        h = driver.LUHandle.from_symbolic(symb, v); t = [h.plan_table()]; h.destroy(); return t
    tree = symb.partition(Pz) if Pz > 1 else None
    comms = grid3d.local_comms(Pr, Pc, Pz)
    out = []
    for r in range(Pr * Pc * Pz):
        h = grid3d.GridHandle.from_symbolic(symb, v, comms[r], tree); out.append(h.plan_table()); h.destroy()
    return out


PARAMS = dict(r_big=a.r_big, r_small=a.r_small, r_panel=a.r_panel, link_wide_us=a.link_wide_us, link_narrow_us=a.link_narrow_us, link_gbs=a.link_gbs, lat_us=a.lat_us)


def predict(tabs, Pz):
    return scale_model.predict(tabs, PARAMS)


flops = symb.flops
print(f"# scale model: {a.N}^3 7-point Poisson, n = {n}, {flops:.4e} flop, grid {a.Pr}x{a.Pc}x{a.Pz}; constants: R_big {a.r_big} R_small {a.r_small} R_panel {a.r_panel} TFLOP/s, "
      f"chain link {a.link_wide_us:.0f} / {a.link_narrow_us:.0f} us, link {a.link_gbs} GB/s per peer and direction, {a.lat_us} us per exchange phase")
if a.no_one_gpu:
    T1, r1 = 0.0, []
else:
    t1_tabs = tables(1, 1, 1)
    T1, _, r1 = predict(t1_tabs, 1)
if not a.no_one_gpu: print(f"one GPU (model): {1e3 * T1:9.1f} ms = {flops / T1 / 1e12:5.1f} TFLOP/s" + (f"   measured {a.t1_ms:.1f} ms (model / measured = {1e3 * T1 / a.t1_ms:.3f})" if a.t1_ms else ""))
for zl, nl, S, C, X, T, red in r1:
    print(f"   Z level {zl}: {nl:4d} DAG levels  Schur {1e3 * S:9.1f} ms  chain {1e3 * C:8.1f} ms  -> {1e3 * T:9.1f} ms")
if P > 1:
    tabs = tables(a.Pr, a.Pc, a.Pz)
    Th, Te, rows = predict(tabs, a.Pz)
    for zl, nl, S, C, X, T, red in rows:
        print(f"   Z level {zl}: {nl:4d} DAG levels  Schur (max over ranks per level) {1e3 * S:9.1f} ms  chain + exchange {1e3 * C:8.1f} ms (exchange alone {1e3 * X:7.1f})  -> {1e3 * T:9.1f} ms   Z reduction after it {1e3 * red:7.1f} ms")
    if a.no_one_gpu and not a.t1_ms:
        T1 = sum(float(t[:, 4].sum()) for t in tabs) / (a.r_big * 1e12) + sum(float(t[:, 6].sum()) for t in tabs) / (a.r_panel * 1e12)
        print(f"one GPU: no device holds this problem; reference time = the ranks' summed Schur + panel work at the model's one-GPU rates = {1e3 * T1:.1f} ms")
    base = a.t1_ms * 1e-3 if a.t1_ms else T1
    for tag, T in (("Z reductions hidden behind the next forest", Th), ("Z reductions fully exposed", Te)):
        print(f"{P} GPUs, {tag}: {1e3 * T:9.1f} ms = {flops / T / 1e12:6.1f} TFLOP/s   speed-up {base / T:5.2f}   predicted_efficiency {base / (P * T):.3f}" + ("  (T1 = measured)" if a.t1_ms else "  (T1 = model)"))
    fl = np.array([t[:, 4].sum() for t in tabs]); by = np.array([t[:, 11].sum() + t[:, 7].sum() for t in tabs])
    print(f"   per rank: Schur flops max / mean = {fl.max() / fl.mean():.3f}; bytes sent in the XY exchanges {by.min() / 1e9:.2f} .. {by.max() / 1e9:.2f} GB; Z reduction {max(abs(t[:, 15]).max() for t in tabs) / 1e9:.2f} GB per sending rank")
print(f"# ({time.time() - t0:.0f} s on the host)")
