"""BASELINE.json configs[3] (pddrive3d on SuiteSparse audikw_1, irregular supernodes) -- the matrix itself cannot be fetched here, so:
a stand-in of ITS SIZE AND SHAPE (matgen.elasticity3d_like: SPD, 3 unknowns per node, 27-point node coupling, n = 943 296, ~75 entries
per row, random numbering) through the library's own pipeline -- graph nested dissection (sluamd_order_nd), symbolic factorisation,
device-side distribution, pdgstrf3d, pdgstrs3d -- at full scale on one GPU and, smaller, on a 2 x 2 x 2 grid; plus the MatrixMarket
path (symmetric storage, expanded like dreadMM.c) that takes the real file when someone has it:
    python bench.py --matrix audikw_1.mtx         examples/pddrive3d_amd audikw_1.mtx
The reference-pipeline parity of irregular matrices (MC64 + MMD + the reference's symbfact, residual against slu_ref_dump) is in
test_gpu_dropin.py at the sizes the CPU reference finishes in seconds."""
import os, subprocess
import numpy as np
import pytest
import grid_cases
from superlu_dist_amd import driver, matgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_audikw_sized_standin_on_one_gpu():
    n, rp, ci, v = matgen.elasticity3d_like(68, drop=0.05, seed=1)
    assert 9.0e5 < n < 9.6e5 and 70 < len(v) / n < 85
    perm = driver.order_nd(n, rp, ci)
    xt, b = matgen.xtrue_rhs(n, rp, ci, v, 1)
    x, info, st = driver.pdgssvx3d(n, rp, ci, v, b, perm, relax=64, maxsup=256)
    res = np.linalg.norm(b - matgen.csr_matvec(n, rp, ci, v, x)) / np.linalg.norm(b)
    assert info == 0 and res < 1e-10
    assert np.abs(x - xt).max() < 1e-8
    print(f"audikw-like n={n} nnz(A)={len(v)} nnz(L+U)={st['nnz_L'] + st['nnz_U']} factor {st['t_factor_ms']:.0f} ms solve {st['t_solve_ms']:.1f} ms residual {res:.1e}")


def test_audikw_like_standin_on_a_2x2x2_grid():
    """configs[3]'s grid shape (ranks = threads sharing the GPU): n = 139 968, irregular supernodes from the graph ordering."""
    n, rp, ci, v = matgen.elasticity3d_like(36, drop=0.05, seed=2)
    perm = driver.order_nd(n, rp, ci)
    grid_cases.check_matrix_on_grid(n, rp, ci, v, perm, (2, 2, 2), nrhs=1, relax=64, maxsup=256)


def test_matrix_market_file_through_the_c_example(tmp_path):
    """A symmetric MatrixMarket file (lower triangle stored, like the SuiteSparse distribution of audikw_1) through the plain-C driver:
    reader + symmetric expansion + graph ordering + factor + solve over the C ABI."""
    exe = os.path.join(ROOT, "examples", "pddrive3d_amd")
    if not os.path.exists(exe):
        pytest.skip("examples/pddrive3d_amd not built")
    n, rp, ci, v = matgen.elasticity3d_like(12, drop=0.1, seed=3)
    matgen.write_matrix_market(str(tmp_path / "a.mtx"), n, rp, ci, v, symmetric=True)
    r = subprocess.run([exe, str(tmp_path / "a.mtx")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert f"n = {n} " in r.stdout
