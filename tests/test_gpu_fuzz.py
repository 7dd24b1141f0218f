"""Randomised own-pipeline parity on the device: irregular patterns (random sparse, perturbed stencils, the elasticity-like stand-in), graph
ordering, random supernode parameters, double and complex16, 1..5 right-hand sides -- every factor value and the solve against the CPU oracle
(oracle/, TEST INFRASTRUCTURE).  Covers what the structured cases do not: ragged tiles of every shape through the per-tile records, tiles
without destination, fused pairs next to unfused ones, the unit records of the sweeps on irregular level structures, and (every fourth case) L blocks whose rows
come in a random order, which the handle sorts at creation."""
import os
import numpy as np
import pytest
import oracle as orc
import grid_cases
from superlu_dist_amd import driver, matgen

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 3
    if kind == 0:
        n = int(rng.integers(200, 900))
        n, rp, ci, v = matgen.random_unsym(n, density=float(rng.uniform(0.004, 0.02)), seed=seed)
    elif kind == 1:
        N = int(rng.integers(7, 13))
        n, rp, ci, v = matgen.stencil3d_unsym(N, drop=float(rng.uniform(0.1, 0.4)), seed=seed, reach=int(rng.integers(1, 3)))
    else:
        N = int(rng.integers(5, 9))
        n, rp, ci, v = matgen.elasticity3d_like(N, dof=3, drop=0.1, seed=seed, shuffle=True)
    relax = int(rng.choice([1, 4, 8, 16, 32, 64]))
    maxsup = int(rng.choice([8, 24, 48, 64, 100, 160, 256]))
    return n, rp, ci, v, max(1, min(relax, maxsup)), maxsup, int(rng.integers(1, 6)), rng


@pytest.mark.parametrize("complex16", [False, True])
@pytest.mark.parametrize("seed", range(int(os.environ.get("SLUAMD_GPU_FUZZ_CASES", "12"))))
def test_random_matrices_match_oracle(seed, complex16):
    n, rp, ci, v, relax, maxsup, nrhs, rng = _case(seed)
    if complex16:
        v = matgen.complex_shift(v, rp, ci, seed=seed)
    perm = driver.order_nd(n, rp, ci, leaf=int(rng.choice([8, 27, 64])))
    symb = driver.Symbolic(n, rp, ci, perm, relax=relax, maxsup=maxsup)
    shuffle = seed % 4 == 3      # every fourth case: the rows INSIDE the L blocks in a random order, as the reference's symbfact leaves them (the handle sorts
                                 # them at creation and permutes the values in its staging buffer, both ways: copy_to_host must return the caller's order)
    if complex16:
        symb.distribute_host(v.real); fr = symb.flat_store()
        symb.distribute_host(v.imag); fi = symb.flat_store()
        if shuffle:
            grid_cases.shuffle_block_rows(fr, seed); grid_cases.shuffle_block_rows(fi, seed)      # same structure, same seed: the same permutations
            assert np.array_equal(fr.Lrowind, fi.Lrowind)
        Lz, Uz = fr.Lnzval + 1j * fi.Lnzval, fr.Unzval + 1j * fi.Unzval
    else:
        symb.distribute_host(v); fr = symb.flat_store()
        if shuffle:
            grid_cases.shuffle_block_rows(fr, seed)
        Lz, Uz = fr.Lnzval, fr.Unzval
    o = orc.LUStore(fr.n, fr.xsup, fr.Lrowind_off, fr.Lrowind, fr.Lnzval_off, Lz, fr.Ufstnz_off, fr.Ufstnz, fr.Unzval_off, Uz)
    fs = driver.FlatStore(fr.n, fr.xsup, fr.Lrowind_off, fr.Lrowind, fr.Lnzval_off, Lz.copy(), fr.Ufstnz_off, fr.Ufstnz, fr.Unzval_off, Uz.copy())
    h = driver.LUHandle.from_store(fs)
    info = h.pdgstrf3d(0.0)
    info_o, _, _ = orc.dfactor(o)
    assert info == info_o
    if info == 0:
        h.copy_to_host()
        scale = max(np.abs(o.Lnzval).max(), np.abs(o.Unzval).max(), 1.0)
        assert np.abs(fs.Lnzval - o.Lnzval).max() <= 1e-11 * scale
        assert np.abs(fs.Unzval - o.Unzval).max() <= 1e-11 * scale
        xp = rng.standard_normal((n, nrhs)) + (1j * rng.standard_normal((n, nrhs)) if complex16 else 0.0)
        xp = np.asfortranarray(xp)
        x = h.pdgstrs3d(xp)
        xo = orc.dsolve(o, xp)
        assert np.abs(x - xo).max() <= 1e-8 * max(np.abs(xo).max(), 1.0)
    h.destroy(); symb.free()
