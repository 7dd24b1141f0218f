"""Pins oracle/slu_oracle.c to the REAL reference: fixtures in tests/golden were recorded from
xiaoyeli/superlu_dist v9.2.1 run in the build container (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import oracle as orc

CASES_1RANK = ["g20_1x1x1", "g20_1x1x1_nrhs3", "g20_1x1x1_legacy", "poisson8_nd", "poisson10_nd", "unsym300",
               "unsym120_tiny",
               # complex16: pzgstrf3d / pzgstrs3d recorded from the reference's SRC/complex16 path
               "z_cg20_1x1x1", "z_cg20_1x1x1_nrhs2", "z_poisson8_nd", "z_unsym200", "z_grid24_nd"]


def _order(g):
    # elimination order of the single forest on a 1x1x1 grid (sForest_t.nodeList)
    return g["r0__forest0_nodeList"]


@pytest.mark.parametrize("case", CASES_1RANK)
def test_factor_matches_reference(golden, case):
    g = golden(case)
    st = orc.LUStore.from_golden(g, 0, "pre")
    info, tiny, flops = orc.dfactor(st, _order(g), bool(g["r0__ReplaceTinyPivot"][0]), float(g["r0__thresh"][0]))
    assert info == int(g["r0__info"][0])
    assert tiny == int(g["r0__TinyPivots"][0])
    scale = max(np.abs(g["r0__Lnzval_pre"]).max(), np.abs(g["r0__Unzval_pre"]).max())
    # summation order differs from the reference only through look-ahead reordering -> 1e-12*||A||
    assert np.abs(st.Lnzval - g["r0__Lnzval_post"]).max() <= 1e-12 * scale
    assert np.abs(st.Unzval - g["r0__Unzval_post"]).max() <= 1e-12 * scale


@pytest.mark.parametrize("case", CASES_1RANK)
def test_solve_matches_reference(golden, case):
    g = golden(case)
    st = orc.LUStore.from_golden(g, 0, "post")
    n = st.n
    pr, pc = g["r0__perm_r"], g["r0__perm_c"]
    nsolve = sum(1 for k in g if k.startswith("r0__solve") and k.endswith("_B_in"))
    assert nsolve >= 1
    for s in range(nsolve):
        nrhs = int(g[f"r0__solve{s}_nrhs"][0])
        B = g[f"r0__solve{s}_B_in"].reshape((n, nrhs), order="F")
        X = g[f"r0__solve{s}_B_out"].reshape((n, nrhs), order="F")
        xp = np.zeros((n, nrhs), order="F", dtype=B.dtype)
        xp[pc[pr], :] = B                       # pdReDistribute3d_B_to_X: row perm_c[perm_r[i]] (pdgstrs3d.c:6329)
        xs = orc.dsolve(st, xp)
        got = xs                                # pdReDistribute3d_X_to_B leaves Y = Pc*X (pdgstrs3d.c:6573-6575);
                                                # pdgssvx3d applies Pc^T afterwards
        assert np.abs(got - X).max() <= 1e-11 * max(1.0, np.abs(X).max())


REFINE_CASES = ["poisson8_nd_refine", "weakdiag150_refine", "weakdiag150_refine_nrhs2"]


@pytest.mark.parametrize("case", REFINE_CASES)
def test_iterative_refinement_matches_reference(golden, case):
    """pdgsrfs3d (IterRefine=SLU_DOUBLE, Equil=NO, NOROWPERM): same step count, berr and refined x as the reference."""
    g = golden(case)
    assert int(g["r0__opt_IterRefine"][0]) == 2 and int(g["r0__opt_Equil"][0]) == 0 and int(g["r0__opt_RowPerm"][0]) == 0
    st = orc.LUStore.from_golden(g, 0, "post")
    n = st.n
    nrhs = int(g["r0__nrhs"][0])
    pc = g["r0__perm_c"]
    rp, ci, v = g["r0__A_rowptr"], g["r0__A_colind"], g["r0__A_nzval"]
    B = g["r0__b"].reshape((n, nrhs), order="F")
    xp = np.zeros((n, nrhs), order="F"); xp[pc, :] = B
    X0 = np.asfortranarray(orc.dsolve(st, xp)[pc, :])          # initial solve: Pc^T (LU)^-1 Pc b
    X, berr, steps = orc.dgsrfs(st, rp, ci, v, pc, B, X0)
    Xref = g["r0__x"].reshape((n, nrhs), order="F")
    assert steps == int(g["r0__RefineSteps"][0])
    assert np.allclose(berr, g["r0__berr"], rtol=0.5, atol=1e-17)      # both at the eps level; ratio within 2x
    assert np.abs(X - Xref).max() <= 1e-12 * max(1.0, np.abs(Xref).max())
    assert np.abs(X - Xref).max() < np.abs(X0 - Xref).max() or np.abs(X0 - Xref).max() < 1e-14
