"""sluamd_plan_table (what scripts/scale_model.py builds its 8-GPU prediction from) against the handle's own statistics and against conservation laws of the
exchange plans -- on the CPU build of the library: the planner is host code."""
import numpy as np
import pytest
from superlu_dist_amd import driver, grid3d, matgen


def _problem(N=16):
    n, rp, ci, v = matgen.poisson3d(N)
    perm = matgen.nd_perm_grid3d(N, N, N, leaf=27)
    return n, rp, ci, v, driver.Symbolic(n, rp, ci, perm, relax=16, maxsup=64)


def test_plan_table_of_one_rank_adds_up_to_the_statistics(emul):
    n, rp, ci, v, symb = _problem()
    h = driver.LUHandle.from_symbolic(symb, v)
    t, st = h.plan_table(), h.stats()
    assert t.shape[1] == driver.PLAN_COLS and t.shape[0] == st["num_levels"]
    assert np.isclose(t[:, 4].sum(), st["flops_schur_exact"], rtol=1e-12)
    assert np.isclose(t[:, 6].sum(), st["flops_panel"], rtol=1e-12)
    assert int(t[:, 5].sum()) == st["schur_tiles"]
    assert int(t[:, 2].sum()) == symb.nsupers
    assert not t[:, 7:].any()                       # no exchange, no reduction on a 1 x 1 x 1 grid
    h.destroy(); symb.free()


@pytest.mark.parametrize("grid", [(2, 2, 2), (2, 1, 2), (1, 2, 1)])
def test_plan_tables_of_a_grid_conserve_flops_and_bytes(emul, grid):
    n, rp, ci, v, symb = _problem()
    h = driver.LUHandle.from_symbolic(symb, v)
    F1 = h.stats()["flops_schur_exact"]
    h.destroy()
    Pr, Pc, Pz = grid
    tree = symb.partition(Pz) if Pz > 1 else None
    comms = grid3d.local_comms(Pr, Pc, Pz)
    tabs = []
    for r in range(Pr * Pc * Pz):
        g = grid3d.GridHandle.from_symbolic(symb, v, comms[r], tree)
        tabs.append(g.plan_table())
        g.destroy()
    # every Schur flop of the one-rank job is planned on exactly one rank
    assert np.isclose(sum(t[:, 4].sum() for t in tabs), F1, rtol=1e-12)
    # what the ranks send in a phase is what the ranks receive in it (bytes and messages), and in the Z reduction
    for snd, rcv in ((7, 9), (8, 10), (11, 13), (12, 14)):
        assert np.isclose(sum(t[:, snd].sum() for t in tabs), sum(t[:, rcv].sum() for t in tabs), rtol=1e-12)
    assert np.isclose(sum(t[:, 15].sum() for t in tabs), 0.0, atol=1e-6)
    if Pr * Pc > 1:
        assert sum(t[:, 11].sum() for t in tabs) > 0
        for t in tabs:      # the busiest peer of a phase carries at most the phase's bytes, and all of them when there is one peer
            assert np.all(t[:, 16] <= t[:, 7] + 1e-9) and np.all(t[:, 18] <= t[:, 11] + 1e-9)
    if Pz > 1:
        assert max(abs(t[:, 15]).max() for t in tabs) > 0
    symb.free()


def test_scale_model_on_plan_tables(emul):
    """superlu_dist_amd/scale_model.py (the harness behind profiles/r06_scale_model.txt and bench.py's `predicted` block): on the plan tables of a real grid the
    prediction is positive, the exposed-reduction figure is not below the hidden one, slower links never make it faster, and a one-rank table has no exchange term."""
    from superlu_dist_amd import scale_model
    n, rp, ci, v, symb = _problem(18)
    h = driver.LUHandle.from_symbolic(symb, v)
    t1 = [h.plan_table()]
    h.destroy()
    T1h, T1e, rows1 = scale_model.predict(t1)
    assert T1h > 0 and T1h == T1e and all(r[4] == 0.0 for r in rows1)          # no exchange, no reduction
    tree = symb.partition(2)
    comms = grid3d.local_comms(2, 2, 2)
    tabs = []
    for r in range(8):
        g = grid3d.GridHandle.from_symbolic(symb, v, comms[r], tree)
        tabs.append(g.plan_table()); g.destroy()
    Th, Te, rows = scale_model.predict(tabs)
    assert 0 < Th <= Te and len(rows) == 2 and rows[0][6] > 0                   # the Z reduction follows Z level 0
    slow_h, slow_e, _ = scale_model.predict(tabs, {"link_gbs": 5.0, "lat_us": 200.0})
    assert slow_h >= Th and slow_e >= Te
    fast_h, _, _ = scale_model.predict(tabs, {"link_gbs": 1e6, "lat_us": 0.0})
    assert fast_h <= Th
    symb.free()
