"""Time model of pdgstrf3d on a process grid from the ranks' plan tables (sluamd_plan_table) -- harness code (scripts/scale_model.py, bench.py's
`predicted_efficiency` at N > 1); nothing of the hot path.

  Schur update of a level on a rank  t_S = ceil(tiles / 512) * (F / tiles) / (R / 512)      512 = 256 CUs x 2 resident tiles; R = rate of the tile configuration
                                                                                             (big: widest supernode >= 96 columns), bench.py's by_configuration
  panel chain of a level             t_P = t_link(width) + F_panel / R_panel                diagonal LU -> panel solves -> urgent tiles of ONE supernode, then throughput
  exchange of a level                t_X = sum over the two phases of [ busiest-peer bytes / B_link + t_lat ]   (a phase without messages costs nothing)
  one forest under the two-level look-ahead schedule (panel chain + exchange of level l + 1 beside the bulk Schur tiles of level l):
                                     T = t_P(0) + t_X(0) + sum_l max( max_r t_S(l), max_r [t_P(l+1) + t_X(l+1)] )
  Z reduction after a Z level        t_R = bytes / B_link; pipelined with the next forest (reduce_ancestors) -- reported hidden and exposed

The kernel constants are calibrated on measured one-GPU runs of round 6 (100^3: model 281.8 ms, measured 277); the link constants are ASSUMPTIONS until a
multi-GPU box runs the job: xGMI 153 GB/s per link = ~76 GB/s per direction, ~80 % of it through ncclSend / ncclRecv."""
import numpy as np

DEFAULTS = dict(r_big=51.0, r_small=37.5, r_panel=9.0, link_wide_us=360.0, link_narrow_us=60.0, link_gbs=60.0, lat_us=20.0)


def level_times(row, p):
    tiles, F, w = row[5], row[4], row[3]
    R = (p["r_big"] if w >= 96 else p["r_small"]) * 1e12
    tS = 0.0 if tiles <= 0 else np.ceil(tiles / 512.0) * (F / tiles) / (R / 512.0)
    tP = 0.0 if row[6] <= 0 else (p["link_wide_us"] if w > 64 else p["link_narrow_us"]) * 1e-6 + row[6] / (p["r_panel"] * 1e12)
    B = p["link_gbs"] * 1e9
    tX = 0.0
    for snd, rcv, ms, mr in ((row[16], row[17], row[8], row[10]), (row[18], row[19], row[12], row[14])):
        if ms + mr > 0:
            tX += max(snd, rcv) / B + p["lat_us"] * 1e-6
    return tS, tP, tX


def predict(tabs, params=None):
    """tabs: one plan table per rank.  Returns (seconds with the Z reductions hidden, seconds with them exposed, rows per Z level:
    (Z level, DAG levels, sum of Schur, sum of chain + exchange, sum of exchange, forest time, reduction after it))."""
    p = dict(DEFAULTS); p.update(params or {})
    nz = int(max(t[:, 0].max() for t in tabs if len(t))) + 1
    total_h = total_e = 0.0
    rows = []
    for zl in range(nz):
        per_rank = [t[t[:, 0] == zl] for t in tabs]
        nl = max(len(q) for q in per_rank)
        if nl == 0:
            continue
        S = np.zeros(nl); C = np.zeros(nl); X = np.zeros(nl); red = 0.0
        for q in per_rank:
            for row in q:
                l = int(row[1]); tS, tP, tX = level_times(row, p)
                S[l] = max(S[l], tS); C[l] = max(C[l], tP + tX); X[l] = max(X[l], tX)
                red = max(red, abs(row[15]) / (p["link_gbs"] * 1e9))
        T = C[0] + sum(max(S[l], C[l + 1] if l + 1 < nl else 0.0) for l in range(nl))
        rows.append((zl, nl, S.sum(), C.sum(), X.sum(), T, red))
        total_h += T; total_e += T + red
    return total_h, total_e, rows
