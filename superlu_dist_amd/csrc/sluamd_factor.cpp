// sluamd_factor.cpp -- numeric drivers: pdgstrf3d (Z-level loop, per-level panel pipeline with look-ahead, XY panel
// exchange, Z ancestor reduction) and pdgstrs3d (level-set forward / backward sweeps with the matching exchanges).
// All device work goes through eng:: (sluamd_kernels.hip), all communication through Comm (sluamd_comm.h).
#include <algorithm>
#include <cstring>
#include "sluamd_comm.h"
#include "sluamd_plan.h"

namespace sluamd {

static void ev_begin(Handle *H, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t &used, hipStream_t s)
{
    if (!H->profile) return;
    if (used == v.size()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); v.emplace_back(a, b); }
    hipEventRecord(v[used].first, s);
}
static void ev_end(Handle *H, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t &used, hipStream_t s)
{
    if (!H->profile) return;
    hipEventRecord(v[used].second, s);
    ++used;
}
static double ev_sum(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t used)
{
    double tot = 0;
    for (size_t i = 0; i < used; ++i) { float ms = 0; hipEventElapsedTime(&ms, v[i].first, v[i].second); tot += ms; }
    return tot;
}
static hipEvent_t next_event(Handle *H)
{
    if (H->ev_pool_used == H->ev_pool.size()) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); H->ev_pool.push_back(e); }
    return H->ev_pool[H->ev_pool_used++];
}

// one grouped exchange of contiguous arena ranges on stream s
static int exchange(Handle *H, const std::vector<XMsg> &sends, const std::vector<XMsg> &recvs, hipStream_t s)
{
    if (sends.empty() && recvs.empty()) return 0;
    Comm *c = H->comm;
    int rc = c->begin();
    if (rc) return rc;
    const int vs = H->z ? 2 : 1;     // doubles per value (arena offsets / lengths are in values)
    for (auto &m : sends) if ((rc = c->send(H->d_val + m.off * vs, m.len * 8 * vs, m.peer))) return rc;
    for (auto &m : recvs) if ((rc = c->recv(H->d_val + m.off * vs, m.len * 8 * vs, m.peer))) return rc;
    return c->end(s);
}

// One elimination forest, level by level.  Serial mode (profiling / deterministic): everything on one stream.
// Look-ahead mode (default): the Schur update of level l is split into the tiles that feed level l+1's panels
// ("urgent", explicit list) and the rest; the panel kernels of level l+1 -- and on an XY layer their two exchange
// phases -- run on a high-priority stream as soon as the urgent tiles are done and overlap with the rest: the GPU form of
// the reference's look-ahead pipeline (dsparseTreeFactor_ASYNC, dtreeFactorization.c:381-706, num_lookaheads).
static hipEvent_t red_wait_event(const Handle *H, int64_t lend, int64_t uend);     // pipelined Z reduction, below

// Inverse of the block triangle of one merged chain group (Handle::groups) into its pair of T.inv: zero fill, gather (diagonal blocks = the members' Linv / Uinv,
// off-diagonal blocks of the panels / skylines into dense images), then the block substitutions by distance as batched dense products -- all on stream st.
static void group_inverse(Handle *H, int gi, hipStream_t st)
{
    const Handle::SolveGroup &G = H->groups[gi];
    hipMemsetAsync(H->T.inv + G.ginv, 0, sizeof(double) * (size_t) 2 * G.nG * G.nG, st);
    hipMemsetAsync(H->d_gscr, 0, sizeof(double) * (size_t) (2 * GRP_SCR), st);       // LG | UG (the sums TL / TU are written before they are read)
    eng::grp_gather(st, H->T, H->d_grpdesc + gi, H->d_gscr);
    for (int q = 0; q + 1 < 7; ++q)
        if (G.tile_off[q + 1] > G.tile_off[q]) { eng::gemm_batched(st, H->T, H->d_gemmdesc, H->d_gemmtiles + G.tile_off[q], G.tile_off[q + 1] - G.tile_off[q], H->d_gscr); H->st.num_launches++; }
    H->st.num_launches += 1;
}

static int run_factor_sched(Handle *H, LevelSched &S, double thresh)
{
    const DevTables &T = H->T;
    const bool xy = H->grid.Pr * H->grid.Pc > 1;
    const bool lookahead = !H->profile && !H->opt.deterministic && !H->env.no_lookahead;
    const bool gemm_panels = !H->env.trsm_panels && !H->z;   // XY layers: the peers of a diagonal block invert the copy they receive (full_inv on SNF_HAS_DIAG)
    hipStream_t s = H->stream, ps = lookahead ? H->pstream : H->stream;
    int rc_x = 0;
    // Tail of the panel chain (1 x 1 layers): on the last `trsm_tail` single-supernode levels -- the top separator, where the trailing
    // update is shorter than the chain diag LU -> Linv / Uinv -> panel GEMM -> urgent tiles -- the panel solves run as blocked
    // substitutions on the 32 x 32 inverses the diagonal kernel leaves behind (k_panel_trsm), so that the full inverses (which the
    // triangular solves still want) leave the chain: they are computed beside it on the bulk stream
    const int trsm_tail = (gemm_panels && !xy) ? H->env.trsm_tail : 0;
    auto tail_level = [&](int l) { return trsm_tail > 0 && l >= S.nlevels - trsm_tail && S.lvl_off[l + 1] - S.lvl_off[l] == 1; };
    auto deferred_inv = [&](hipStream_t st, int l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        eng::full_inv(st, T, S.d_nodes + n0, S.d_finv_prefix + po, nn, S.finv_prefix[po + nn], S.max_nsupc[l]);
        H->st.num_launches++;
    };
    int cur_level = 0, cur_pass = 0;
    // per-tile records of the list schedules (k_schur mmode 1 / 2): built once per schedule, on its first factorisation
    if (S.maps_state == 0) {
        S.maps_state = -1;
        if (!H->env.no_tile_maps && !H->opt.deterministic && !S.ulist.empty()) {
            S.m_off.assign(2 * S.nlevels + 1, 0);
            for (int l = 0; l < S.nlevels; ++l)
                for (int g = 0; g < 2; ++g) {
                    const int64_t cnt = S.u_off[(2 * l + g) * 4 + 4] - S.u_off[(2 * l + g) * 4];
                    S.m_off[2 * l + g + 1] = S.m_off[2 * l + g] + cnt * eng::schur_rec_ints(g == 0 ? 0 : 2, H->z);
                }
            const size_t bytes = sizeof(int) * (size_t) S.m_off[2 * S.nlevels];
            size_t fr = 0, tot = 0;
            hipMemGetInfo(&fr, &tot);
            // the records are an accelerator, not a requirement: leave a tenth of the device free.  (Chunks the arena pool holds for later handles count as free:
            // they go back to the driver when the records need the room.)
            const size_t cached = devpool_cached_bytes(H->device);
            if (bytes > 0 && bytes + tot / 10 >= fr && bytes + tot / 10 < fr + cached) { devpool_trim(H->device); hipMemGetInfo(&fr, &tot); }
            if (bytes > 0 && bytes + tot / 10 < fr && hipMalloc((void **) &S.d_tmaps, bytes) == hipSuccess) {
                H->d_misc.push_back(S.d_tmaps);
                for (int l = 0; l < S.nlevels; ++l)
                    for (int g = 0; g < 2; ++g) {
                        const int u0 = S.u_off[(2 * l + g) * 4], nu = S.u_off[(2 * l + g) * 4 + 4] - u0;
                        if (!nu) continue;
                        if (H->z) eng::zschur(s, g == 0 ? 0 : 1, T, nullptr, nullptr, 0, 0, nu, H->d_info, S.d_ulist + u0, 0, S.d_tmaps + S.m_off[2 * l + g], 1);
                        else eng::schur(s, g == 0 ? (H->env.schur_4waves ? 1 : 0) : 2, T, nullptr, nullptr, 0, 0, nu, H->d_info, S.d_ulist + u0, 0, S.d_tmaps + S.m_off[2 * l + g], 1);
                    }
                S.maps_state = 1;
                H->st.bytes_device += (int64_t) bytes;
            } else (void) hipGetLastError();
        }
    }
    auto schur = [&](hipStream_t st, bool big, int ntile, const int *nodes, const int *prefix, int nn, int id_base,
                     const int4 *ulist, int prio = 0, const int *tmaps = nullptr, int ksplit = 1, const int *xoff = nullptr, int xmax = 0) {
        if (H->profile && H->env.profile_dump) H->schur_rec.push_back({cur_level, cur_pass, big ? 1 : 0, ntile, S.max_nsupc[cur_level]});
        if (H->profile) { if (H->ev_schur_big.size() <= H->ev_schur_used) H->ev_schur_big.resize(H->ev_schur_used + 1); H->ev_schur_big[H->ev_schur_used] = big ? 1 : 0; }
        ev_begin(H, H->ev_schur, H->ev_schur_used, st);
        if (H->z) eng::zschur(st, big ? 0 : 1, T, nodes, prefix, nn, id_base, ntile, H->d_info, ulist, prio, tmaps, tmaps ? 2 : 0, xoff, xmax);     // complex16: k_schur on the real embedding
        else eng::schur(st, big ? (H->env.schur_4waves ? 1 : 0) : 2, T, nodes, prefix, nn, id_base, ntile, H->d_info, ulist, prio, tmaps, tmaps ? 2 : 0, ksplit, xoff, xmax);
        ev_end(H, H->ev_schur, H->ev_schur_used, st);
        H->st.num_launches++; H->st.schur_launches++; H->st.schur_tiles += ntile;
    };
    // panel(l) in two parts: A needs only the diagonal blocks of level l to be up to date, B the whole panels
    // pipelined Z reduction: the own L / U slots of DAG level l end at these arena offsets (own slots are laid out level by level)
    std::vector<int64_t> lv_lend, lv_uend;
    if (!H->red_events.empty()) {
        lv_lend.assign(S.nlevels, -1); lv_uend.assign(S.nlevels, -1);
        int64_t le = -1, ue = -1;
        for (int l = 0; l < S.nlevels; ++l) {
            for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
                const int k = S.nodes[i];
                if ((H->h_flags[k] & SNF_L_OWN) && H->hs.lval_len[k]) le = std::max(le, H->hs.lval_off[k] + H->hs.lval_len[k]);
                if ((H->h_flags[k] & SNF_U_OWN) && H->hs.uval_len[k]) ue = std::max(ue, H->hs.uval_off[k] + H->hs.uval_len[k]);
            }
            lv_lend[l] = le; lv_uend[l] = ue;
        }
    }
    else if (H->red_all) hipStreamWaitEvent(s, H->red_all, 0);     // nothing chunk-wise pending: whatever an earlier reduction still adds must be in
    auto panelA = [&](int l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        const int *nodes = S.d_nodes + n0;
        const int mx = S.max_nsupc[l];
        if (!lv_lend.empty()) {   // the panels of this level must hold the partner layer's contributions before they are factored
            hipEvent_t e = red_wait_event(H, lv_lend[l], lv_uend[l]);
            if (e) hipStreamWaitEvent(ps, e, 0);
        }
        ev_begin(H, H->ev_panel, H->ev_panel_used, ps);
        if (H->z) {   // Local_Zgstrf2 (pzgstrf2.c); the complex panel solves substitute on the factored block: no inverses
            eng::zdiag_lu(ps, T, nodes, nn, mx, H->opt.replace_tiny_pivot, thresh, H->d_info);
            if (xy) {   // zDiagFactIBCast (ztrfCommWrapper.c): packed diagonal blocks down the process column and along the process row
                eng::pack_diag(ps, T, nodes, S.d_dg_prefix + po, S.d_dg_off + po, nn, S.dg_prefix[po + nn], H->d_val + 2 * S.dg_stage_off[l], 2);
                ev_begin(H, H->ev_xchg, H->ev_xchg_used, ps);
                if (!rc_x) rc_x = exchange(H, S.x_diag_send[l], S.x_diag_recv[l], ps);
                ev_end(H, H->ev_xchg, H->ev_xchg_used, ps);
            }
            ev_end(H, H->ev_panel, H->ev_panel_used, ps);
            H->st.num_launches += 1 + (xy ? 1 : 0);
            return;
        }
        // bit 2: the single-supernode levels at the top of the tree get the diagonal kernel built for the whole register file (no scratch)
        const int big_regs = (H->env.diag_tail > 0 && nn == 1 && l >= S.nlevels - H->env.diag_tail) ? 4 : 0;
        eng::diag_lu(ps, T, nodes, nn, mx, (H->opt.replace_tiny_pivot ? 1 : 0) | (H->env.diag_v1 ? 2 : 0) | big_regs, thresh, H->d_info);   // Local_Dgstrf2 (+ dinv of the owned blocks)
        if (xy) {   // dDiagFactIBCast (dtrfCommWrapper.c:32-118): diagonal blocks down the process column and along the process row
            eng::pack_diag(ps, T, nodes, S.d_dg_prefix + po, S.d_dg_off + po, nn, S.dg_prefix[po + nn], H->d_val + S.dg_stage_off[l]);
            ev_begin(H, H->ev_xchg, H->ev_xchg_used, ps);
            if (!rc_x) rc_x = exchange(H, S.x_diag_send[l], S.x_diag_recv[l], ps);
            ev_end(H, H->ev_xchg, H->ev_xchg_used, ps);
            eng::diag_inv(ps, T, nodes, S.d_inv_prefix + po, nn, S.inv_prefix[po + nn]);   // column / row peers invert the diagonal blocks they received
        }
        const bool inv_here = gemm_panels && !tail_level(l);
        if (inv_here) eng::full_inv(ps, T, nodes, S.d_finv_prefix + po, nn, S.finv_prefix[po + nn], mx);   // Linv / Uinv (the solve uses the owner's too)
        ev_end(H, H->ev_panel, H->ev_panel_used, ps);
        H->st.num_launches += 1 + (inv_here ? 1 : 0) + (xy ? 2 : 0);
    };
    // Split panel solves (LevelSched::ps_units, look-ahead schedule only): part 0 = the strips / chunks the level's part-0 tiles read, on the panel stream;
    // part 1 = all the others, on the urgent-tile stream beside the next level's diagonal LU.  The GPU form of what the reference's look-ahead window does
    // with its panel factorisations (dsparseTreeFactor_ASYNC, dtreeFactorization.c:381-470: the panels of the look-ahead supernodes first).
    auto level_split = [&](int l) { return lookahead && !S.ps_off.empty() && S.ps_off[4 * l + 4] > S.ps_off[4 * l]; };
    auto panelB_part = [&](hipStream_t st, int l, int part) {
        const int mx = S.max_nsupc[l];
        const int o0 = S.ps_off[4 * l + 2 * part], o1 = S.ps_off[4 * l + 2 * part + 1], o2 = S.ps_off[4 * l + 2 * part + 2];
        const int nl = o1 - o0, nu = o2 - o1;
        if (nl + nu == 0) return;
        if (gemm_panels && !tail_level(l)) eng::panel_gemm(st, T, nullptr, nullptr, nullptr, 0, nl, nu, mx, S.d_ps_units + o0);
        else eng::panel_trsm(st, T, nullptr, nullptr, nullptr, 0, nl, nu, 64, mx, S.d_ps_units + o0);
        H->st.num_launches++;
    };
    auto panelB = [&](int l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        const int *nodes = S.d_nodes + n0;
        const int mx = S.max_nsupc[l];
        const int nl = S.ltr_prefix[po + nn], nu = S.utr_prefix[po + nn];
        ev_begin(H, H->ev_panel, H->ev_panel_used, ps);
        if (H->z) {   // zLPanelTrSolve / zUPanelTrSolve (ztrfCommWrapper.c): 64-row strips / 64-column chunks
            const int znl = S.zltr_prefix[po + nn], znu = S.bwd_prefix[po + nn];
            eng::zpanel_trsm(ps, T, nodes, S.d_zltr_prefix + po, S.d_bwd_prefix + po, nn, znl, znu, mx);
            if (xy) {
                ev_begin(H, H->ev_xchg, H->ev_xchg_used, ps);
                if (!rc_x) rc_x = exchange(H, S.x_panel_send[l], S.x_panel_recv[l], ps);   // zIBcastRecvLPanel / zIBcastRecvUPanel
                ev_end(H, H->ev_xchg, H->ev_xchg_used, ps);
            }
            ev_end(H, H->ev_panel, H->ev_panel_used, ps);
            H->st.num_launches += (znl + znu > 0);
            return;
        }
        if (gemm_panels && !tail_level(l)) eng::panel_gemm(ps, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, nl, nu, mx);   // chain-free GEMM form of dLPanelTrSolve / dUPanelTrSolve
        else eng::panel_trsm(ps, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, nl, nu, trsm_rs(*H, (mx + 31) & ~31), mx);
        if (xy) {
            ev_begin(H, H->ev_xchg, H->ev_xchg_used, ps);
            if (!rc_x) rc_x = exchange(H, S.x_panel_send[l], S.x_panel_recv[l], ps);   // dIBcastRecvLPanel / dIBcastRecvUPanel
            ev_end(H, H->ev_xchg, H->ev_xchg_used, ps);
        }
        ev_end(H, H->ev_panel, H->ev_panel_used, ps);
        H->st.num_launches += (nl + nu > 0);
    };
    // Stream program.  Serial mode: everything in order on one stream.  Look-ahead mode, four streams:
    //   s   (bulk)     : the bulk of Schur(l), level after level                                        -- never waits for a later panel
    //   us  (urgent 1) : U1(l) [tiles that feed level-(l+1) panels] minus its diagonal-block tiles, which run on ps in front of diag_lu(l+1)
    //   u2s (urgent 2) : U2(l) [tiles that feed level-(l+2) panels]
    //   ps  (panels)   : panel(l+1) = diag LU / inverses after U1's diagonal part, panel solves after all of U1(l)
    // panel(l+1) needs U1(l), U2(l-1) and the bulk of every level <= l-2: while the bulk of one level runs, the panels of the
    // next TWO levels and the urgent tiles between them are factored beside it.
    hipStream_t us = lookahead ? H->ustream : H->stream, u2s = lookahead ? H->u2stream : H->stream;
    if (lookahead && S.nlevels) {
        hipEvent_t e = next_event(H);    // the side streams must see everything queued so far on the main stream
        hipEventRecord(e, s); hipStreamWaitEvent(ps, e, 0); hipStreamWaitEvent(us, e, 0); hipStreamWaitEvent(u2s, e, 0);
    }
    // parts p0 .. p1 of level l's tile lists (0 / 1: feed level-(l+1) panels, diagonal blocks first; 2: feed level-(l+2) panels;
    // 3: bulk) as ONE launch per tile-size group -- the parts are contiguous in the list
    auto list_launch = [&](hipStream_t st, int l, int p0, int p1) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0;
        const int nbig = S.n_big[l];
        for (int g = 0; g < 2; ++g) {
            const int cnt = g == 0 ? nbig : nn - nbig;
            if (!cnt) continue;
            const int u0 = S.u_off[(2 * l + g) * 4 + p0], nu = S.u_off[(2 * l + g) * 4 + p1 + 1] - u0;
            const int *tm = S.maps_state == 1 ? S.d_tmaps + S.m_off[2 * l + g] + (int64_t) (u0 - S.u_off[(2 * l + g) * 4]) * eng::schur_rec_ints(g == 0 ? 0 : 2, H->z) : nullptr;
            // the diagonal-block tiles of the next level (part 0, on the panel stream) when they are few: split K over several workgroups per tile
            const int ks = (lookahead && p0 == 0 && p1 == 0 && g == 0 && nu <= 64 && !H->z && !H->env.schur_4waves) ? H->env.ksplit : 1;
            // a launch of the bulk alone: the XCDs' ranges of equal modelled cost (balance_bulk)
            const bool bal = p0 == 3 && p1 == 3 && S.d_x_off && !S.x_off.empty() && S.x_off[10 * (size_t) (2 * l + g) + 9] > 0;
            if (nu) schur(st, g == 0, nu, nullptr, nullptr, 0, 0, S.d_ulist + u0, (lookahead && p1 < 3) ? 1 : 0, tm, ks, bal ? S.d_x_off + 10 * (size_t) (2 * l + g) : nullptr,
                          bal ? S.x_off[10 * (size_t) (2 * l + g) + 9] : 0);
        }
    };
    // deterministic mode: one supernode per launch over its full tile grid -- tiles of one k hit distinct destinations, the
    // summation order is fixed
    auto grid_launch = [&](hipStream_t st, int l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0;
        const int *nodes = S.d_nodes + n0;
        const int nbig = S.n_big[l];
        for (int g = 0; g < 2; ++g) {
            const int cnt = g == 0 ? nbig : nn - nbig;
            if (!cnt) continue;
            const int so = S.lvl_soff[l] + (g == 0 ? 0 : nbig + 1);
            const int *gn = nodes + (g == 0 ? 0 : nbig);
            for (int i = 0; i < cnt; ++i) {
                const int c = S.tile_prefix[so + i + 1] - S.tile_prefix[so + i];
                if (c) schur(st, g == 0, c, gn, S.d_tile_prefix + so, cnt, S.tile_prefix[so + i], nullptr);
            }
        }
    };
    auto wait_on = [&](hipStream_t waiter, hipStream_t on) -> hipEvent_t {   // waiter continues after everything queued on `on`
        hipEvent_t e = next_event(H);
        hipEventRecord(e, on); hipStreamWaitEvent(waiter, e, 0);
        return e;
    };
    hipEvent_t e_split_urgent = nullptr; // level l's panels were solved in two parts: the urgent part's event (the rest is queued on us)
    hipEvent_t e_u2_prev = nullptr;      // U2(l-1) done
    hipEvent_t e_bulk_prev = nullptr, e_bulk_prev2 = nullptr;   // bulk(l-1), bulk(l-2) done (stream order: and every earlier one)
    if (S.nlevels) { panelA(0); panelB(0); if (!lookahead && tail_level(0)) deferred_inv(s, 0); }
    for (int l = 0; l < S.nlevels; ++l) {
        cur_level = l;
        const bool more = l + 1 < S.nlevels;
        if (!lookahead) {
            // serial partition (also the measurement harness of the Schur kernel, as in rounds 1-2): one launch per level and
            // tile-size group; only K-fused levels need their level-(l+1) urgent tiles launched apart
            if (H->opt.deterministic) { cur_pass = 3; grid_launch(s, l); }
            // (the bulk as its own launch: its XCD ranges are balanced, LevelSched::x_off)
            else {
                const bool bal = !S.x_off.empty() && (S.x_off[10 * (size_t) (2 * l) + 9] > 0 || S.x_off[10 * (size_t) (2 * l + 1) + 9] > 0);
                if (T.defer && S.lvl_defer[l]) { cur_pass = 0; list_launch(s, l, 0, 0); cur_pass = 1; list_launch(s, l, 1, 1); cur_pass = 3; if (bal) { list_launch(s, l, 2, 2); list_launch(s, l, 3, 3); } else list_launch(s, l, 2, 3); }
                else if (bal) { cur_pass = 3; list_launch(s, l, 0, 2); list_launch(s, l, 3, 3); }
                else { cur_pass = 3; list_launch(s, l, 0, 3); }
            }
            if (more) {
                panelA(l + 1); panelB(l + 1);
                if (tail_level(l + 1)) deferred_inv(s, l + 1);
                if (!H->lvl_groups.empty() && &S == &H->sched[0]) for (int gi : H->lvl_groups[l + 1]) group_inverse(H, gi, s);
            }
            if (rc_x) return rc_x;
            continue;
        }
        if (e_split_urgent) {
            // level l was solved in two parts: its urgent strips on ps (event e_split_urgent), the others on us, queued there already.  The part-0 tiles
            // below follow the urgent strips in stream order; everything else needs both parts
            hipStreamWaitEvent(us, e_split_urgent, 0);
            hipEvent_t e_rest = next_event(H);
            hipEventRecord(e_rest, us);
            hipStreamWaitEvent(u2s, e_split_urgent, 0); hipStreamWaitEvent(u2s, e_rest, 0);
            hipStreamWaitEvent(s, e_split_urgent, 0); hipStreamWaitEvent(s, e_rest, 0);
            e_split_urgent = nullptr;
        } else {
            hipEvent_t e_p = next_event(H);  // panel(l) done
            hipEventRecord(e_p, ps);
            hipStreamWaitEvent(us, e_p, 0); hipStreamWaitEvent(u2s, e_p, 0); hipStreamWaitEvent(s, e_p, 0);
        }
        if (tail_level(l)) deferred_inv(s, l);        // off the chain: the bulk stream waits for panel(l) anyway
        if (!H->lvl_groups.empty() && &S == &H->sched[0] && !H->lvl_groups[l].empty()) {
            // merged chain groups whose last member was factored at this level: their inverses on a stream of their own, after the members' panels and inverses
            hipEvent_t e_g = next_event(H);
            hipEventRecord(e_g, s); hipStreamWaitEvent(H->gstream, e_g, 0);
            for (int gi : H->lvl_groups[l]) group_inverse(H, gi, H->gstream);
        }
        cur_pass = 0; list_launch(ps, l, 0, 0);   // on the panel stream itself: diag_lu(l+1) follows in stream order, no event hop
        cur_pass = 1; list_launch(us, l, 1, 1);
        hipEvent_t e_u1 = next_event(H); hipEventRecord(e_u1, us);
        cur_pass = 2; list_launch(u2s, l, 2, 2);
        hipEvent_t e_u2 = next_event(H); hipEventRecord(e_u2, u2s);
        cur_pass = 3; list_launch(s, l, 3, 3);
        hipEvent_t e_bulk = next_event(H); hipEventRecord(e_bulk, s);
        if (more) {
            if (e_u2_prev) hipStreamWaitEvent(ps, e_u2_prev, 0);
            if (e_bulk_prev2) hipStreamWaitEvent(ps, e_bulk_prev2, 0);
            if (xy && e_bulk_prev) hipStreamWaitEvent(ps, e_bulk_prev, 0);   // XY layer: the received panels of level l-1 share the scratch copy (level parity) that panel(l+1)'s exchange fills
            panelA(l + 1);
            if (level_split(l + 1)) {
                hipEvent_t e_pa = next_event(H); hipEventRecord(e_pa, ps);     // diagonal blocks (and inverses) of level l + 1 done
                hipStreamWaitEvent(ps, e_u1, 0);
                panelB_part(ps, l + 1, 0);                                     // urgent strips: what diag_lu(l + 2) waits for, through the part-0 tiles
                e_split_urgent = next_event(H); hipEventRecord(e_split_urgent, ps);
                hipStreamWaitEvent(us, e_pa, 0);                               // (after U1(l) in stream order; U2(l - 1) and the bulk of levels <= l - 2 through e_pa)
                panelB_part(us, l + 1, 1);                                     // the other strips, beside diag_lu(l + 2)
            } else {
                hipStreamWaitEvent(ps, e_u1, 0);
                panelB(l + 1);
            }
        }
        e_u2_prev = e_u2; e_bulk_prev2 = e_bulk_prev; e_bulk_prev = e_bulk;
        if (rc_x) return rc_x;
    }
    if (lookahead && S.nlevels) { wait_on(s, ps); wait_on(s, us); wait_on(s, u2s); if (!H->lvl_groups.empty() && H->gstream) wait_on(s, H->gstream); }
    HIPCHK(hipGetLastError());
    return rc_x;
}

static int ensure_xtmp(Handle *H, int64_t doubles)
{
    if (doubles <= H->xtmp_cap) return 0;
    if (H->d_xtmp) hipFree(H->d_xtmp);
    H->d_xtmp = nullptr; H->xtmp_cap = 0;
    if (hipMalloc((void **) &H->d_xtmp, sizeof(double) * (size_t) doubles) != hipSuccess) { set_error("hipMalloc of the exchange buffer failed"); return SLUAMD_ENOMEM; }
    H->xtmp_cap = doubles;
    return 0;
}

// own-slot ranges [L | U] of the ancestor forests above Z level zl: contiguous by construction of the arena
static void ancestor_ranges(const Handle *H, int zl, std::vector<std::pair<int64_t, int64_t>> &out)
{
    out.clear();
    const HostStruct &hs = H->hs;
    const Grid &g = H->grid;
    for (int pass = 0; pass < 2; ++pass) {
        int64_t first = -1, len = 0;
        for (size_t a = zl + 1; a < H->forest_nodes.size(); ++a)
            for (int k : H->forest_nodes[a]) {
                if (pass == 0 && g.kcol(k) == g.c && hs.lval_len[k]) { if (first < 0 || hs.lval_off[k] < first) first = hs.lval_off[k]; len += hs.lval_len[k]; }
                if (pass == 1 && g.krow(k) == g.r && hs.uval_len[k]) { if (first < 0 || hs.uval_off[k] < first) first = hs.uval_off[k]; len += hs.uval_len[k]; }
            }
        if (len) out.emplace_back(first, len);
    }
}

// dreduceAllAncestors3d (pd3dcomm.c:1046-1081) after Z level zl: the layer myz + 2^zl sends its copies of every ancestor
// forest to layer myz (myz % 2^(zl+1) == 0), which adds them (dzRecvLPanel / dzRecvUPanel: daxpy) -- whole arena ranges in
// bounded chunks through one staging buffer.
// PIPELINED with the factorisation of the next forest (the reference overlaps the same way through its look-ahead / Isend,
// pdgstrf3d.c:333-385): receives and additions run on a stream of their own (rstream), L and U chunks alternate in arena order
// (= Z level, DAG level, supernode), every chunk records an event; a DAG level of the next forest waits only for the chunks that
// cover ITS own L and U slots (run_factor_sched).  The additions are fp64 atomics: the Schur tiles of the levels already running
// scatter into panels of later levels whose chunks are still to come -- sums commute, the two kinds of update must only not tear.
static hipEvent_t red_event(Handle *H)
{
    if (H->red_pool_used == H->red_pool.size()) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); H->red_pool.push_back(e); }
    return H->red_pool[H->red_pool_used++];
}

static int reduce_ancestors(Handle *H, int zl)
{
    const Grid &g = H->grid;
    const int step = 1 << zl;
    H->red_events.clear();
    if (zl + 1 >= (int) H->forest_nodes.size() || (g.z % step) != 0) return 0;
    std::vector<std::pair<int64_t, int64_t>> rg;
    ancestor_ranges(H, zl, rg);
    const bool receiver = (g.z % (2 * step)) == 0;
    const int peer = g.rank_of(g.r, g.c, receiver ? g.z + step : g.z - step);
    if (receiver && g.z + step >= g.Pz) return 0;
    const int vs = H->z ? 2 : 1;            // doubles per value (complex16 panels are reduced as pairs of doubles)
    const int64_t CH = (int64_t) 1 << 24;   // values per message: <= 256 MiB
    Comm *c = H->comm;
    // serial (profiling / deterministic / SLUAMD_NO_LOOKAHEAD) runs keep the reduction on the main stream: t_reduce_ms then measures it
    const bool pipelined = H->rstream && !H->profile && !H->opt.deterministic && !H->env.no_lookahead;
    hipStream_t s = H->stream, rs = pipelined ? H->rstream : H->stream;
    // what is sent / added to must be complete: everything the factorisation queued so far (all look-ahead streams were joined
    // into s), and on the sender also the additions of an earlier reduction that are still running on rstream
    {
        hipEvent_t e = red_event(H);
        HIPCHK(hipEventRecord(e, s)); HIPCHK(hipStreamWaitEvent(rs, e, 0));
    }
    // chunk sequence: L and U alternate, so that the first DAG levels of the next forest are complete early in BOTH ranges
    struct Ch { int r; int64_t o, len; };
    std::vector<Ch> seq;
    {
        std::vector<int64_t> pos(rg.size(), 0);
        for (bool any = true; any;) {
            any = false;
            for (size_t r = 0; r < rg.size(); ++r)
                if (pos[r] < rg[r].second) { const int64_t len = std::min(CH, rg[r].second - pos[r]); seq.push_back({(int) r, pos[r], len}); pos[r] += len; any = true; }
        }
    }
    int rc;
    if (receiver) {
        int64_t mx = 0;
        for (auto &q : seq) mx = std::max(mx, q.len);
        if ((rc = ensure_xtmp(H, std::max<int64_t>(mx * vs, 1)))) return rc;
    }
    int64_t done[2] = {-1, -1};            // arena offset covered so far in the L range (rg[0]) and the U range (rg[1]); -1: nothing to wait for
    // ancestor_ranges() returns [L range][U range] (either may be missing: then its kind is complete from the start)
    bool has_l = false, has_u = false;
    {
        // tell the kinds apart by the arena layout: own L slots come first, own U slots after hs.nnzL
        for (size_t r = 0; r < rg.size(); ++r) { if (rg[r].first < H->hs.nnzL) has_l = true; else has_u = true; }
    }
    for (auto &q : seq) {
        const auto &r = rg[q.r];
        if ((rc = c->begin())) return rc;
        if (receiver) {
            if ((rc = c->recv(H->d_xtmp, q.len * 8 * vs, peer))) return rc;
            if ((rc = c->end(rs))) return rc;
            eng::add_atomic(rs, q.len * vs, H->d_xtmp, H->d_val + (r.first + q.o) * vs);   // the next chunk's receive into the staging buffer is ordered behind this on rs
            const bool is_l = r.first < H->hs.nnzL;
            done[is_l ? 0 : 1] = r.first + q.o + q.len;
            hipEvent_t e = red_event(H);
            HIPCHK(hipEventRecord(e, rs));
            H->red_events.push_back({has_l ? done[0] : INT64_MAX, has_u ? done[1] : INT64_MAX, e});
        } else {
            if ((rc = c->send(H->d_val + (r.first + q.o) * vs, q.len * 8 * vs, peer))) return rc;
            if ((rc = c->end(rs))) return rc;
        }
    }
    if (!pipelined) { H->red_events.clear(); return 0; }      // everything is on s already
    HIPCHK(hipEventRecord(H->red_all, rs));
    if (!receiver) HIPCHK(hipStreamWaitEvent(s, H->red_all, 0));    // a sender's later work (the solve) follows its sends
    return 0;
}

// the chunk event a DAG level of the forest being factored has to wait for: the first one whose coverage reaches the end of the
// level's own L and U slots (nullptr: nothing pending)
static hipEvent_t red_wait_event(const Handle *H, int64_t lend, int64_t uend)
{
    for (auto &e : H->red_events)
        if ((lend < 0 || e.lend >= lend) && (uend < 0 || e.uend >= uend)) return e.ev;
    return H->red_events.empty() ? nullptr : H->red_events.back().ev;
}

int run_factor(Handle *H, double thresh, int *info)
{
    HIPCHK(hipSetDevice(H->device));
    const Grid &g = H->grid;
    if (g.size() > 1 && !H->comm) { set_error("handle of a multi-rank grid has no communicator"); return SLUAMD_EINVAL; }
    int init[8] = {0x7fffffff, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(H->d_info, init, sizeof(init), hipMemcpyHostToDevice, H->stream));
    H->st.num_launches = 0; H->st.schur_launches = 0; H->st.schur_tiles = 0;
    H->profile = H->opt.verbose >= 2 || H->env.profile;
    H->ev_schur_used = H->ev_panel_used = H->ev_xchg_used = H->ev_red_used = 0;
    H->schur_rec.clear();
    H->ev_pool_used = 0;
    H->red_pool_used = 0; H->red_events.clear();
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    // Z levels in order (pdgstrf3d.c:333-385): factor my forest of the level, then the ancestor reduction
    int rc = 0;
    for (size_t zl = 0; zl < H->sched.size(); ++zl) {
        if (H->z_active[zl]) {
            rc = run_factor_sched(H, H->sched[zl], thresh);     // double and complex16 alike: the look-ahead schedule is type-independent
            if (rc) return rc;
        }
        if (g.Pz > 1) {
            ev_begin(H, H->ev_red, H->ev_red_used, H->stream);
            rc = reduce_ancestors(H, (int) zl);
            ev_end(H, H->ev_red, H->ev_red_used, H->stream);
            if (rc) return rc;
        }
    }
    if (H->red_all) { HIPCHK(hipStreamWaitEvent(H->stream, H->red_all, 0)); H->red_events.clear(); }   // the last reduction's additions belong to the factorisation
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    int res[8];
    HIPCHK(hipMemcpyAsync(res, H->d_info, sizeof(res), hipMemcpyDeviceToHost, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_factor_ms = ms;
    if (getenv("SLUAMD_EXP_COUNT")) fprintf(stderr, "[sluamd] K chunks of the Schur tiles (instrumented kernel build only): 128 x 128 clean %d other %d; 64 x 64 clean %d other %d\n", res[4], res[5], res[6], res[7]);
    H->dinv_ready = true; H->inv_ready = !H->env.trsm_panels && !H->z; H->factored = true;
    if (H->profile && H->env.profile_dump && H->schur_rec.size() == H->ev_schur_used)
        for (size_t i = 0; i < H->ev_schur_used; ++i) {
            float ems = 0; hipEventElapsedTime(&ems, H->ev_schur[i].first, H->ev_schur[i].second);
            const auto &r = H->schur_rec[i];
            fprintf(stderr, "SCHUR level %d pass %d big %d tiles %d max_nsupc %d ms %.4f\n", r.level, r.pass, r.big, r.ntiles, r.mx, ems);
        }
    H->st.t_schur_ms = H->profile ? ev_sum(H->ev_schur, H->ev_schur_used) : 0.0;
    H->st.t_schur_big_ms = 0.0;
    if (H->profile)
        for (size_t i = 0; i < H->ev_schur_used && i < H->ev_schur_big.size(); ++i)
            if (H->ev_schur_big[i]) { float ms = 0; hipEventElapsedTime(&ms, H->ev_schur[i].first, H->ev_schur[i].second); H->st.t_schur_big_ms += ms; }
    H->st.t_panel_ms = H->profile ? ev_sum(H->ev_panel, H->ev_panel_used) : 0.0;
    H->st.t_exchange_ms = H->profile ? ev_sum(H->ev_xchg, H->ev_xchg_used) : 0.0;     // XY panel-exchange phases (inside t_panel_ms)
    H->st.t_reduce_ms = H->profile ? ev_sum(H->ev_red, H->ev_red_used) : 0.0;         // Z ancestor reduction
    H->st.tiny_pivots = res[1];
    // zero-pivot rule: the smallest column (the reference's DOCUMENTED meaning of info, pdgstrf2.c:493-497) or, SLUAMD_INFO_LAST=1, what its code leaves:
    // each rank keeps the zero pivot it met last (Local_Dgstrf2 overwrites *info, :568-571; supernodes in elimination order), pdgstrf3d takes the MIN over ranks
    int linfo = H->env.info_last ? res[4] : ((res[0] == 0x7fffffff) ? 0 : res[0]);
    int missing = res[2];
    if (g.size() > 1) {   // info = first zero pivot over the whole grid (MPI_Allreduce MIN, pdgstrf3d.c:388-392)
        int v[2] = {linfo ? linfo : 0x7fffffff, -missing};   // one collective for both, on the library's stream
        if ((rc = H->comm->allreduce_min(v, 2, H->stream))) return rc;
        linfo = (v[0] == 0x7fffffff) ? 0 : v[0];
        missing = -v[1];
    }
    if (info) *info = linfo;
    if (missing) { set_error("Schur update found no destination block for " + std::to_string(missing) + " tiles (structure not closed)"); return SLUAMD_ESTRUCT; }
    return 0;
}

// ================================================================================================
//                                     triangular solves
// ================================================================================================
int ensure_dinv(Handle *H)
{
    if (H->dinv_ready) return 0;
    // factors were uploaded already factored: build the diagonal sub-block inverses once (owners only have the blocks)
    if (H->grid.Pr * H->grid.Pc > 1) { set_error("solve before factorisation is not supported on an XY grid"); return SLUAMD_EINVAL; }
    for (auto &S : H->sched)
        for (int l = 0; l < S.nlevels; ++l) {
            const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
            eng::diag_inv(H->stream, H->T, S.d_nodes + n0, S.d_inv_prefix + po, nn, S.inv_prefix[po + nn]);
        }
    H->dinv_ready = true;
    return 0;
}

// Linv / Uinv of every owned diagonal block, once per factorisation (pdCompute_Diag_Inv, pdgstrs.c:842)
int ensure_inv(Handle *H)
{
    if (H->inv_ready) return 0;
    int rc = ensure_dinv(H);
    if (rc) return rc;
    for (auto &S : H->sched)
        for (int l = 0; l < S.nlevels; ++l) {
            const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
            eng::full_inv(H->stream, H->T, S.d_nodes + n0, S.d_finv_prefix + po, nn, S.finv_prefix[po + nn], S.max_nsupc[l]);
        }
    for (size_t gi = 0; gi < H->groups.size(); ++gi) group_inverse(H, (int) gi, H->stream);
    H->inv_ready = true;
    return 0;
}

// device image of a run list: (row0, nrows, rows before) triples, uploaded once per list and kept with the handle
static int runs_on_device(Handle *H, const LevelSched::XSeg &m, const int **d)
{
    if (!m.d_runs) {
        std::vector<int> h;
        int64_t tot = 0;
        for (auto &r : m.runs) { h.push_back(r.first); h.push_back(r.second); h.push_back((int) tot); tot += r.second; }
        int *p = nullptr;
        if (upload(H->d_misc, h, &p)) return SLUAMD_EHIP;
        m.d_runs = p;
    }
    *d = m.d_runs;
    return 0;
}

// exchange of x segments with a set of peers: send = pack (mode 0, or 3 = pack and clear) -> grouped send/recv -> unpack
// (mode 1 = overwrite, 2 = accumulate)
static int xseg_exchange(Handle *H, double *d_x, int64_t ldx, int nrhs, const std::vector<LevelSched::XSeg> &snd, int pack_mode,
                         const std::vector<LevelSched::XSeg> &rcv, int unpack_mode, hipStream_t s)
{
    if (snd.empty() && rcv.empty()) return 0;
    int64_t need = 0;
    for (auto &m : snd) need += m.total * nrhs;
    for (auto &m : rcv) need += m.total * nrhs;
    int rc = ensure_xtmp(H, need);
    if (rc) return rc;
    int64_t off = 0;
    Comm *c = H->comm;
    std::vector<int64_t> so(snd.size()), ro(rcv.size());
    for (size_t i = 0; i < snd.size(); ++i) {
        const int *dr;
        if ((rc = runs_on_device(H, snd[i], &dr))) return rc;
        so[i] = off;
        eng::xseg_copy(s, d_x, ldx, nrhs, dr, (int) snd[i].runs.size(), snd[i].total, H->d_xtmp + off, pack_mode);
        off += snd[i].total * nrhs;
    }
    for (size_t i = 0; i < rcv.size(); ++i) { ro[i] = off; off += rcv[i].total * nrhs; }
    if ((rc = c->begin())) return rc;
    for (size_t i = 0; i < snd.size(); ++i) if ((rc = c->send(H->d_xtmp + so[i], snd[i].total * nrhs * 8, snd[i].peer))) return rc;
    for (size_t i = 0; i < rcv.size(); ++i) if ((rc = c->recv(H->d_xtmp + ro[i], rcv[i].total * nrhs * 8, rcv[i].peer))) return rc;
    if ((rc = c->end(s))) return rc;
    for (size_t i = 0; i < rcv.size(); ++i) {
        const int *dr;
        if ((rc = runs_on_device(H, rcv[i], &dr))) return rc;
        eng::xseg_copy(s, d_x, ldx, nrhs, dr, (int) rcv[i].runs.size(), rcv[i].total, H->d_xtmp + ro[i], unpack_mode);
    }
    // the staging buffer is reused by the next exchange: a stream-ordered transport (RCCL) packs, moves and unpacks in the order of
    // stream s already; the host-driven ones read and write the staging buffer from the host side
    if (!c->stream_ordered()) HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// forward / backward block solves of one Z level (one elimination forest): DAG levels ascending / descending.
// XY layers: before the diagonal solves of a level the partial sums of x_k held by the process row k % Pr are reduced to
// the diagonal owner (dlsum_fmod_inv's lsum reduction, pdgstrs_lsum.c:414-960 / dlsumReducePrK), afterwards x_k goes down
// the process column k % Pc (dbCastXk2Pck, pdgstrs3d.c).  Non-owners use their entries of x as the lsum accumulators.
// 1 x 1 layers, real: the update units of a level that feed the next level's diagonal blocks (LevelSched::fwd_units /
// bwd_units, urgent part) run first; the others (levels >= l+2 only) share ONE launch with the next level's diagonal solves
// (eng::sweep_step), so the diagonal solve of a chain supernode hides behind the far updates of its predecessor.  The
// reference's solve gets this overlap from its message-driven fmod / bmod counters (pdgstrs_lsum.c:414-960); here it is static.
// (Measured and rejected: the far units on side streams -- each event record / wait costs the chain ~6 us; the feeding units
// run by the diagonal workgroup itself -- serial 64-row strips, 7.05 -> 8.1 ms.)

static int max_rhs_chunk(const Handle *H);
// second vector of the sweeps below: the diagonal solves are out of place (strips of one supernode = independent workgroups)
static int ensure_w(Handle *H, int64_t doubles)
{
    if (doubles <= H->w_cap) return 0;
    if (H->d_w) hipFree(H->d_w);
    H->d_w = nullptr; H->w_cap = 0;
    if (hipMalloc((void **) &H->d_w, sizeof(double) * (size_t) doubles) != hipSuccess) { set_error("hipMalloc of the solve work vector failed"); return SLUAMD_ENOMEM; }
    H->w_cap = doubles;
    return 0;
}

// Joined links (LevelSched::join): ONE launch per level.  Forward: the joined units of level l + 1 apply the level-l updates to their own block of the
// right-hand side and ADD their share of y_j = Linv_j (...) into w (zeroed first); every other row of the level-l panels is updated by the regular units in
// the same launch.  Backward: the joined units of level l subtract U(k, columns of level l + 1) x themselves and ADD x_k = Uinv_k (...) into d_x (the forest's
// rows zeroed first; the top level keeps the storing strips).
static void zero_forest(Handle *H, LevelSched &S, double *x, int64_t ldx, int nrhs)
{
    if ((int) S.nodes.size() == H->hs.nsupers) { for (int q = 0; q < nrhs; ++q) hipMemsetAsync(x + (int64_t) q * ldx, 0, sizeof(double) * (size_t) H->hs.xsup[H->hs.nsupers], H->stream); }
    else eng::zero_nodes(H->stream, H->T, S.d_nodes, (int) S.nodes.size(), x, ldx, nrhs);
}
// Per level m: joined when it holds at most SLUAMD_JOIN_MAX_NODES supernodes (the recomputation-free joined units win where a level is a chain of round
// trips; levels of hundreds of supernodes with several sources each keep the two-launch form).  The two forms meet in any order: a level's diagonal blocks
// are either stored by strips or added by joined units into zeroed rows, and the level below hands over exactly the rows / columns its successor's form expects.
static inline bool level_joined(const Handle *H, const LevelSched &S, int m, int nrhs = 1)
{
    if (nrhs >= 4) return false;       // blocks of right-hand sides take the two-launch links: their units read the factor entries once per FOUR right-hand sides (k_sweep / k_fwd_update / k_bwd_update <.., RK = 4>), the joined units once per right-hand side
    if (!S.lvl_has_group.empty() && S.lvl_has_group[m]) return false;      // merged groups: strips of the group inverse, two launches for up to four levels
    return S.lvl_off[m + 1] - S.lvl_off[m] <= H->env.join_max_nodes;
}
static int solve_fwd_join(Handle *H, LevelSched &S, double *d_x, int64_t ldx, int nrhs)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    const int nl = S.nlevels;
    if (nl == 0) return 0;
    double *w = H->d_w;
    const int4 *fr = S.d_fwd_recs, *dr = S.d_diag_recs;
    zero_forest(H, S, w, ldx, nrhs);
    if (level_joined(H, S, 0, nrhs)) eng::sweep_join(s, true, T, S.d_jf_recs + 8 * (size_t) S.jf_off[0], S.jf_off[1] - S.jf_off[0], S.d_jf_aux, nullptr, 0, d_x, w, ldx, nrhs, S.max_nsupc[0]);
    else eng::sweep_step(s, true, T, S.d_diag_units + S.du_off[0], S.du_off[1] - S.du_off[0], nullptr, 0, d_x, w, ldx, nrhs, S.max_nsupc[0], dr + 2 * (size_t) S.du_off[0], nullptr);
    H->st.solve_launches += 1;
    for (int l = 0; l < nl; ++l) {      // the panels of level l, and the diagonal blocks of level l + 1
        const int mx = std::max(S.max_nsupc[l], l + 1 < nl ? S.max_nsupc[l + 1] : 0);
        if (nrhs < 4 && (l + 1 == nl || level_joined(H, S, l + 1))) {
            const int j0 = l + 1 < nl ? S.jf_off[l + 1] : 0, nj = l + 1 < nl ? S.jf_off[l + 2] - j0 : 0;
            eng::sweep_join(s, true, T, S.d_jf_recs + 8 * (size_t) j0, nj, S.d_jf_aux, S.d_jfu_recs + 2 * (size_t) S.jfu_off[l], S.jfu_off[l + 1] - S.jfu_off[l], d_x, w, ldx, nrhs, mx);
            H->st.solve_launches += 1;
        } else {
            const int u0 = S.fu_off[2 * l], u1 = S.fu_off[2 * l + 1], u2 = S.fu_off[2 * l + 2];
            const int nd = l + 1 < nl ? S.du_off[l + 2] - S.du_off[l + 1] : 0;       // (the last level's panels update other forests' rows only: no diagonal strips beside them)
            eng::fwd_update(s, T, nullptr, nullptr, 0, u1 - u0, w, d_x, ldx, nrhs, S.max_nsupc[l], S.d_fwd_units + u0, fr + 2 * (size_t) u0);
            eng::sweep_step(s, true, T, S.d_diag_units + S.du_off[l + 1], nd, S.d_fwd_units + u1, u2 - u1, d_x, w, ldx, nrhs, mx, dr + 2 * (size_t) S.du_off[l + 1], fr + 2 * (size_t) u1);
            H->st.solve_launches += 2;
        }
    }
    return 0;
}
static int solve_bwd_join(Handle *H, LevelSched &S, double *d_x, int64_t ldx, int nrhs)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    const int nl = S.nlevels;
    if (nl == 0) return 0;
    double *w = H->d_w;
    const int4 *br = S.d_bwd_recs, *dr = S.d_diag_recs;
    zero_forest(H, S, d_x, ldx, nrhs);
    // the chunks of a level whose columns lie beyond the next level run beside the diagonal blocks of the level ABOVE; the top level's (columns of ancestors in
    // other forests, solved before this sweep) run first, alone.  In which form a level's chunks come follows from its own form: joined -> every chunk, the
    // columns of the next level skipped (its joined units apply them); two-launch -> the far chunks here, the urgent ones in their own launch later.
    auto chunks = [&](int l, const int4 *&recs, const int2 *&units, int &n) {
        if (l < 0) { recs = nullptr; units = nullptr; n = 0; }
        else if (level_joined(H, S, l, nrhs)) { recs = S.d_jbu_recs + 2 * (size_t) S.jbu_off[l]; units = nullptr; n = S.jbu_off[l + 1] - S.jbu_off[l]; }
        else { const int b1 = S.bu_off[2 * l + 1], b2 = S.bu_off[2 * l + 2]; recs = br + 2 * (size_t) b1; units = S.d_bwd_units + b1; n = b2 - b1; }
    };
    {
        const int4 *recs; const int2 *units; int n;
        chunks(nl - 1, recs, units, n);
        eng::sweep_step(s, false, T, nullptr, 0, units, n, d_x, w, ldx, nrhs, S.max_nsupc[nl - 1], nullptr, recs);
        H->st.solve_launches += 1;
    }
    for (int l = nl - 1; l >= 0; --l) {
        const int mx = std::max(S.max_nsupc[l], l > 0 ? S.max_nsupc[l - 1] : 0);
        const int4 *recs; const int2 *units; int n;
        chunks(l - 1, recs, units, n);
        if (level_joined(H, S, l, nrhs)) {
            eng::sweep_join(s, false, T, S.d_jb_recs + 4 * (size_t) S.jb_off[l], S.jb_off[l + 1] - S.jb_off[l], S.d_jb_aux, recs, n, d_x, w, ldx, nrhs, mx);
            H->st.solve_launches += 1;
        } else {
            const int u0 = S.bu_off[2 * l], u1 = S.bu_off[2 * l + 1];
            eng::bwd_update(s, T, nullptr, nullptr, 0, u1 - u0, d_x, w, ldx, nrhs, S.max_nsupc[l], S.d_bwd_units + u0, br + 2 * (size_t) u0);
            eng::sweep_step(s, false, T, S.d_diag_units + S.du_off[l], S.du_off[l + 1] - S.du_off[l], units, n, d_x, w, ldx, nrhs, mx, dr + 2 * (size_t) S.du_off[l], recs);
            H->st.solve_launches += 2;
        }
    }
    return 0;
}

// Forward links.  x (= d_x) holds the right-hand side minus the updates applied so far; w receives the solved blocks y_k = Linv (x_k)
// and is what the updates read.  After the forward sweeps of all Z levels w holds y for every supernode solved on this rank.
static int solve_fwd_links(Handle *H, LevelSched &S, double *d_x, int64_t ldx, int nrhs)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    const int nl = S.nlevels;
    if (nl == 0) return 0;
    double *w = H->d_w;
    const int4 *fr = S.d_fwd_recs, *dr = S.d_diag_recs;     // unit records (null on complex handles)
    eng::sweep_step(s, true, T, S.d_diag_units + S.du_off[0], S.du_off[1] - S.du_off[0], nullptr, 0, d_x, w, ldx, nrhs, S.max_nsupc[0], dr ? dr + 2 * (size_t) S.du_off[0] : nullptr, nullptr);
    for (int l = 0; l < nl; ++l) {
        const int u0 = S.fu_off[2 * l], u1 = S.fu_off[2 * l + 1], u2 = S.fu_off[2 * l + 2];
        const int nd = (l + 1 < nl) ? S.du_off[l + 2] - S.du_off[l + 1] : 0;
        const int mx = std::max(S.max_nsupc[l], l + 1 < nl ? S.max_nsupc[l + 1] : 0);
        eng::fwd_update(s, T, nullptr, nullptr, 0, u1 - u0, w, d_x, ldx, nrhs, S.max_nsupc[l], S.d_fwd_units + u0, fr ? fr + 2 * (size_t) u0 : nullptr);
        eng::sweep_step(s, true, T, S.d_diag_units + (nd ? S.du_off[l + 1] : 0), nd, S.d_fwd_units + u1, u2 - u1, d_x, w, ldx, nrhs, mx,
                        dr ? dr + 2 * (size_t) (nd ? S.du_off[l + 1] : 0) : nullptr, fr ? fr + 2 * (size_t) u1 : nullptr);
        H->st.solve_launches += 2;
    }
    return 0;
}

// Backward links: w_k (the forward solution) minus the updates U(k, :) x accumulates in w; the final x_k = Uinv w_k goes to d_x.
static int solve_bwd_links(Handle *H, LevelSched &S, double *d_x, int64_t ldx, int nrhs)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    const int nl = S.nlevels;
    if (nl == 0) return 0;
    double *w = H->d_w;
    const int4 *br = S.d_bwd_recs, *dr = S.d_diag_recs;
    {   // far chunks of the top level (columns of ancestors in other forests, solved before this sweep)
        const int u1 = S.bu_off[2 * (nl - 1) + 1], u2 = S.bu_off[2 * (nl - 1) + 2];
        eng::sweep_step(s, false, T, nullptr, 0, S.d_bwd_units + u1, u2 - u1, d_x, w, ldx, nrhs, S.max_nsupc[nl - 1], nullptr, br ? br + 2 * (size_t) u1 : nullptr);
        H->st.solve_launches += 1;
    }
    for (int l = nl - 1; l >= 0; --l) {
        const int u0 = S.bu_off[2 * l], u1 = S.bu_off[2 * l + 1];
        const int nd = S.du_off[l + 1] - S.du_off[l];
        const int b1 = l > 0 ? S.bu_off[2 * (l - 1) + 1] : 0, b2 = l > 0 ? S.bu_off[2 * (l - 1) + 2] : 0;   // far chunks of level l-1: x of levels >= l+1 only
        const int mx = std::max(S.max_nsupc[l], l > 0 ? S.max_nsupc[l - 1] : 0);
        eng::bwd_update(s, T, nullptr, nullptr, 0, u1 - u0, d_x, w, ldx, nrhs, S.max_nsupc[l], S.d_bwd_units + u0, br ? br + 2 * (size_t) u0 : nullptr);
        eng::sweep_step(s, false, T, S.d_diag_units + S.du_off[l], nd, S.d_bwd_units + b1, b2 - b1, d_x, w, ldx, nrhs, mx,
                        dr ? dr + 2 * (size_t) S.du_off[l] : nullptr, br ? br + 2 * (size_t) b1 : nullptr);
        H->st.solve_launches += 2;
    }
    return 0;
}

// the strips of a merged group stage the group's right-hand sides in LDS (nG x nrhs doubles): wider blocks of right-hand sides take the ungrouped schedule
static inline bool groups_fit(const Handle *H, int nrhs)
{
    int w = 0;
    for (const Handle::SolveGroup &G : H->groups) w = std::max(w, G.nG);
    return (size_t) w * nrhs * sizeof(double) <= 48 * 1024;
}
static int solve_fwd_z(Handle *H, int z, double *d_x, int64_t ldx, int nrhs)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    LevelSched &S = H->sched[z];
    const bool xy = H->grid.Pr * H->grid.Pc > 1;
    if (!xy && !H->z && !H->profile) {
        int rc = ensure_w(H, ldx * (int64_t) max_rhs_chunk(H));
        if (!rc && !H->ssched.empty() && H->ssched[z].join && groups_fit(H, nrhs)) return solve_fwd_join(H, H->ssched[z], d_x, ldx, nrhs);     // merged chain groups: the contracted schedule
        if (!rc && S.join) return solve_fwd_join(H, S, d_x, ldx, nrhs);
        return rc ? rc : solve_fwd_links(H, S, d_x, ldx, nrhs);
    }
    for (int l = 0; l < S.nlevels; ++l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        int rc;
        if (H->z) {   // complex16: d_x holds doublecomplex, ldx in complex values; the exchanges see 2 ldx doubles (run lists in doubles)
            if (xy && (rc = xseg_exchange(H, d_x, 2 * ldx, nrhs, S.xs_red_send[l], 3, S.xs_red_recv[l], 2, s))) return rc;
            eng::zsolve_diag(s, true, T, S.d_nodes + n0, nn, d_x, ldx, nrhs, S.max_nsupc[l]);
            if (xy && (rc = xseg_exchange(H, d_x, 2 * ldx, nrhs, S.xs_bc_send[l], 0, S.xs_bc_recv[l], 1, s))) return rc;
            eng::zfwd_update(s, T, S.d_nodes + n0, S.d_zfwd_prefix + po, nn, S.zfwd_prefix[po + nn], d_x, ldx, nrhs, S.max_nsupc[l]);
            continue;
        }
        if (xy && (rc = xseg_exchange(H, d_x, ldx, nrhs, S.xs_red_send[l], 3, S.xs_red_recv[l], 2, s))) return rc;
        eng::solve_diag(s, true, T, S.d_nodes + n0, nn, d_x, ldx, nrhs, S.max_nsupc[l]);
        if (xy && (rc = xseg_exchange(H, d_x, ldx, nrhs, S.xs_bc_send[l], 0, S.xs_bc_recv[l], 1, s))) return rc;
        eng::fwd_update(s, T, S.d_nodes + n0, S.d_fwd_prefix + po, nn, S.fwd_prefix[po + nn], d_x, d_x, ldx, nrhs, S.max_nsupc[l]);
    }
    return 0;
}
static int solve_bwd_z(Handle *H, int z, double *d_x, int64_t ldx, int nrhs)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    LevelSched &S = H->sched[z];
    const bool xy = H->grid.Pr * H->grid.Pc > 1;
    if (!xy && !H->z && !H->profile) {
        int rc = ensure_w(H, ldx * (int64_t) max_rhs_chunk(H));      // (already there: the forward sweep ran first)
        if (!rc && !H->ssched.empty() && H->ssched[z].join && groups_fit(H, nrhs)) return solve_bwd_join(H, H->ssched[z], d_x, ldx, nrhs);
        if (!rc && S.join) return solve_bwd_join(H, S, d_x, ldx, nrhs);
        return rc ? rc : solve_bwd_links(H, S, d_x, ldx, nrhs);
    }
    for (int l = S.nlevels - 1; l >= 0; --l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        int rc;
        if (H->z) {
            eng::zbwd_update(s, T, S.d_nodes + n0, S.d_bwd_prefix + po, nn, S.bwd_prefix[po + nn], d_x, ldx, nrhs);
            if (xy && (rc = xseg_exchange(H, d_x, 2 * ldx, nrhs, S.xs_red_send[l], 3, S.xs_red_recv[l], 2, s))) return rc;
            eng::zsolve_diag(s, false, T, S.d_nodes + n0, nn, d_x, ldx, nrhs, S.max_nsupc[l]);
            if (xy && (rc = xseg_exchange(H, d_x, 2 * ldx, nrhs, S.xs_bc_send[l], 0, S.xs_bc_recv[l], 1, s))) return rc;
            continue;
        }
        eng::bwd_update(s, T, S.d_nodes + n0, S.d_bwd_prefix + po, nn, S.bwd_prefix[po + nn], d_x, d_x, ldx, nrhs, S.max_nsupc[l]);
        if (xy && (rc = xseg_exchange(H, d_x, ldx, nrhs, S.xs_red_send[l], 3, S.xs_red_recv[l], 2, s))) return rc;
        eng::solve_diag(s, false, T, S.d_nodes + n0, nn, d_x, ldx, nrhs, S.max_nsupc[l]);
        if (xy && (rc = xseg_exchange(H, d_x, ldx, nrhs, S.xs_bc_send[l], 0, S.xs_bc_recv[l], 1, s))) return rc;
    }
    return 0;
}

static int max_rhs_chunk(const Handle *H)
{   // x_k is staged in LDS by the diagonal solve / forward update: max_nsupc x nrhs values next to <= 50 KiB of static arrays
    // (k_sweep), 160 KiB per workgroup
    const int per = H->max_nsupc * (H->z ? 16 : 8);
    return std::max(1, (96 * 1024) / std::max(per, 1));
}

// complex16 on one rank, every supernode <= 64 columns: ONE launch per level and sweep (eng::zsweep_fused) instead of two -- the complex path's stand-in for the
// joined links of the double path (it keeps no inverses).  Forward: x holds the accumulated right-hand side, w receives y; backward: w accumulates, x receives the solution.
static bool zsweeps_fused(const Handle *H) { return H->z && H->grid.size() == 1 && H->max_nsupc <= 64 && H->env.z_fuse_max_nodes > 0 && !H->profile && H->d_ztickets; }
// Per level, the same form in both sweeps: fused (levels of few supernodes -- the chain at the top of the tree, where a level is launch latency) or the two in-place
// launches (levels of thousands of supernodes: every strip re-solving its diagonal block and every chunk paying an agent-scope release cost more than a launch:
// all levels fused 8.2 ms against 3.14 ms on the 1000 x 1000 configuration).  The forms meet in any order: right-hand-side accumulators always live in x; a fused
// level keeps its y_k / backward accumulators in w, an in-place level in x_k itself; final solutions are always in x.
static int zsolve_fused(Handle *H, double *x, int64_t ldx, int nr)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    auto fused = [&](const LevelSched &S, int l) { return S.lvl_off[l + 1] - S.lvl_off[l] <= H->env.z_fuse_max_nodes; };
    for (auto &S : H->sched)
        for (int l = 0; l < S.nlevels; ++l) {
            const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
            if (fused(S, l)) {
                eng::zsweep_fused(s, true, T, S.d_nodes + n0, S.d_zffu_prefix + po, nn, S.zffu_prefix[po + nn], x, H->d_w, ldx, nr, S.max_nsupc[l], nullptr);
                H->st.solve_launches += 1;
            } else {
                eng::zsolve_diag(s, true, T, S.d_nodes + n0, nn, x, ldx, nr, S.max_nsupc[l]);
                eng::zfwd_update(s, T, S.d_nodes + n0, S.d_zfwd_prefix + po, nn, S.zfwd_prefix[po + nn], x, ldx, nr, S.max_nsupc[l]);
                H->st.solve_launches += 2;
            }
        }
    for (int z = (int) H->sched.size() - 1; z >= 0; --z) {
        LevelSched &S = H->sched[z];
        for (int l = S.nlevels - 1; l >= 0; --l) {
            const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
            if (fused(S, l)) {
                eng::zsweep_fused(s, false, T, S.d_nodes + n0, S.d_zbfu_prefix + po, nn, S.zbfu_prefix[po + nn], x, H->d_w, ldx, nr, S.max_nsupc[l], H->d_ztickets);
                H->st.solve_launches += 1;
            } else {
                eng::zbwd_update(s, T, S.d_nodes + n0, S.d_bwd_prefix + po, nn, S.bwd_prefix[po + nn], x, ldx, nr);
                eng::zsolve_diag(s, false, T, S.d_nodes + n0, nn, x, ldx, nr, S.max_nsupc[l]);
                H->st.solve_launches += 2;
            }
        }
    }
    return 0;
}

int run_solve_local(Handle *H, double *d_x, int64_t ldx, int nrhs)
{
    int rc = H->z ? 0 : ensure_inv(H);
    if (rc) return rc;
    H->st.solve_launches = 0;
    const int ch = max_rhs_chunk(H);
    const int vs = H->z ? 2 : 1;
    const bool zf = zsweeps_fused(H);
    if (zf && (rc = ensure_w(H, 2 * ldx * (int64_t) ch))) return rc;
    static const bool dbg = getenv("SLUAMD_SOLVE_DEBUG") != nullptr;
    const double t_enq0 = dbg ? SetupTimer::now() : 0.0;
    struct EnqTimer { bool on; double t0; Handle *H; ~EnqTimer() { if (on) fprintf(stderr, "[sluamd solve] host enqueue of %d launches: %.3f ms\n", H->st.solve_launches, 1e3 * (SetupTimer::now() - t0)); } } enq_timer{dbg, t_enq0, H};
    for (int j0 = 0; j0 < nrhs; j0 += ch) {
        const int nr = std::min(ch, nrhs - j0);
        double *x = d_x + (size_t) j0 * ldx * vs;
        if (zf) { if ((rc = zsolve_fused(H, x, ldx, nr))) return rc; continue; }
        for (int z = 0; z < (int) H->sched.size(); ++z) if ((rc = solve_fwd_z(H, z, x, ldx, nr))) return rc;
        for (int z = (int) H->sched.size() - 1; z >= 0; --z) if ((rc = solve_bwd_z(H, z, x, ldx, nr))) return rc;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// rows of the forests of Z levels [a0, a1) whose x entries this rank holds in role `role`:
//   0 = lsum accumulators / b of the process row (k % Pr == myrow), 1 = solved x of the process column (k % Pc == mycol),
//   2 = diagonal owner only
static void forest_runs_build(const Handle *H, int a0, int a1, int role, LevelSched::XSeg &out)
{
    const HostStruct &hs = H->hs;
    const Grid &g = H->grid;
    std::vector<int> ks;
    for (int a = a0; a < a1 && a < (int) H->forest_nodes.size(); ++a)
        for (int k : H->forest_nodes[a]) {
            const bool rr = g.krow(k) == g.r, cc = g.kcol(k) == g.c;
            if ((role == 0 && rr) || (role == 1 && cc) || (role == 2 && rr && cc)) ks.push_back(k);
        }
    std::sort(ks.begin(), ks.end());
    out.runs.clear(); out.total = 0;
    const int vs = H->z ? 2 : 1;   // runs are in doubles of the right-hand side viewed as a real array (complex16: 2 per row)
    for (int k : ks) {
        const int row0 = hs.xsup[k] * vs, nr = (hs.xsup[k + 1] - hs.xsup[k]) * vs;
        if (!out.runs.empty() && out.runs.back().first + out.runs.back().second == row0) out.runs.back().second += nr;
        else out.runs.emplace_back(row0, nr);
        out.total += nr;
    }
}

// cached: the forests of a handle never change.  role 3 = diagonal owner on the levels this layer factors (where b is
// consumed and x is final)
static const LevelSched::XSeg &forest_runs(Handle *H, int a0, int role)
{
    const int key = a0 * 8 + role;
    auto it = H->xseg_cache.find(key);
    if (it != H->xseg_cache.end()) return it->second;
    LevelSched::XSeg seg;
    const int nzl = (int) H->sched.size();
    if (role == 3) {
        const Grid &g = H->grid;
        std::vector<int> ks;
        for (int zl = 0; zl < nzl; ++zl) if (H->z_active[zl]) for (int k : H->forest_nodes[zl]) if (g.krow(k) == g.r && g.kcol(k) == g.c) ks.push_back(k);
        std::sort(ks.begin(), ks.end());
        const int vs = H->z ? 2 : 1;
        for (int k : ks) {
            const int row0 = H->hs.xsup[k] * vs, n1 = (H->hs.xsup[k + 1] - H->hs.xsup[k]) * vs;
            if (!seg.runs.empty() && seg.runs.back().first + seg.runs.back().second == row0) seg.runs.back().second += n1; else seg.runs.emplace_back(row0, n1);
            seg.total += n1;
        }
    } else forest_runs_build(H, a0, nzl, role, seg);
    const LevelSched::XSeg &c = H->xseg_cache.emplace(key, std::move(seg)).first->second;
    const int *dr;
    runs_on_device(H, c, &dr);      // upload now: copies of the cached entry (with a peer filled in) share the device image
    return c;
}

// every rank's owner rows -- the rows whose x is final on that rank after the sweeps (diagonal owner, on the layer that factors
// the forest) -- learnt once per handle: all-to-all of the run lists over the host-buffer channel
static int ensure_owner_runs(Handle *H)
{
    if (H->owner_ready) return 0;
    const Grid &g = H->grid;
    const int P = g.size(), me = g.rank();
    LevelSched::XSeg mine = forest_runs(H, 0, 3);
    H->owner_runs.assign(P, LevelSched::XSeg());
    mine.peer = me;
    H->owner_runs[me] = mine;
    if (P > 1) {
        Comm *c = H->comm;
        int rc;
        std::vector<int> flat;
        for (auto &r : mine.runs) { flat.push_back(r.first); flat.push_back(r.second); }
        std::vector<int64_t> lens(P, 0);
        int64_t mylen = (int64_t) flat.size();
        if ((rc = c->hbegin())) return rc;
        for (int p = 0; p < P; ++p) if (p != me) { if ((rc = c->hsend(&mylen, 8, p))) return rc; if ((rc = c->hrecv(&lens[p], 8, p))) return rc; }
        if ((rc = c->hend())) return rc;
        std::vector<std::vector<int>> all(P);
        if ((rc = c->hbegin())) return rc;
        for (int p = 0; p < P; ++p) {
            if (p == me) continue;
            all[p].resize((size_t) lens[p]);
            if (mylen && (rc = c->hsend(flat.data(), mylen * 4, p))) return rc;
            if (lens[p] && (rc = c->hrecv(all[p].data(), lens[p] * 4, p))) return rc;
        }
        if ((rc = c->hend())) return rc;
        for (int p = 0; p < P; ++p) {
            if (p == me) continue;
            LevelSched::XSeg &m = H->owner_runs[p];
            m.peer = p;
            for (size_t i = 0; i + 1 < all[p].size(); i += 2) { m.runs.emplace_back(all[p][i], all[p][i + 1]); m.total += all[p][i + 1]; }
        }
    }
    H->owner_ready = true;
    return 0;
}

// all-gather of the solution: every rank packs its owner rows ONCE and sends them to every other rank directly, one grouped
// exchange (xGMI is a full mesh: P - 1 concurrent links per GPU) -- not a gather on rank 0 followed by P - 1 full-vector sends
static int allgather_solution(Handle *H, double *x, int64_t ldxd, int nr, hipStream_t s)
{
    int rc = ensure_owner_runs(H);
    if (rc) return rc;
    const Grid &g = H->grid;
    const int P = g.size(), me = g.rank();
    Comm *c = H->comm;
    const LevelSched::XSeg &mine = H->owner_runs[me];
    int64_t need = mine.total * nr;
    for (int p = 0; p < P; ++p) if (p != me) need += H->owner_runs[p].total * nr;
    if ((rc = ensure_xtmp(H, std::max<int64_t>(need, 1)))) return rc;
    const int *dr;
    if (mine.total) {
        if ((rc = runs_on_device(H, mine, &dr))) return rc;
        eng::xseg_copy(s, x, ldxd, nr, dr, (int) mine.runs.size(), mine.total, H->d_xtmp, 0);
    }
    std::vector<int64_t> ro(P, 0);
    int64_t off = mine.total * nr;
    if ((rc = c->begin())) return rc;
    for (int p = 0; p < P; ++p) {
        if (p == me) continue;
        if (mine.total && (rc = c->send(H->d_xtmp, mine.total * nr * 8, p))) return rc;
        const int64_t t = H->owner_runs[p].total * nr;
        ro[p] = off;
        if (t && (rc = c->recv(H->d_xtmp + off, t * 8, p))) return rc;
        off += t;
    }
    if ((rc = c->end(s))) return rc;
    for (int p = 0; p < P; ++p) {
        const LevelSched::XSeg &m = H->owner_runs[p];
        if (p == me || !m.total) continue;
        if ((rc = runs_on_device(H, m, &dr))) return rc;
        eng::xseg_copy(s, x, ldxd, nr, dr, (int) m.runs.size(), m.total, H->d_xtmp + ro[p], 1);
    }
    if (!c->stream_ordered()) HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// forward and backward sweeps of one right-hand-side chunk on the grid.  On entry x holds b at the rows this rank consumes (diagonal
// owner on the factoring layer) and zeros elsewhere; on return x_k is final at those rows.  Z sweeps as pdgsTrForwardSolve3d /
// pdgsTrBackSolve3d (pdgstrs3d.c:7312, :7564): forward reduction of the ancestor rows to the partner layer (dfsolveReduceLsum3d
// :1646), backward hand-down of the solved ancestors (dp2pSolvedX3d :1596).
static int grid_sweeps(Handle *H, double *x, int64_t ldx, int nr)
{
    const Grid &g = H->grid;
    const int nzl = (int) H->sched.size();
    hipStream_t s = H->stream;
    const int vs = H->z ? 2 : 1;
    const int64_t ldxd = ldx * vs;
    int rc;
    // ---- forward sweep, leaves to root ----
    for (int zl = 0; zl < nzl; ++zl) {
        const int step = 1 << zl;
        if (g.z % step) break;
        if (H->z_active[zl] && (rc = solve_fwd_z(H, zl, x, ldx, nr))) return rc;
        if (zl + 1 < nzl) {
            const bool receiver = (g.z % (2 * step)) == 0;
            if (receiver && g.z + step >= g.Pz) continue;
            LevelSched::XSeg seg = forest_runs(H, zl + 1, 0);        // (copy shares the cached device image)
            seg.peer = g.rank_of(g.r, g.c, receiver ? g.z + step : g.z - step);
            std::vector<LevelSched::XSeg> one(1, seg), none;
            if (seg.total && (rc = receiver ? xseg_exchange(H, x, ldxd, nr, none, 0, one, 2, s) : xseg_exchange(H, x, ldxd, nr, one, 0, none, 0, s))) return rc;
        }
    }
    // ---- backward sweep, root to leaves ----
    for (int zl = nzl - 1; zl >= 0; --zl) {
        const int step = 1 << zl;
        if (g.z % step) continue;
        if (zl + 1 < nzl) {
            const bool sender = (g.z % (2 * step)) == 0;
            if (!(sender && g.z + step >= g.Pz)) {
                LevelSched::XSeg seg = forest_runs(H, zl + 1, 1);
                seg.peer = g.rank_of(g.r, g.c, sender ? g.z + step : g.z - step);
                std::vector<LevelSched::XSeg> one(1, seg), none;
                if (seg.total && (rc = sender ? xseg_exchange(H, x, ldxd, nr, one, 0, none, 0, s) : xseg_exchange(H, x, ldxd, nr, none, 0, one, 1, s))) return rc;
            }
        }
        if (H->z_active[zl] && (rc = solve_bwd_z(H, zl, x, ldx, nr))) return rc;
    }
    return 0;
}

static int grid_solve_checks(Handle *H)
{
    const Grid &g = H->grid;
    if (!H->comm) { set_error("handle of a multi-rank grid has no communicator"); return SLUAMD_EINVAL; }
    if (!H->dinv_ready) { set_error("grid solve needs the factorisation to have run on this handle"); return SLUAMD_EINVAL; }
    return H->z ? 0 : ensure_inv(H);
}

// pdgstrs3d on the grid, replicated form: d_x holds the COMPLETE permuted right-hand side on entry (on every rank) and the
// complete solution on return.  (sluamd_pdgstrs3d_dist keeps B distributed like the reference.)
int run_solve_dev(Handle *H, double *d_x, int64_t ldx, int nrhs)
{
    const Grid &g = H->grid;
    if (g.size() == 1) return run_solve_local(H, d_x, ldx, nrhs);
    int rc = grid_solve_checks(H);
    if (rc) return rc;
    hipStream_t s = H->stream;
    const int ch = max_rhs_chunk(H);
    H->st.solve_launches = 0;     // launches of THIS solve (the grid sweeps only ever add)
    // complex16: the exchanges see the right-hand sides as real arrays of 2 n rows (run lists in doubles, leading dimension ldxd);
    // the sweeps get the complex view (ldx)
    const int vs = H->z ? 2 : 1;
    const int64_t ldxd = ldx * vs;
    for (int j0 = 0; j0 < nrhs; j0 += ch) {
        const int nr = std::min(ch, nrhs - j0);
        double *x = d_x + (size_t) j0 * ldxd;
        // keep b only where it is consumed: at the diagonal owner, on the layer that factors the forest; everything else
        // starts as a zero accumulator (rows of other layers' forests are never touched)
        {
            const LevelSched::XSeg &keep = forest_runs(H, 0, 3);
            if ((rc = ensure_xtmp(H, std::max<int64_t>(keep.total * nr, 1)))) return rc;
            const int *dr;
            if ((rc = runs_on_device(H, keep, &dr))) return rc;
            eng::xseg_copy(s, x, ldxd, nr, dr, (int) keep.runs.size(), keep.total, H->d_xtmp, 0);
            for (int q = 0; q < nr; ++q) HIPCHK(hipMemsetAsync(x + (size_t) q * ldxd, 0, sizeof(double) * (size_t) H->hs.n * vs, s));
            eng::xseg_copy(s, x, ldxd, nr, dr, (int) keep.runs.size(), keep.total, H->d_xtmp, 1);
            if (!H->comm->stream_ordered()) HIPCHK(hipStreamSynchronize(s));
        }
        if ((rc = grid_sweeps(H, x, ldx, nr))) return rc;
        // every x_k is final at its diagonal owner on the layer that factored its forest: all-gather
        if ((rc = allgather_solution(H, x, ldxd, nr, s))) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return 0;
}

// ================================================================================================
//        distributed right-hand side at the solve boundary: pdReDistribute3d_B_to_X / X_to_B
// ================================================================================================
// the plan is keyed by the permutations THEMSELVES (kept with the plan and compared element by element: a hash could collide and
// silently reuse the wrong route)
static bool same_perm(const std::vector<int> &kept, bool kept_null, const int *perm, int64_t n)
{
    if (kept_null != (perm == nullptr)) return false;
    if (!perm) return true;
    return (int64_t) kept.size() == n && std::memcmp(kept.data(), perm, sizeof(int) * (size_t) n) == 0;
}

static void free_dist_plan(Handle *H)
{
    for (void *p : H->dist.bufs) hipFree(p);
    H->dist = Handle::DistPlan();
}

// Routing of the caller's rows of B (row-block distribution over the ranks of layer 0: m_loc rows from fst_row, NRformat_loc) to
// the ranks that consume them in the sweeps (through perm_in) and of the solution back (through perm_out).  Both sides of a pair
// derive matching lists from replicated data (the permutation, every rank's range and owner rows): the rows between one pair of
// ranks travel in ascending order of the row of x.
static int build_route(Handle *H, const std::vector<int> &src_rank, const std::vector<int> &owner, int64_t m_loc, int64_t fst_row,
                       const int *perm, Handle::DistRoute &R, std::vector<void *> &bufs)
{
    const Grid &g = H->grid;
    const int P = g.size(), me = g.rank();
    const int64_t n = H->hs.n;
    std::vector<int> iperm(n, -1);
    for (int64_t i = 0; i < n; ++i) {
        const int q = perm ? perm[i] : (int) i;
        if (q < 0 || q >= n || iperm[q] >= 0) { set_error("sluamd_pdgstrs3d_dist: perm is not a permutation"); return SLUAMD_EINVAL; }
        iperm[q] = (int) i;
    }
    std::vector<std::vector<std::pair<int, int>>> bs(P);     // my rows of B: (row of x, local row), bucketed by the owner of the row of x
    for (int64_t i = 0; i < m_loc; ++i) {
        const int q = perm ? perm[fst_row + i] : (int) (fst_row + i);
        if (owner[q] < 0) { set_error("sluamd_pdgstrs3d_dist: a row of the system has no owner in this grid"); return SLUAMD_ESTRUCT; }
        bs[owner[q]].emplace_back(q, (int) i);
    }
    std::vector<std::vector<int>> xs(P);                     // my owner rows of x (ascending), bucketed by the rank that holds the paired row of B
    const int ovs = H->z ? 2 : 1;                            // (owner runs are in doubles of the right-hand side seen as a real array)
    for (auto &r : H->owner_runs[me].runs) for (int q = r.first / ovs; q < (r.first + r.second) / ovs; ++q) xs[src_rank[iperm[q]]].push_back(q);
    R.cnt_b.assign(P, 0); R.cnt_x.assign(P, 0); R.d_bidx.assign(P, nullptr); R.d_xidx.assign(P, nullptr);
    for (int p = 0; p < P; ++p) {
        std::sort(bs[p].begin(), bs[p].end());
        std::vector<int> bi(bs[p].size());
        for (size_t j = 0; j < bi.size(); ++j) bi[j] = bs[p][j].second;
        R.cnt_b[p] = (int64_t) bi.size(); R.cnt_x[p] = (int64_t) xs[p].size();
        if (!bi.empty() && upload(bufs, bi, &R.d_bidx[p])) return SLUAMD_EHIP;
        if (!xs[p].empty() && upload(bufs, xs[p], &R.d_xidx[p])) return SLUAMD_EHIP;
    }
    if (R.cnt_b[me] != R.cnt_x[me]) { set_error("sluamd_pdgstrs3d_dist: inconsistent routing tables"); return SLUAMD_ESTRUCT; }
    return 0;
}

static int build_dist_plan(Handle *H, int64_t m_loc, int64_t fst_row, const int *perm_in, const int *perm_out)
{
    const Grid &g = H->grid;
    const int P = g.size(), me = g.rank();
    const int64_t n = H->hs.n;
    int rc = ensure_owner_runs(H);
    if (rc) return rc;
    free_dist_plan(H);
    Handle::DistPlan &D = H->dist;
    // every rank's range
    std::vector<int64_t> rng(2 * (size_t) P, 0);
    rng[2 * me] = fst_row; rng[2 * me + 1] = m_loc;
    if (P > 1) {
        Comm *c = H->comm;
        if ((rc = c->hbegin())) return rc;
        for (int p = 0; p < P; ++p) if (p != me) { if ((rc = c->hsend(&rng[2 * me], 16, p))) return rc; if ((rc = c->hrecv(&rng[2 * p], 16, p))) return rc; }
        if ((rc = c->hend())) return rc;
    }
    std::vector<int> src_rank(n, -1);       // original row -> the rank that holds it in B
    int64_t covered = 0;
    for (int p = 0; p < P; ++p) {
        const int64_t f = rng[2 * p], m = rng[2 * p + 1];
        if (m < 0 || f < 0 || f + m > n) { set_error("sluamd_pdgstrs3d_dist: a rank's row range lies outside the matrix"); return SLUAMD_EINVAL; }
        for (int64_t i = f; i < f + m; ++i) { if (src_rank[i] >= 0) { set_error("sluamd_pdgstrs3d_dist: the row ranges of two ranks overlap"); return SLUAMD_EINVAL; } src_rank[i] = p; }
        covered += m;
    }
    if (covered != n) { set_error("sluamd_pdgstrs3d_dist: the ranks' row ranges do not cover the matrix"); return SLUAMD_EINVAL; }
    std::vector<int> owner(n, -1);          // row of x -> the rank where it is consumed / final
    const int ovs = H->z ? 2 : 1;
    for (int p = 0; p < P; ++p)
        for (auto &r : H->owner_runs[p].runs) for (int i = r.first / ovs; i < (r.first + r.second) / ovs; ++i) owner[i] = p;
    D.null_in = !perm_in; D.null_out = !perm_out;
    if (perm_in) D.perm_in.assign(perm_in, perm_in + n);
    if (perm_out) D.perm_out.assign(perm_out, perm_out + n);
    D.same = same_perm(D.perm_in, D.null_in, perm_out, n);
    if ((rc = build_route(H, src_rank, owner, m_loc, fst_row, perm_in, D.in, D.bufs))) return rc;
    if (!D.same && (rc = build_route(H, src_rank, owner, m_loc, fst_row, perm_out, D.out, D.bufs))) return rc;
    D.m_loc = m_loc; D.fst_row = fst_row; D.ready = true;
    return 0;
}

// one direction of the redistribution for one chunk of right-hand sides: b2x = rows of d_b (local, ld ldb) -> rows of x; else back
static int redistribute(Handle *H, bool b2x, double *d_b, int64_t ldb, double *x, int64_t ldx, int nr, hipStream_t s)
{
    const Grid &g = H->grid;
    const int P = g.size(), me = g.rank();
    const Handle::DistRoute &R = (b2x || H->dist.same) ? H->dist.in : H->dist.out;
    const std::vector<int64_t> &cs = b2x ? R.cnt_b : R.cnt_x, &cr = b2x ? R.cnt_x : R.cnt_b;
    const int vs = H->z ? 2 : 1;      // doubles per value: the staging offsets below are in doubles, ldb / ldx in values
    int64_t need = 0;
    for (int p = 0; p < P; ++p) need += (cs[p] + (p == me ? 0 : cr[p])) * nr * vs;
    int rc = ensure_xtmp(H, std::max<int64_t>(need, 1));
    if (rc) return rc;
    std::vector<int64_t> so(P), ro(P);
    int64_t off = 0;
    for (int p = 0; p < P; ++p) {     // pack
        so[p] = off; off += cs[p] * nr * vs;
        if (b2x) eng::rows_copy(s, d_b, ldb, nr, R.d_bidx[p], cs[p], H->d_xtmp + so[p], 0, vs);
        else eng::rows_copy(s, x, ldx, nr, R.d_xidx[p], cs[p], H->d_xtmp + so[p], 0, vs);
    }
    for (int p = 0; p < P; ++p) { ro[p] = (p == me) ? so[p] : off; if (p != me) off += cr[p] * nr * vs; }     // my own rows need no transport
    if (P > 1) {
        Comm *c = H->comm;
        if ((rc = c->begin())) return rc;
        for (int p = 0; p < P; ++p) {
            if (p == me) continue;
            if (cs[p] && (rc = c->send(H->d_xtmp + so[p], cs[p] * nr * 8 * vs, p))) return rc;
            if (cr[p] && (rc = c->recv(H->d_xtmp + ro[p], cr[p] * nr * 8 * vs, p))) return rc;
        }
        if ((rc = c->end(s))) return rc;
    }
    for (int p = 0; p < P; ++p) {     // unpack
        if (b2x) eng::rows_copy(s, x, ldx, nr, R.d_xidx[p], cr[p], H->d_xtmp + ro[p], 1, vs);
        else eng::rows_copy(s, d_b, ldb, nr, R.d_bidx[p], cr[p], H->d_xtmp + ro[p], 1, vs);
    }
    if (P > 1 && !H->comm->stream_ordered()) HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// pdgstrs3d with B distributed as in the reference (pdgstrs3d.c:6604-6933): the ranks of layer 0 hold m_loc consecutive rows of B
// from fst_row (ORIGINAL row order), the other layers none; perm[i] = row of the factored system that row i of B belongs to
// (perm_c[perm_r[i]], pdReDistribute3d_B_to_X :6265).  d_b: the local rows on the device (ld ldb), overwritten by the local rows
// of the solution: row i = row perm_out[i] of the solved vector (pdReDistribute3d_X_to_B :6404 returns the rows of the solution of
// the PERMUTED system, perm_out = identity; pdgssvx3d applies Pc^T afterwards).  Collective.
int run_solve_dist(Handle *H, double *d_b, int64_t ldb, int nrhs, int64_t m_loc, int64_t fst_row, const int *perm, const int *perm_out)
{
    const Grid &g = H->grid;
    int rc;
    if (g.size() > 1) { if ((rc = grid_solve_checks(H))) return rc; }
    else if (!H->z && (rc = ensure_inv(H))) return rc;
    const int vs = H->z ? 2 : 1;      // d_b, H->d_x: values of vs doubles; ldb in values
    hipStream_t s = H->stream;
    {
        // build_dist_plan is COLLECTIVE (it exchanges every rank's row range): the decision to rebuild must be too -- a rank whose own
        // range and permutations are unchanged still has to take part when a peer's partition of B changed
        int keep[1] = {(H->dist.ready && H->dist.m_loc == m_loc && H->dist.fst_row == fst_row && same_perm(H->dist.perm_in, H->dist.null_in, perm, H->hs.n) &&
                        same_perm(H->dist.perm_out, H->dist.null_out, perm_out, H->hs.n)) ? 1 : 0};
        if (g.size() > 1 && (rc = H->comm->allreduce_min(keep, 1, s))) return rc;
        if (!keep[0] && (rc = build_dist_plan(H, m_loc, fst_row, perm, perm_out))) return rc;
    }
    H->st.solve_launches = 0;
    const int64_t n = H->hs.n;
    const int ch = max_rhs_chunk(H);
    const int64_t need = n * std::min(ch, nrhs) * vs;
    if (need > H->x_cap) {
        if (H->d_x) hipFree(H->d_x);
        H->d_x = nullptr; H->x_cap = 0;
        HIPCHK(hipMalloc((void **) &H->d_x, sizeof(double) * (size_t) need));
        H->x_cap = need;
    }
    for (int j0 = 0; j0 < nrhs; j0 += ch) {
        const int nr = std::min(ch, nrhs - j0);
        double *b = d_b + (size_t) j0 * ldb * vs;
        HIPCHK(hipMemsetAsync(H->d_x, 0, sizeof(double) * (size_t) n * nr * vs, s));       // zero accumulators everywhere but the consumed rows
        if ((rc = redistribute(H, true, b, ldb, H->d_x, n, nr, s))) return rc;
        if (g.size() > 1) { if ((rc = grid_sweeps(H, H->d_x, n, nr))) return rc; }
        else {
            for (int z = 0; z < (int) H->sched.size(); ++z) if ((rc = solve_fwd_z(H, z, H->d_x, n, nr))) return rc;
            for (int z = (int) H->sched.size() - 1; z >= 0; --z) if ((rc = solve_bwd_z(H, z, H->d_x, n, nr))) return rc;
        }
        if ((rc = redistribute(H, false, b, ldb, H->d_x, n, nr, s))) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace sluamd
