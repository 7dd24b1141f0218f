// sluamd_plan.cpp -- creation-time planning: slot structures -> level schedules (elimination DAG), value-arena layout
// (own slots ordered by Z level, DAG level, supernode so that every exchange moves ONE contiguous range), device block
// tables / tile lists (dSchurComplementSetup's job, dtrfAux.c:102-491, done once), XY panel-exchange plans.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <queue>
#include <functional>
#include <cmath>
#include "sluamd_comm.h"
#include "sluamd_plan.h"

namespace sluamd {

static inline int nsupc_of(const HostStruct &hs, int k) { return hs.xsup[k + 1] - hs.xsup[k]; }

// ---- DAG levels of one forest: lvl[j] = 1 + max lvl[k] over the supernodes k of the list that update j -------------
static void dag_levels(const SlotInput &in, const std::vector<int> &list, int ns, std::vector<int> &lvl, int &nlevels)
{
    lvl.assign(ns, -1);
    for (int k : list) lvl[k] = 0;
    int maxl = 0;
    for (int k : list) {   // ascending: every predecessor of k has a smaller id
        const int l1 = lvl[k] + 1;
        for (int g : in.succ[k]) if (lvl[g] >= 0 && lvl[g] < l1) lvl[g] = l1;
        maxl = std::max(maxl, lvl[k]);
    }
    nlevels = list.empty() ? 0 : maxl + 1;
}

// XY layers: the panels a rank receives for one DAG level live in the exchange scratch until the level's tiles are done, so the
// scratch is as large as the LARGEST level -- the leaf level of a nested-dissection tree, tens of thousands of independent small
// supernodes (100^3: 9 142 of 10^5 supernodes).  The supernodes of one level are independent, so a level may be cut into
// consecutive sub-levels (ascending supernode id) without touching any dependency: levels with more than
// cap = ceil(largest level / 4) supernodes (forests whose largest level has at least 4 x 4096 [SLUAMD_LEVEL_SPLIT_MIN] of them: 150^3 and up) are exchanged in sub-batches of at most `cap` (<= 1/4 of the largest level),
// and the look-ahead schedule pipelines the sub-batches like any other levels.  The rule reads only the forest's global node lists and
// levels (identical on every rank of the layer), so all ranks cut alike.  (leaf-level remote-panel scratch in
// <= 1/4-level sub-batches; the reference keeps every received panel of its look-ahead window, dtreeFactorization.c:295-716.)
// Round 4 (second half): the largest level BY BYTES is not the leaf level at 150^3 and up but the levels of the 128 .. 512-node separators
// (few supernodes, long panels), which the count rule never cuts.  Second rule, same cut: w(k) = nsupc(k) x (nsupc(k) + sum of the widths of
// the block rows of k) -- an upper bound of the panel of k in values, read from the global block graph, so identical on every rank -- and a
// level heavier than 1 / wdiv of the FOREST's total weight (forests below `wmin` values in total are left alone) is cut into as many sub-levels
// as that takes: three scratch copies of at most total / wdiv each, whatever the size of the problem.  OPT-IN (SLUAMD_LEVEL_SPLIT_WDIV): a K-fused pair needs its
// two supernodes in consecutive levels, and cutting the separator levels takes most pairs apart -- measured at 150^3 on 2 x 2 x 2 with wdiv = 128: allocated / values
// 1.27-1.29 -> 1.17-1.24, levels 219 -> 293, fused pairs per rank 830 -> 164, planned Schur tile executions + 20 % (profiles/r04_grid_work_and_footprint.txt).
static void split_wide_levels(const SlotInput &in, const std::vector<int> &xsup, const std::vector<int> &list, std::vector<int> &lvl, int &nlevels, int min_cap,
                              int wdiv, double wmin)
{
    if (nlevels <= 0) return;
    std::vector<int> cnt(nlevels, 0);
    std::vector<double> wl(nlevels, 0.0);
    for (int k : list) {
        cnt[lvl[k]]++;
        double rows = xsup[k + 1] - xsup[k];
        for (int g : in.succ[k]) rows += xsup[g + 1] - xsup[g];
        wl[lvl[k]] += rows * (xsup[k + 1] - xsup[k]);
    }
    const int nmax = *std::max_element(cnt.begin(), cnt.end());
    double wtot = 0.0;
    for (double w : wl) wtot += w;
    const bool by_count = nmax >= 4 * min_cap;        // small forests: the scratch is small and every extra level costs two exchange phases of latency
    const bool by_weight = wdiv > 1 && wtot >= wmin;
    if (!by_count && !by_weight) return;
    const int cap = std::max(min_cap, (nmax + 3) / 4);
    const double wcap = wtot / std::max(wdiv, 1);
    std::vector<int> first(nlevels + 1, 0), per(nlevels, 1);        // new id of the first sub-level of every level; sub-level size
    for (int l = 0; l < nlevels; ++l) {
        int chunks = 1;
        if (by_count) chunks = std::max(chunks, (cnt[l] + cap - 1) / cap);
        if (by_weight) chunks = std::max(chunks, std::min(cnt[l], (int) std::ceil(wl[l] / wcap - 1e-9)));
        chunks = std::max(chunks, 1);
        per[l] = (cnt[l] + chunks - 1) / chunks;
        first[l + 1] = first[l] + (cnt[l] + std::max(per[l], 1) - 1) / std::max(per[l], 1);
    }
    if (first[nlevels] == nlevels) return;
    std::vector<int> seen(nlevels, 0);
    for (int k : list) {   // ascending supernode id
        const int l = lvl[k];
        lvl[k] = first[l] + seen[l] / std::max(per[l], 1);
        seen[l]++;
    }
    nlevels = first[nlevels];
}

// ---- block tables, tile lists, flop tallies (host images of DevTables) ---------------------------------------------
// XY layers run the panel solves in their GEMM form too: the row / column peers of a diagonal block compute its full inverses
// from the block they receive (SLUAMD_TRSM_PANELS=1: the blocked substitution of round 1)
static bool xy_gemm_panels(const Handle &H) { return H.grid.Pr * H.grid.Pc > 1 && !H.env.trsm_panels && !H.z; }

static int build_tables(Handle &H, HostTables &t)
{
    const HostStruct &hs = H.hs;
    const Grid &g = H.grid;
    const int ns = hs.nsupers;
    const bool xy_gemm = xy_gemm_panels(H);
    t.sn_lval.resize(ns); t.sn_uval.resize(ns); t.sn_lidx.resize(ns); t.sn_uidx.resize(ns); t.sn_dinv.assign(ns, 0); t.sn_dptr.assign(ns, 0); t.sn_inv.assign(ns, 0);
    t.sn_nsupr.assign(ns, 0); t.sn_flags.assign(ns, 0); t.sn_ldiag.assign(ns, 0); t.sn_dlda.assign(ns, 1); t.sn_ldu.assign(ns, 0); t.sn_ncolu.assign(ns, 0);
    t.sn_lb_off.resize(ns); t.sn_nlb.assign(ns, 0); t.sn_ub_off.resize(ns); t.sn_nub.assign(ns, 0);
    t.sn_rt_off.resize(ns); t.sn_nrt.assign(ns, 0); t.sn_ct_off.resize(ns); t.sn_nct.assign(ns, 0);
    t.ucolptr.assign(hs.uidx.size(), 0); t.unzcol.assign(hs.uidx.size(), 0);
    t.sn_big.assign(ns, 0);
    t.sn_rows_sorted.assign(ns, 0); t.sn_ucols_sorted.assign(ns, 0);
    t.sn_flops_exact.assign(ns, 0.0); t.sn_bytes_alg.assign(ns, 0.0);
    t.sn_lrow.assign(ns, 0); t.sn_ucol.assign(ns, 0);
    H.max_nsupc = 0;
    auto &st = H.st;
    st.flops_schur_padded = st.flops_schur_exact = st.flops_panel = 0;
    st.schur_bytes_alg = 0;
    st.flops_schur_exact_big = st.schur_bytes_alg_big = 0;
    // Two passes over the supernodes, each embarrassingly parallel (a supernode's table entries depend on its own two index arrays only):
    // count (and validate) -> serial prefix sums -> fill.  Pass 1 also settles the tile configuration (it needs the block sizes only).
    struct SnCnt { int nb = 0, nsupr = 0, nub = 0, ncol = 0, ldu = 0, nrt = 0, nct = 0, ldiag = 0, fl = 0; double exact = 0; uint8_t big = 0; };
    std::vector<SnCnt> cnt(ns);
    std::atomic<int> ecode{0};
    const char *emsg = nullptr;
    auto fail = [&](const char *m, int code) { int z = 0; if (ecode.compare_exchange_strong(z, code)) emsg = m; return false; };
    const int tmr_big = H.z ? 64 : 128;   // complex16: a tile is 64 / 32 panel rows (128 / 64 real rows of the embedding) x 128 / 64 columns
    auto count_one = [&](int k) -> bool {
        SnCnt &c = cnt[k];
        if (!hs.present[k]) return true;
        const int nsupc = nsupc_of(hs, k), klst = hs.xsup[k + 1];
        if (nsupc > 256) return fail("supernodes wider than 256 columns are not supported yet (set SUPERLU_MAXSUP <= 256)", SLUAMD_EINVAL);
        const bool l_own = g.kcol(k) == g.c, u_own = g.krow(k) == g.r;
        c.fl = (l_own ? SNF_L_OWN : 0) | (u_own ? SNF_U_OWN : 0) | ((l_own && u_own) ? SNF_OWN_DIAG : 0) | ((l_own || u_own) ? SNF_HAS_DIAG : 0);
        const int *li = hs.lidx.data() + hs.lidx_off[k];      // always >= BC_HEADER ints (empty slots carry {0, 0})
        const int nb = li[0], nsupr = li[1];
        c.nb = nb; c.nsupr = nsupr;
        const int ldiag = (u_own && nb > 0) ? nsupc : 0;      // the process row of k holds the diagonal block at the top of its part
        c.ldiag = ldiag;
        int p = BC_HEADER, rowoff = 0;
        long t128r = 0, t64r = 0;
        for (int b = 0; b < nb; ++b) {
            const int gid = li[p], nbrow = li[p + 1];
            if (gid < k || gid >= ns || nbrow <= 0 || rowoff + nbrow > nsupr) return fail("malformed L block", SLUAMD_ESTRUCT);
            if (g.krow(gid) != g.r) return fail("L block stored on the wrong process row", SLUAMD_ESTRUCT);
            if (b == 0 && u_own && (gid != k || nbrow != nsupc)) return fail("diagonal block must be the first L block of its panel", SLUAMD_ESTRUCT);
            if (b > 0 && gid == k) return fail("diagonal block must be the first L block of its panel", SLUAMD_ESTRUCT);
            if (b >= (ldiag ? 1 : 0)) { t128r += (nbrow + tmr_big - 1) / tmr_big; t64r += (nbrow + tmr_big / 2 - 1) / (tmr_big / 2); }
            rowoff += nbrow; p += LB_DESCRIPTOR + nbrow;
        }
        if (rowoff != nsupr) return fail("L panel row count mismatch", SLUAMD_ESTRUCT);
        if ((int64_t) nsupr * nsupc != hs.lval_len[k]) return fail("L panel value count mismatch", SLUAMD_ESTRUCT);
        long t128c = 0, t64c = 0;
        if (hs.uidx_off[k + 1] > hs.uidx_off[k]) {
            const int *ui = hs.uidx.data() + hs.uidx_off[k];
            c.nub = ui[0];
            int iukp = BR_HEADER; int64_t rukp = 0; int prev = k;
            for (int b = 0; b < c.nub; ++b) {
                const int jb = ui[iukp];
                if (jb <= prev || jb >= ns) return fail("U blocks must be sorted by block column", SLUAMD_ESTRUCT);
                if (g.kcol(jb) != g.c) return fail("U block stored on the wrong process column", SLUAMD_ESTRUCT);
                prev = jb;
                const int nsj = nsupc_of(hs, jb);
                int nc = 0;
                for (int jj = 0; jj < nsj; ++jj) {
                    const int seg = klst - ui[iukp + UB_DESCRIPTOR + jj];
                    if (seg < 0 || seg > nsupc) return fail("bad U segment", SLUAMD_ESTRUCT);
                    if (seg) { ++nc; rukp += seg; c.ldu = std::max(c.ldu, seg); c.exact += seg; }
                }
                c.ncol += nc;
                t128c += (nc + 127) / 128; t64c += (nc + 63) / 64;
                iukp += UB_DESCRIPTOR + nsj;
            }
            if (rukp != hs.uval_len[k]) return fail("U value count mismatch", SLUAMD_ESTRUCT);
        }
        const double cells = (double) (nsupr - ldiag) * c.ncol;
        const double util128 = (t128r * t128c) ? cells / ((double) t128r * t128c * tmr_big * 128.0) : 0.0;
        c.big = nsupc >= (H.z ? 48 : H.env.big_min_cols) && util128 >= 0.01 * H.env.big_util_pct && !H.env.no_big_tiles;
        c.nrt = (int) (c.big ? t128r : t64r); c.nct = (int) (c.big ? t128c : t64c);
        return true;
    };
    parallel_chunks(ns, 256, [&](int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1 && !ecode; ++k) if (!count_one((int) k)) return; });
    if (ecode) { set_error(emsg); return ecode; }
    {   // offsets, totals and the statistics (summed in supernode order: the same values whatever the thread count)
        int64_t lb = 0, ub = 0, rt = 0, ct = 0, lrow = 0, ucol = 0;
        for (int k = 0; k < ns; ++k) {
            const SnCnt &c = cnt[k];
            const int nsupc = nsupc_of(hs, k);
            t.sn_lval[k] = hs.lval_off[k]; t.sn_uval[k] = hs.uval_off[k];
            t.sn_lidx[k] = hs.lidx_off[k]; t.sn_uidx[k] = hs.uidx_off[k];
            if (std::max(std::max(lb, ub), std::max(rt, ct)) > 0x7fffffff) { set_error("block / tile tables too large for 32-bit offsets"); return SLUAMD_ESTRUCT; }
            t.sn_lb_off[k] = (int) lb; t.sn_ub_off[k] = (int) ub; t.sn_rt_off[k] = (int) rt; t.sn_ct_off[k] = (int) ct;
            t.sn_dinv[k] = t.dinv_total; t.sn_inv[k] = t.inv_total;
            t.sn_lrow[k] = lrow; t.sn_ucol[k] = ucol;
            if (!hs.present[k]) continue;
            H.max_nsupc = std::max(H.max_nsupc, nsupc);
            t.sn_flags[k] = c.fl;
            if (c.fl & SNF_HAS_DIAG) t.dinv_total += (int64_t) 2 * ((nsupc + 31) / 32) * 32 * 32;
            if (c.fl & (xy_gemm ? SNF_HAS_DIAG : SNF_OWN_DIAG)) t.inv_total += (int64_t) 2 * nsupc * nsupc;   // Linv / Uinv: the diagonal owner (solve, panel solves); on an XY
                                                                                                        // layer also the row / column peers (GEMM-form panel solves with the block they receive)
            t.sn_nsupr[k] = c.nsupr; t.sn_nlb[k] = c.nb; t.sn_ldiag[k] = c.ldiag;
            t.sn_nub[k] = c.nub; t.sn_ldu[k] = c.ldu; t.sn_ncolu[k] = c.ncol;
            t.sn_big[k] = c.big; t.sn_nrt[k] = c.nrt; t.sn_nct[k] = c.nct;
            lb += c.nb; ub += c.nub; rt += c.nrt; ct += c.nct; lrow += c.nsupr; ucol += c.ncol;
            const double rrows = c.nsupr - c.ldiag;
            st.flops_schur_padded += 2.0 * rrows * c.ldu * c.ncol;
            st.schur_bytes_alg += 16.0 * rrows * c.ncol;   // read-modify-write of every updated destination element
            st.flops_schur_exact += 2.0 * rrows * c.exact;
            t.sn_flops_exact[k] = 2.0 * rrows * c.exact; t.sn_bytes_alg[k] = 16.0 * rrows * c.ncol;      // attributed to a tile configuration once the K-fused groups are known (build_schedule)
            if (c.fl & SNF_OWN_DIAG) st.flops_panel += (2.0 / 3.0) * nsupc * (double) nsupc * nsupc;
            if (c.fl & SNF_L_OWN) st.flops_panel += (double) nsupc * nsupc * rrows;
            if (c.fl & SNF_U_OWN) st.flops_panel += (double) nsupc * c.exact;
        }
        t.lb_gid.resize(lb); t.lb_nbrow.resize(lb); t.lb_rowoff.resize(lb); t.lb_lptr.resize(lb); t.lbs_gid.resize(lb); t.lbs_idx.resize(lb);
        t.ub_gid.resize(ub); t.ub_ncols.resize(ub); t.ub_iukp.resize(ub); t.ub_stcol.resize(ub);
        t.lrow.resize(lrow); t.ucol_cp.resize(ucol); t.ucol_ld.resize(ucol); t.ucol_gc.resize(ucol);
        t.rtile.resize(rt); t.rt_info.resize(rt); t.ctile.resize(ct); t.ct_info.resize(ct);
    }
    auto fill_one = [&](int k, std::vector<std::pair<int, int>> &dir) -> bool {
        if (!hs.present[k]) return true;
        const SnCnt &c = cnt[k];
        const int nsupc = nsupc_of(hs, k), klst = hs.xsup[k + 1];
        const int *li = hs.lidx.data() + hs.lidx_off[k];
        const int nb = c.nb, nsupr = c.nsupr, nub = c.nub, ldiag = c.ldiag;
        const int lb0 = t.sn_lb_off[k], ub0 = t.sn_ub_off[k];
        int *lrow = t.lrow.data() + t.sn_lrow[k];
        int p = BC_HEADER, rowoff = 0;
        dir.clear();
        for (int b = 0; b < nb; ++b) {
            const int gid = li[p], nbrow = li[p + 1];
            t.lb_gid[lb0 + b] = gid; t.lb_nbrow[lb0 + b] = nbrow; t.lb_rowoff[lb0 + b] = rowoff; t.lb_lptr[lb0 + b] = p + LB_DESCRIPTOR;
            std::copy(li + p + LB_DESCRIPTOR, li + p + LB_DESCRIPTOR + nbrow, lrow + rowoff);
            dir.emplace_back(gid, b);
            rowoff += nbrow; p += LB_DESCRIPTOR + nbrow;
        }
        {
            bool asc = true;
            for (int i = 1; i < nsupr && asc; ++i) asc = lrow[i - 1] < lrow[i];
            t.sn_rows_sorted[k] = asc ? 1 : 0;
        }
        std::sort(dir.begin(), dir.end());
        for (int b = 0; b < nb; ++b) { t.lbs_gid[lb0 + b] = dir[b].first; t.lbs_idx[lb0 + b] = dir[b].second; }
        // U block row
        int ncol_tot = 0;
        if (nub) {
            const int *ui = hs.uidx.data() + hs.uidx_off[k];
            int *cp = t.ucolptr.data() + hs.uidx_off[k];
            int *nz = t.unzcol.data() + hs.uidx_off[k];
            int *ucp = t.ucol_cp.data() + t.sn_ucol[k], *uld = t.ucol_ld.data() + t.sn_ucol[k], *ugc = t.ucol_gc.data() + t.sn_ucol[k];
            int iukp = BR_HEADER; int64_t rukp = 0;
            for (int b = 0; b < nub; ++b) {
                const int jb = ui[iukp];
                const int nsj = nsupc_of(hs, jb);
                int nc = 0;
                for (int jj = 0; jj < nsj; ++jj) {
                    const int seg = klst - ui[iukp + UB_DESCRIPTOR + jj];
                    cp[iukp + UB_DESCRIPTOR + jj] = (int) rukp;
                    if (seg) {
                        ucp[ncol_tot + nc] = (int) rukp; uld[ncol_tot + nc] = nsupc - seg; ugc[ncol_tot + nc] = hs.xsup[jb] + jj;
                        nz[iukp + UB_DESCRIPTOR + nc] = jj; ++nc; rukp += seg;
                    }
                }
                t.ub_gid[ub0 + b] = jb; t.ub_ncols[ub0 + b] = nc; t.ub_iukp[ub0 + b] = iukp + UB_DESCRIPTOR; t.ub_stcol[ub0 + b] = ncol_tot;
                ncol_tot += nc;
                iukp += UB_DESCRIPTOR + nsj;
            }
        }
        {
            const int *gc = t.ucol_gc.data() + t.sn_ucol[k];
            bool asc = true;
            for (int i = 1; i < ncol_tot && asc; ++i) asc = gc[i - 1] < gc[i];
            t.sn_ucols_sorted[k] = asc ? 1 : 0;
        }
        {   // tile lists of supernode k in its configuration
            const int bfirst = ldiag ? 1 : 0;
            const int tm = c.big ? 128 : 64;
            const int tmr = H.z ? tm / 2 : tm;    // complex16: 64 (32) panel rows = 128 (64) real rows of the embedding
            int rt = t.sn_rt_off[k], ct = t.sn_ct_off[k];
            for (int b = bfirst; b < nb; ++b) {
                const int nbrow = t.lb_nbrow[lb0 + b], ro = t.lb_rowoff[lb0 + b];
                for (int r0 = 0; r0 < nbrow; r0 += tmr, ++rt) {
                    t.rtile[rt] = make_int4(b, r0, std::min(tmr, nbrow - r0), ro + r0);
                    const int64_t lo = t.sn_lidx[k] + t.lb_lptr[lb0 + b] + r0;
                    if (lo > 0x7fffffff) return fail("index arena too large for 32-bit tile descriptors", SLUAMD_ESTRUCT);
                    t.rt_info[rt] = make_int2(t.lb_gid[lb0 + b], (int) lo);
                }
            }
            for (int b = 0; b < nub; ++b) {
                const int nc = t.ub_ncols[ub0 + b];
                for (int c0 = 0; c0 < nc; c0 += tm, ++ct) {
                    t.ctile[ct] = make_int4(b, c0, std::min(tm, nc - c0), 0);
                    const int64_t uo = t.sn_uidx[k] + t.ub_iukp[ub0 + b];
                    if (uo > 0x7fffffff) return fail("index arena too large for 32-bit tile descriptors", SLUAMD_ESTRUCT);
                    t.ct_info[ct] = make_int4(t.ub_gid[ub0 + b], (int) uo, t.ub_stcol[ub0 + b] + c0, 0);
                }
            }
            if (rt != t.sn_rt_off[k] + c.nrt || ct != t.sn_ct_off[k] + c.nct) return fail("internal: tile counts of the two table passes differ", SLUAMD_ESTRUCT);
        }
        return true;
    };
    parallel_chunks(ns, 256, [&](int64_t k0, int64_t k1) {
        std::vector<std::pair<int, int>> dir;
        for (int64_t k = k0; k < k1 && !ecode; ++k) if (!fill_one((int) k, dir)) return;
    });
    if (ecode) { set_error(emsg); return ecode; }
    H.setup.lap("tables.blocks_and_tiles");
    // ---- merged row tiles ----
    // The update L(ib, k) U(k, jb) of every block row ib >= jb lands in ONE destination panel (jb), and the rows of those blocks are contiguous in
    // the slot of k (blocks ascending): tiles may run across the block boundaries instead of ending with a ragged tile per block -- a 144-row
    // block costs two 128-row tiles of which the second is 1/8 full and still takes half the time of a full one (the chunk period has a latency
    // floor).  Per U block the rows of the L blocks at and below it are re-cut into tiles when that saves a tile; the destination row map of such
    // a tile is a search in the destination panel's (ascending) row list instead of one block's row list.  Real arithmetic, list schedules only.
    // (two passes over the supernodes on the planner's threads, like the block tables above: count -> prefix sums -> fill; the tables come out as the serial loop wrote them)
    t.ub_mrt_off.assign(t.ub_gid.size(), 0); t.ub_mrt_cnt.assign(t.ub_gid.size(), 0);
    if (!H.env.no_merge_tiles && !H.opt.deterministic) {
        auto rows_of = [&](int k, int64_t base, bool write) -> int64_t {
            if (!hs.present[k] || !t.sn_nrt[k] || !t.sn_nct[k] || !t.sn_rows_sorted[k]) return 0;
            const int lb0 = t.sn_lb_off[k], ub0 = t.sn_ub_off[k], nb = t.sn_nlb[k], nub = t.sn_nub[k];
            const int bfirst = t.sn_ldiag[k] ? 1 : 0;
            const int tm = (t.sn_big[k] ? 128 : 64) / (H.z ? 2 : 1);      // complex16: a tile is 64 / 32 panel rows (128 / 64 real rows of the embedding)
            int64_t cnt = 0;
            for (int u = 0; u < nub; ++u) {
                const int jb = t.ub_gid[ub0 + u];
                if (!t.sn_rows_sorted[jb] || !(t.sn_flags[jb] & SNF_L_OWN)) continue;
                int bs = bfirst;
                while (bs < nb && t.lb_gid[lb0 + bs] < jb) ++bs;
                if (nb - bs < 2) continue;                                    // one block: nothing to merge
                int regular = 0;
                for (int b = bs; b < nb; ++b) regular += (t.lb_nbrow[lb0 + b] + tm - 1) / tm;
                const int row0 = t.lb_rowoff[lb0 + bs], rows = t.sn_nsupr[k] - row0;
                const int merged = (rows + tm - 1) / tm;
                if (merged >= regular) continue;
                if (write) {
                    t.ub_mrt_off[ub0 + u] = (int) (base + cnt); t.ub_mrt_cnt[ub0 + u] = merged;
                    int b = bs;
                    int64_t w = base + cnt;
                    for (int r0 = 0; r0 < rows; r0 += tm, ++w) {
                        while (b + 1 < nb && t.lb_rowoff[lb0 + b + 1] <= row0 + r0) ++b;        // block that holds the tile's first row
                        t.rtile[w] = make_int4(b, row0 + r0 - t.lb_rowoff[lb0 + b], std::min(tm, rows - r0), row0 + r0);
                        t.rt_info[w] = make_int2(t.lb_gid[lb0 + b], 0);
                    }
                }
                cnt += merged;
            }
            return cnt;
        };
        std::vector<int64_t> off(ns + 1, 0);
        parallel_chunks(ns, 64, [&](int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; ++k) off[k + 1] = rows_of((int) k, 0, false); });
        const int64_t base0 = (int64_t) t.rtile.size();
        for (int k = 0; k < ns; ++k) off[k + 1] += off[k];
        t.rtile.resize(base0 + off[ns]); t.rt_info.resize(base0 + off[ns]);
        parallel_chunks(ns, 64, [&](int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; ++k) rows_of((int) k, base0 + off[k], true); });
    }
    // ... and merged COLUMN tiles: L(ib, k) U(k, jb) of every block column jb > ib lands in ONE destination U row (ib); the non-empty columns of
    // those U blocks are contiguous in the U slot of k.  Destination column map = rank of the global column in row ib's ascending column list.
    t.lb_mct_off.assign(t.lb_gid.size(), 0); t.lb_mct_cnt.assign(t.lb_gid.size(), 0);
    if (!H.env.no_merge_tiles && !H.opt.deterministic) {
        auto cols_of = [&](int k, int64_t base, bool write) -> int64_t {
            if (!hs.present[k] || !t.sn_nrt[k] || !t.sn_nct[k] || !t.sn_ucols_sorted[k]) return 0;
            const int lb0 = t.sn_lb_off[k], ub0 = t.sn_ub_off[k], nb = t.sn_nlb[k], nub = t.sn_nub[k];
            const int bfirst = t.sn_ldiag[k] ? 1 : 0;
            const int tn = t.sn_big[k] ? 128 : 64;
            int64_t cnt = 0;
            for (int b = bfirst; b < nb; ++b) {
                const int ib = t.lb_gid[lb0 + b];
                if (!t.sn_ucols_sorted[ib] || !(t.sn_flags[ib] & SNF_U_OWN)) continue;
                int us = 0;
                while (us < nub && t.ub_gid[ub0 + us] <= ib) ++us;
                if (nub - us < 2) continue;
                int regular = 0;
                for (int u = us; u < nub; ++u) regular += (t.ub_ncols[ub0 + u] + tn - 1) / tn;
                const int col0 = t.ub_stcol[ub0 + us], cols = t.sn_ncolu[k] - col0;
                const int merged = (cols + tn - 1) / tn;
                if (merged >= regular) continue;
                if (write) {
                    t.lb_mct_off[lb0 + b] = (int) (base + cnt); t.lb_mct_cnt[lb0 + b] = merged;
                    int u = us;
                    int64_t w = base + cnt;
                    for (int c0 = 0; c0 < cols; c0 += tn, ++w) {
                        while (u + 1 < nub && t.ub_stcol[ub0 + u + 1] <= col0 + c0) ++u;        // U block that holds the tile's first column
                        t.ctile[w] = make_int4(u, col0 + c0 - t.ub_stcol[ub0 + u], std::min(tn, cols - c0), 0);
                        t.ct_info[w] = make_int4(t.ub_gid[ub0 + u], (int) (t.sn_uidx[k] + t.ub_iukp[ub0 + u]), col0 + c0, 0);
                    }
                }
                cnt += merged;
            }
            return cnt;
        };
        std::vector<int64_t> off(ns + 1, 0);
        parallel_chunks(ns, 64, [&](int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; ++k) off[k + 1] = cols_of((int) k, 0, false); });
        const int64_t base0 = (int64_t) t.ctile.size();
        for (int k = 0; k < ns; ++k) off[k + 1] += off[k];
        t.ctile.resize(base0 + off[ns]); t.ct_info.resize(base0 + off[ns]);
        parallel_chunks(ns, 64, [&](int64_t k0, int64_t k1) { for (int64_t k = k0; k < k1; ++k) cols_of((int) k, base0 + off[k], true); });
    }
    H.setup.lap("tables.merged_tiles");
    if (H.z) { st.flops_schur_padded *= 4; st.flops_schur_exact *= 4; st.flops_panel *= 4; st.schur_bytes_alg *= 2; }   // complex multiply-add = 8 flop
    H.h_nsupr = t.sn_nsupr; H.h_ldu = t.sn_ldu; H.h_ncolu = t.sn_ncolu; H.h_flags = t.sn_flags; H.h_ldiag = t.sn_ldiag;
    return 0;
}

// K-fused source a of supernode b (a < b members of one chain): b's tiles will also accumulate a's deferred update.  Needs
// every row / column of a's structure beyond b to exist in b's structure (true when b is an ancestor of a in the supernodal
// elimination tree); builds the row map (per panel row of b: row in a's panel or -1) and the column info (per non-empty U
// column of row b: value offset and leading zeros inside a's U row).  Rejects pairs whose a is much smaller than b (the
// fused tiles would multiply mostly zeros).
static bool build_pair_maps(const Handle &H, const HostTables &t, int a, int b, std::vector<int> &rowmap, std::vector<int> &colinfo)
{
    const HostStruct &hs = H.hs;
    const int sa = nsupc_of(hs, a);
    const int nsupr_b = t.sn_nsupr[b], ncolu_b = t.sn_ncolu[b];
    rowmap.assign(nsupr_b, -1);
    colinfo.assign(2 * (size_t) ncolu_b, 0);
    for (int c = 0; c < ncolu_b; ++c) colinfo[2 * c + 1] = sa;
    const int la = t.sn_lb_off[a], lb = t.sn_lb_off[b];
    int rows_a = 0, cols_a = 0;
    for (int x = t.sn_ldiag[a] ? 1 : 0; x < t.sn_nlb[a]; ++x) {     // (the diagonal block heads the slot only on the supernode's own process row)
        const int g = t.lb_gid[la + x];
        if (g <= b) continue;                      // a's updates of the chain members up to b: their urgent tiles
        int y = -1;
        for (int q = t.sn_ldiag[b] ? 1 : 0; q < t.sn_nlb[b]; ++q) if (t.lb_gid[lb + q] == g) { y = q; break; }
        if (y < 0) return false;
        const int *ra = hs.lidx.data() + hs.lidx_off[a] + t.lb_lptr[la + x], *rb = hs.lidx.data() + hs.lidx_off[b] + t.lb_lptr[lb + y];
        const int na = t.lb_nbrow[la + x], nb = t.lb_nbrow[lb + y];
        const int *walk = rb;                      // both lists ascending (every store the handle builds): one merge walk instead of a search per row
        for (int i = 0; i < na; ++i) {
            while (walk < rb + nb && *walk < ra[i]) ++walk;
            const int *f = walk;
            if (f == rb + nb || *f != ra[i]) {
                f = std::lower_bound(rb, rb + nb, ra[i]);
                if (f == rb + nb || *f != ra[i]) f = std::find(rb, rb + nb, ra[i]);     // (a caller's unsorted block: linear search)
                if (f == rb + nb) return false;
                walk = rb;                         // order broken: the walk restarts, the searches above carry the block
            }
            rowmap[t.lb_rowoff[lb + y] + (int) (f - rb)] = t.lb_rowoff[la + x] + i;
        }
        rows_a += na;
    }
    const int ua = t.sn_ub_off[a], ub = t.sn_ub_off[b];
    const int klst_a = hs.xsup[a + 1];
    for (int x = 0; x < t.sn_nub[a]; ++x) {
        const int g = t.ub_gid[ua + x];
        if (g <= b) continue;
        int y = -1;
        for (int q = 0; q < t.sn_nub[b]; ++q) if (t.ub_gid[ub + q] == g) { y = q; break; }
        if (y < 0) return false;
        const int64_t pa = hs.uidx_off[a] + t.ub_iukp[ua + x], pb = hs.uidx_off[b] + t.ub_iukp[ub + y];
        const int *ca = t.unzcol.data() + pa, *cb = t.unzcol.data() + pb;
        const int na = t.ub_ncols[ua + x], nb = t.ub_ncols[ub + y];
        const int *walk = cb;
        for (int i = 0; i < na; ++i) {
            while (walk < cb + nb && *walk < ca[i]) ++walk;                         // non-empty columns of a U block: ascending by construction (build_tables)
            const int *f = walk;
            if (f == cb + nb || *f != ca[i]) { f = std::lower_bound(cb, cb + nb, ca[i]); walk = cb; }
            if (f == cb + nb || *f != ca[i]) return false;
            const int c = t.ub_stcol[ub + y] + (int) (f - cb), jj = ca[i];
            colinfo[2 * c] = t.ucolptr[pa + jj];
            colinfo[2 * c + 1] = sa - (klst_a - hs.uidx[pa + jj]);
        }
        cols_a += na;
    }
    // the Schur loader fetches two neighbouring rows of b's tile with ONE 16-byte load from a's panel (k_schur, SLUAMD_SCHUR_FETCH3): rows that both exist
    // there must be neighbours there too -- true whenever both panels list their blocks in the same order (every store the handle builds); otherwise no pair
    for (int i = 0; i + 1 < nsupr_b; ++i)
        if (rowmap[i] >= 0 && rowmap[i + 1] >= 0 && rowmap[i + 1] != rowmap[i] + 1) return false;
    const int rows_b = nsupr_b - t.sn_ldiag[b];
    const int pct = H.env.fuse_min_pct;
    return rows_a > 0 && cols_a > 0 && 100 * (int64_t) rows_a >= (int64_t) pct * rows_b && 100 * (int64_t) cols_a >= (int64_t) pct * ncolu_b;
}

// destination block of the update L(ib, k) U(k, jb): index into the lb tables (block ib of panel jb, ib >= jb) or into the ub
// tables (block jb of U row ib); -1 when this rank has no such block (structure not closed / not an own slot)
static int dest_block(const HostTables &t, int ib, int jb)
{
    if (ib >= jb) {
        const int o = t.sn_lb_off[jb], nb = t.sn_nlb[jb];
        const int *g = t.lbs_gid.data() + o;
        const int *p = std::lower_bound(g, g + nb, ib);
        return (p != g + nb && *p == ib) ? o + t.lbs_idx[o + (int) (p - g)] : -1;
    }
    const int o = t.sn_ub_off[ib], nb = t.sn_nub[ib];
    const int *g = t.ub_gid.data() + o;
    const int *p = std::lower_bound(g, g + nb, jb);
    return (p != g + nb && *p == jb) ? o + (int) (p - g) : -1;
}

static void build_tile_lists(const HostTables &t, const std::vector<int> &lvl, const std::vector<int> &defer, LevelSched &S, bool zrows = false /* complex16: row tiles of 64 / 32 panel rows */)
{
    // every Schur tile of the schedule as (supernode, absolute row tile, absolute column tile, destination block): ONE load gives a
    // workgroup what it otherwise chases through a prefix search and six tables.  Per level and tile-size group, four parts:
    //   part 0: destination is the DIAGONAL block of a level-(l+1) supernode  -> diag_lu(l+1) may start after these alone
    //   part 1: the rest of the block row / block column of those supernodes  -> the panel solves of l+1 wait for these too
    //   part 2: destination in a panel of level l+2 (and not l+1)             -> with these done early the bulk of level l
    //           may still be running while the panels of levels l+1 AND l+2 are factored (two-level look-ahead)
    //   part 3: the bulk, supernode after supernode in bands of 8 row tiles, column-major inside a band: the ~64 tiles an XCD
    //           runs at a time form an 8 x 8 block that shares 8 L row tiles and 8 U column tiles in that XCD's L2.
    // A deferred (K-fused) supernode runs parts 0 and 1 only; its partner on the next level applies everything else.
    S.sn_level = lvl;
    S.u_off.assign(8 * S.nlevels + 1, 0);
    const bool plan_debug = getenv("SLUAMD_PLAN_DEBUG") != nullptr;
    double dbg_exact[2] = {0, 0}, dbg_exec[2] = {0, 0}, dbg_full[2] = {0, 0}, dbg_tiles[2] = {0, 0};
    struct Cand { int a, c, w; };
    // per supernode, independent of every other one: its tiles in the four parts (parallel over the planner's threads, dynamic chunks -- a leaf has a
    // handful of tiles, a supernode of the top separator tens of thousands), concatenated afterwards in schedule order
    struct NodeTiles { std::vector<int4> part[4]; };
    std::vector<NodeTiles> nt(S.nodes.size());
    parallel_chunks((int64_t) S.nodes.size(), 16, [&](int64_t i0, int64_t i1) {
        std::vector<int8_t> cflag;
        std::vector<int> dcache;
        std::vector<Cand> cand, bulk;
        for (int64_t i = i0; i < i1; ++i) {
            const int k = S.nodes[i], l = lvl[k];
            std::vector<int4> *bucket = nt[i].part;
            const int nrt = t.sn_nrt[k], nct = t.sn_nct[k];
            if (!nrt || !nct) continue;
            const bool deferred = !defer.empty() && defer[k];               // runs parts 0 and 1 only
            const int r0 = t.sn_rt_off[k], c0 = t.sn_ct_off[k], nub = t.sn_nub[k], ub0 = t.sn_ub_off[k];
            const int tmk = (t.sn_big[k] ? 128 : 64) / (zrows ? 2 : 1);
            // the tiles of k as (absolute row tile, column tile) pairs: per column tile the block pairs' own row tiles, or -- where the U block
            // has merged row tiles -- those for the block rows at and below the U block's supernode and the own tiles for the ones above it
            cand.clear();
            cflag.assign(nct, 0);   // 1: level l+1, 2: level l+2
            for (int c = 0; c < nct; ++c) { const int d = lvl[t.ct_info[c0 + c].x] - l; cflag[c] = (d == 1 || d == 2) ? d : 0; }
            auto level_flag = [&](int a) { const int d = lvl[t.rt_info[a].x] - l; return (int8_t) ((d == 1 || d == 2) ? d : 0); };
            auto col_flag = [&](int c) { if (c >= 0 && c < nct) return cflag[c]; const int d = lvl[t.ct_info[c0 + c].x] - l; return (int8_t) ((d == 1 || d == 2) ? d : 0); };
            const int lb0 = t.sn_lb_off[k];
            for (int c = 0; c < nct; ++c) {
                const int ub = ub0 + t.ctile[c0 + c].x, jb = t.ct_info[c0 + c].x;
                const int mcnt = t.ub_mrt_cnt.empty() ? 0 : t.ub_mrt_cnt[ub];
                for (int r = 0; r < nrt; ++r) {
                    const int ib = t.rt_info[r0 + r].x;
                    if (ib >= jb) { if (mcnt) continue; }                                                         // covered by the merged ROW tiles of this U block
                    else if (!t.lb_mct_cnt.empty() && t.lb_mct_cnt[lb0 + t.rtile[r0 + r].x]) continue;           // covered by the merged COLUMN tiles of this L block
                    cand.push_back({r0 + r, c, t.rtile[r0 + r].w});
                }
                for (int m = 0; m < mcnt; ++m) cand.push_back({t.ub_mrt_off[ub] + m, c, t.rtile[t.ub_mrt_off[ub] + m].w});
            }
            if (!t.lb_mct_cnt.empty())
                for (int r = 0; r < nrt; ++r) {
                    const int lb = lb0 + t.rtile[r0 + r].x;
                    for (int m = 0; m < t.lb_mct_cnt[lb]; ++m) cand.push_back({r0 + r, t.lb_mct_off[lb] + m - c0, t.rtile[r0 + r].w});     // column index relative to c0 (beyond nct)
                }
            dcache.assign((size_t) t.sn_nlb[k] * std::max(nub, 1), -2);
            auto entry = [&](const Cand &q) {
                const bool merged = q.a < r0 || q.a >= r0 + nrt;
                int d = -3;                                                   // merged rows: the destination panel is searched row by row
                if (q.c < 0 || q.c >= nct) d = -4;                            // merged columns: the destination U row is searched column by column
                else if (!merged) {
                    int &dc = dcache[(size_t) t.rtile[q.a].x * nub + t.ctile[c0 + q.c].x];
                    if (dc == -2) dc = dest_block(t, t.rt_info[q.a].x, t.ct_info[c0 + q.c].x);
                    d = dc;
                }
                return make_int4(k, q.a, c0 + q.c, d);
            };
            std::sort(cand.begin(), cand.end(), [&](const Cand &x, const Cand &y) { return x.w != y.w ? x.w < y.w : x.c < y.c; });
            bulk.clear();
            for (auto &q : cand) {
                const int8_t rf = level_flag(q.a);
                const int8_t cf = col_flag(q.c);
                if (!(rf || cf)) { if (!deferred) bulk.push_back(q); continue; }
                const bool next = rf == 1 || cf == 1;          // feeds a level-(l+1) panel
                if (!next) { if (!deferred) bucket[2].push_back(entry(q)); continue; }
                const bool diag = t.rt_info[q.a].x == t.ct_info[c0 + q.c].x;      // (holds rows of) the diagonal block of a level-(l+1) supernode
                bucket[diag ? 0 : 1].push_back(entry(q));
            }
            // the bulk: bands of 8 row tiles' worth of slot rows, column tile after column tile inside a band
            std::stable_sort(bulk.begin(), bulk.end(), [&](const Cand &x, const Cand &y) {
                const int bx = x.w / (8 * tmk), by = y.w / (8 * tmk);
                if (bx != by) return bx < by;
                if (x.c != y.c) return x.c < y.c;
                return x.w < y.w; });
            for (auto &q : bulk) bucket[3].push_back(entry(q));
        }
    });
    {
        size_t tot = 0;
        for (auto &q : nt) for (auto &v : q.part) tot += v.size();
        S.ulist.reserve(S.ulist.size() + tot);
    }
    for (int l = 0; l < S.nlevels; ++l) {
        const int nbig = S.n_big[l];
        for (int g = 0; g < 2; ++g) {
            const int b = S.lvl_off[l] + (g == 0 ? 0 : nbig), e = (g == 0) ? S.lvl_off[l] + nbig : S.lvl_off[l + 1];
            for (int part = 0; part < 4; ++part) {
                const size_t first = S.ulist.size();
                for (int i = b; i < e; ++i) S.ulist.insert(S.ulist.end(), nt[i].part[part].begin(), nt[i].part[part].end());
                S.u_off[(2 * l + g) * 4 + part + 1] = (int) S.ulist.size();
                if (plan_debug)
                    for (size_t q = first; q < S.ulist.size(); ++q) {       // executed MFMA area (in the granularity at which idle waves skip) against the useful area, weighted by the source width
                        const int4 u = S.ulist[q];
                        const int nr = t.rtile[u.y].z, nc = t.ctile[u.z].z;
                        const int ksz = (int) (t.sn_ldu[u.x]);
                        dbg_exact[g] += (double) nr * nc * ksz;
                        dbg_exec[g] += (g == 0 ? (double) ((nr + 31) / 32 * 32) * ((nc + 63) / 64 * 64) : (double) ((nr + 31) / 32 * 32) * ((nc + 31) / 32 * 32)) * ksz;
                        dbg_full[g] += (g == 0 ? 128.0 * 128.0 : 64.0 * 64.0) * ksz;
                        dbg_tiles[g] += 1;
                    }
            }
        }
    }
    if (plan_debug)
        for (int g = 0; g < 2; ++g)
            fprintf(stderr, "[sluamd_plan] tile lists, %s configuration: %.0f tiles, useful area x K %.4g, executed (wave-skip granularity) %.4g = %.3f x, full tiles %.4g = %.3f x\n",
                    g == 0 ? "128 x 128" : "64 x 64", dbg_tiles[g], dbg_exact[g], dbg_exec[g], dbg_exact[g] > 0 ? dbg_exec[g] / dbg_exact[g] : 0.0, dbg_full[g],
                    dbg_exact[g] > 0 ? dbg_full[g] / dbg_exact[g] : 0.0);
}

// Modelled cost of one Schur tile in K-chunk periods: the chunks of all its sources (the K-fused predecessors run in the same tile) x the share of its waves that run
// MFMAs (a wave whose part of a ragged tile is empty skips them: k_schur `wave_on`; the loader and the barriers stay) + a fixed part (record, prologue, scatter).
static inline double tile_cost(const Handle &H, const HostTables &t, const int4 &u, int g, double ovh)
{
    const auto &hs = H.hs;
    auto chunks_of = [&](int s) { const int nss = hs.xsup[s + 1] - hs.xsup[s]; const int kb = (nss - t.sn_ldu[s]) & ~3; return (nss - kb + KC - 1) / KC; };
    const int k = u.x, nr = t.rtile[u.y].z * (H.z ? 2 : 1), nc = t.ctile[u.z].z;
    int ch = chunks_of(k);
    if (!H.h_fuse_prev.empty())
        for (int j = 0; j < 3 && H.h_fuse_prev[3 * (size_t) k + j] >= 0; ++j) ch += chunks_of(H.h_fuse_prev[3 * (size_t) k + j]);
    const double frac = g == 0 ? ((nr + 31) / 32) * ((nc + 63) / 64) / 8.0 : ((nr + 31) / 32) * ((nc + 31) / 32) / 4.0;
    return ovh + ch * (0.35 + 0.65 * std::min(frac, 1.0));
}

// Balanced bulk launches (LevelSched::x_off).  k_schur gives every XCD one contiguous range of a launch's tile list (the 8 x 8 bands of a range share their L row
// tiles and U column tiles in that XCD's L2) and the hardware hands an XCD's workgroups to its free slots in list order.  With ranges of equal COUNT the launch
// ends with its slowest XCD: where a level mixes K-fused groups (tiles of 2-3 sources) with plain supernodes, or wide with ragged tiles, the busiest eighth
// is 4-15 % above the mean (SLUAMD_PLAN_BALANCE: 100^3, levels 11 / 14 / 19 / 29 / 39).  Here: the supernodes of a bulk list are ordered longest tiles first
// (their order inside a level is free: the scatter is atomic) and the list is cut into eight ranges of equal modelled cost; a range's tail then holds its shortest tiles.
static void balance_bulk(const Handle &H, const HostTables &t, LevelSched &S)
{
    S.x_off.assign(20 * (size_t) S.nlevels, 0);
    if (H.env.balance_min_tiles <= 0 || H.opt.deterministic) return;
    struct Seg { int a, n; double c; };
    std::vector<Seg> segs;
    std::vector<int4> tmp;
    std::vector<double> pre;
    for (int l = 0; l < S.nlevels; ++l)
        for (int g = 0; g < 2; ++g) {
            const int u0 = S.u_off[(2 * l + g) * 4 + 3], n = S.u_off[(2 * l + g) * 4 + 4] - u0;
            if (n < H.env.balance_min_tiles) continue;
            // the supernodes' runs, longest tiles first (stable: equal ones keep the schedule order)
            segs.clear();
            for (int i = 0; i < n;) {
                int j = i;
                while (j < n && S.ulist[u0 + j].x == S.ulist[u0 + i].x) ++j;
                int4 full = S.ulist[u0 + i];
                segs.push_back({i, j - i, tile_cost(H, t, full, g, 0.0)});     // (the first tile of a run: the K of the run; its ragged tiles are priced below)
                i = j;
            }
            std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.c > y.c; });
            tmp.clear(); tmp.reserve(n);
            for (const Seg &sg : segs) tmp.insert(tmp.end(), S.ulist.begin() + u0 + sg.a, S.ulist.begin() + u0 + sg.a + sg.n);
            std::copy(tmp.begin(), tmp.end(), S.ulist.begin() + u0);
            pre.assign(n + 1, 0.0);
            for (int i = 0; i < n; ++i) pre[i + 1] = pre[i] + tile_cost(H, t, S.ulist[u0 + i], g, H.env.balance_ovh);
            int *xo = S.x_off.data() + 10 * (size_t) (2 * l + g);
            xo[0] = 0; xo[8] = n;
            for (int x = 1; x < 8; ++x) {
                const double want = pre[n] * x / 8.0;
                int pos = (int) (std::lower_bound(pre.begin(), pre.end(), want) - pre.begin());
                if (pos > 0 && want - pre[pos - 1] < pre[pos] - want) --pos;         // the nearer boundary
                xo[x] = std::max(xo[x - 1], std::min(n, pos));
            }
            int mx = 0;
            for (int x = 0; x < 8; ++x) mx = std::max(mx, xo[x + 1] - xo[x]);
            xo[9] = mx;
        }
}

// SLUAMD_PLAN_BALANCE (diagnostic, plan time, CPU build too): how evenly the bulk tile lists fill the device -- per bulk launch the modelled makespan (greedy
// in-order list scheduling of every XCD's range on its slots) against the same work spread perfectly.
static void report_balance(const Handle &H, const HostTables &t, const LevelSched &S)
{
    const double ovh = H.env.balance_ovh;
    double tot_ideal[2] = {0, 0}, tot_xcd[2] = {0, 0}, tot_sim[2] = {0, 0};
    int nlaunch[2] = {0, 0}, nbal[2] = {0, 0};
    std::vector<double> cost;
    for (int l = 0; l < S.nlevels; ++l)
        for (int g = 0; g < 2; ++g) {
            const int u0 = S.u_off[(2 * l + g) * 4 + 3], u1 = S.u_off[(2 * l + g) * 4 + 4];
            const int n = u1 - u0;
            if (n <= 0) continue;
            const int slots = g == 0 ? 64 : 160;
            cost.resize(n);
            for (int i = 0; i < n; ++i) cost[i] = tile_cost(H, t, S.ulist[u0 + i], g, ovh);
            const int *xo = S.x_off.empty() ? nullptr : S.x_off.data() + 10 * (size_t) (2 * l + g);
            const bool bal = xo && xo[9] > 0;
            const int chunk = (n + 7) >> 3;
            double sum = 0, mx_x = 0, mk = 0;
            for (int x = 0; x < 8; ++x) {
                const int a = bal ? xo[x] : std::min(n, x * chunk), b = bal ? xo[x + 1] : std::min(n, (x + 1) * chunk);
                double sx = 0;
                std::priority_queue<double, std::vector<double>, std::greater<double>> q;
                for (int s2 = 0; s2 < slots; ++s2) q.push(0.0);
                double end = 0;
                for (int i = a; i < b; ++i) { sx += cost[i]; const double st = q.top(); q.pop(); q.push(st + cost[i]); end = std::max(end, st + cost[i]); }
                sum += sx; mx_x = std::max(mx_x, sx / slots); mk = std::max(mk, end);
            }
            const double ideal = sum / (8.0 * slots);
            tot_ideal[g] += ideal; tot_xcd[g] += mx_x; tot_sim[g] += mk; nlaunch[g] += 1; nbal[g] += bal;
            if (getenv("SLUAMD_PLAN_BALANCE_V"))
                fprintf(stderr, "[balance] level %3d %s%s: %6d tiles, ideal %8.1f, busiest XCD %8.1f (%.3f), in-order makespan %8.1f (%.3f)\n", l, g ? "64" : "128", bal ? " balanced" : "", n, ideal,
                        mx_x, mx_x / ideal, mk, mk / ideal);
        }
    {   // tiles and K chunks by the number of sources a tile runs (its own supernode + K-fused predecessors): every source restarts the tile's load pipeline
        double nt[2][4] = {{0}}, nch[2][4] = {{0}};
        auto chunks_of = [&](int s2) { const int nss = H.hs.xsup[s2 + 1] - H.hs.xsup[s2]; const int kb = (nss - t.sn_ldu[s2]) & ~3; return (nss - kb + KC - 1) / KC; };
        for (int l = 0; l < S.nlevels; ++l)
            for (int g = 0; g < 2; ++g)
                for (int u = S.u_off[(2 * l + g) * 4]; u < S.u_off[(2 * l + g) * 4 + 4]; ++u) {
                    const int k = S.ulist[u].x;
                    int ns2 = 0, ch = chunks_of(k);
                    if (!H.h_fuse_prev.empty())
                        for (int j = 0; j < 3 && H.h_fuse_prev[3 * (size_t) k + j] >= 0; ++j) { ++ns2; ch += chunks_of(H.h_fuse_prev[3 * (size_t) k + j]); }
                    nt[g][ns2] += 1; nch[g][ns2] += ch;
                }
        for (int g = 0; g < 2; ++g)
            fprintf(stderr, "[balance] %s tiles by sources 1/2/3/4: %.0f %.0f %.0f %.0f; their K chunks: %.0f %.0f %.0f %.0f\n", g ? "64 x 64" : "128 x 128", nt[g][0], nt[g][1], nt[g][2], nt[g][3],
                    nch[g][0], nch[g][1], nch[g][2], nch[g][3]);
    }
    for (int g = 0; g < 2; ++g)
        if (nlaunch[g])
            fprintf(stderr, "[balance] %s x %s bulk launches: %d (%d balanced), sum of ideal spans %.0f chunk periods, busiest-XCD spans %.0f (%.3f x), in-order makespans %.0f (%.3f x)\n",
                    g ? "64" : "128", g ? "64" : "128", nlaunch[g], nbal[g], tot_ideal[g], tot_xcd[g], tot_xcd[g] / tot_ideal[g], tot_sim[g], tot_sim[g] / tot_ideal[g]);
}

// Split panel solves (LevelSched::ps_units): per level the strips of L(:, k) / chunks of U(k, :) its part-0 tiles read.  Derived from the tile lists
// themselves -- merged row / column tiles run across block boundaries, so "the rows of the blocks whose supernode is in the next level" would miss rows.
// Levels that are split: 1 x 1 layers, real arithmetic, 64-high units, not the first level (its panels are solved before the loop) nor the last (no
// successor to hurry for), at most SLUAMD_PANEL_SPLIT supernodes, and at least one strip on each side of the cut.
static void build_panel_split(const Handle &H, const HostTables &t, LevelSched &S)
{
    S.ps_units.clear();
    S.ps_off.assign(4 * (size_t) S.nlevels + 1, 0);
    const bool eligible = H.grid.Pr * H.grid.Pc == 1 && !H.z && !H.opt.deterministic && H.env.panel_split_max_nodes > 0 && !H.env.trsm_panels;
    std::vector<uint8_t> lurg, uurg;
    std::vector<int> loff, uoff;     // per node of the level: first strip / chunk in lurg / uurg
    std::vector<int> idx_of(H.hs.nsupers, -1);   // position of a supernode inside its level
    for (int l = 0; l < S.nlevels; ++l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0;
        const size_t base = S.ps_units.size();
        bool split = eligible && l > 0 && l + 1 < S.nlevels && nn <= H.env.panel_split_max_nodes && trsm_rs(H, (S.max_nsupc[l] + 31) & ~31) == 64;
        if (split) {
            loff.assign(nn + 1, 0); uoff.assign(nn + 1, 0);
            for (int i = 0; i < nn; ++i) {
                const int k = S.nodes[n0 + i];
                idx_of[k] = i;
                loff[i + 1] = loff[i] + (t.sn_nsupr[k] - t.sn_ldiag[k] + 63) / 64;
                uoff[i + 1] = uoff[i] + (t.sn_ncolu[k] + 63) / 64;
            }
            lurg.assign(loff[nn], 0); uurg.assign(uoff[nn], 0);
            for (int g = 0; g < 2; ++g)
                for (int u = S.u_off[(2 * l + g) * 4]; u < S.u_off[(2 * l + g) * 4 + 1]; ++u) {      // part 0 of the level's two tile-size groups
                    const int4 e = S.ulist[u];
                    const int k = e.x;
                    const int idx = idx_of[k];
                    const int4 rt = t.rtile[e.y], ct = t.ctile[e.z];
                    const int r0 = rt.w - t.sn_ldiag[k], r1 = r0 + rt.z - 1;                          // slot rows below the diagonal block
                    for (int s = r0 / 64; s <= r1 / 64; ++s) lurg[loff[idx] + s] = 1;
                    const int c0 = t.ct_info[e.z].z, c1 = c0 + ct.z - 1;                               // ranks among the non-empty columns of U(k, :)
                    for (int c = c0 / 64; c <= c1 / 64; ++c) uurg[uoff[idx] + c] = 1;
                }
            size_t nu_l = 0, nu_u = 0;
            for (uint8_t f : lurg) nu_l += f;
            for (uint8_t f : uurg) nu_u += f;
            if (nu_l + nu_u == 0 || nu_l + nu_u == lurg.size() + uurg.size()) split = false;         // nothing to hurry for, or nothing left to defer
        }
        if (split)
            for (int part = 0; part < 2; ++part) {            // [urgent L | urgent U | rest L | rest U]
                for (int i = 0; i < nn; ++i)
                    for (int s = loff[i]; s < loff[i + 1]; ++s) if ((lurg[s] != 0) == (part == 0)) S.ps_units.push_back(make_int2(S.nodes[n0 + i], s - loff[i]));
                S.ps_off[4 * l + 2 * part + 1] = (int) S.ps_units.size();
                for (int i = 0; i < nn; ++i)
                    for (int c = uoff[i]; c < uoff[i + 1]; ++c) if ((uurg[c] != 0) == (part == 0)) S.ps_units.push_back(make_int2(S.nodes[n0 + i], c - uoff[i]));
                S.ps_off[4 * l + 2 * part + 2] = (int) S.ps_units.size();
            }
        else for (int q = 1; q <= 4; ++q) S.ps_off[4 * l + q] = (int) base;
        if (l + 1 < S.nlevels) S.ps_off[4 * (l + 1)] = S.ps_off[4 * l + 4];
    }
    if (getenv("SLUAMD_PLAN_DEBUG")) {
        int nsplit = 0; int64_t urg = 0, rest = 0;
        for (int l = 0; l < S.nlevels; ++l) if (S.ps_off[4 * l + 4] > S.ps_off[4 * l]) { ++nsplit; urg += S.ps_off[4 * l + 2] - S.ps_off[4 * l]; rest += S.ps_off[4 * l + 4] - S.ps_off[4 * l + 2]; }
        fprintf(stderr, "[sluamd_plan] split panel solves: %d of %d levels, %lld urgent / %lld other 64-high units\n", nsplit, S.nlevels, (long long) urg, (long long) rest);
    }
}

// Level schedule of one forest (lvl / nlevels from dag_levels): node order, per-level work-unit prefix arrays, urgent tile
// lists, K-fused pairs.
static void build_schedule(Handle &H, const HostTables &t, const std::vector<int> &list, const std::vector<int> &lvl, int nlevels, LevelSched &S)
{
    const HostStruct &hs = H.hs;
    const int ns = hs.nsupers;
    const bool xy_gemm = xy_gemm_panels(H);
    S.nlevels = nlevels;
    S.lvl_off.assign(S.nlevels + 1, 0);
    for (int k : list) S.lvl_off[lvl[k] + 1]++;
    for (int l = 0; l < S.nlevels; ++l) S.lvl_off[l + 1] += S.lvl_off[l];
    S.nodes.resize(list.size());
    std::vector<int> fill(S.lvl_off.begin(), S.lvl_off.end() - (S.nlevels ? 1 : 0));
    for (int pass = 1; pass >= 0; --pass)   // big-tile supernodes first inside each level
        for (int k : list) if ((int) t.sn_big[k] == pass) S.nodes[fill[lvl[k]]++] = k;
    S.n_big.assign(S.nlevels, 0);
    for (int k : list) if (t.sn_big[k]) S.n_big[lvl[k]]++;
    S.lvl_soff.assign(S.nlevels + 1, 0);
    for (int l = 0; l < S.nlevels; ++l) S.lvl_soff[l + 1] = S.lvl_soff[l] + (S.lvl_off[l + 1] - S.lvl_off[l]) + 2;
    S.lvl_poff.assign(S.nlevels + 1, 0);
    for (int l = 0; l < S.nlevels; ++l) S.lvl_poff[l + 1] = S.lvl_poff[l] + (S.lvl_off[l + 1] - S.lvl_off[l]) + 1;
    const int psz = S.lvl_poff[S.nlevels];
    S.tile_prefix.assign(S.lvl_soff[S.nlevels], 0); S.ltr_prefix.assign(psz, 0); S.utr_prefix.assign(psz, 0);
    S.fwd_prefix.assign(psz, 0); S.bwd_prefix.assign(psz, 0); S.inv_prefix.assign(psz, 0); S.zltr_prefix.assign(psz, 0);
    S.dg_prefix.assign(psz, 0); S.dg_off.assign(psz, 0);
    S.finv_prefix.assign(psz, 0); S.zfwd_prefix.assign(psz, 0);
    S.zffu_prefix.assign(psz, 0); S.zbfu_prefix.assign(psz, 0);
    S.max_nsupc.assign(S.nlevels, 0);
    for (int l = 0; l < S.nlevels; ++l)
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i)
            S.max_nsupc[l] = std::max(S.max_nsupc[l], nsupc_of(hs, S.nodes[i]));
    for (int l = 0; l < S.nlevels; ++l) {
        int po = S.lvl_poff[l];
        int so = S.lvl_soff[l];
        const int rs = trsm_rs(H, (S.max_nsupc[l] + 31) & ~31);   // strip height of this level's TRSM launch
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i, ++po, ++so) {
            const int k = S.nodes[i];
            if (i - S.lvl_off[l] == S.n_big[l]) ++so;      // start of the small group: its own prefix, from 0
            S.tile_prefix[so + 1] = S.tile_prefix[so] + t.sn_nrt[k] * t.sn_nct[k];
            const int nsupc = nsupc_of(hs, k);
            const int fl = t.sn_flags[k];
            const int rrows = t.sn_nsupr[k] - t.sn_ldiag[k];
            const int lrows = (fl & SNF_L_OWN) ? rrows : 0;                 // L TRSM / forward update: the owner of the L slot
            const int ucols = (fl & SNF_U_OWN) ? t.sn_ncolu[k] : 0;         // U TRSM / backward update: the owner of the U slot
            S.ltr_prefix[po + 1] = S.ltr_prefix[po] + (lrows + rs - 1) / rs;
            S.utr_prefix[po + 1] = S.utr_prefix[po] + (ucols + rs - 1) / rs;
            S.inv_prefix[po + 1] = S.inv_prefix[po] + ((fl & SNF_HAS_DIAG) ? 2 * ((nsupc + 31) / 32) : 0);
            S.zltr_prefix[po + 1] = S.zltr_prefix[po] + (lrows + 63) / 64;
            S.fwd_prefix[po + 1] = S.fwd_prefix[po] + (lrows + 63) / 64;
            S.zfwd_prefix[po + 1] = S.zfwd_prefix[po] + (lrows + 255) / 256;
            S.zffu_prefix[po + 1] = S.zffu_prefix[po] + ((fl & SNF_OWN_DIAG) ? std::max(1, (lrows + 255) / 256) : (lrows + 255) / 256);
            S.zbfu_prefix[po + 1] = S.zbfu_prefix[po] + ((fl & SNF_OWN_DIAG) ? std::max(1, (ucols + 63) / 64) : (ucols + 63) / 64);
            S.bwd_prefix[po + 1] = S.bwd_prefix[po] + (ucols + 63) / 64;
            S.finv_prefix[po + 1] = S.finv_prefix[po] + ((fl & (xy_gemm ? SNF_HAS_DIAG : SNF_OWN_DIAG)) ? 2 * ((nsupc + 15) / 16) : 0);    // FIS = 16-row identity strips
        }
    }
    // solve units as explicit lists, per level [urgent | bulk]: a strip of L(:,k) is urgent when one of its rows belongs to a supernode
    // of level l+1 (its diagonal solve is the next link of the chain), a chunk of U(k,:) when one of its columns does (solved just
    // before level l in the backward sweep); the bulk units touch levels >= l+2 only and share a launch with the next diagonal solves
    S.fu_off.assign(2 * S.nlevels + 1, 0); S.bu_off.assign(2 * S.nlevels + 1, 0);
    {
        std::vector<uint8_t> urg;
        for (int l = 0; l < S.nlevels; ++l)
            for (int part = 0; part < 2; ++part) {
                for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
                    const int k = S.nodes[i], fl = t.sn_flags[k];
                    const int ldiag = t.sn_ldiag[k];
                    const int nstrip = (fl & SNF_L_OWN) ? (t.sn_nsupr[k] - ldiag + 63) / 64 : 0;
                    urg.assign(nstrip, 0);
                    for (int b = 0; b < t.sn_nlb[k] && nstrip; ++b) {
                        const int bi = t.sn_lb_off[k] + b;
                        if (t.lb_gid[bi] == k || lvl[t.lb_gid[bi]] != l + 1) continue;
                        const int r0 = t.lb_rowoff[bi] - ldiag, r1 = r0 + t.lb_nbrow[bi] - 1;
                        for (int sidx = r0 / 64; sidx <= r1 / 64; ++sidx) urg[sidx] = 1;
                    }
                    for (int sidx = 0; sidx < nstrip; ++sidx) if ((urg[sidx] != 0) == (part == 0)) S.fwd_units.push_back(make_int2(k, sidx));
                    const int nchunk = (fl & SNF_U_OWN) ? (t.sn_ncolu[k] + 63) / 64 : 0;
                    urg.assign(nchunk, 0);
                    for (int b = 0; b < t.sn_nub[k] && nchunk; ++b) {
                        const int bi = t.sn_ub_off[k] + b;
                        if (!t.ub_ncols[bi] || lvl[t.ub_gid[bi]] != l + 1) continue;
                        const int c0 = t.ub_stcol[bi], c1 = c0 + t.ub_ncols[bi] - 1;
                        for (int c = c0 / 64; c <= c1 / 64; ++c) urg[c] = 1;
                    }
                    for (int c = 0; c < nchunk; ++c) if ((urg[c] != 0) == (part == 0)) S.bwd_units.push_back(make_int2(k, c));
                }
                S.fu_off[2 * l + part + 1] = (int) S.fwd_units.size();
                S.bu_off[2 * l + part + 1] = (int) S.bwd_units.size();
            }
    }
    S.diag_units.clear(); S.du_off.assign(S.nlevels + 1, 0);
    for (int l = 0; l < S.nlevels; ++l) {
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
            const int k = S.nodes[i];
            if (!(t.sn_flags[k] & SNF_OWN_DIAG)) continue;
            for (int st = 0; st < (nsupc_of(hs, k) + 63) / 64; ++st) S.diag_units.push_back(make_int2(k, st));
        }
        S.du_off[l + 1] = (int) S.diag_units.size();
    }
    H.setup.lap("sched.prefixes_sweep_units");
    // K-fused chain groups of up to four supernodes (a, a+1, a+2, a+3) in consecutive levels: every member but the last
    // runs only its urgent tiles; every member's executed tiles accumulate all earlier members' deferred updates.
    // XY layers (round 4): a deferred supernode's received panels outlive one more exchange -- three scratch copies by level modulo 3
    // (plan_and_upload); the pair maps are in slot rows / non-empty slot columns of THIS rank's parts of the two supernodes.
    S.lvl_defer.assign(S.nlevels, 0);
    if (H.h_fuse_prev.empty()) { H.h_fuse_prev.assign(3 * (size_t) ns, -1); H.h_defer.assign(ns, 0); H.h_pair_roff.assign(3 * (size_t) ns, -1); H.h_pair_coff.assign(3 * (size_t) ns, -1); }
    if (!H.env.no_fuse && !H.opt.deterministic && !H.z) {
        const int maxprev_env = H.env.fuse_max_prev;   // measured: pairs beat groups of 3-4 end to end (longer urgent tiles sit on the panel chain) ...
        // per level: the candidates (b, its predecessor a = b - 1 in the level below) are independent of each other -- only a's OWN group, formed one level
        // earlier, is read -- so their maps are built on the planner's threads and committed in schedule order
        struct PairCand { int b = 0, nsrc = 0, srcs[3] = {-1, -1, -1}; bool ok = false; std::vector<int> rowmap[3], colinfo[3]; };
        std::vector<PairCand> cands;
        for (int l = 0; l + 1 < S.nlevels; ++l) {
            // ... so groups of more than two only form where the bulk launches hide the chain: below the last `fuse_tail_guard` levels
            // (and whose level holds at least `fuse_group_min_nodes` supernodes: 100^3 -> the bottom ~20 levels, whatever the depth of the tree)
            // XY layers: pairs only -- the three scratch copies keep a deferred supernode's received panels for ONE more level
            const bool xy_layer = H.grid.Pr * H.grid.Pc > 1;
            const int maxprev = (!xy_layer && S.nlevels - (l + 1) > H.env.fuse_tail_guard && S.lvl_off[l + 2] - S.lvl_off[l + 1] >= H.env.fuse_group_min_nodes) ? maxprev_env : 1;
            cands.clear();
            for (int i = S.lvl_off[l + 1]; i < S.lvl_off[l + 2]; ++i) {
                const int b = S.nodes[i], a = b - 1;
                if (a < 0 || lvl[a] != l) continue;
                if (!H.env.fuse_small && (!t.sn_big[a] || !t.sn_big[b])) continue;      // (SLUAMD_FUSE_SMALL: pairs whose successor runs the 64 x 64 configuration too)
                PairCand c;
                c.b = b; c.srcs[0] = a; c.nsrc = 1;
                for (int j = 0; j < 3 && H.h_fuse_prev[3 * (size_t) a + j] >= 0; ++j) {
                    if (c.nsrc == maxprev) { c.nsrc = -1; break; }       // a already closes a full group: b starts a new one later
                    c.srcs[c.nsrc++] = H.h_fuse_prev[3 * (size_t) a + j];
                }
                if (c.nsrc < 0) continue;
                cands.push_back(std::move(c));
            }
            parallel_chunks((int64_t) cands.size(), 4, [&](int64_t c0, int64_t c1) {
                for (int64_t ci = c0; ci < c1; ++ci) {
                    PairCand &c = cands[ci];
                    bool ok = true;
                    for (int j = 0; j < c.nsrc && ok; ++j) ok = build_pair_maps(H, t, c.srcs[j], c.b, c.rowmap[j], c.colinfo[j]);
                    if (!ok) {   // the far members do not fit b: fall back to the plain pair when a is not fused itself
                        if (c.nsrc > 1 || !build_pair_maps(H, t, c.srcs[0], c.b, c.rowmap[0], c.colinfo[0])) continue;
                    }
                    c.ok = true;
                }
            });
            for (PairCand &c : cands) {
                if (!c.ok) continue;
                const int b = c.b, a = c.srcs[0];
                for (int j = 0; j < c.nsrc; ++j) {
                    const size_t pj = 3 * (size_t) b + j;
                    H.h_fuse_prev[pj] = c.srcs[j];
                    H.h_pair_roff[pj] = (int) H.h_pair_rowmap.size(); H.h_pair_coff[pj] = (int) (H.h_pair_colinfo.size() / 2);
                    H.h_pair_rowmap.insert(H.h_pair_rowmap.end(), c.rowmap[j].begin(), c.rowmap[j].end());
                    H.h_pair_colinfo.insert(H.h_pair_colinfo.end(), c.colinfo[j].begin(), c.colinfo[j].end());
                }
                H.h_defer[a] = 1;
                S.lvl_defer[l] = 1;
                H.fused_pairs += 1;
            }
        }
    }
    H.setup.lap("sched.fused_pair_maps");
    build_tile_lists(t, lvl, H.h_defer, S, H.z);
    H.setup.lap("sched.tile_lists");
    balance_bulk(H, t, S);
    if (getenv("SLUAMD_PLAN_BALANCE")) report_balance(H, t, S);
    build_panel_split(H, t, S);
    // by-configuration accounting: a supernode's Schur flops run in ITS tile configuration, except a deferred (K-fused) one's, whose update is applied by
    // the tiles of the first non-deferred successor of its chain (the few urgent tiles it runs itself are counted there too)
    // per level, this rank's share of the work (sluamd_plan_table: the scaling model of scripts/scale_model.py reads it)
    S.lvl_flops_schur.assign(S.nlevels, 0.0); S.lvl_flops_panel.assign(S.nlevels, 0.0);
    for (int k : list) {
        const int l = lvl[k];
        if (l < 0 || l >= S.nlevels) continue;
        const double zf = H.z ? 4.0 : 1.0, nsupc = nsupc_of(hs, k), rrows = t.sn_nsupr[k] - t.sn_ldiag[k];
        S.lvl_flops_schur[l] += zf * t.sn_flops_exact[k];
        const int fl = t.sn_flags[k];
        if (fl & SNF_OWN_DIAG) S.lvl_flops_panel[l] += zf * (2.0 / 3.0) * nsupc * nsupc * nsupc;
        if (fl & SNF_L_OWN) S.lvl_flops_panel[l] += zf * nsupc * nsupc * rrows;
        if ((fl & SNF_U_OWN) && rrows > 0) S.lvl_flops_panel[l] += zf * nsupc * (t.sn_flops_exact[k] / (2.0 * rrows));
    }
    for (int k : list) {
        int ex = k;
        while (ex + 1 < ns && !H.h_defer.empty() && H.h_defer[ex]) ++ex;
        if (t.sn_big[ex]) { H.st.flops_schur_exact_big += (H.z ? 4.0 : 1.0) * t.sn_flops_exact[k]; H.st.schur_bytes_alg_big += (H.z ? 2.0 : 1.0) * t.sn_bytes_alg[k]; }
    }
}

// Joined sweeps (LevelSched::join): near flags, joined diagonal units, regular unit records per level.  Returns false (nothing kept) when a panel's rows
// inside one block are not ascending -- the rows of a column block of the target would not be one range.
// Merged chain groups (Handle::groups; S is then the contracted schedule Handle::ssched): a group is ONE node of its level, solved by the strips of the GROUP's
// inverse (up to 1024 columns) in the two-launch form -- joined units would read every source panel once per strip of the group (16 x); rows / columns of a member's
// panel / U row that belong to a later member of the same group are nobody's to update (the group's inverse couples the members): flagged 2 ("dead"), skipped by the
// regular units of either form (record bit 16 = skip every flagged entry, bit 17 = skip the dead ones only).
static bool build_join(Handle &H, LevelSched &S, const HostTables &t)
{
    const HostStruct &hs = H.hs;
    const std::vector<int> &lev = S.sn_level;
    const int nl = S.nlevels;
    auto lohi = [](int64_t v, int &lo, int &hi) { lo = (int) (uint32_t) v; hi = (int) (v >> 32); };
    auto grp = [&](int k) { return H.grp_of.empty() ? -1 : H.grp_of[k]; };
    struct Src { int k, bi; };
    std::vector<std::vector<Src>> srcs(hs.nsupers);
    std::vector<int> near_off(hs.nsupers, 0), near_cnt(hs.nsupers, 0);      // per supernode: its near columns in S.jb_aux
    // near flags + the sources of every supernode
    for (int k : S.nodes) {
        const int l = lev[k], fl = t.sn_flags[k], gk = grp(k);
        if (fl & SNF_L_OWN)
            for (int b = 0; b < t.sn_nlb[k]; ++b) {
                const int bi = t.sn_lb_off[k] + b, g = t.lb_gid[bi];
                if (g == k) continue;
                const bool dead = gk >= 0 && grp(g) == gk;
                if (!dead && lev[g] != l + 1) continue;
                const int64_t r0 = t.sn_lrow[k] + t.lb_rowoff[bi];
                for (int r = 0; r < t.lb_nbrow[bi]; ++r) {
                    if (r && t.lrow[r0 + r] <= t.lrow[r0 + r - 1]) return false;
                    H.h_lrow_near[r0 + r] = dead ? 2 : 1;
                }
                if (!dead) srcs[g].push_back({k, bi});
            }
        if (fl & SNF_U_OWN)
            for (int b = 0; b < t.sn_nub[k]; ++b) {
                const int bi = t.sn_ub_off[k] + b, g = t.ub_gid[bi];
                if (!t.ub_ncols[bi]) continue;
                const bool dead = gk >= 0 && grp(g) == gk;
                if (!dead && lev[g] != l + 1) continue;
                for (int c = 0; c < t.ub_ncols[bi]; ++c) H.h_ucol_near[t.sn_ucol[k] + t.ub_stcol[bi] + c] = dead ? 2 : 1;
            }
    }
    S.jf_off.assign(nl + 1, 0); S.jb_off.assign(nl + 1, 0); S.jfu_off.assign(nl + 1, 0); S.jbu_off.assign(nl + 1, 0);
    for (int l = 0; l < nl; ++l) {
        // regular units of every supernode of the level (members of groups included), and the near-column lists of the backward joined units
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
            const int k = S.nodes[i], fl = t.sn_flags[k];
            const int fst = hs.xsup[k], ns = hs.xsup[k + 1] - fst;
            if (fl & SNF_L_OWN) {
                const int ldiag = t.sn_ldiag[k], lda = t.sn_nsupr[k];
                for (int row0 = ldiag; row0 < lda; row0 += 64) {
                    int nnear = 0;
                    const int nr = std::min(64, lda - row0);
                    for (int r = 0; r < nr; ++r) nnear += H.h_lrow_near[t.sn_lrow[k] + row0 + r] != 0;
                    if (nnear == nr) continue;
                    int4 b; lohi(t.sn_lval[k] + row0, b.x, b.y); lohi(t.sn_lrow[k] + row0, b.z, b.w);
                    S.jfu_recs.push_back(make_int4(fst, ns | (nnear ? 1 << 16 : 0), lda, row0)); S.jfu_recs.push_back(b);
                }
            }
            if (fl & SNF_U_OWN) {
                const int ncolu = t.sn_ncolu[k];
                for (int c0 = 0; c0 < ncolu; c0 += 64) {
                    int nnear = 0;
                    const int nc = std::min(64, ncolu - c0);
                    for (int c = 0; c < nc; ++c) nnear += H.h_ucol_near[t.sn_ucol[k] + c0 + c] != 0;
                    if (nnear == nc) continue;
                    int4 b; lohi(t.sn_ucol[k] + c0, b.x, b.y); lohi(t.sn_uval[k], b.z, b.w);
                    S.jbu_recs.push_back(make_int4(fst, ns | (nnear ? 1 << 16 : 0), nc, 0)); S.jbu_recs.push_back(b);
                }
                near_off[k] = (int) S.jb_aux.size();
                for (int c = 0; c < ncolu; ++c) {
                    const int64_t ci = t.sn_ucol[k] + c;
                    if (H.h_ucol_near[ci] == 1) S.jb_aux.push_back(make_int4(t.ucol_ld[ci], t.ucol_cp[ci], t.ucol_gc[ci], 0));
                }
                near_cnt[k] = (int) S.jb_aux.size() - near_off[k];
            }
        }
        // joined units: one node = one supernode, or one merged group (emitted at its first member; all members are in this level)
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
            const int k = S.nodes[i];
            if (!(t.sn_flags[k] & SNF_OWN_DIAG)) continue;
            const int gk = grp(k);
            int nm = 1, mem[4] = {k, 0, 0, 0}, mo[4] = {0, 0, 0, 0}, mw[4] = {hs.xsup[k + 1] - hs.xsup[k], 0, 0, 0};
            int64_t linv = t.sn_inv[k], uinv = t.sn_inv[k] + (int64_t) mw[0] * mw[0];
            if (gk >= 0) continue;                                // a level that holds a merged group keeps the two-launch form (level_joined)
            const int fst = hs.xsup[k], ns = mo[nm - 1] + mw[nm - 1], nb = (ns + 63) / 64;
            auto member_of_block = [&](int c) { int m = 0; while (m + 1 < nm && mo[m + 1] <= 64 * c) ++m; return m; };
            // forward: block (s, c), c <= s, sources = the panels of the previous level with rows in column block c (none at level 0)
            for (int c = 0; c < nb; ++c) {
                const int m = member_of_block(c);
                std::vector<int4> sv;
                for (const Src &q : srcs[mem[m]]) {
                    const int64_t r0 = t.sn_lrow[q.k] + t.lb_rowoff[q.bi];
                    const int nbr = t.lb_nbrow[q.bi];
                    const int *rows = t.lrow.data() + r0;
                    const int a = (int) (std::lower_bound(rows, rows + nbr, fst + 64 * c) - rows), e = (int) (std::lower_bound(rows, rows + nbr, fst + 64 * c + 64) - rows);
                    if (e <= a) continue;
                    const int fk = hs.xsup[q.k];
                    int4 v; lohi(t.sn_lval[q.k] + t.lb_rowoff[q.bi] + a, v.x, v.y); lohi(r0 + a, v.z, v.w);
                    sv.push_back(make_int4(fk, hs.xsup[q.k + 1] - fk, t.sn_nsupr[q.k], e - a)); sv.push_back(v);
                }
                const int nsrc = (int) sv.size() / 2;
                int ovf = 0;
                if (nsrc > 3) { ovf = (int) S.jf_aux.size() / 2; S.jf_aux.insert(S.jf_aux.end(), sv.begin() + 6, sv.end()); }
                for (int st = c; st < nb; ++st) {
                    int4 b; lohi(linv, b.x, b.y); b.z = nsrc; b.w = ovf;
                    S.jf_recs.push_back(make_int4(fst, ns, st, c)); S.jf_recs.push_back(b);
                    for (int q = 0; q < 6; ++q) S.jf_recs.push_back(q < (int) sv.size() ? sv[q] : make_int4(0, 0, 0, 0));
                }
            }
            // backward: block (s, c), s <= c; near columns = every column of the next level in the U row of the member that holds block c
            for (int c = 0; c < nb; ++c) {
                const int m = member_of_block(c), km = mem[m];
                for (int st = 0; st <= c; ++st) {
                    int4 b; lohi(uinv, b.x, b.y); b.z = near_off[km]; b.w = near_cnt[km];
                    int4 u; lohi(t.sn_uval[km], u.x, u.y); u.z = mo[m]; u.w = mw[m];
                    S.jb_recs.push_back(make_int4(fst, ns, st, c)); S.jb_recs.push_back(b); S.jb_recs.push_back(u); S.jb_recs.push_back(make_int4(0, 0, 0, 0));
                }
            }
        }
        S.jf_off[l + 1] = (int) S.jf_recs.size() / 8; S.jb_off[l + 1] = (int) S.jb_recs.size() / 4;
        S.jfu_off[l + 1] = (int) S.jfu_recs.size() / 2; S.jbu_off[l + 1] = (int) S.jbu_recs.size() / 2;
    }
    return true;
}

// Merged chain groups: detection, storage of their inverses behind the per-supernode pairs of T.inv, the dense products that build them (eng::gemm_batched
// descriptors + tiles per stage) and the solve levels of the DAG with every group contracted to one node.
//   lower:  X_ij = -Linv_ii sum_{k = j .. i-1} L_ik X_kj   (i > j; by distance d = i - j)        upper:  X_ij = -Uinv_ii sum_{k = i+1 .. j} U_ik X_kj   (i < j)
// with the block rows of L_G / U_G gathered into dense images (eng::grp_gather): T_ij = LG[rows of i, columns of j .. i-1] LinvG[rows of j .. i-1, columns of j] is ONE product.
static void build_solve_groups(Handle &H, const SlotInput &in, HostTables &t, const std::vector<int> &list, const std::vector<int> &lvl, const LevelSched &S,
                               std::vector<int> &slev, int &nslev, std::vector<GemmDesc> &descs, std::vector<int4> &tiles)
{
    const HostStruct &hs = H.hs;
    const int ns = hs.nsupers;
    H.grp_of.assign(ns, -1);
    std::vector<int> parent(ns, -1), nchild(ns, 0);
    for (int k : list) {
        int p = -1;
        for (int g : in.succ[k]) if (g > k && (p < 0 || g < p)) p = g;
        parent[k] = p;
        if (p >= 0) nchild[p]++;
    }
    auto level_nodes = [&](int l) { return S.lvl_off[l + 1] - S.lvl_off[l]; };
    const int cap = H.env.solve_group_level_nodes;
    for (size_t ii = 0; ii < list.size(); ++ii) {
        const int k0 = list[ii];
        if (H.grp_of[k0] >= 0 || !(t.sn_flags[k0] & SNF_OWN_DIAG) || level_nodes(lvl[k0]) > cap) continue;
        Handle::SolveGroup G;
        int cur = k0, off = 0;
        auto width = [&](int k) { return hs.xsup[k + 1] - hs.xsup[k]; };
        if (width(k0) % 16) continue;
        G.k[0] = k0; G.o[0] = 0; G.w[0] = width(k0); G.nm = 1; off = width(k0);
        while (G.nm < 4) {
            const int nx = cur + 1;
            if (nx >= ns || parent[cur] != nx || nchild[nx] != 1 || lvl[nx] != lvl[cur] + 1 || level_nodes(lvl[nx]) > cap || H.grp_of[nx] >= 0) break;
            if (width(cur) % 64 || width(nx) % 16 || off + width(nx) > 1024 || !(t.sn_flags[nx] & SNF_OWN_DIAG)) break;
            G.k[G.nm] = nx; G.o[G.nm] = off; G.w[G.nm] = width(nx); off += width(nx); ++G.nm; cur = nx;
        }
        if (G.nm < 2) continue;
        G.nG = off;
        G.ginv = t.inv_total; t.inv_total += (int64_t) 2 * G.nG * G.nG;
        G.last_level = lvl[G.k[G.nm - 1]];
        const int gi = (int) H.groups.size();
        for (int m = 0; m < G.nm; ++m) H.grp_of[G.k[m]] = gi;
        // the products, stage by stage: 2 (d - 1) = sums T, 2 (d - 1) + 1 = X = -inv T, both triangles in one launch
        const int nG = G.nG;
        const int64_t LinvG = G.ginv, UinvG = G.ginv + (int64_t) nG * nG, LGs = 0, UGs = GRP_SCR, TLs = 2 * GRP_SCR, TUs = 3 * GRP_SCR;
        auto at = [&](int64_t base, int r, int c) { return base + r + (int64_t) c * nG; };
        auto add = [&](const GemmDesc &d) {
            const int di = (int) descs.size();
            descs.push_back(d);
            for (int tc = 0; tc < (d.N + 31) / 32; ++tc) for (int tr = 0; tr < (d.M + 31) / 32; ++tr) tiles.push_back(make_int4(di, tr, tc, 0));      // 32 x 32 tiles (eng::gemm_batched)
        };
        G.tile_off[0] = (int) tiles.size();
        for (int d = 1; d < G.nm; ++d) {
            for (int i = d; i < G.nm; ++i) {          // lower sums
                const int j = i - d;
                add(GemmDesc{at(LGs, G.o[i], G.o[j]), at(LinvG, G.o[j], G.o[j]), at(TLs, G.o[i], G.o[j]), nG, nG, nG, G.w[i], G.w[j], G.o[i] - G.o[j], 1, 0, 1, 0});
            }
            for (int i = 0; i + d < G.nm; ++i) {      // upper sums
                const int j = i + d;
                add(GemmDesc{at(UGs, G.o[i], G.o[i + 1]), at(UinvG, G.o[i + 1], G.o[j]), at(TUs, G.o[i], G.o[j]), nG, nG, nG, G.w[i], G.w[j], G.o[j] + G.w[j] - G.o[i + 1], 1, 0, 1, 0});
            }
            G.tile_off[2 * d - 1] = (int) tiles.size();
            for (int i = d; i < G.nm; ++i) {
                const int j = i - d;
                add(GemmDesc{at(LinvG, G.o[i], G.o[i]), at(TLs, G.o[i], G.o[j]), at(LinvG, G.o[i], G.o[j]), nG, nG, nG, G.w[i], G.w[j], G.w[i], 0, 1, 0, 1});
            }
            for (int i = 0; i + d < G.nm; ++i) {
                const int j = i + d;
                add(GemmDesc{at(UinvG, G.o[i], G.o[i]), at(TUs, G.o[i], G.o[j]), at(UinvG, G.o[i], G.o[j]), nG, nG, nG, G.w[i], G.w[j], G.w[i], 0, 1, 0, 1});
            }
            G.tile_off[2 * d] = (int) tiles.size();
        }
        for (int q = 2 * G.nm - 1; q < 7; ++q) G.tile_off[q] = (int) tiles.size();
        H.groups.push_back(G);
    }
    H.lvl_groups.assign(S.nlevels, {});
    for (size_t gi = 0; gi < H.groups.size(); ++gi) H.lvl_groups[H.groups[gi].last_level].push_back((int) gi);
    // solve levels: longest path in the DAG with every group contracted (members are consecutive supernodes: every external predecessor of a member precedes the first)
    std::vector<int> nlev_of(ns, -1);
    auto node = [&](int k) { return H.grp_of[k] >= 0 ? H.groups[H.grp_of[k]].k[0] : k; };
    for (int k : list) nlev_of[k] = 0;
    int maxl = 0;
    for (int k : list) {
        const int nk = node(k), l1 = nlev_of[nk] + 1;
        for (int g : in.succ[k]) { if (nlev_of[g] < 0) continue; const int ng = node(g); if (ng != nk && nlev_of[ng] < l1) nlev_of[ng] = l1; }
        maxl = std::max(maxl, nlev_of[nk]);
    }
    slev.assign(ns, -1);
    for (int k : list) slev[k] = nlev_of[node(k)];
    nslev = list.empty() ? 0 : maxl + 1;
}

// The schedule of the sweeps on the contracted DAG: the fields of LevelSched the solve drivers and upload_schedule read (node lists, unit lists of the two-launch
// form, diagonal strips); the factorisation keeps its own schedule.
static void build_solve_sched(Handle &H, const HostTables &t, const std::vector<int> &list, const std::vector<int> &slev, int nslev, LevelSched &S)
{
    const HostStruct &hs = H.hs;
    S.nlevels = nslev;
    S.sn_level = slev;
    S.lvl_off.assign(nslev + 1, 0);
    for (int k : list) S.lvl_off[slev[k] + 1]++;
    for (int l = 0; l < nslev; ++l) S.lvl_off[l + 1] += S.lvl_off[l];
    S.nodes.resize(list.size());
    std::vector<int> fill(S.lvl_off.begin(), S.lvl_off.end() - (nslev ? 1 : 0));
    for (int k : list) S.nodes[fill[slev[k]]++] = k;
    S.n_big.assign(nslev, 0);
    S.lvl_soff.assign(nslev + 1, 0); S.lvl_poff.assign(nslev + 1, 0);
    for (int l = 0; l < nslev; ++l) { S.lvl_soff[l + 1] = S.lvl_soff[l] + (S.lvl_off[l + 1] - S.lvl_off[l]) + 2; S.lvl_poff[l + 1] = S.lvl_poff[l] + (S.lvl_off[l + 1] - S.lvl_off[l]) + 1; }
    S.max_nsupc.assign(nslev, 0);
    for (int l = 0; l < nslev; ++l)
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
            const int k = S.nodes[i];
            const int w = H.grp_of[k] >= 0 ? H.groups[H.grp_of[k]].nG : nsupc_of(hs, k);      // the strips of a group stage the group's whole right-hand side
            S.max_nsupc[l] = std::max(S.max_nsupc[l], w);
        }
    S.fu_off.assign(2 * nslev + 1, 0); S.bu_off.assign(2 * nslev + 1, 0);
    std::vector<uint8_t> urg;
    for (int l = 0; l < nslev; ++l)
        for (int part = 0; part < 2; ++part) {
            for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
                const int k = S.nodes[i], fl = t.sn_flags[k];
                const int ldiag = t.sn_ldiag[k];
                const int nstrip = (fl & SNF_L_OWN) ? (t.sn_nsupr[k] - ldiag + 63) / 64 : 0;
                urg.assign(nstrip, 0);
                for (int b = 0; b < t.sn_nlb[k] && nstrip; ++b) {
                    const int bi = t.sn_lb_off[k] + b;
                    if (t.lb_gid[bi] == k || slev[t.lb_gid[bi]] != l + 1) continue;
                    const int r0 = t.lb_rowoff[bi] - ldiag, r1 = r0 + t.lb_nbrow[bi] - 1;
                    for (int sidx = r0 / 64; sidx <= r1 / 64; ++sidx) urg[sidx] = 1;
                }
                for (int sidx = 0; sidx < nstrip; ++sidx) if ((urg[sidx] != 0) == (part == 0)) S.fwd_units.push_back(make_int2(k, sidx));
                const int nchunk = (fl & SNF_U_OWN) ? (t.sn_ncolu[k] + 63) / 64 : 0;
                urg.assign(nchunk, 0);
                for (int b = 0; b < t.sn_nub[k] && nchunk; ++b) {
                    const int bi = t.sn_ub_off[k] + b;
                    if (!t.ub_ncols[bi] || slev[t.ub_gid[bi]] != l + 1) continue;
                    const int c0 = t.ub_stcol[bi], c1 = c0 + t.ub_ncols[bi] - 1;
                    for (int c = c0 / 64; c <= c1 / 64; ++c) urg[c] = 1;
                }
                for (int c = 0; c < nchunk; ++c) if ((urg[c] != 0) == (part == 0)) S.bwd_units.push_back(make_int2(k, c));
            }
            S.fu_off[2 * l + part + 1] = (int) S.fwd_units.size();
            S.bu_off[2 * l + part + 1] = (int) S.bwd_units.size();
        }
    S.diag_units.clear(); S.du_off.assign(nslev + 1, 0);
    for (int l = 0; l < nslev; ++l) {
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i) {
            const int k = S.nodes[i];
            if (!(t.sn_flags[k] & SNF_OWN_DIAG)) continue;
            int w = nsupc_of(hs, k);
            if (H.grp_of[k] >= 0) { const Handle::SolveGroup &G = H.groups[H.grp_of[k]]; if (G.k[0] != k) continue; w = G.nG; }      // the group's strips, at its first member
            for (int st = 0; st < (w + 63) / 64; ++st) S.diag_units.push_back(make_int2(k, st));
        }
        S.du_off[l + 1] = (int) S.diag_units.size();
    }
    S.lvl_has_group.assign(nslev, 0);
    for (int k : list) if (H.grp_of[k] >= 0) S.lvl_has_group[slev[k]] = 1;
    S.lvl_defer.assign(nslev, 0);
    S.u_off.assign(8 * nslev + 1, 0);
    S.ps_off.assign(4 * (size_t) nslev + 1, 0);
    const int psz = S.lvl_poff[nslev];
    for (auto *v : {&S.tile_prefix}) v->assign(S.lvl_soff[nslev], 0);
    for (auto *v : {&S.ltr_prefix, &S.utr_prefix, &S.fwd_prefix, &S.bwd_prefix, &S.inv_prefix, &S.zltr_prefix, &S.dg_prefix, &S.finv_prefix, &S.zfwd_prefix, &S.zffu_prefix, &S.zbfu_prefix})
        v->assign(psz, 0);
    S.dg_off.assign(psz, 0);
}

static int upload_schedule(Handle &H, LevelSched &S, const HostTables &t)
{
    if (upload(H.d_misc, S.nodes, &S.d_nodes)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.tile_prefix, &S.d_tile_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.ltr_prefix, &S.d_ltr_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.utr_prefix, &S.d_utr_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.fwd_prefix, &S.d_fwd_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.bwd_prefix, &S.d_bwd_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.inv_prefix, &S.d_inv_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.zltr_prefix, &S.d_zltr_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.sn_level, &S.d_sn_level)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.ulist, &S.d_ulist)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.x_off, &S.d_x_off)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.dg_prefix, &S.d_dg_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.zfwd_prefix, &S.d_zfwd_prefix)) return SLUAMD_EHIP;
    if (H.z && (upload(H.d_misc, S.zffu_prefix, &S.d_zffu_prefix) || upload(H.d_misc, S.zbfu_prefix, &S.d_zbfu_prefix))) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.finv_prefix, &S.d_finv_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.dg_off, &S.d_dg_off)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.fwd_units, &S.d_fwd_units)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.bwd_units, &S.d_bwd_units)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.diag_units, &S.d_diag_units)) return SLUAMD_EHIP;
    if (!S.ps_units.empty() && upload(H.d_misc, S.ps_units, &S.d_ps_units)) return SLUAMD_EHIP;
    H.setup.lap("upload.schedule_lists");
    if (!H.z) {
        // unit records (k_sweep / k_fwd_update / k_bwd_update): the scalars of every unit in the order of the unit lists
        const HostStruct &hs = H.hs;
        auto lohi = [](int64_t v, int &lo, int &hi) { lo = (int) (uint32_t) v; hi = (int) (v >> 32); };
        S.fwd_recs.resize(2 * S.fwd_units.size()); S.bwd_recs.resize(2 * S.bwd_units.size()); S.diag_recs.resize(2 * S.diag_units.size());
        const bool grouped = !S.lvl_has_group.empty() && !H.grp_of.empty() && !H.h_lrow_near.empty();
        auto grp = [&](int k) { return grouped ? H.grp_of[k] : -1; };
        if (grouped)        // the dead rows / columns of the members (see build_join), before the records that announce them
            for (int k : S.nodes) {
                const int gk = grp(k);
                if (gk < 0) continue;
                if (t.sn_flags[k] & SNF_L_OWN)
                    for (int b = 0; b < t.sn_nlb[k]; ++b) {
                        const int bi = t.sn_lb_off[k] + b, g = t.lb_gid[bi];
                        if (g != k && grp(g) == gk) for (int r = 0; r < t.lb_nbrow[bi]; ++r) H.h_lrow_near[t.sn_lrow[k] + t.lb_rowoff[bi] + r] = 2;
                    }
                if (t.sn_flags[k] & SNF_U_OWN)
                    for (int b = 0; b < t.sn_nub[k]; ++b) {
                        const int bi = t.sn_ub_off[k] + b;
                        if (t.ub_ncols[bi] && grp(t.ub_gid[bi]) == gk) for (int c = 0; c < t.ub_ncols[bi]; ++c) H.h_ucol_near[t.sn_ucol[k] + t.ub_stcol[bi] + c] = 2;
                    }
            }
        for (size_t u = 0; u < S.fwd_units.size(); ++u) {
            const int k = S.fwd_units[u].x, strip = S.fwd_units[u].y;
            const int fst = hs.xsup[k], ns = hs.xsup[k + 1] - fst, row0 = t.sn_ldiag[k] + strip * 64;
            int4 b; lohi(t.sn_lval[k] + row0, b.x, b.y); lohi(t.sn_lrow[k] + row0, b.z, b.w);
            int dead = 0;
            if (grp(k) >= 0) for (int r = row0; r < std::min(row0 + 64, t.sn_nsupr[k]); ++r) dead |= H.h_lrow_near[t.sn_lrow[k] + r] == 2;
            S.fwd_recs[2 * u] = make_int4(fst, ns | (dead ? 2 << 16 : 0), t.sn_nsupr[k], row0); S.fwd_recs[2 * u + 1] = b;
        }
        for (size_t u = 0; u < S.bwd_units.size(); ++u) {
            const int k = S.bwd_units[u].x, chunk = S.bwd_units[u].y;
            const int fst = hs.xsup[k], ns = hs.xsup[k + 1] - fst, nc = std::min(64, t.sn_ncolu[k] - chunk * 64);
            int4 b; lohi(t.sn_ucol[k] + (int64_t) chunk * 64, b.x, b.y); lohi(t.sn_uval[k], b.z, b.w);
            int dead = 0;
            if (grp(k) >= 0) for (int c = 0; c < nc; ++c) dead |= H.h_ucol_near[t.sn_ucol[k] + (int64_t) chunk * 64 + c] == 2;
            S.bwd_recs[2 * u] = make_int4(fst, ns | (dead ? 2 << 16 : 0), nc, 0); S.bwd_recs[2 * u + 1] = b;
        }
        for (size_t u = 0; u < S.diag_units.size(); ++u) {
            const int k = S.diag_units[u].x, strip = S.diag_units[u].y;
            const int fst = hs.xsup[k];
            int ns = hs.xsup[k + 1] - fst;
            int64_t li = t.sn_inv[k];
            if (grp(k) >= 0) { const Handle::SolveGroup &G = H.groups[grp(k)]; ns = G.nG; li = G.ginv; }
            int4 b; lohi(li, b.x, b.y); lohi(li + (int64_t) ns * ns, b.z, b.w);
            S.diag_recs[2 * u] = make_int4(fst, ns, strip, 0); S.diag_recs[2 * u + 1] = b;
        }
        if (upload(H.d_misc, S.fwd_recs, &S.d_fwd_recs) || upload(H.d_misc, S.bwd_recs, &S.d_bwd_recs) || upload(H.d_misc, S.diag_recs, &S.d_diag_recs)) return SLUAMD_EHIP;
        std::vector<int4>().swap(S.fwd_recs); std::vector<int4>().swap(S.bwd_recs); std::vector<int4>().swap(S.diag_recs);
        H.setup.lap("upload.sweep_unit_records");
        S.join = false;
        if (H.grid.Pr * H.grid.Pc == 1 && H.env.solve_join && !H.h_lrow_near.empty() && !S.no_join) {
            const std::vector<uint8_t> keep_l = H.h_lrow_near, keep_u = H.h_ucol_near;
            if (build_join(H, S, t)) {
                S.join = true;
                if (upload(H.d_misc, S.jf_recs, &S.d_jf_recs) || upload(H.d_misc, S.jf_aux, &S.d_jf_aux) || upload(H.d_misc, S.jb_recs, &S.d_jb_recs) ||
                    upload(H.d_misc, S.jb_aux, &S.d_jb_aux) || upload(H.d_misc, S.jfu_recs, &S.d_jfu_recs) || upload(H.d_misc, S.jbu_recs, &S.d_jbu_recs)) return SLUAMD_EHIP;
            } else { H.h_lrow_near = keep_l; H.h_ucol_near = keep_u; }
            for (auto *v : {&S.jf_recs, &S.jf_aux, &S.jb_recs, &S.jb_aux, &S.jfu_recs, &S.jbu_recs}) std::vector<int4>().swap(*v);
            H.setup.lap("upload.joined_sweep_tables");
        }
    }
    return 0;
}

static void add_runs(std::vector<std::pair<int, int>> &runs, int64_t &total, int row0, int nrows)
{
    if (!runs.empty() && runs.back().first + runs.back().second == row0) runs.back().second += nrows;
    else runs.emplace_back(row0, nrows);
    total += nrows;
}

int plan_and_upload(Handle *H, SlotInput &in, HostTables &t)
{
    HostStruct &hs = H->hs;
    const Grid &g = H->grid;
    const int ns = hs.nsupers;
    const bool xy = g.Pr * g.Pc > 1;
    const size_t upload_mark = upload_bytes();

    const int nz = (int) in.lists.size();
    H->Pz = g.Pz; H->myz = g.z;
    H->forest_nodes = in.lists;
    H->z_active = in.z_active;

    // ---- 1. index arenas (slot images) ----
    hs.lidx_off.assign(ns + 1, 0); hs.uidx_off.assign(ns + 1, 0);
    for (int k = 0; k < ns; ++k) {
        hs.lidx_off[k + 1] = hs.lidx_off[k] + (hs.present[k] ? std::max<int64_t>((int64_t) in.lidx[k].size(), BC_HEADER) : 0);
        hs.uidx_off[k + 1] = hs.uidx_off[k] + (hs.present[k] ? (int64_t) in.uidx[k].size() : 0);
    }
    hs.lidx.assign(hs.lidx_off[ns], 0); hs.uidx.assign(hs.uidx_off[ns], 0);
    hs.lval_len.assign(ns, 0); hs.uval_len.assign(ns, 0);
    {
        std::atomic<int> bad{0};      // 1: L index array, 2: U index array
        parallel_chunks(ns, 64, [&](int64_t k0, int64_t k1) {
            for (int k = (int) k0; k < (int) k1; ++k) {
                if (!hs.present[k]) continue;
                if (!in.lidx[k].empty()) {
                    if ((int64_t) in.lidx[k].size() != BC_HEADER + (int64_t) in.lidx[k][0] * LB_DESCRIPTOR + in.lidx[k][1]) { bad = 1; continue; }
                    std::copy(in.lidx[k].begin(), in.lidx[k].end(), hs.lidx.begin() + hs.lidx_off[k]);
                    hs.lval_len[k] = (int64_t) in.lidx[k][1] * nsupc_of(hs, k);
                }
                if (!in.uidx[k].empty()) {
                    if (in.uidx[k].size() < (size_t) BR_HEADER || in.uidx[k][2] != (int) in.uidx[k].size()) { bad = 2; continue; }
                    std::copy(in.uidx[k].begin(), in.uidx[k].end(), hs.uidx.begin() + hs.uidx_off[k]);
                    hs.uval_len[k] = in.uidx[k][1];
                }
                in.lidx[k] = std::vector<int>(); in.uidx[k] = std::vector<int>();
            }
        });
        if (bad == 1) { set_error("L index array length mismatch"); return SLUAMD_ESTRUCT; }
        if (bad == 2) { set_error("U index array length mismatch"); return SLUAMD_ESTRUCT; }
    }

    H->setup.lap("index_arenas");
    // ---- 2. DAG levels per Z level ----
    std::vector<std::vector<int>> lvl(nz);
    std::vector<int> nlev(nz, 0);
    for (int zl = 0; zl < nz; ++zl) {
        dag_levels(in, in.lists[zl], ns, lvl[zl], nlev[zl]);
        if (xy && !H->env.no_level_split)
            split_wide_levels(in, hs.xsup, in.lists[zl], lvl[zl], nlev[zl], std::max(1, H->env.level_split_min), H->env.level_split_wdiv, H->env.level_split_wmin);
    }

    // ---- 3. value arena layout ----
    // own slots: [L: zl 0 (level 0 | level 1 | ...) | zl 1 ... ][U: same order] -> one contiguous range per (zl, level) and
    // per ancestor forest; then, on XY layers, two parity copies of the per-level scratch for received slots / diagonal blocks
    hs.lval_off.assign(ns + 1, 0); hs.uval_off.assign(ns + 1, 0);
    std::vector<std::vector<std::vector<int>>> lev_nodes(nz);   // [zl][level] ascending supernodes
    for (int zl = 0; zl < nz; ++zl) {
        lev_nodes[zl].assign(std::max(nlev[zl], 1), {});
        for (int k : in.lists[zl]) lev_nodes[zl][std::max(lvl[zl][k], 0)].push_back(k);
    }
    int64_t cur = 0;
    H->own_l_order.clear(); H->own_u_order.clear();
    for (int pass = 0; pass < 2; ++pass) {
        for (int zl = 0; zl < nz; ++zl)
            for (auto &nodes : lev_nodes[zl])
                for (int k : nodes) {
                    if (pass == 0 && g.kcol(k) == g.c) { hs.lval_off[k] = cur; cur += hs.lval_len[k]; H->own_l_order.push_back(k); }
                    if (pass == 1 && g.krow(k) == g.r) { hs.uval_off[k] = cur; cur += hs.uval_len[k]; H->own_u_order.push_back(k); }
                }
        if (pass == 0) hs.nnzL = cur; else hs.nnzU = cur - hs.nnzL;
    }
    H->own_len = cur;
    // scratch sizes: remote L + remote U slots of the level in flight, in NB copies by level modulo NB, each as large as the largest level
    // of ITS class (round 3: two copies of the overall maximum); diagonal blocks (own staging + received), two copies by level parity.
    // NB = 3 where K-fused pairs may form on the layer: the received panels of a deferred supernode of level l are read again by its
    // partner's tiles at level l + 1, so they must outlive the exchange of level l + 2 -- with three copies the exchange of level m
    // overwrites level m - 3, whose last readers are the tiles of level m - 2: exactly what panel(m) waits for already.
    const bool xy_fuse = xy && !H->env.no_fuse && !H->opt.deterministic && !H->z;
    const int NB = xy_fuse ? 3 : 2;
    H->xy_scratch_copies = xy ? NB : 0;
    int64_t rmaxc[3] = {0, 0, 0}, dmax = 0;
    if (xy)
        for (int zl = 0; zl < nz; ++zl) {
            if (!in.z_active[zl]) continue;
            for (auto &nodes : lev_nodes[zl]) {
                int64_t rr = 0, dd = 0;
                for (int k : nodes) {
                    const bool l_own = g.kcol(k) == g.c, u_own = g.krow(k) == g.r;
                    if (!l_own) rr += hs.lval_len[k];
                    if (!u_own) rr += hs.uval_len[k];
                    if (l_own || u_own) dd += (int64_t) nsupc_of(hs, k) * nsupc_of(hs, k);
                }
                const int cls = (int) (&nodes - &lev_nodes[zl][0]) % NB;
                rmaxc[cls] = std::max(rmaxc[cls], rr); dmax = std::max(dmax, dd);
                if (getenv("SLUAMD_PLAN_DEBUG") && atoi(getenv("SLUAMD_PLAN_DEBUG")) > 1)
                    fprintf(stderr, "[sluamd plan] zl %d level %d nodes %zu received %.3f GB\n", zl, (int) (&nodes - &lev_nodes[zl][0]), nodes.size(), 8.0 * rr / 1e9);
            }
        }
    const int64_t rtot = rmaxc[0] + rmaxc[1] + rmaxc[2];
    const int64_t rbase[3] = {cur, cur + rmaxc[0], cur + rmaxc[0] + rmaxc[1]};
    const int64_t dbase[2] = {cur + rtot, cur + rtot + dmax};
    H->arena_len = cur + rtot + 2 * dmax;
    H->xy_scratch_len = rtot;
    // The value arena is allocated and zero-filled by a helper thread from here on, beside the rest of the planning: on a freshly leased device the first
    // large hipMalloc + fill costs 0.4-0.6 s for 17 GB (the driver hands out cleared pages; later processes on the same box pay 0.02 s), which is as long
    // as all the host-side planning below.  Joined before the first upload (step 7), and by the guard on every error return.
    const size_t esz = H->z ? 16 : 8;
    const size_t arena_bytes = esz * (size_t) std::max<int64_t>(H->arena_len, 1);
    std::atomic<int> arena_rc{0};
    struct Joiner { std::thread th; ~Joiner() { if (th.joinable()) th.join(); } } arena_job;
    arena_job.th = std::thread([H, arena_bytes, &arena_rc] {
        if (hipSetDevice(H->device) != hipSuccess) { arena_rc = 3; return; }
        // single-rank handles: from the pool of physical chunks (a later handle of the process re-maps what an earlier one released); grids: plain hipMalloc
        const bool pooled = H->grid.size() == 1 && !H->comm;
        if (pooled ? devpool_alloc((void **) &H->d_val, arena_bytes, H->device) != 0 : hipMalloc((void **) &H->d_val, arena_bytes) != hipSuccess) {
            (void) hipGetLastError();
            devpool_trim(H->device);          // what the pool holds may be what is missing
            if (hipMalloc((void **) &H->d_val, arena_bytes) != hipSuccess) { H->d_val = nullptr; arena_rc = 1; return; }
        }
        if (hipMemset(H->d_val, 0, arena_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) arena_rc = 2;
    });

    // ---- 4. per-level exchange plan + offsets of the remote slots / scratch diagonal blocks ----
    H->sched.assign(nz, LevelSched());
    std::vector<int64_t> dptr(ns, 0);
    std::vector<int> dlda(ns, 1);
    for (int k = 0; k < ns; ++k) { dptr[k] = hs.lval_off[k]; dlda[k] = 1; }
    struct LevelX { std::vector<XMsg> ds, dr, ps, pr; int64_t dstage = 0; std::vector<LevelSched::XSeg> rs, rr, bs, br; };
    std::vector<std::vector<LevelX>> lx(nz);
    for (int zl = 0; zl < nz; ++zl) {
        if (!in.z_active[zl]) continue;
        lx[zl].resize(nlev[zl]);
        for (int l = 0; l < nlev[zl]; ++l) {
            const auto &nodes = lev_nodes[zl][l];
            LevelX &X = lx[zl][l];
            if (!xy) continue;
            const int par = l & 1;
            // -- phase 1: diagonal blocks (dDiagFactIBCast): owner -> its process column and its process row
            int64_t doff = dbase[par];
            X.dstage = doff;
            int64_t own_d = 0;
            for (int k : nodes) if (g.kcol(k) == g.c && g.krow(k) == g.r) own_d += (int64_t) nsupc_of(hs, k) * nsupc_of(hs, k);
            if (own_d) {
                for (int r2 = 0; r2 < g.Pr; ++r2) if (r2 != g.r) X.ds.push_back({g.rank_of(r2, g.c, g.z), doff, own_d});
                for (int c2 = 0; c2 < g.Pc; ++c2) if (c2 != g.c) X.ds.push_back({g.rank_of(g.r, c2, g.z), doff, own_d});
            }
            doff += own_d;
            for (int r2 = 0; r2 < g.Pr; ++r2) {   // from the owners in my process column
                if (r2 == g.r) continue;
                int64_t len = 0;
                for (int k : nodes) if (g.kcol(k) == g.c && g.krow(k) == r2) { const int s = nsupc_of(hs, k); dptr[k] = doff + len; dlda[k] = s; len += (int64_t) s * s; }
                if (len) X.dr.push_back({g.rank_of(r2, g.c, g.z), doff, len});
                doff += len;
            }
            for (int c2 = 0; c2 < g.Pc; ++c2) {   // from the owners in my process row
                if (c2 == g.c) continue;
                int64_t len = 0;
                for (int k : nodes) if (g.krow(k) == g.r && g.kcol(k) == c2) { const int s = nsupc_of(hs, k); dptr[k] = doff + len; dlda[k] = s; len += (int64_t) s * s; }
                if (len) X.dr.push_back({g.rank_of(g.r, c2, g.z), doff, len});
                doff += len;
            }
            // -- phase 2: L slots along the process row, U slots down the process column (dIBcast_LPanel / dIBcast_UPanel)
            int64_t lfirst = -1, llen = 0, ufirst = -1, ulen = 0;
            for (int k : nodes) {
                if (g.kcol(k) == g.c) { if (lfirst < 0) lfirst = hs.lval_off[k]; llen += hs.lval_len[k]; }
                if (g.krow(k) == g.r) { if (ufirst < 0) ufirst = hs.uval_off[k]; ulen += hs.uval_len[k]; }
            }
            if (llen) for (int c2 = 0; c2 < g.Pc; ++c2) if (c2 != g.c) X.ps.push_back({g.rank_of(g.r, c2, g.z), lfirst, llen});
            if (ulen) for (int r2 = 0; r2 < g.Pr; ++r2) if (r2 != g.r) X.ps.push_back({g.rank_of(r2, g.c, g.z), ufirst, ulen});
            int64_t roff = rbase[l % NB];
            for (int c2 = 0; c2 < g.Pc; ++c2) {
                if (c2 == g.c) continue;
                int64_t len = 0;
                for (int k : nodes) if (g.kcol(k) == c2) { hs.lval_off[k] = roff + len; len += hs.lval_len[k]; }
                if (len) X.pr.push_back({g.rank_of(g.r, c2, g.z), roff, len});
                roff += len;
            }
            for (int r2 = 0; r2 < g.Pr; ++r2) {
                if (r2 == g.r) continue;
                int64_t len = 0;
                for (int k : nodes) if (g.krow(k) == r2) { hs.uval_off[k] = roff + len; len += hs.uval_len[k]; }
                if (len) X.pr.push_back({g.rank_of(r2, g.c, g.z), roff, len});
                roff += len;
            }
            // -- solve: lsum of x_k reduced along process row k % Pr to the diagonal owner; x_k broadcast down column k % Pc
            // (run lists in DOUBLES of the right-hand side seen as a real array: complex16 rows count twice)
            const int xvs = H->z ? 2 : 1;
            for (int c2 = 0; c2 < g.Pc; ++c2) {
                if (c2 == g.c) continue;
                LevelSched::XSeg snd, rcv; snd.peer = rcv.peer = g.rank_of(g.r, c2, g.z);
                for (int k : nodes) {
                    if (g.krow(k) != g.r) continue;
                    if (g.kcol(k) == c2) add_runs(snd.runs, snd.total, hs.xsup[k] * xvs, nsupc_of(hs, k) * xvs);
                    if (g.kcol(k) == g.c) add_runs(rcv.runs, rcv.total, hs.xsup[k] * xvs, nsupc_of(hs, k) * xvs);
                }
                if (snd.total) X.rs.push_back(std::move(snd));
                if (rcv.total) X.rr.push_back(std::move(rcv));
            }
            for (int r2 = 0; r2 < g.Pr; ++r2) {
                if (r2 == g.r) continue;
                LevelSched::XSeg snd, rcv; snd.peer = rcv.peer = g.rank_of(r2, g.c, g.z);
                for (int k : nodes) {
                    if (g.kcol(k) != g.c) continue;
                    if (g.krow(k) == g.r) add_runs(snd.runs, snd.total, hs.xsup[k] * xvs, nsupc_of(hs, k) * xvs);
                    if (g.krow(k) == r2) add_runs(rcv.runs, rcv.total, hs.xsup[k] * xvs, nsupc_of(hs, k) * xvs);
                }
                if (snd.total) X.bs.push_back(std::move(snd));
                if (rcv.total) X.br.push_back(std::move(rcv));
            }
        }
    }

    H->setup.lap("levels_layout_exchange_plan");
    // ---- 5. block tables, tiles ----
    int rc = build_tables(*H, t);
    if (rc) return rc;
    H->setup.lap("tables.rest");
    if (xy_gemm_panels(*H)) {
        // Linv / Uinv stores.  The diagonal OWNER keeps its pair for the triangular solves; the row / column peers of a diagonal block need
        // theirs only between full_inv(l) and panel_gemm(l) of the block's own level -- two consecutive launches on the panel stream -- so all
        // peer pairs of a level share ONE scratch region behind the owners' (round 3 kept every peer's pair for the lifetime of the handle:
        // 1.0 GB beside 2.2 GB of factor values on the off-diagonal ranks of a 2 x 2 x 2 grid at 100^3, profiles/r04_grid_footprint.txt)
        int64_t tot = 0, pmax = 0;
        for (int k = 0; k < ns; ++k)
            if (hs.present[k] && (t.sn_flags[k] & SNF_OWN_DIAG)) { t.sn_inv[k] = tot; tot += (int64_t) 2 * nsupc_of(hs, k) * nsupc_of(hs, k); }
        for (int zl = 0; zl < nz; ++zl)
            for (auto &nodes : lev_nodes[zl]) {
                int64_t off = 0;
                for (int k : nodes)
                    if (hs.present[k] && (t.sn_flags[k] & SNF_HAS_DIAG) && !(t.sn_flags[k] & SNF_OWN_DIAG)) { t.sn_inv[k] = tot + off; off += (int64_t) 2 * nsupc_of(hs, k) * nsupc_of(hs, k); }
                pmax = std::max(pmax, off);
            }
        t.inv_total = tot + pmax;
    }
    for (int k = 0; k < ns; ++k) {
        if (!hs.present[k]) continue;
        // the owner factors the diagonal block in place at the top of its own L slot; its column / row peers read the image
        // received in exchange phase 1 (the L slot image of a row peer arrives only in phase 2, after its U TRSM)
        if (t.sn_flags[k] & SNF_OWN_DIAG) { t.sn_dptr[k] = hs.lval_off[k]; t.sn_dlda[k] = std::max(t.sn_nsupr[k], 1); }
        else { t.sn_dptr[k] = dptr[k]; t.sn_dlda[k] = dlda[k]; }
    }
    H->h_dptr = t.sn_dptr;

    // The block / tile tables are final here: a helper thread uploads them (pageable copies, ~4 GB/s) while this one builds the schedules.  Nothing below
    // writes the vectors it reads or touches H->d_misc / H->T until the join in step 7.
    int table_rc = 0;
    size_t table_bytes = 0;
    struct TJoiner { std::thread th; ~TJoiner() { if (th.joinable()) th.join(); } } table_job;
    static const bool sync_tables = getenv("SLUAMD_SYNC_TABLE_UPLOAD") != nullptr;      // development: the upload at its old place, on this thread
    auto table_upload = [H, &t, &hs, &table_rc, &table_bytes] {
        table_rc = [&]() -> int {
            HIPCHK(hipSetDevice(H->device));
            auto &K = H->d_misc;
            DevTables &T = H->T;
            const size_t mark = upload_bytes();
            if (upload(K, hs.lidx, &H->d_lidx) || upload(K, hs.uidx, &H->d_uidx) || upload(K, t.ucolptr, &H->d_ucolptr) ||
                upload(K, t.unzcol, &H->d_unzcol) || upload(K, hs.xsup, &H->d_xsup)) return SLUAMD_EHIP;
            T.lidx = H->d_lidx; T.uidx = H->d_uidx; T.ucolptr = H->d_ucolptr; T.unzcol = H->d_unzcol; T.xsup = H->d_xsup;
#define UP(field, vec, type) { type *p_; if (upload(K, vec, &p_)) return SLUAMD_EHIP; T.field = p_; }
            UP(sn_lval, t.sn_lval, int64_t) UP(sn_uval, t.sn_uval, int64_t) UP(sn_lidx, t.sn_lidx, int64_t) UP(sn_uidx, t.sn_uidx, int64_t)
            UP(sn_dinv, t.sn_dinv, int64_t) UP(sn_dptr, t.sn_dptr, int64_t) UP(sn_inv, t.sn_inv, int64_t)
            UP(sn_nsupr, t.sn_nsupr, int) UP(sn_flags, t.sn_flags, int) UP(sn_ldiag, t.sn_ldiag, int) UP(sn_dlda, t.sn_dlda, int)
            UP(sn_ldu, t.sn_ldu, int) UP(sn_ncolu, t.sn_ncolu, int)
            UP(sn_lb_off, t.sn_lb_off, int) UP(sn_nlb, t.sn_nlb, int) UP(sn_ub_off, t.sn_ub_off, int) UP(sn_nub, t.sn_nub, int)
            UP(sn_rt_off, t.sn_rt_off, int) UP(sn_nrt, t.sn_nrt, int) UP(sn_ct_off, t.sn_ct_off, int) UP(sn_nct, t.sn_nct, int)
            UP(lb_gid, t.lb_gid, int) UP(lb_nbrow, t.lb_nbrow, int) UP(lb_rowoff, t.lb_rowoff, int) UP(lb_lptr, t.lb_lptr, int)
            UP(lbs_gid, t.lbs_gid, int) UP(lbs_idx, t.lbs_idx, int)
            UP(ub_gid, t.ub_gid, int) UP(ub_ncols, t.ub_ncols, int) UP(ub_iukp, t.ub_iukp, int) UP(ub_stcol, t.ub_stcol, int)
            UP(rtile, t.rtile, int4) UP(ctile, t.ctile, int4) UP(rt_info, t.rt_info, int2) UP(ct_info, t.ct_info, int4)
            UP(lrow, t.lrow, int) UP(sn_lrow, t.sn_lrow, int64_t) UP(ucol_cp, t.ucol_cp, int) UP(ucol_ld, t.ucol_ld, int) UP(ucol_gc, t.ucol_gc, int) UP(sn_ucol, t.sn_ucol, int64_t)
#undef UP
            table_bytes = upload_bytes() - mark;
            return 0;
        }();
    };
    if (!sync_tables) table_job.th = std::thread(table_upload);

    // ---- 6. schedules ----
    int nlevtot = 0;
    std::vector<GemmDesc> grp_descs;
    std::vector<int4> grp_tiles;
    for (int zl = 0; zl < nz; ++zl) {
        if (!in.z_active[zl]) continue;
        LevelSched &S = H->sched[zl];
        build_schedule(*H, t, in.lists[zl], lvl[zl], nlev[zl], S);
        nlevtot += S.nlevels;
        // merged chain groups of the sweeps: one rank, one forest, real arithmetic, the list schedules (not the deterministic mode)
        if (H->env.solve_groups && !xy && nz == 1 && g.size() == 1 && !H->z && !H->opt.deterministic && H->env.solve_join) {
            std::vector<int> slev;
            int nslev = 0;
            build_solve_groups(*H, in, t, in.lists[zl], lvl[zl], S, slev, nslev, grp_descs, grp_tiles);
            if (!H->groups.empty()) {
                H->ssched.assign(nz, LevelSched());
                build_solve_sched(*H, t, in.lists[zl], slev, nslev, H->ssched[zl]);
                S.no_join = true;
            } else H->grp_of.clear();
        }
        // exchange plan into the schedule; own diagonal blocks of each level packed in ascending supernode order
        if (xy) {
            S.x_diag_send.resize(S.nlevels); S.x_diag_recv.resize(S.nlevels); S.x_panel_send.resize(S.nlevels); S.x_panel_recv.resize(S.nlevels);
            S.xs_red_send.resize(S.nlevels); S.xs_red_recv.resize(S.nlevels); S.xs_bc_send.resize(S.nlevels); S.xs_bc_recv.resize(S.nlevels);
            S.dg_stage_off.assign(S.nlevels, 0);
            for (int l = 0; l < S.nlevels; ++l) {
                LevelX &X = lx[zl][l];
                S.x_diag_send[l] = X.ds; S.x_diag_recv[l] = X.dr; S.x_panel_send[l] = X.ps; S.x_panel_recv[l] = X.pr;
                S.xs_red_send[l] = X.rs; S.xs_red_recv[l] = X.rr; S.xs_bc_send[l] = X.bs; S.xs_bc_recv[l] = X.br;
                S.dg_stage_off[l] = X.dstage;
                // pack offsets follow ASCENDING supernode order (what the receivers assume), whatever the launch order
                const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
                std::vector<int64_t> off_of(nn, 0);
                {
                    std::vector<int> order(nn);
                    std::iota(order.begin(), order.end(), 0);
                    std::sort(order.begin(), order.end(), [&](int a, int b) { return S.nodes[n0 + a] < S.nodes[n0 + b]; });
                    int64_t o = 0;
                    for (int i : order) {
                        const int k = S.nodes[n0 + i];
                        off_of[i] = o;
                        if (t.sn_flags[k] & SNF_OWN_DIAG) o += (int64_t) nsupc_of(hs, k) * nsupc_of(hs, k);
                    }
                }
                for (int i = 0; i < nn; ++i) {
                    const int k = S.nodes[n0 + i];
                    const int64_t sz = (t.sn_flags[k] & SNF_OWN_DIAG) ? (int64_t) nsupc_of(hs, k) * nsupc_of(hs, k) : 0;
                    S.dg_prefix[po + i + 1] = S.dg_prefix[po + i] + (int) ((sz + 1023) / 1024);
                    S.dg_off[po + i] = off_of[i];
                }
            }
        }
    }
    H->st.num_levels = nlevtot;
    H->setup.lap("sched.rest");
    H->st.chain_levels = 0; H->st.chain_units = 0;

    // ---- 7. device allocations + uploads (the value arena is still being allocated and zero-filled by its helper thread: joined at the end, nothing here touches it) ----
    if (H->env.reserve_cus > 0) {
        // keep `reserve_cus` compute units out of the main (Schur tile) stream: the panel kernels of the look-ahead stream then
        // always find a free CU (LDS for a whole TRSM strip / diagonal block) instead of waiting for a Schur workgroup to retire
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, H->device));
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0);
        for (int i = 0; i < ncu; ++i) if (i >= H->env.reserve_cus) mask[i >> 5] |= 1u << (i & 31);
        HIPCHK(hipExtStreamCreateWithCUMask(&H->stream, (uint32_t) mask.size(), mask.data()));
    } else HIPCHK(hipStreamCreate(&H->stream));
    {
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&H->pstream, hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&H->ustream, hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&H->u2stream, hipStreamNonBlocking, hi));
        if (H->grid.Pz > 1) { HIPCHK(hipStreamCreateWithPriority(&H->rstream, hipStreamNonBlocking, lo)); HIPCHK(hipEventCreateWithFlags(&H->red_all, hipEventDisableTiming)); }
    }
    HIPCHK(hipEventCreate(&H->ev0)); HIPCHK(hipEventCreate(&H->ev1));
    auto &K = H->d_misc;
    DevTables &T = H->T;
    if (sync_tables) { table_upload(); table_bytes = 0; } else table_job.th.join();          // the block / tile tables went up beside the schedule construction
    if (table_rc) return table_rc;
    upload_bytes() += table_bytes;
    H->h_sn_dinv = t.sn_dinv;
    {
        double *dv;
        auto malloc_retry = [&](void **q, size_t bytes) {      // (once more after the arena pool gave its unused chunks back to the driver)
            if (hipMalloc(q, bytes) == hipSuccess) return true;
            (void) hipGetLastError();
            devpool_trim(H->device);
            return hipMalloc(q, bytes) == hipSuccess;
        };
        if (!malloc_retry((void **) &dv, esz * (size_t) std::max<int64_t>(t.dinv_total, 1))) { set_error("hipMalloc(dinv) failed"); return SLUAMD_ENOMEM; }
        K.push_back(dv); T.dinv = dv;
    }
    if (!H->z) {   // Linv / Uinv of the owned diagonal blocks (solve); complex handles use their own first-correct solves
        double *iv;
        if (hipMalloc((void **) &iv, sizeof(double) * (size_t) std::max<int64_t>(t.inv_total, 1)) != hipSuccess) {
            (void) hipGetLastError();
            devpool_trim(H->device);
            if (hipMalloc((void **) &iv, sizeof(double) * (size_t) std::max<int64_t>(t.inv_total, 1)) != hipSuccess) { set_error("hipMalloc(inv) failed"); return SLUAMD_ENOMEM; }
        }
        K.push_back(iv); T.inv = iv;
    }
    if (!H->z && g.Pr * g.Pc == 1 && H->env.solve_join) { H->h_lrow_near.assign(std::max<size_t>(t.lrow.size(), 1), 0); H->h_ucol_near.assign(std::max<size_t>(t.ucol_gc.size(), 1), 0); }
    H->setup.lap("upload.block_tile_tables");
    for (auto &S : H->sched) if (upload_schedule(*H, S, t)) return SLUAMD_EHIP;
    for (auto &S : H->ssched) if (S.nlevels && upload_schedule(*H, S, t)) return SLUAMD_EHIP;
    if (!H->groups.empty()) {
        std::vector<GrpDesc> gd(H->groups.size());
        for (size_t gi = 0; gi < gd.size(); ++gi) {
            const Handle::SolveGroup &G = H->groups[gi];
            gd[gi].nm = G.nm; gd[gi].nG = G.nG; gd[gi].ginv = G.ginv;
            for (int m = 0; m < 4; ++m) { gd[gi].k[m] = G.k[m]; gd[gi].o[m] = G.o[m]; gd[gi].w[m] = G.w[m]; }
        }
        if (upload(K, gd, &H->d_grpdesc) || upload(K, grp_descs, &H->d_gemmdesc) || upload(K, grp_tiles, &H->d_gemmtiles)) return SLUAMD_EHIP;
        if (hipMalloc((void **) &H->d_gscr, sizeof(double) * (size_t) (4 * GRP_SCR)) != hipSuccess) { set_error("hipMalloc of the group scratch failed"); return SLUAMD_ENOMEM; }
        K.push_back(H->d_gscr);
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&H->gstream, hipStreamNonBlocking, lo));
        if (getenv("SLUAMD_PLAN_DEBUG")) {
            int64_t cols = 0, vals = 0;
            for (auto &G : H->groups) { cols += G.nG; vals += (int64_t) 2 * G.nG * G.nG; }
            fprintf(stderr, "[sluamd_plan] merged chain groups of the sweeps: %zu groups, %lld columns, %.3f GB of group inverses, sweep levels %d -> %d\n", H->groups.size(),
                    (long long) cols, 8.0 * vals / 1e9, H->sched[0].nlevels, H->ssched[0].nlevels);
        }
    }
    T.lrow_near = nullptr; T.ucol_near = nullptr;
    if (!H->h_lrow_near.empty()) {
        uint8_t *p0, *p1;
        if (upload(K, H->h_lrow_near, &p0) || upload(K, H->h_ucol_near, &p1)) return SLUAMD_EHIP;
        T.lrow_near = p0; T.ucol_near = p1;
        std::vector<uint8_t>().swap(H->h_lrow_near); std::vector<uint8_t>().swap(H->h_ucol_near);
    }
    if (H->fused_pairs) {
        int *p0, *p1, *p2, *p3, *p4, *p5;
        if (upload(K, H->h_fuse_prev, &p0) || upload(K, H->h_defer, &p1) || upload(K, H->h_pair_roff, &p2) ||
            upload(K, H->h_pair_coff, &p3) || upload(K, H->h_pair_rowmap, &p4) || upload(K, H->h_pair_colinfo, &p5)) return SLUAMD_EHIP;
        T.fuse_prev = p0; T.defer = p1; T.pair_roff = p2; T.pair_coff = p3; T.pair_rowmap = p4; T.pair_colinfo = p5;
    }
    H->st.reserved_i = H->fused_pairs;   // K-fused supernode pairs (diagnostic)
    H->st.schur_tiles = 0;               // until the first factorisation: the PLANNED tile executions of one factorisation (list schedules)
    for (auto &S : H->sched) H->st.schur_tiles += (int64_t) S.ulist.size();
    HIPCHK(hipMalloc((void **) &H->d_info, 8 * sizeof(int)));
    if (H->z) { std::vector<int> zero(std::max(ns, 1), 0); if (upload(K, zero, &H->d_ztickets)) return SLUAMD_EHIP; }
    rc = eng::setup();
    if (rc) return rc;
    H->setup.lap("upload.rest");
    arena_job.th.join();
    if (arena_rc == 1) {
        (void) hipGetLastError();
        size_t fr = 0, tot = 0;
        hipMemGetInfo(&fr, &tot);
        char msg[512];
        snprintf(msg, sizeof msg, "hipMalloc of the value arena failed: rank (%d,%d,%d) of the %d x %d x %d grid needs %.1f GB (own L/U slots %.1f GB + exchange scratch %.1f GB) "
                 "and %.1f GB of %.1f GB are free on device %d -- use a larger process grid (scripts/capacity.py prints the per-rank storage of a grid)",
                 H->grid.r, H->grid.c, H->grid.z, H->grid.Pr, H->grid.Pc, H->grid.Pz, esz * (double) H->arena_len / 1e9, esz * (double) H->own_len / 1e9,
                 esz * (double) (H->arena_len - H->own_len) / 1e9, fr / 1e9, tot / 1e9, H->device);
        set_error(msg);
        return SLUAMD_ENOMEM;
    }
    if (arena_rc) { set_error("allocation / zero fill of the value arena failed"); return SLUAMD_EHIP; }
    H->setup.lap("arena_alloc_zero_wait");
    T.val = H->d_val;
    H->st.nnz_L = hs.nnzL; H->st.nnz_U = hs.nnzU;
    // everything this handle allocated on the device: the value arena, the inverse stores and every uploaded table (index images, block / tile
    // tables, tile lists, unit lists and records, pair maps; round 3 counted the index images only)
    const size_t tables = upload_bytes() - upload_mark;
    H->st.bytes_device = (int64_t) ((size_t) H->arena_len * esz + tables + (size_t) (t.dinv_total + t.inv_total) * 8);
    if (getenv("SLUAMD_PLAN_DEBUG"))
        fprintf(stderr, "[sluamd_plan] rank (%d,%d,%d): own values %.3f GB, exchange scratch %.3f GB (%d copies), diagonal scratch + own tail %.3f GB, inverses %.3f GB, "
                        "tables %.3f GB, levels %d, fused pairs %d\n", g.r, g.c, g.z, esz * (double) H->own_len / 1e9, esz * (double) H->xy_scratch_len / 1e9, H->xy_scratch_copies,
                esz * (double) (H->arena_len - H->own_len - H->xy_scratch_len) / 1e9, 8.0 * (double) (t.dinv_total + t.inv_total) / 1e9, tables / 1e9, H->st.num_levels, H->fused_pairs);
    return 0;
}

}  // namespace sluamd
