// oracle/emul/engine_cpu.cpp -- TEST INFRASTRUCTURE: serial CPU restatement of the eng:: interface
// (superlu_dist_amd/csrc/sluamd_internal.h), one plain loop nest per kernel of sluamd_kernels.hip, consuming the SAME
// device tables (DevTables, level schedules, tile lists, exchange staging).  Linked with the library's host sources into
// oracle/libsluamd_emul.so so that `-m "not gpu"` tests can run the planning and the multi-rank orchestration
// (Z forests + ancestor reduction, XY block-cyclic panel exchange, distributed solve) against the golden fixtures
// without a GPU.  The product library (libsluamd.so) never links this file.
//
// Arithmetic follows the reference routines each kernel replaces: Local_Dgstrf2 (pdgstrf2.c:508-601), dLPanelTrSolve
// (dtrfCommWrapper.c:120-223), dTrs2_GatherTrsmScatter (pdgstrf2.c:757-840), dblock_gemm_scatter + dscatter_l +
// scatter_u (dscatter3d.c:81-189, dscatter.c:109-194, dscatter3d.c:555-631), dlsum_fmod_inv / dlsum_bmod_inv
// (pdgstrs_lsum.c:414, :1362) -- with the kernels' blocking by 32 (inverted diagonal sub-blocks) kept.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "sluamd_internal.h"

namespace sluamd {
namespace eng {
// the restated kernels; the eng:: entry points at the end of the file hand them to the emulated streams (emul_rt.cpp)
namespace impl {

static int find_node(const int *prefix, int nn, int id)
{
    int lo = 0, hi = nn;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (prefix[mid] <= id) lo = mid; else hi = mid; }
    return lo;
}

static void inv_block(const DevTables &T, int k, int typ, int b);

void diag_lu(hipStream_t, const DevTables &T, const int *nodes, int nn, int, int flags, double thresh, int *info)
{
    const int replace_tiny = flags & 1;
    for (int i = 0; i < nn; ++i) {
        const int k = nodes[i];
        if (!(T.sn_flags[k] & SNF_OWN_DIAG)) continue;
        const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst, lda = T.sn_dlda[k];
        double *A = T.val + T.sn_dptr[k];
        for (int j = 0; j < ns; ++j) {
            double p = A[j + (size_t) j * lda];
            if (replace_tiny && std::fabs(p) < thresh) { p = (p < 0) ? -thresh : thresh; A[j + (size_t) j * lda] = p; info[1] += 1; }
            if (p == 0.0) { info[0] = std::min(info[0], fst + j + 1); info[4] = std::max(info[4], fst + j + 1); }
            const double rinv = (p != 0.0) ? 1.0 / p : 1.0;
            for (int r = j + 1; r < ns; ++r) A[r + (size_t) j * lda] *= rinv;
            for (int c = j + 1; c < ns; ++c) {
                const double u = A[j + (size_t) c * lda];
                for (int r = j + 1; r < ns; ++r) A[r + (size_t) c * lda] -= A[r + (size_t) j * lda] * u;
            }
        }
        for (int typ = 0; typ < 2; ++typ) for (int b = 0; b < (ns + DB - 1) / DB; ++b) inv_block(T, k, typ, b);   // contract: dinv of the owned blocks
    }
}

static void inv_block(const DevTables &T, int k, int typ, int b)
{
    const int ns = T.xsup[k + 1] - T.xsup[k], nblk = (ns + DB - 1) / DB, o = b * DB;
    const int lda = T.sn_dlda[k];
    const double *A = T.val + T.sn_dptr[k];
    double Bs[DB][DB], X[DB][DB];
    for (int i = 0; i < DB; ++i)
        for (int c = 0; c < DB; ++c) {
            double v = (i == c) ? 1.0 : 0.0;
            if (o + i < ns && o + c < ns && i <= c) {
                if (typ == 0) v = A[o + i + (size_t) (o + c) * lda];
                else if (i < c) v = A[o + c + (size_t) (o + i) * lda];
            }
            Bs[i][c] = v;
        }
    for (int c = 0; c < DB; ++c)
        for (int i = c; i >= 0; --i) {
            double a = (i == c) ? 1.0 : 0.0;
            for (int jj = i + 1; jj <= c; ++jj) a -= Bs[i][jj] * X[jj][c];
            X[i][c] = a / Bs[i][i];
        }
    double *dst = T.dinv + T.sn_dinv[k] + (size_t) (typ * nblk + b) * DB * DB;
    for (int c = 0; c < DB; ++c) for (int i = 0; i < DB; ++i) dst[c * DB + i] = (i <= c) ? X[i][c] : 0.0;
}

void diag_inv(hipStream_t, const DevTables &T, const int *nodes, const int *prefix, int nn, int ntask)
{
    for (int task = 0; task < ntask; ++task) {
        const int ni = find_node(prefix, nn, task);
        const int k = nodes[ni];
        if (!(T.sn_flags[k] & SNF_HAS_DIAG)) continue;
        const int ns = T.xsup[k + 1] - T.xsup[k], nblk = (ns + DB - 1) / DB;
        const int rem = task - prefix[ni], typ = rem / nblk;
        inv_block(T, k, typ, rem - typ * nblk);
    }
}

// X (1 x ns row vector, zero-padded to nsp) <- X * T^-1, T upper triangular given by tfun(kk, cc) and its inverted 32x32
// diagonal blocks D(kk, cc) = dinv[blk][cc * 32 + kk]
template <class TF>
static void row_trsm(double *x, int ns, int nsp, const double *dinv, TF tfun)
{
    for (int jb = 0; jb < nsp; jb += DB) {
        double rhs[DB], out[DB];
        for (int c = 0; c < DB; ++c) {
            double a = x[jb + c];
            for (int kk = 0; kk < jb; ++kk) a -= x[kk] * tfun(kk, jb + c);
            rhs[c] = a;
        }
        const double *D = dinv + (size_t) (jb / DB) * DB * DB;
        for (int c = 0; c < DB; ++c) { double a = 0; for (int kk = 0; kk < DB; ++kk) a += rhs[kk] * D[c * DB + kk]; out[c] = a; }
        for (int c = 0; c < DB; ++c) x[jb + c] = out[c];
    }
    (void) ns;
}

void panel_trsm(hipStream_t, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int rs, int, const int2 *units)
{
    std::vector<double> x;
    for (int w = 0; w < nl + nu; ++w) {
        const bool lmode = w < nl;
        const int id = lmode ? w : w - nl;
        const int ni = units ? 0 : find_node(lmode ? lprefix : uprefix, nn, id);
        const int k = units ? units[w].x : nodes[ni];
        const int strip = units ? units[w].y : id - (lmode ? lprefix : uprefix)[ni];
        const int klst = T.xsup[k + 1], ns = klst - T.xsup[k], nsp = (ns + DB - 1) & ~(DB - 1), nblk = nsp / DB;
        const int lda = T.sn_nsupr[k], ldd = T.sn_dlda[k];
        double *A = T.val + T.sn_lval[k];
        const double *Dg = T.val + T.sn_dptr[k];
        double *Uv = T.val + T.sn_uval[k];
        x.assign(nsp, 0.0);
        if (lmode) {
            const double *dinv = T.dinv + T.sn_dinv[k];
            auto tf = [&](int kk, int cc) { return (kk < ns && cc < ns) ? Dg[kk + (size_t) cc * ldd] : 0.0; };
            for (int r = 0; r < rs; ++r) {
                const int row = T.sn_ldiag[k] + strip * rs + r;
                if (row >= lda) break;
                for (int c = 0; c < nsp; ++c) x[c] = c < ns ? A[row + (size_t) c * lda] : 0.0;
                row_trsm(x.data(), ns, nsp, dinv, tf);
                for (int c = 0; c < ns; ++c) A[row + (size_t) c * lda] = x[c];
            }
        } else {
            const double *dinv = T.dinv + T.sn_dinv[k] + (size_t) nblk * DB * DB;
            auto tf = [&](int kk, int cc) { return (kk < ns && cc < ns) ? Dg[cc + (size_t) kk * ldd] : 0.0; };   // L_kk^T
            for (int r = 0; r < rs; ++r) {
                const int cr = strip * rs + r;
                if (cr >= T.sn_ncolu[k]) break;
                const int ub0 = T.sn_ub_off[k], nub = T.sn_nub[k];
                int lo = 0, hi = nub;
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (T.ub_stcol[ub0 + mid] <= cr) lo = mid; else hi = mid; }
                const int b = ub0 + lo;
                const int64_t u0 = T.sn_uidx[k] + T.ub_iukp[b];
                const int jj = T.unzcol[u0 + (cr - T.ub_stcol[b])];
                const int ld = ns - (klst - T.uidx[u0 + jj]);
                const int cp = T.ucolptr[u0 + jj];
                for (int c = 0; c < nsp; ++c) x[c] = (c >= ld && c < ns) ? Uv[cp + (c - ld)] : 0.0;
                row_trsm(x.data(), ns, nsp, dinv, tf);
                for (int c = ld; c < ns; ++c) Uv[cp + (c - ld)] = x[c];
            }
        }
    }
}

void panel_gemm(hipStream_t, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int, const int2 *units)
{
    std::vector<double> x, o;
    for (int w = 0; w < nl + nu; ++w) {
        const bool lmode = w < nl;
        const int id = lmode ? w : w - nl;
        const int ni = units ? 0 : find_node(lmode ? lprefix : uprefix, nn, id);
        const int k = units ? units[w].x : nodes[ni];
        const int strip = units ? units[w].y : id - (lmode ? lprefix : uprefix)[ni];
        const int klst = T.xsup[k + 1], ns = klst - T.xsup[k];
        const int lda = T.sn_nsupr[k];
        double *A = T.val + T.sn_lval[k];
        double *Uv = T.val + T.sn_uval[k];
        const double *Li = T.inv + T.sn_inv[k], *Ui = Li + (size_t) ns * ns;
        x.assign(ns, 0.0); o.assign(ns, 0.0);
        for (int r = 0; r < 64; ++r) {
            if (lmode) {
                const int row = T.sn_ldiag[k] + strip * 64 + r;
                if (row >= lda) break;
                for (int c = 0; c < ns; ++c) x[c] = A[row + (size_t) c * lda];
                for (int n = 0; n < ns; ++n) { double a = 0; for (int kk = 0; kk <= n; ++kk) a += x[kk] * Ui[kk + (size_t) n * ns]; o[n] = a; }
                for (int c = 0; c < ns; ++c) A[row + (size_t) c * lda] = o[c];
            } else {
                const int cr = strip * 64 + r;
                if (cr >= T.sn_ncolu[k]) break;
                const int ub0 = T.sn_ub_off[k], nub = T.sn_nub[k];
                int lo = 0, hi = nub;
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (T.ub_stcol[ub0 + mid] <= cr) lo = mid; else hi = mid; }
                const int b = ub0 + lo;
                const int64_t u0 = T.sn_uidx[k] + T.ub_iukp[b];
                const int jj = T.unzcol[u0 + (cr - T.ub_stcol[b])];
                const int ld = ns - (klst - T.uidx[u0 + jj]), cp = T.ucolptr[u0 + jj];
                for (int c = 0; c < ns; ++c) x[c] = (c >= ld) ? Uv[cp + (c - ld)] : 0.0;
                for (int n = 0; n < ns; ++n) { double a = 0; for (int kk = 0; kk <= n; ++kk) a += Li[n + (size_t) kk * ns] * x[kk]; o[n] = a; }
                for (int c = ld; c < ns; ++c) Uv[cp + (c - ld)] = o[c];
            }
        }
    }
}

// V = double or std::complex<double>: the value arena is addressed in elements of V (offsets are type independent)
template <class V>
static void schur_t(const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info, const int4 *ulist)
{
    V *val = reinterpret_cast<V *>(T.val);
    std::vector<V> acc;
    std::vector<int> rowmap, colmap;
    // workgroups of one launch run in no particular order on the device: under the adversarial scheduler, in a shuffled one here
    std::vector<int> order(ntiles);
    for (int i = 0; i < ntiles; ++i) order[i] = i;
    if (unsigned sd = emul_launch_seed()) { std::mt19937 rng(sd); std::shuffle(order.begin(), order.end(), rng); }
    for (int it = 0; it < ntiles; ++it) {
        const int bid0 = order[it];
        const int bid = bid0 + id_base;
        int k, rt, ct;
        if (ulist) { k = ulist[bid].x; rt = ulist[bid].y - T.sn_rt_off[k]; ct = ulist[bid].z - T.sn_ct_off[k]; }
        else {
            const int ni = find_node(prefix, nn, bid);
            k = nodes[ni];
            const int local = bid - prefix[ni], nct = T.sn_nct[k];
            rt = local / nct; ct = local - rt * nct;
        }
        const int4 R = T.rtile[T.sn_rt_off[k] + rt], C = T.ctile[T.sn_ct_off[k] + ct];
        const int lb = T.sn_lb_off[k] + R.x, ub = T.sn_ub_off[k] + C.x;
        const int nr = R.z, nc = C.z;
        const int ib = T.lb_gid[lb], jb = T.ub_gid[ub];
        const int klst = T.xsup[k + 1], ns = klst - T.xsup[k];
        // merged row tile (list entry with destination -3): rows of several L blocks, all of them with gid >= jb; global ids from the flat row map
        const bool merged = ulist && ulist[bid].w == -3;
        const bool mergedU = ulist && ulist[bid].w == -4;     // merged column tile: columns of several U blocks (all with gid > ib), destination U row ib
        const int *lsub = merged ? T.lrow + T.sn_lrow[k] + R.w : T.lidx + T.sn_lidx[k] + T.lb_lptr[lb] + R.y;
        const int64_t uix0 = T.sn_uidx[k] + T.ub_iukp[ub];
        acc.assign((size_t) nr * nc, V(0));
        int nprev = 0;
        if (T.fuse_prev) while (nprev < 3 && T.fuse_prev[3 * k + nprev] >= 0) ++nprev;
        for (int src = 0; src <= nprev; ++src) {
            if (src < nprev) {
                const int pj = 3 * k + (nprev - 1 - src);
                const int ks = T.fuse_prev[pj];
                const int nss = T.xsup[ks + 1] - T.xsup[ks], ldas = T.sn_nsupr[ks];
                const int *cinfo = T.pair_colinfo + 2 * (size_t) (T.pair_coff[pj] + T.ub_stcol[ub] + C.y);
                const V *Ls = val + T.sn_lval[ks], *Us = val + T.sn_uval[ks];
                for (int c = 0; c < nc; ++c) {
                    const int cp = cinfo[2 * c], lead = cinfo[2 * c + 1];
                    for (int r = 0; r < nr; ++r) {
                        const int ra = T.pair_rowmap[T.pair_roff[pj] + R.w + r];
                        if (ra < 0) continue;
                        V a(0);
                        for (int kk = lead; kk < nss; ++kk) a += Ls[ra + (size_t) kk * ldas] * Us[cp + (kk - lead)];
                        acc[r + (size_t) c * nr] += a;
                    }
                }
            } else {
                const int lda = T.sn_nsupr[k];
                const V *Lp = val + T.sn_lval[k] + R.w, *Uv = val + T.sn_uval[k];
                for (int c = 0; c < nc; ++c) {
                    int lead, cp;
                    if (mergedU) {   // columns of several U blocks: the flat per-non-empty-column maps of the slot
                        const int64_t f = T.sn_ucol[k] + T.ub_stcol[ub] + C.y + c;
                        lead = T.ucol_ld[f]; cp = T.ucol_cp[f];
                    } else {
                        const int jj = T.unzcol[uix0 + C.y + c];
                        lead = ns - (klst - T.uidx[uix0 + jj]); cp = T.ucolptr[uix0 + jj];
                    }
                    for (int r = 0; r < nr; ++r) {
                        V a(0);
                        for (int kk = lead; kk < ns; ++kk) a += Lp[r + (size_t) kk * lda] * Uv[cp + (kk - lead)];
                        acc[r + (size_t) c * nr] += a;
                    }
                }
            }
        }
        if (mergedU) {
            const int2 ri = T.rt_info[ulist[bid].y];
            const int4 ci = T.ct_info[ulist[bid].z];
            if (ri.x != ib || ci.x != jb || ib >= jb || T.lidx + ri.y != lsub || ci.y != uix0 || ci.z != T.ub_stcol[ub] + C.y || ci.z + nc > T.sn_ncolu[k]) { std::fprintf(stderr, "engine_cpu: merged column tile entry disagrees with the block tables\n"); std::abort(); }
            // destination of every column: LINEAR search of its global id among ALL columns of U row ib (block directory + fstnz: independent of the
            // kernel's binary search in the flat maps)
            const int o2 = T.sn_ub_off[ib], nb2 = T.sn_nub[ib];
            V *dst = val + T.sn_uval[ib];
            std::vector<int64_t> cmap(nc, INT64_MIN);
            bool all = true;
            for (int c = 0; c < nc; ++c) {
                const int gc = T.ucol_gc[T.sn_ucol[k] + T.ub_stcol[ub] + C.y + c];
                if (gc < T.xsup[ib + 1]) { std::fprintf(stderr, "engine_cpu: merged column tile column left of its destination row\n"); std::abort(); }
                for (int q = 0; q < nb2 && cmap[c] == INT64_MIN; ++q) {
                    const int jq = T.ub_gid[o2 + q];
                    if (gc < T.xsup[jq] || gc >= T.xsup[jq + 1]) continue;
                    const int64_t d0 = T.sn_uidx[ib] + T.ub_iukp[o2 + q];
                    const int jj = gc - T.xsup[jq];
                    if (T.uidx[d0 + jj] < T.xsup[ib + 1]) cmap[c] = (int64_t) T.ucolptr[d0 + jj] - T.uidx[d0 + jj];
                }
                all = all && cmap[c] != INT64_MIN;
            }
            if (!all) { info[2] += 1; continue; }
            for (int c = 0; c < nc; ++c)
                for (int r = 0; r < nr; ++r) dst[cmap[c] + lsub[r]] -= acc[r + (size_t) c * nr];
            continue;
        }
        if (merged) {
            // every row must belong to a block row >= jb and exist in panel jb (found by a LINEAR search of the whole panel: independent of the
            // kernel's binary search and of the planner's sortedness test); the tile list entry must agree with the tables
            const int2 ri = T.rt_info[ulist[bid].y];
            const int4 ci = T.ct_info[ulist[bid].z];
            if (ri.x != ib || ci.x != jb || ib < jb || ci.y != uix0 || ci.z != T.ub_stcol[ub] + C.y || R.w + nr > T.sn_nsupr[k]) { std::fprintf(stderr, "engine_cpu: merged tile entry disagrees with the block tables\n"); std::abort(); }
            const int *prow = T.lrow + T.sn_lrow[jb];
            const int pn = T.sn_nsupr[jb];
            V *dst = val + T.sn_lval[jb];
            bool all = true;
            std::vector<int> pos(nr, -1);
            for (int r = 0; r < nr; ++r) {
                if (lsub[r] < T.xsup[jb]) { std::fprintf(stderr, "engine_cpu: merged tile row above its destination panel\n"); std::abort(); }
                for (int q = 0; q < pn; ++q) if (prow[q] == lsub[r]) { pos[r] = q; break; }
                all = all && pos[r] >= 0;
            }
            if (!all) { info[2] += 1; continue; }
            for (int r = 0; r < nr; ++r)
                for (int c = 0; c < nc; ++c) {
                    const int jj = T.unzcol[uix0 + C.y + c];
                    dst[pos[r] + (size_t) jj * pn] -= acc[r + (size_t) c * nr];
                }
            continue;
        }
        // destination lookup + scatter
        const bool ldest = ib >= jb;
        const int o = ldest ? T.sn_lb_off[jb] : T.sn_ub_off[ib];
        const int nb = ldest ? T.sn_nlb[jb] : T.sn_nub[ib];
        const int *dir = ldest ? T.lbs_gid : T.ub_gid;
        const int want = ldest ? ib : jb;
        int pos = -1;
        for (int q = 0; q < nb; ++q) if (dir[o + q] == want) { pos = q; break; }
        if (ulist) {   // the host-resolved destination block and tile descriptors must agree with the tables
            const int want_d = pos < 0 ? -1 : (ldest ? o + T.lbs_idx[o + pos] : o + pos);
            const int2 ri = T.rt_info[ulist[bid].y];
            const int4 ci = T.ct_info[ulist[bid].z];
            if (ulist[bid].w != want_d || ri.x != ib || ci.x != jb || T.lidx + ri.y != lsub || ci.y != uix0 || ci.z != T.ub_stcol[ub] + C.y) {
                std::fprintf(stderr, "engine_cpu: tile list entry disagrees with the block tables\n"); std::abort();
            }
        }
        if (pos < 0) { info[2] += 1; continue; }
        if (ldest) {
            const int d = o + T.lbs_idx[o + pos];
            const int rowoff = T.lb_rowoff[d], dn = T.lb_nbrow[d];
            const int *drows = T.lidx + T.sn_lidx[jb] + T.lb_lptr[d];
            V *dst = val + T.sn_lval[jb];
            const int ldv = T.sn_nsupr[jb];
            for (int r = 0; r < nr; ++r) {
                int di = -1;
                for (int q = 0; q < dn; ++q) if (drows[q] == lsub[r]) { di = q; break; }
                if (di < 0) { std::fprintf(stderr, "engine_cpu: destination row missing\n"); std::abort(); }
                for (int c = 0; c < nc; ++c) {
                    const int jj = T.unzcol[uix0 + C.y + c];
                    dst[rowoff + di + (size_t) jj * ldv] -= acc[r + (size_t) c * nr];
                }
            }
        } else {
            const int64_t d0 = T.sn_uidx[ib] + T.ub_iukp[o + pos];
            V *dst = val + T.sn_uval[ib];
            for (int c = 0; c < nc; ++c) {
                const int jj = T.unzcol[uix0 + C.y + c];
                const int cm = T.ucolptr[d0 + jj] - T.uidx[d0 + jj];
                for (int r = 0; r < nr; ++r) dst[cm + lsub[r]] -= acc[r + (size_t) c * nr];
            }
        }
    }
}

void schur(hipStream_t, int, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info,
           const int4 *ulist, int)
{
    schur_t<double>(T, nodes, prefix, nn, id_base, ntiles, info, ulist);
}

void full_inv(hipStream_t, const DevTables &T, const int *nodes, const int *prefix, int nn, int, int)
{
    for (int i0 = 0; i0 < nn; ++i0) {
        const int k = nodes[i0];
        if (prefix[i0 + 1] == prefix[i0]) continue;          // no work units: the block is not available here (or not wanted: XY peers in TRSM mode)
        if (!(T.sn_flags[k] & SNF_HAS_DIAG)) { std::fprintf(stderr, "engine_cpu: full_inv unit without the diagonal block\n"); std::abort(); }
        const int ns = T.xsup[k + 1] - T.xsup[k], lda = T.sn_dlda[k], nblk = (ns + DB - 1) / DB;
        const double *A = T.val + T.sn_dptr[k];
        for (int typ = 0; typ < 2; ++typ) {
            const double *dinv = T.dinv + T.sn_dinv[k] + (size_t) typ * nblk * DB * DB;
            double *out = T.inv + T.sn_inv[k] + (typ == 0 ? (size_t) ns * ns : 0);
            auto M = [&](int r, int c) { return (r < ns && c < ns) ? (typ == 0 ? A[r + (size_t) c * lda] : A[c + (size_t) r * lda]) : 0.0; };
            auto Xat = [&](int r, int c) { return typ == 0 ? out + r + (size_t) c * ns : out + c + (size_t) r * ns; };
            for (int e = 0; e < ns * ns; ++e) { const int r = e % ns, c = e / ns; if (r > c) *Xat(r, c) = 0.0; }
            for (int i = nblk - 1; i >= 0; --i) {
                const double *D = dinv + (size_t) i * DB * DB;
                for (int kk = 0; kk < DB; ++kk) for (int cc = 0; cc < DB; ++cc)
                    if (i * DB + kk < ns && i * DB + cc < ns) *Xat(i * DB + kk, i * DB + cc) = D[cc * DB + kk];
                for (int j = i + 1; j < nblk; ++j) {
                    double S[DB][DB];
                    for (int r = 0; r < DB; ++r) for (int c = 0; c < DB; ++c) {
                        double a = 0.0;
                        if (j * DB + c < ns) for (int q = (i + 1) * DB; q < std::min((j + 1) * DB, ns); ++q) a += M(i * DB + r, q) * *Xat(q, j * DB + c);
                        S[r][c] = a;
                    }
                    for (int r = 0; r < DB; ++r) for (int c = 0; c < DB; ++c)
                        if (i * DB + r < ns && j * DB + c < ns) { double a = 0.0; for (int q = 0; q < DB; ++q) a += D[q * DB + r] * S[q][c]; *Xat(i * DB + r, j * DB + c) = -a; }
                }
            }
        }
    }
}

void solve_diag(hipStream_t, bool lower, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs, int)
{
    std::vector<double> xs;
    for (int i0 = 0; i0 < nn; ++i0) {
        const int k = nodes[i0];
        if (!(T.sn_flags[k] & SNF_OWN_DIAG)) continue;
        const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
        const double *Ti = T.inv + T.sn_inv[k] + (lower ? 0 : (size_t) ns * ns);
        for (int q = 0; q < nrhs; ++q) {
            double *xk = x + fst + (int64_t) q * ldx;
            xs.assign(xk, xk + ns);
            for (int i = 0; i < ns; ++i) {
                double a = 0.0;
                if (lower) for (int j = 0; j <= i; ++j) a += Ti[i + (size_t) j * ns] * xs[j];
                else for (int j = i; j < ns; ++j) a += Ti[i + (size_t) j * ns] * xs[j];
                xk[i] = a;
            }
        }
    }
}

template <class V, int STRIP>
static void fwd_update_t(const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const V *xsrc, V *x, int64_t ldx, int nrhs, const int2 *units)
{
    std::vector<int> order(std::max(nwork, 0));
    for (int i = 0; i < nwork; ++i) order[i] = i;
    if (unsigned sd = emul_launch_seed()) { std::mt19937 rng(sd); std::shuffle(order.begin(), order.end(), rng); }
    for (int it = 0; it < nwork; ++it) {
        const int w = order[it];
        int k, strip;
        if (units) { k = units[w].x; strip = units[w].y; }
        else { const int ni = find_node(prefix, nn, w); k = nodes[ni]; strip = w - prefix[ni]; }
        const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst, lda = T.sn_nsupr[k];
        const int *lsub = T.lidx + T.sn_lidx[k];
        for (int t = 0; t < STRIP; ++t) {
            const int row = T.sn_ldiag[k] + strip * STRIP + t;
            if (row >= lda) break;
            const int grow = T.lrow[T.sn_lrow[k] + row];
            const V *L = reinterpret_cast<const V *>(T.val) + T.sn_lval[k] + row;
            for (int r = 0; r < nrhs; ++r) {
                V acc(0);
                for (int kk = 0; kk < ns; ++kk) acc += L[(size_t) kk * lda] * xsrc[fst + kk + (int64_t) r * ldx];
                x[grow + (int64_t) r * ldx] -= acc;
            }
        }
    }
}

template <class V>
static void bwd_update_t(const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const V *xcols, V *x, int64_t ldx, int nrhs, const int2 *units)
{
    std::vector<int> order(std::max(nwork, 0));
    for (int i = 0; i < nwork; ++i) order[i] = i;
    if (unsigned sd = emul_launch_seed()) { std::mt19937 rng(sd); std::shuffle(order.begin(), order.end(), rng); }
    for (int it = 0; it < nwork; ++it) {
        const int w = order[it];
        int k, chunk;
        if (units) { k = units[w].x; chunk = units[w].y; }
        else { const int ni = find_node(prefix, nn, w); k = nodes[ni]; chunk = w - prefix[ni]; }
        const int fst = T.xsup[k], klst = T.xsup[k + 1], ns = klst - fst;
        const int ncol = std::min(64, T.sn_ncolu[k] - chunk * 64);
        const V *Uv = reinterpret_cast<const V *>(T.val) + T.sn_uval[k];
        for (int t = 0; t < ncol; ++t) {
            const int c = chunk * 64 + t;
            const int64_t cidx = T.sn_ucol[k] + c;
            const int ld = T.ucol_ld[cidx], cp = T.ucol_cp[cidx], gc = T.ucol_gc[cidx];
            for (int r = 0; r < nrhs; ++r) {
                const V xv = xcols[gc + (int64_t) r * ldx];
                for (int i = ld; i < ns; ++i) x[fst + i + (int64_t) r * ldx] -= Uv[cp + (i - ld)] * xv;
            }
        }
    }
}

void fwd_update(hipStream_t, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xsrc, double *x, int64_t ldx, int nrhs, int,
                const int2 *units)
{
    fwd_update_t<double, 64>(T, nodes, prefix, nn, nwork, xsrc, x, ldx, nrhs, units);
}

void bwd_update(hipStream_t, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xcols, double *x, int64_t ldx, int nrhs, int,
                const int2 *units)
{
    bwd_update_t<double>(T, nodes, prefix, nn, nwork, xcols, x, ldx, nrhs, units);
}

// out-of-place diagonal-solve strips (diag_strip_body): xout_k[64-row strip] = inverse[strip rows, :] xin_k, in a shuffled order under the
// adversarial modes -- the strips of one supernode are independent only because xin != xout
static void diag_strips(bool lower, const DevTables &T, const int2 *dunits, int ndu, const double *xin, double *xout, int64_t ldx, int nrhs, const int4 *drecs = nullptr)
{
    std::vector<int> order(std::max(ndu, 0));
    for (int i = 0; i < ndu; ++i) order[i] = i;
    if (unsigned sd = emul_launch_seed()) { std::mt19937 rng(sd + 17); std::shuffle(order.begin(), order.end(), rng); }
    for (int it = 0; it < ndu; ++it) {
        int k = dunits[order[it]].x, strip = dunits[order[it]].y;
        int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
        const double *Ti = T.inv + T.sn_inv[k] + (lower ? 0 : (size_t) ns * ns);
        if (drecs) {      // unit records (as the kernel): a merged group's strips carry the GROUP's width and inverse
            const int4 a = drecs[2 * (size_t) order[it]], b = drecs[2 * (size_t) order[it] + 1];
            fst = a.x; ns = a.y; strip = a.z;
            Ti = T.inv + (lower ? (((int64_t) b.y << 32) | (uint32_t) b.x) : (((int64_t) b.w << 32) | (uint32_t) b.z));
        }
        for (int q = 0; q < nrhs; ++q)
            for (int i = strip * 64; i < std::min(ns, strip * 64 + 64); ++i) {
                double a = 0.0;
                const int j0 = lower ? 0 : i, j1 = lower ? i + 1 : ns;
                for (int j = j0; j < j1; ++j) a += Ti[i + (size_t) j * ns] * xin[fst + j + (int64_t) q * ldx];
                xout[fst + i + (int64_t) q * ldx] = a;
            }
    }
}

// the far units and the diagonal strips of one launch in either order (seed parity): a dependency inside one launch that should
// not be there shows up under one of the two orders
static void fwd_unit_rec(const DevTables &T, const int4 *rec, const double *xsrc, double *x, int64_t ldx, int nrhs);
static void bwd_unit_rec(const DevTables &T, const int4 *rec, const double *xcols, double *x, int64_t ldx, int nrhs);
void sweep_step(hipStream_t s, bool lower, const DevTables &T, const int2 *dunits, int ndu, const int2 *units, int nunits,
                double *xa, double *xb, int64_t ldx, int nrhs, int mx, const int4 *urecs = nullptr, const int4 *drecs = nullptr)
{
    const bool diag_first = emul_launch_seed() & 1;
    for (int pass = 0; pass < 2; ++pass) {
        if ((pass == 0) == diag_first) { if (ndu > 0) diag_strips(lower, T, dunits, ndu, lower ? xa : xb, lower ? xb : xa, ldx, nrhs, drecs); }
        else if (urecs) { for (int u = 0; u < nunits; ++u) { if (lower) fwd_unit_rec(T, urecs + 2 * (size_t) u, xb, xa, ldx, nrhs); else bwd_unit_rec(T, urecs + 2 * (size_t) u, xa, xb, ldx, nrhs); } }
        else if (lower) fwd_update(s, T, nullptr, nullptr, 0, nunits, xb, xa, ldx, nrhs, mx, units);
        else bwd_update(s, T, nullptr, nullptr, 0, nunits, xa, xb, ldx, nrhs, mx, units);
    }
}

// ---- joined links (LevelSched::join; k_sweep's joined units) -------------------------------------------------------------------------------------------
static inline int64_t rec64(int lo, int hi) { return ((int64_t) hi << 32) | (uint32_t) lo; }
// regular units given by their records; near rows / columns are not theirs
static void fwd_unit_rec(const DevTables &T, const int4 *rec, const double *xsrc, double *x, int64_t ldx, int nrhs)
{
    const int fst = rec[0].x, ns = rec[0].y & 0xffff, lda = rec[0].z, row0 = rec[0].w;
    const int chk = rec[0].y >> 16;      // as the kernel: the near flags are read only where the planner announced near rows (1: any flag, 2: dead rows only)
    const int64_t loff = rec64(rec[1].x, rec[1].y), roff = rec64(rec[1].z, rec[1].w);
    for (int r = 0; r < 64 && row0 + r < lda; ++r) {
        if (chk && T.lrow_near[roff + r] >= chk) continue;
        const int grow = T.lrow[roff + r];
        for (int q = 0; q < nrhs; ++q) {
            double acc = 0.0;
            for (int kk = 0; kk < ns; ++kk) acc += T.val[loff + r + (size_t) kk * lda] * xsrc[fst + kk + (int64_t) q * ldx];
            x[grow + (int64_t) q * ldx] -= acc;
        }
    }
}
static void bwd_unit_rec(const DevTables &T, const int4 *rec, const double *xcols, double *x, int64_t ldx, int nrhs)
{
    const int fst = rec[0].x, ns = rec[0].y & 0xffff, ncol = rec[0].z;
    const int chk = rec[0].y >> 16;
    const int64_t ci0 = rec64(rec[1].x, rec[1].y), uoff = rec64(rec[1].z, rec[1].w);
    for (int c = 0; c < ncol; ++c) {
        if (chk && T.ucol_near[ci0 + c] >= chk) continue;
        const int ld = T.ucol_ld[ci0 + c], cp = T.ucol_cp[ci0 + c], gc = T.ucol_gc[ci0 + c];
        for (int q = 0; q < nrhs; ++q) {
            const double xv = xcols[gc + (int64_t) q * ldx];
            for (int i = ld; i < ns; ++i) x[fst + i + (int64_t) q * ldx] -= T.val[uoff + cp + (i - ld)] * xv;
        }
    }
}
// a launch of regular units given by their records, in a shuffled order under the adversarial modes
static void units_by_records(bool lower, const DevTables &T, const int4 *recs, int n, const double *xsrc, double *x, int64_t ldx, int nrhs)
{
    std::vector<int> order(std::max(n, 0));
    for (int i = 0; i < n; ++i) order[i] = i;
    if (unsigned sd = emul_launch_seed()) { std::mt19937 rng(sd); std::shuffle(order.begin(), order.end(), rng); }
    for (int it = 0; it < n; ++it) { if (lower) fwd_unit_rec(T, recs + 2 * (size_t) order[it], xsrc, x, ldx, nrhs); else bwd_unit_rec(T, recs + 2 * (size_t) order[it], xsrc, x, ldx, nrhs); }
}
// forward joined unit (s, c) of supernode j: t = b_j[block c] - (rows of the level-l panels in block c) x_k, then y_j[strip s] += Linv[s, c] t
static void join_fwd_unit(const DevTables &T, const int4 *rec, const int4 *jaux, const double *xa, double *xb, int64_t ldx, int nrhs)
{
    const int fst = rec[0].x, ns = rec[0].y, st = rec[0].z, c = rec[0].w, nsrc = rec[1].z, ovf = rec[1].w;
    const double *Li = T.inv + rec64(rec[1].x, rec[1].y);
    if (c > st || 64 * st >= ns) { fprintf(stderr, "[emul] joined forward unit (%d, %d) of a %d-column supernode\n", st, c, ns); abort(); }
    const int nc = std::min(64, ns - 64 * c);
    for (int q = 0; q < nrhs; ++q) {
        double t[64];
        for (int i = 0; i < nc; ++i) t[i] = xa[fst + 64 * c + i + (int64_t) q * ldx];
        for (int si = 0; si < nsrc; ++si) {
            const int4 *sr = si < 3 ? rec + 2 + 2 * si : jaux + 2 * (size_t) (ovf + si - 3);
            const int fk = sr[0].x, nk = sr[0].y, lda = sr[0].z, nr = sr[0].w;
            const int64_t loff = rec64(sr[1].x, sr[1].y), roff = rec64(sr[1].z, sr[1].w);
            for (int r = 0; r < nr; ++r) {
                const int pos = T.lrow[roff + r] - fst - 64 * c;
                if (pos < 0 || pos >= nc || !T.lrow_near || !T.lrow_near[roff + r]) { fprintf(stderr, "[emul] joined forward unit: source row outside its column block / not flagged near\n"); abort(); }
                double acc = 0.0;
                for (int kk = 0; kk < nk; ++kk) acc += T.val[loff + r + (size_t) kk * lda] * xb[fk + kk + (int64_t) q * ldx];
                t[pos] -= acc;
            }
        }
        for (int r = 64 * st; r < std::min(ns, 64 * st + 64); ++r) {
            double a = 0.0;
            for (int cc = 0; cc < nc; ++cc) if (64 * c + cc <= r) a += Li[r + (size_t) (64 * c + cc) * ns] * t[cc];
            xb[fst + r + (int64_t) q * ldx] += a;
        }
    }
}
// backward joined unit (s, c) of supernode k: t = w_k[block c] - U(k rows of block c, columns of level l+1) x, then x_k[strip s] += Uinv[s, c] t
static void join_bwd_unit(const DevTables &T, const int4 *rec, const int4 *jaux, double *xa, const double *xb, int64_t ldx, int nrhs)
{
    const int fst = rec[0].x, ns = rec[0].y, st = rec[0].z, c = rec[0].w, noff = rec[1].z, ncnt = rec[1].w;
    const double *Ui = T.inv + rec64(rec[1].x, rec[1].y);
    const int64_t uoff = rec64(rec[2].x, rec[2].y);
    if (st > c || 64 * c >= ns) { fprintf(stderr, "[emul] joined backward unit (%d, %d) of a %d-column supernode\n", st, c, ns); abort(); }
    const int nc = std::min(64, ns - 64 * c);
    for (int q = 0; q < nrhs; ++q) {
        double t[64];
        for (int i = 0; i < nc; ++i) t[i] = xb[fst + 64 * c + i + (int64_t) q * ldx];
        for (int e = 0; e < ncnt; ++e) {
            const int4 col = jaux[noff + e];
            const double xv = xa[col.z + (int64_t) q * ldx];
            // rows of block c inside the supernode whose U row this is: a member of a merged group starts at column rec[2].z of the group and is rec[2].w wide
            const int moff = rec[2].z, nsm = rec[2].w;
            for (int i = std::max(col.x, 64 * c - moff); i < std::min(nsm, 64 * c - moff + nc); ++i) t[i + moff - 64 * c] -= T.val[uoff + col.y + (i - col.x)] * xv;
        }
        for (int r = 64 * st; r < std::min(ns, 64 * st + 64); ++r) {
            double a = 0.0;
            for (int cc = 0; cc < nc; ++cc) if (64 * c + cc >= r) a += Ui[r + (size_t) (64 * c + cc) * ns] * t[cc];
            xa[fst + r + (int64_t) q * ldx] += a;
        }
    }
}
// one joined link: the joined units and the regular units of the launch in ONE shuffled order under the adversarial modes (they are independent)
void sweep_join(hipStream_t, bool lower, const DevTables &T, const int4 *jrecs, int nj, const int4 *jaux, const int4 *urecs, int nunits, double *xa, double *xb,
                int64_t ldx, int nrhs, int)
{
    const int total = std::max(nj, 0) + std::max(nunits, 0);
    std::vector<int> order(total);
    for (int i = 0; i < total; ++i) order[i] = i;
    if (unsigned sd = emul_launch_seed()) { std::mt19937 rng(sd + 29); std::shuffle(order.begin(), order.end(), rng); }
    for (int it = 0; it < total; ++it) {
        const int w = order[it];
        if (w < nj) {
            if (lower) join_fwd_unit(T, jrecs + 8 * (size_t) w, jaux, xa, xb, ldx, nrhs);
            else join_bwd_unit(T, jrecs + 4 * (size_t) w, jaux, xa, xb, ldx, nrhs);
        } else if (lower) fwd_unit_rec(T, urecs + 2 * (size_t) (w - nj), xb, xa, ldx, nrhs);
        else bwd_unit_rec(T, urecs + 2 * (size_t) (w - nj), xa, xb, ldx, nrhs);
    }
}
void zero_nodes(hipStream_t, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs)
{
    for (int i = 0; i < nn; ++i)
        for (int q = 0; q < nrhs; ++q)
            for (int r = T.xsup[nodes[i]]; r < T.xsup[nodes[i] + 1]; ++r) x[r + (int64_t) q * ldx] = 0.0;
}

void scatter_values(hipStream_t, double *val, const int64_t *pos, const double *a, int64_t nnz)
{
    for (int64_t e = 0; e < nnz; ++e) val[pos[e]] = a[e];
}

void rfs_residual(hipStream_t, int n, const int *rp, const int *ci, const double *av, const double *x, const double *b, const int *pc,
                  double *r_perm, unsigned long long *s_out, double safe1, double safe2)
{
    double best = 0;
    for (int i = 0; i < n; ++i) {
        double ax = 0, t = 0;
        for (int e = rp[i]; e < rp[i + 1]; ++e) { ax += av[e] * x[ci[e]]; t += std::fabs(av[e]) * std::fabs(x[ci[e]]); }
        const double r = b[i] - ax;
        t += std::fabs(b[i]);
        r_perm[pc[i]] = r;
        double q = 0;
        if (t > safe2) q = std::fabs(r) / t; else if (t != 0.0) q = (safe1 + std::fabs(r)) / t;
        best = std::max(best, q);
    }
    double cur; std::memcpy(&cur, s_out, 8);
    if (best > cur) std::memcpy(s_out, &best, 8);
}

void rfs_update(hipStream_t, int n, const int *pc, const double *dx_perm, double *x)
{
    for (int i = 0; i < n; ++i) x[i] += dx_perm[pc[i]];
}

void add_atomic(hipStream_t, int64_t n, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) y[i] += x[i];
}

void axpy(hipStream_t, int64_t n, double a, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}

void pack_diag(hipStream_t, const DevTables &T, const int *nodes, const int *prefix, const int64_t *off, int nn, int nwork, double *stage, int vs)
{
    for (int w = 0; w < nwork; ++w) {
        const int ni = find_node(prefix, nn, w);
        const int k = nodes[ni], ns = T.xsup[k + 1] - T.xsup[k];
        const int e0 = (w - prefix[ni]) * 1024, e1 = std::min(e0 + 1024, ns * ns);
        const double *A = T.val + T.sn_dptr[k] * vs;
        const int lda = T.sn_dlda[k];
        double *S = stage + off[ni] * vs;
        for (int e = e0; e < e1; ++e) for (int v = 0; v < vs; ++v) S[(size_t) e * vs + v] = A[((e % ns) + (size_t) (e / ns) * lda) * vs + v];
    }
}

void xseg_copy(hipStream_t, double *x, int64_t ldx, int nrhs, const int *runs, int nruns, int64_t total, double *buf, int mode)
{
    for (int q = 0; q < nrhs; ++q)
        for (int r = 0; r < nruns; ++r)
            for (int i = 0; i < runs[3 * r + 1]; ++i) {
                double *xp = x + runs[3 * r] + i + (int64_t) q * ldx;
                double *bp = buf + runs[3 * r + 2] + i + (int64_t) q * total;
                if (mode == 0) *bp = *xp; else if (mode == 1) *xp = *bp; else if (mode == 2) *xp += *bp; else { *bp = *xp; *xp = 0.0; }
            }
}

void rows_copy(hipStream_t, double *v, int64_t ldv, int nrhs, const int *idx, int64_t cnt, double *buf, int mode, int vs)
{
    for (int q = 0; q < nrhs; ++q)
        for (int64_t j = 0; j < cnt; ++j)
            for (int t = 0; t < vs; ++t) {
                double *vp = v + (idx[j] + (int64_t) q * ldv) * vs + t;
                if (mode == 0) buf[(j + q * cnt) * vs + t] = *vp; else *vp = buf[(j + q * cnt) * vs + t];
            }
}

}  // namespace impl

int setup() { return 0; }

void diag_lu(hipStream_t s, const DevTables &T, const int *nodes, int nn, int max_nsupc, int flags, double thresh, int *info)
{
    emul_enqueue(s, [=] { impl::diag_lu(s, T, nodes, nn, max_nsupc, flags, thresh, info); });
}

void diag_inv(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int ntask)
{
    emul_enqueue(s, [=] { impl::diag_inv(s, T, nodes, prefix, nn, ntask); });
}

void panel_trsm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int rs, int max_nsupc, const int2 *units)
{
    emul_enqueue(s, [=] { impl::panel_trsm(s, T, nodes, lprefix, uprefix, nn, nl, nu, rs, max_nsupc, units); });
}

void panel_gemm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int max_nsupc, const int2 *units)
{
    emul_enqueue(s, [=] { impl::panel_gemm(s, T, nodes, lprefix, uprefix, nn, nl, nu, max_nsupc, units); });
}

void schur(hipStream_t s, int cfg, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info, const int4 *ulist, int prio,
           const int *, int mmode, int /* ksplit: shares of K per tile on the device; the restatement computes a tile once */, const int *, int /* XCD ranges: a device mapping */)
{
    if (mmode == 1) return;     // build pass of the per-tile records: the restatement reads the tables every time
    emul_enqueue(s, [=] { impl::schur(s, cfg, T, nodes, prefix, nn, id_base, ntiles, info, ulist, prio); });
}

void full_inv(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, int max_nsupc)
{
    emul_enqueue(s, [=] { impl::full_inv(s, T, nodes, prefix, nn, nwork, max_nsupc); });
}

void solve_diag(hipStream_t s, bool lower, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs, int max_nsupc)
{
    emul_enqueue(s, [=] { impl::solve_diag(s, lower, T, nodes, nn, x, ldx, nrhs, max_nsupc); });
}

void fwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xsrc, double *x, int64_t ldx, int nrhs, int max_nsupc, const int2 *units,
                const int4 *recs)      // unit records where the caller has them (as the kernel), the tables otherwise
{
    if (recs) { emul_enqueue(s, [=] { impl::units_by_records(true, T, recs, nwork, xsrc, x, ldx, nrhs); }); return; }
    emul_enqueue(s, [=] { impl::fwd_update(s, T, nodes, prefix, nn, nwork, xsrc, x, ldx, nrhs, max_nsupc, units); });
}

void bwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xcols, double *x, int64_t ldx, int nrhs, int max_nsupc, const int2 *units,
                const int4 *recs)
{
    if (recs) { emul_enqueue(s, [=] { impl::units_by_records(false, T, recs, nwork, xcols, x, ldx, nrhs); }); return; }
    emul_enqueue(s, [=] { impl::bwd_update(s, T, nodes, prefix, nn, nwork, xcols, x, ldx, nrhs, max_nsupc, units); });
}

void sweep_step(hipStream_t s, bool lower, const DevTables &T, const int2 *dunits, int ndu, const int2 *units, int nunits, double *xa, double *xb, int64_t ldx, int nrhs, int max_nsupc,
                const int4 *drecs, const int4 *urecs)
{
    if (ndu + nunits <= 0) return;
    emul_enqueue(s, [=] { impl::sweep_step(s, lower, T, dunits, ndu, units, nunits, xa, xb, ldx, nrhs, max_nsupc, urecs, drecs); });
}

void sweep_join(hipStream_t s, bool lower, const DevTables &T, const int4 *jrecs, int nj, const int4 *jaux, const int4 *urecs, int nunits, double *xa, double *xb,
                int64_t ldx, int nrhs, int max_nsupc)
{
    if (nj + nunits <= 0) return;
    emul_enqueue(s, [=] { impl::sweep_join(s, lower, T, jrecs, nj, jaux, urecs, nunits, xa, xb, ldx, nrhs, max_nsupc); });
}

void zero_nodes(hipStream_t s, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs)
{
    if (nn <= 0) return;
    emul_enqueue(s, [=] { impl::zero_nodes(s, T, nodes, nn, x, ldx, nrhs); });
}

void scatter_values(hipStream_t s, double *val, const int64_t *pos, const double *a, int64_t nnz)
{
    emul_enqueue(s, [=] { impl::scatter_values(s, val, pos, a, nnz); });
}

void rfs_residual(hipStream_t s, int n, const int *rp, const int *ci, const double *av, const double *x, const double *b, const int *pc, double *r_perm, unsigned long long *s_out, double safe1, double safe2)
{
    emul_enqueue(s, [=] { impl::rfs_residual(s, n, rp, ci, av, x, b, pc, r_perm, s_out, safe1, safe2); });
}

void rfs_update(hipStream_t s, int n, const int *pc, const double *dx_perm, double *x)
{
    emul_enqueue(s, [=] { impl::rfs_update(s, n, pc, dx_perm, x); });
}

void axpy(hipStream_t s, int64_t n, double a, const double *x, double *y)
{
    emul_enqueue(s, [=] { impl::axpy(s, n, a, x, y); });
}

void add_atomic(hipStream_t s, int64_t n, const double *x, double *y)
{
    if (n <= 0) return;
    emul_enqueue(s, [=] { impl::add_atomic(s, n, x, y); });
}

void pack_diag(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, const int64_t *off, int nn, int nwork, double *stage, int vs)
{
    emul_enqueue(s, [=] { impl::pack_diag(s, T, nodes, prefix, off, nn, nwork, stage, vs); });
}

void xseg_copy(hipStream_t s, double *x, int64_t ldx, int nrhs, const int *runs, int nruns, int64_t total, double *buf, int mode)
{
    emul_enqueue(s, [=] { impl::xseg_copy(s, x, ldx, nrhs, runs, nruns, total, buf, mode); });
}

void rows_copy(hipStream_t s, double *v, int64_t ldv, int nrhs, const int *idx, int64_t cnt, double *buf, int mode, int vs)
{
    if (cnt <= 0) return;
    emul_enqueue(s, [=] { impl::rows_copy(s, v, ldv, nrhs, idx, cnt, buf, mode, vs); });
}

int mfma_selftest(const double *A, const double *B, double *D)
{
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double a = 0; for (int k = 0; k < 4; ++k) a += A[i * 4 + k] * B[k * 16 + j]; D[i * 16 + j] = a; }
    return 0;
}

// ---- complex16 twins (sluamd_zkernels.inc; 1 x 1 layers): the generic pieces above instantiated for std::complex<double>, the
// panel kernels as plain substitutions on the factored diagonal block (the complex path has no inverted blocks) ----
namespace impl {
typedef std::complex<double> zc;
static zc z_div(zc a, zc b)   // slud_z_div (Smith), SRC/complex16/dcomplex_dist.c:29-58 -- as the device kernels divide
{
    double ratio, den, cr, ci;
    if (std::fabs(b.real()) <= std::fabs(b.imag())) {
        ratio = b.real() / b.imag(); den = b.imag() * (1 + ratio * ratio);
        cr = (a.real() * ratio + a.imag()) / den; ci = (a.imag() * ratio - a.real()) / den;
    } else {
        ratio = b.imag() / b.real(); den = b.real() * (1 + ratio * ratio);
        cr = (a.real() + a.imag() * ratio) / den; ci = (a.imag() - a.real() * ratio) / den;
    }
    return zc(cr, ci);
}

// Local_Zgstrf2 (pzgstrf2.c): unpivoted right-looking LU of the diagonal block; |re|+|im| < thresh with both parts non-zero ->
// (sign(re) thresh, 0); a zero pivot is reported in info[0] and leaves its column unscaled
static void zdiag_lu(const DevTables &T, const int *nodes, int nn, int replace_tiny, double thresh, int *info)
{
    for (int q = 0; q < nn; ++q) {
        const int k = nodes[q];
        if (!(T.sn_flags[k] & SNF_OWN_DIAG)) continue;
        const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst, lda = T.sn_dlda[k];
        zc *A = reinterpret_cast<zc *>(T.val) + T.sn_dptr[k];
        for (int j = 0; j < ns; ++j) {
            zc p = A[j + (size_t) j * lda];
            if (replace_tiny && (std::fabs(p.real()) + std::fabs(p.imag())) < thresh && p.real() != 0.0 && p.imag() != 0.0) {
                p = zc(p.real() < 0 ? -thresh : thresh, 0.0);
                A[j + (size_t) j * lda] = p; info[1] += 1;
            }
            const bool zero = p == zc(0.0, 0.0);
            if (zero) { info[0] = std::min(info[0], fst + j + 1); info[4] = std::max(info[4], fst + j + 1); }
            const zc rinv = zero ? zc(1.0, 0.0) : z_div(zc(1.0, 0.0), p);
            for (int i = j + 1; i < ns; ++i) {
                zc l = A[i + (size_t) j * lda];
                if (!zero) l *= rinv;
                A[i + (size_t) j * lda] = l;
                for (int c = j + 1; c < ns; ++c) A[i + (size_t) c * lda] -= l * A[j + (size_t) c * lda];
            }
        }
    }
}

// zLPanelTrSolve / zUPanelTrSolve (ztrfCommWrapper.c): L(off-diagonal rows, :) <- L inv(U_kk), U(k, :) <- inv(L_kk) U(k, :) on the skyline
static void zpanel_trsm(const DevTables &T, const int *nodes, int nn)
{
    for (int q = 0; q < nn; ++q) {
        const int k = nodes[q];
        const int ns = T.xsup[k + 1] - T.xsup[k], lda = T.sn_nsupr[k];
        zc *A = reinterpret_cast<zc *>(T.val) + T.sn_lval[k];
        const zc *D = reinterpret_cast<const zc *>(T.val) + T.sn_dptr[k];      // factored diagonal block: own slot or the received image
        const int ldd = T.sn_dlda[k];
        if (T.sn_flags[k] & SNF_L_OWN)
            for (int row = T.sn_ldiag[k]; row < lda; ++row)
                for (int j = 0; j < ns; ++j) {
                    zc acc = A[row + (size_t) j * lda];
                    for (int kk = 0; kk < j; ++kk) acc -= A[row + (size_t) kk * lda] * D[kk + (size_t) j * ldd];
                    A[row + (size_t) j * lda] = z_div(acc, D[j + (size_t) j * ldd]);
                }
        if (T.sn_flags[k] & SNF_U_OWN) {
            zc *Uv = reinterpret_cast<zc *>(T.val) + T.sn_uval[k];
            for (int c = 0; c < T.sn_ncolu[k]; ++c) {
                const int64_t ci = T.sn_ucol[k] + c;
                const int ld = T.ucol_ld[ci];
                zc *col = Uv + T.ucol_cp[ci] - ld;       // col[i] = U(i, column), rows ld .. ns-1
                for (int i = ld; i < ns; ++i) {
                    zc acc = col[i];
                    for (int kk = ld; kk < i; ++kk) acc -= D[i + (size_t) kk * ldd] * col[kk];
                    col[i] = acc;
                }
            }
        }
    }
}

static void zsolve_diag(bool lower, const DevTables &T, const int *nodes, int nn, zc *x, int64_t ldx, int nrhs)
{
    for (int q = 0; q < nn; ++q) {
        const int k = nodes[q];
        if (!(T.sn_flags[k] & SNF_OWN_DIAG)) continue;
        const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst, lda = T.sn_dlda[k];
        const zc *A = reinterpret_cast<const zc *>(T.val) + T.sn_dptr[k];
        for (int r = 0; r < nrhs; ++r) {
            zc *xk = x + fst + (int64_t) r * ldx;
            if (lower) { for (int j = 0; j < ns; ++j) for (int i = j + 1; i < ns; ++i) xk[i] -= A[i + (size_t) j * lda] * xk[j]; }
            else for (int j = ns - 1; j >= 0; --j) { xk[j] = z_div(xk[j], A[j + (size_t) j * lda]); for (int i = 0; i < j; ++i) xk[i] -= A[i + (size_t) j * lda] * xk[j]; }
        }
    }
}
}  // namespace impl

void zdiag_lu(hipStream_t s, const DevTables &T, const int *nodes, int nn, int, int replace_tiny, double thresh, int *info)
{
    emul_enqueue(s, [=] { impl::zdiag_lu(T, nodes, nn, replace_tiny, thresh, info); });
}
void zpanel_trsm(hipStream_t s, const DevTables &T, const int *nodes, const int *, const int *, int nn, int, int, int)
{
    emul_enqueue(s, [=] { impl::zpanel_trsm(T, nodes, nn); });
}
void zschur(hipStream_t s, int, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info, const int4 *ulist, int,
            const int *, int mmode, const int *, int)
{
    if (ntiles <= 0 || mmode == 1) return;
    emul_enqueue(s, [=] { impl::schur_t<impl::zc>(T, nodes, prefix, nn, id_base, ntiles, info, ulist); });
}
void zsolve_diag(hipStream_t s, bool lower, const DevTables &T, const int *nodes, int nn, void *x, int64_t ldx, int nrhs, int)
{
    emul_enqueue(s, [=] { impl::zsolve_diag(lower, T, nodes, nn, static_cast<impl::zc *>(x), ldx, nrhs); });
}
void zfwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, int64_t ldx, int nrhs, int)
{
    emul_enqueue(s, [=] { impl::fwd_update_t<impl::zc, 256>(T, nodes, prefix, nn, nwork, static_cast<const impl::zc *>(x), static_cast<impl::zc *>(x), ldx, nrhs, nullptr); });
}
void zbwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, int64_t ldx, int nrhs)
{
    emul_enqueue(s, [=] { impl::bwd_update_t<impl::zc>(T, nodes, prefix, nn, nwork, static_cast<const impl::zc *>(x), static_cast<impl::zc *>(x), ldx, nrhs, nullptr); });
}
// fused links of the complex16 sweeps: the same results as the two launches they replace, out of place (forward: y to w; backward: accumulators in w, solution to x)
void zsweep_fused(hipStream_t s, bool lower, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *xv, void *wv, int64_t ldx, int nrhs, int, int *)
{
    if (nwork <= 0) return;
    emul_enqueue(s, [=] {
        impl::zc *x = static_cast<impl::zc *>(xv), *w = static_cast<impl::zc *>(wv);
        auto copy_blocks = [&](const impl::zc *src, impl::zc *dst) {
            for (int i = 0; i < nn; ++i) {
                const int k = nodes[i];
                if (!(T.sn_flags[k] & SNF_OWN_DIAG)) continue;
                for (int r = 0; r < nrhs; ++r) for (int c = T.xsup[k]; c < T.xsup[k + 1]; ++c) dst[c + (int64_t) r * ldx] = src[c + (int64_t) r * ldx];
            }
        };
        // units of the fused prefixes: strip / chunk `u` of a supernode beyond its real count is the placeholder of a supernode without off-diagonal part
        std::vector<int> real(nn + 1, 0);
        if (lower) {
            copy_blocks(x, w);
            impl::zsolve_diag(true, T, nodes, nn, w, ldx, nrhs);
            for (int i = 0; i < nn; ++i) { const int k = nodes[i]; const int lrows = (T.sn_flags[k] & SNF_L_OWN) ? T.sn_nsupr[k] - T.sn_ldiag[k] : 0; real[i + 1] = real[i] + (lrows + 255) / 256; }
            impl::fwd_update_t<impl::zc, 256>(T, nodes, real.data(), nn, real[nn], w, x, ldx, nrhs, nullptr);
        } else {
            for (int i = 0; i < nn; ++i) { const int k = nodes[i]; const int ucols = (T.sn_flags[k] & SNF_U_OWN) ? T.sn_ncolu[k] : 0; real[i + 1] = real[i] + (ucols + 63) / 64; }
            impl::bwd_update_t<impl::zc>(T, nodes, real.data(), nn, real[nn], x, w, ldx, nrhs, nullptr);
            copy_blocks(w, x);
            impl::zsolve_diag(false, T, nodes, nn, x, ldx, nrhs);
        }
        (void) prefix;
    });
}
// merged chain groups: the same gather and products as the device kernels, element by element
void grp_gather(hipStream_t s, const DevTables &T, const GrpDesc *gd, double *scr)
{
    emul_enqueue(s, [=] {
        const GrpDesc g = *gd;
        const int nG = g.nG;
        double *LinvG = T.inv + g.ginv, *UinvG = LinvG + (int64_t) nG * nG, *LG = scr, *UG = scr + GRP_SCR;
        for (int t = 0; t < g.nm; ++t) {
            const int k = g.k[t], o = g.o[t], w = g.w[t];
            const double *Li = T.inv + T.sn_inv[k], *Ui = Li + (int64_t) w * w;
            for (int c = 0; c < w; ++c) for (int r = 0; r < w; ++r) {
                LinvG[(o + r) + (int64_t) (o + c) * nG] = Li[r + (int64_t) c * w];
                UinvG[(o + r) + (int64_t) (o + c) * nG] = Ui[r + (int64_t) c * w];
            }
        }
        for (int i = 1; i < g.nm; ++i)
            for (int kk = 0; kk < i; ++kk) {
                {   // L block of member i in panel kk
                    const int mi = g.k[i], mk = g.k[kk], oi = g.o[i], ok = g.o[kk], wk = g.w[kk];
                    const int lb0 = T.sn_lb_off[mk], nb = T.sn_nlb[mk];
                    for (int q = 0; q < nb; ++q) {
                        if (T.lb_gid[lb0 + q] != mi) continue;
                        const int b = lb0 + q, nbrow = T.lb_nbrow[b], ro = T.lb_rowoff[b], lda = T.sn_nsupr[mk], f = T.xsup[mi];
                        for (int c = 0; c < wk; ++c) for (int r = 0; r < nbrow; ++r)
                            LG[(oi + T.lrow[T.sn_lrow[mk] + ro + r] - f) + (int64_t) (ok + c) * nG] = T.val[T.sn_lval[mk] + ro + r + (int64_t) c * lda];
                    }
                }
                {   // U block (row member kk, column member i)
                    const int mr = g.k[kk], mc = g.k[i], orr = g.o[kk], oc = g.o[i], wr = g.w[kk];
                    const int ub0 = T.sn_ub_off[mr], nub = T.sn_nub[mr];
                    for (int q = 0; q < nub; ++q) {
                        if (T.ub_gid[ub0 + q] != mc) continue;
                        const int b = ub0 + q, ncol = T.ub_ncols[b], st = T.ub_stcol[b], f = T.xsup[mc];
                        for (int cq = 0; cq < ncol; ++cq) {
                            const int64_t ci = T.sn_ucol[mr] + st + cq;
                            const int ld = T.ucol_ld[ci];
                            for (int r = ld; r < wr; ++r) UG[(orr + r) + (int64_t) (oc + T.ucol_gc[ci] - f) * nG] = T.val[T.sn_uval[mr] + T.ucol_cp[ci] + (r - ld)];
                        }
                    }
                }
            }
    });
}
void gemm_batched(hipStream_t s, const DevTables &T, const GemmDesc *descs, const int4 *tiles, int ntiles, double *scr)
{
    if (ntiles <= 0) return;
    emul_enqueue(s, [=] {
        for (int t = 0; t < ntiles; ++t) {
            const int4 tl = tiles[t];
            const GemmDesc d = descs[tl.x];
            const double *A = (d.abase ? scr : T.inv) + d.a, *B = (d.bbase ? scr : T.inv) + d.b;
            double *C = (d.cbase ? scr : T.inv) + d.c;
            for (int col = 32 * tl.z; col < std::min(d.N, 32 * tl.z + 32); ++col)
                for (int row = 32 * tl.y; row < std::min(d.M, 32 * tl.y + 32); ++row) {
                    double acc = 0.0;
                    for (int k = 0; k < d.K; ++k) acc += A[row + (int64_t) k * d.lda] * B[k + (int64_t) col * d.ldb];
                    C[row + (int64_t) col * d.ldc] = d.neg ? -acc : acc;
                }
        }
    });
}
void zscatter_values(hipStream_t s, void *val, const int64_t *pos, const void *a, int64_t nnz)
{
    emul_enqueue(s, [=] {
        impl::zc *v = static_cast<impl::zc *>(val); const impl::zc *av = static_cast<const impl::zc *>(a);
        for (int64_t e = 0; e < nnz; ++e) v[pos[e]] = av[e];
    });
}

}  // namespace eng
}  // namespace sluamd
