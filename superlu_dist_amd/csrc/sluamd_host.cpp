// sluamd_host.cpp -- host side of the MI355X (gfx950) implementation of the 3D supernodal LU hot path.
//
// Flattens the caller's reference-format L/U store (superlu_dist_amd.h) into per-supernode "slots", uploads it ONCE to
// HBM (index arenas + value arena stay resident for factor and solve), builds device-side block directories, tile lists,
// an elimination-DAG level schedule per Z level and -- on XY block-cyclic layers -- the per-level panel exchange plan.
// The numeric drivers (pdgstrf3d, pdgstrs3d, pdgsrfs3d) queue kernels through the eng:: interface (sluamd_kernels.hip)
// and panel exchanges through Comm (sluamd_comm.h).
//
// There is NO CPU fallback in the product library: every entry point fails when no HIP device is present.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>
#include "sluamd_comm.h"
#include "sluamd_internal.h"
#include "sluamd_plan.h"

namespace sluamd {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
double SetupTimer::now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
const std::string &get_error() { return g_err; }

int check_device(int dev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible: this library has no CPU fallback"); return SLUAMD_ENODEVICE; }
    if (dev >= 0) { HIPCHK(hipSetDevice(dev)); }
    return 0;
}

void read_env(Handle::Env &e)
{
    e.no_lookahead = getenv("SLUAMD_NO_LOOKAHEAD") != nullptr;
    e.no_tile_maps = getenv("SLUAMD_NO_TILE_MAPS") != nullptr;
    if (const char *il = getenv("SLUAMD_INFO_LAST")) e.info_last = atoi(il) != 0;   // overrides sluamd_options_t::info_rule (the callers set e.info_last from it first)
    if (const char *sg = getenv("SLUAMD_SOLVE_GROUPS")) e.solve_groups = atoi(sg) != 0;
    if (const char *sg = getenv("SLUAMD_SOLVE_GROUP_LEVEL_NODES")) e.solve_group_level_nodes = std::max(1, atoi(sg));
    if (const char *zf = getenv("SLUAMD_ZFUSE_MAX_NODES")) e.z_fuse_max_nodes = std::max(0, atoi(zf));
    if (const char *sp = getenv("SLUAMD_PANEL_SPLIT")) e.panel_split_max_nodes = std::max(0, atoi(sp));
    e.no_level_split = getenv("SLUAMD_NO_LEVEL_SPLIT") != nullptr;
    e.no_merge_tiles = getenv("SLUAMD_NO_MERGE_TILES") != nullptr;
    if (const char *v = getenv("SLUAMD_FUSE_TAIL_GUARD")) e.fuse_tail_guard = atoi(v);
    if (const char *v = getenv("SLUAMD_FUSE_GROUP_MIN_NODES")) e.fuse_group_min_nodes = atoi(v);
    if (const char *v = getenv("SLUAMD_FUSE_SMALL")) e.fuse_small = atoi(v) != 0;
    if (const char *v = getenv("SLUAMD_SOLVE_JOIN")) e.solve_join = atoi(v) != 0;
    if (const char *v = getenv("SLUAMD_SORT_BLOCK_ROWS")) e.sort_block_rows = atoi(v) != 0;
    if (const char *v = getenv("SLUAMD_JOIN_MAX_NODES")) e.join_max_nodes = atoi(v);
    if (const char *v = getenv("SLUAMD_KSPLIT")) e.ksplit = std::max(1, std::min(16, atoi(v)));
    if (const char *v = getenv("SLUAMD_BIG_UTIL_PCT")) e.big_util_pct = atoi(v);
    if (const char *v = getenv("SLUAMD_BIG_MIN_COLS")) e.big_min_cols = atoi(v);
    if (const char *v = getenv("SLUAMD_LEVEL_SPLIT_MIN")) e.level_split_min = atoi(v);
    if (const char *v = getenv("SLUAMD_LEVEL_SPLIT_WDIV")) e.level_split_wdiv = atoi(v);
    if (const char *v = getenv("SLUAMD_LEVEL_SPLIT_WMIN")) e.level_split_wmin = atof(v);
    e.no_fuse = getenv("SLUAMD_NO_FUSE") != nullptr;
    e.no_big_tiles = getenv("SLUAMD_NO_BIG_TILES") != nullptr;
    e.schur_4waves = getenv("SLUAMD_SCHUR_4WAVES") != nullptr;
    e.trsm_rs32 = getenv("SLUAMD_TRSM_RS32") != nullptr;
    e.profile = getenv("SLUAMD_PROFILE") != nullptr;
    e.profile_dump = getenv("SLUAMD_PROFILE_DUMP") != nullptr;   // per-launch Schur table on stderr after a profiled factorisation
    e.diag_v1 = getenv("SLUAMD_DIAG_V1") != nullptr;             // round-1 right-looking diagonal LU kernel
    e.trsm_panels = getenv("SLUAMD_TRSM_PANELS") != nullptr;    // blocked-substitution panel kernels instead of the GEMM form (1 x 1 layers)
    if (const char *v = getenv("SLUAMD_TRSM_TAIL")) e.trsm_tail = atoi(v);
    if (const char *v = getenv("SLUAMD_DIAG_TAIL")) e.diag_tail = atoi(v);
    if (const char *v = getenv("SLUAMD_FUSE_MIN_PCT")) e.fuse_min_pct = atoi(v);
    if (const char *v = getenv("SLUAMD_FUSE_MAX_PREV")) e.fuse_max_prev = std::max(1, std::min(3, atoi(v)));
    if (const char *v = getenv("SLUAMD_RESERVE_CUS")) e.reserve_cus = std::max(0, atoi(v));
    if (const char *v = getenv("SLUAMD_BALANCE_MIN_TILES")) e.balance_min_tiles = std::max(0, atoi(v));
    if (const char *v = getenv("SLUAMD_BALANCE_OVH")) e.balance_ovh = atof(v);
}

int trsm_rs(const Handle &H, int nsp) { return (nsp > 128 && H.env.trsm_rs32) ? 32 : 64; }

// ================================================================================================
//                           slot structures from the caller's view
// ================================================================================================
// Z-level node lists of this layer's path: forests (dtrf3Dpartition_t) or one list with every supernode
static int lists_from_forests(const sluamd_forest_view_t *forests, int nsupers, int npdep, SlotInput &in)
{
    in.lists.clear(); in.z_active.clear();
    if (forests && forests->maxLvl > 0 && forests->nodeList) {
        for (int l = 0; l < forests->maxLvl; ++l) {
            std::vector<int> v;
            const int f = forests->myTreeIdxs[l];
            if (f >= 0 && f < forests->numForests && forests->nNodes[f] > 0 && forests->nodeList[f])
                v.assign(forests->nodeList[f], forests->nodeList[f] + forests->nNodes[f]);
            std::sort(v.begin(), v.end());
            in.lists.push_back(std::move(v));
            in.z_active.push_back(forests->myZeroTrIdxs[l] ? 0 : 1);
        }
    } else {
        if (npdep > 1) { set_error("npdep > 1 needs the elimination forests (sluamd_forest_view_t) of this layer"); return SLUAMD_EINVAL; }
        std::vector<int> v(nsupers);
        std::iota(v.begin(), v.end(), 0);
        in.lists.push_back(std::move(v));
        in.z_active.push_back(1);
    }
    for (auto &l : in.lists)
        for (int k : l)
            if (k < 0 || k >= nsupers) { set_error("forest node list names a supernode outside [0, nsupers)"); return SLUAMD_ESTRUCT; }
    return 0;
}

static void add_succ_from_slot(SlotInput &in, int k, const std::vector<int> &li, const std::vector<int> &ui, int nsupers, const int *xsup)
{
    auto &s = in.succ[k];
    if (li.size() >= (size_t) BC_HEADER) {
        int p = BC_HEADER;
        for (int b = 0; b < li[0] && p + 1 < (int) li.size(); ++b) { if (li[p] > k) s.push_back(li[p]); p += LB_DESCRIPTOR + li[p + 1]; }
    }
    if (ui.size() >= (size_t) BR_HEADER) {
        int p = BR_HEADER;
        for (int b = 0; b < ui[0] && p < (int) ui.size(); ++b) {
            const int jb = ui[p];
            if (jb < 0 || jb >= nsupers) break;
            s.push_back(jb);
            p += UB_DESCRIPTOR + (xsup[jb + 1] - xsup[jb]);
        }
    }
}

static void finish_succ(SlotInput &in)
{
    for (auto &s : in.succ) { std::sort(s.begin(), s.end()); s.erase(std::unique(s.begin(), s.end()), s.end()); }
}

// message = [count][k, len, ints...]*
static void pack_slots(const std::vector<std::vector<int>> &idx, const std::vector<int> &ks, std::vector<int> &msg)
{
    msg.clear();
    msg.push_back((int) ks.size());
    for (int k : ks) { msg.push_back(k); msg.push_back((int) idx[k].size()); msg.insert(msg.end(), idx[k].begin(), idx[k].end()); }
}
static int unpack_slots(const std::vector<int> &msg, std::vector<std::vector<int>> &idx, int nsupers)
{
    size_t p = 0;
    if (msg.empty()) return 0;
    const int cnt = msg[p++];
    for (int i = 0; i < cnt; ++i) {
        if (p + 2 > msg.size()) { set_error("truncated structure message"); return SLUAMD_ESTRUCT; }
        const int k = msg[p++], len = msg[p++];
        if (k < 0 || k >= nsupers || len < 0 || p + len > msg.size()) { set_error("bad structure message"); return SLUAMD_ESTRUCT; }
        idx[k].assign(msg.begin() + p, msg.begin() + p + len);
        p += len;
    }
    return 0;
}

// one grouped exchange of variable-length int messages: sends[i] -> peers_s[i]; receives one message from every peers_r[i]
static int exchange_ints(Comm *comm, const std::vector<int> &peers_s, const std::vector<const std::vector<int> *> &sends,
                         const std::vector<int> &peers_r, std::vector<std::vector<int>> &recvs)
{
    std::vector<int64_t> slen(peers_s.size()), rlen(peers_r.size(), 0);
    int rc = comm->hbegin();
    if (rc) return rc;
    for (size_t i = 0; i < peers_s.size(); ++i) { slen[i] = (int64_t) sends[i]->size(); if ((rc = comm->hsend(&slen[i], sizeof(int64_t), peers_s[i]))) return rc; }
    for (size_t i = 0; i < peers_r.size(); ++i) if ((rc = comm->hrecv(&rlen[i], sizeof(int64_t), peers_r[i]))) return rc;
    if ((rc = comm->hend())) return rc;
    recvs.resize(peers_r.size());
    if ((rc = comm->hbegin())) return rc;
    for (size_t i = 0; i < peers_s.size(); ++i) if ((rc = comm->hsend(sends[i]->data(), slen[i] * (int64_t) sizeof(int), peers_s[i]))) return rc;
    for (size_t i = 0; i < peers_r.size(); ++i) { recvs[i].resize((size_t) rlen[i]); if ((rc = comm->hrecv(recvs[i].data(), rlen[i] * (int64_t) sizeof(int), peers_r[i]))) return rc; }
    return comm->hend();
}

// Wide supernodes (257..512 columns) of a 1 x 1 layer -> chains of <= 256-column pieces (SplitMap): rewrites the slot index
// arrays, the block graph and the forest lists in the internal numbering and records where every internal slot's values live
// inside the caller's arrays.
static int split_wide_supernodes(Handle &H, SlotInput &in)
{
    HostStruct &hs = H.hs;
    const int nso = hs.nsupers;
    const std::vector<int> ox = hs.xsup;
    int wmax = 0;
    for (int k = 0; k < nso; ++k) wmax = std::max(wmax, ox[k + 1] - ox[k]);
    if (wmax <= 256) return 0;
    if (wmax > 512) { set_error("supernodes wider than 512 columns (MAX_SUPER_SIZE) are not supported"); return SLUAMD_EINVAL; }
    // XY layers: the pieces of a supernode stay with its owners (Grid::own) -- every rank refines the slots it works with (own and
    // received index arrays alike) by the same rule, so senders and receivers of a panel agree on the refined layout
    const bool xy = H.grid.Pr * H.grid.Pc > 1;
    const Grid g0 = H.grid;                    // ownership of the caller's (unrefined) supernodes
    // complex16: the pieces are ordinary supernodes to the complex kernels (values move as 16-byte elements)
    SplitMap &M = H.split;
    M.active = true; M.oxsup = ox; M.first.assign(nso + 1, 0); M.owner.clear();
    std::vector<int> nx(1, 0);                 // internal xsup
    for (int k = 0; k < nso; ++k) {
        const int w = ox[k + 1] - ox[k];
        M.first[k] = (int) nx.size() - 1;
        const int np = (w + 255) / 256;
        const int pw = (((w + np - 1) / np) + 31) & ~31;           // piece width: even split rounded up to the 32-column blocking
        for (int c = 0; c < w; c += pw) { nx.push_back(ox[k] + std::min(w, c + pw)); M.owner.push_back(k); }
    }
    M.first[nso] = (int) nx.size() - 1;
    const int ns = M.first[nso];
    std::vector<std::vector<int>> nl(ns), nu(ns), nsucc(ns);
    std::vector<uint8_t> npresent(ns, 0);
    M.lsrc.assign(ns, {}); M.usrc.assign(ns, {});
    for (int ko = 0; ko < nso; ++ko) {
        if (!hs.present[ko]) continue;
        const std::vector<int> &li = in.lidx[ko], &ui = in.uidx[ko];
        // this rank's part of the panel: the rows of its process row -- with the diagonal block on top on the process row of ko
        const bool has_l = li.size() >= (size_t) BC_HEADER && li[0] > 0;
        const bool has_diag = has_l && li[BC_HEADER] == ko;
        const int x0 = ox[ko], w = ox[ko + 1] - x0, nsupr_o = has_l ? li[1] : 0;
        if (g0.krow(ko) == g0.r && !has_diag) { set_error("L panel with the diagonal block missing"); return SLUAMD_ESTRUCT; }
        if (has_diag && (g0.krow(ko) != g0.r || li[BC_HEADER + 1] != w)) { set_error("diagonal block must be the first L block of its panel"); return SLUAMD_ESTRUCT; }
        const int f = M.first[ko], np = M.first[ko + 1] - f;
        // original U row: column start offsets
        std::vector<int64_t> ucol0;    // per (block, jj) flattened in walk order: value offset of the column's segment
        for (int p = 0; p < np; ++p) {
            const int id = f + p, c0 = nx[id] - x0, c1 = nx[id + 1] - x0, h = c1 - c0;
            npresent[id] = 1;
            // ---- L slot: rows of the original panel from diagonal row c0 on, columns [c0, c1) ----
            std::vector<int> &o = nl[id];
            o.assign(BC_HEADER, 0);
            int nb = 0, nr = 0;
            for (int q = p; q < np && has_diag; ++q) {           // the rest of the original diagonal block, piece by piece
                const int r0 = nx[f + q], r1 = nx[f + q + 1];
                o.push_back(f + q); o.push_back(r1 - r0);
                for (int r = r0; r < r1; ++r) o.push_back(r);
                ++nb; nr += r1 - r0;
            }
            // host row (position inside the caller's panel) of every row of the internal slot: the rest of the diagonal block in
            // place, then every off-diagonal block split by the pieces of its supernode (stable: the reference does not sort the
            // rows inside a block, symbfact.c keeps discovery order)
            std::vector<int> hostrow(nr);
            for (int i = 0; i < nr; ++i) hostrow[i] = c0 + i;
            int pp = BC_HEADER + (has_diag ? LB_DESCRIPTOR + w : 0), rowbase = has_diag ? w : 0;
            for (int b = has_diag ? 1 : 0; b < (has_l ? li[0] : 0); ++b) {
                const int g = li[pp], nbrow = li[pp + 1];
                const int *rows = li.data() + pp + LB_DESCRIPTOR;
                for (int q = M.first[g]; q < M.first[g + 1]; ++q) {
                    int cnt = 0;
                    for (int i = 0; i < nbrow; ++i) cnt += rows[i] >= nx[q] && rows[i] < nx[q + 1];
                    if (!cnt) continue;
                    o.push_back(q); o.push_back(cnt);
                    for (int i = 0; i < nbrow; ++i) if (rows[i] >= nx[q] && rows[i] < nx[q + 1]) { o.push_back(rows[i]); hostrow.push_back(rowbase + i); }
                    nsucc[id].push_back(q);
                    ++nb; nr += cnt;
                }
                pp += LB_DESCRIPTOR + nbrow; rowbase += nbrow;
            }
            o[0] = nb; o[1] = nr;
            if (nr != (has_diag ? nsupr_o - c0 : nsupr_o) || (int) hostrow.size() != nr) { set_error("L panel row count mismatch"); return SLUAMD_ESTRUCT; }
            if (!nb) o.clear();
            for (int j = c0; j < c1; ++j)
                for (int i = 0; i < nr;) {       // maximal runs of consecutive host rows
                    int e = i + 1;
                    while (e < nr && hostrow[e] == hostrow[e - 1] + 1) ++e;
                    M.lsrc[id].push_back({0, ko, (int64_t) j * nsupr_o + hostrow[i], (int64_t) (e - i)});
                    i = e;
                }
            for (int q = p + 1; q < np; ++q) nsucc[id].push_back(f + q);
            // ---- U slot: rows [x0 + c0, x0 + c1) ----
            std::vector<int> &u = nu[id];
            u.assign(BR_HEADER, 0);
            int nub = 0; int64_t nnz = 0;
            const int klst = x0 + c1, row0 = x0 + c0;
            for (int q = p + 1; q < np && g0.kcol(ko) == g0.c; ++q) {   // U(piece p, piece q): inside the original diagonal block, full segments (process column of ko)
                const int cq0 = nx[f + q] - x0, cq1 = nx[f + q + 1] - x0;
                u.push_back(f + q); u.push_back(h * (cq1 - cq0));
                for (int j = cq0; j < cq1; ++j) { u.push_back(row0); M.usrc[id].push_back({0, ko, (int64_t) j * nsupr_o + c0, (int64_t) h}); }
                ++nub; nnz += (int64_t) h * (cq1 - cq0);
            }
            if (ui.size() >= (size_t) BR_HEADER) {
                int iukp = BR_HEADER; int64_t rukp = 0;
                for (int b = 0; b < ui[0]; ++b) {
                    const int jbo = ui[iukp], wj = ox[jbo + 1] - ox[jbo];
                    for (int q = M.first[jbo]; q < M.first[jbo + 1]; ++q) {
                        const int cq0 = nx[q] - ox[jbo], cq1 = nx[q + 1] - ox[jbo];
                        std::vector<int> fst(cq1 - cq0);
                        std::vector<SplitMap::Piece> pcs;
                        int bn = 0;
                        int64_t off = rukp;
                        for (int jj = 0; jj < cq0; ++jj) off += ox[ko + 1] - ui[iukp + UB_DESCRIPTOR + jj];
                        for (int jj = cq0; jj < cq1; ++jj) {
                            const int fo = ui[iukp + UB_DESCRIPTOR + jj], sego = ox[ko + 1] - fo;
                            const int fn = std::max(fo, row0);
                            const int seg = std::max(0, klst - fn);
                            fst[jj - cq0] = seg ? fn : klst;
                            if (seg) { pcs.push_back({1, ko, off + (fn - fo), (int64_t) seg}); bn += seg; }
                            off += sego;
                        }
                        if (bn) {
                            u.push_back(q); u.push_back(bn);
                            u.insert(u.end(), fst.begin(), fst.end());
                            M.usrc[id].insert(M.usrc[id].end(), pcs.begin(), pcs.end());
                            nsucc[id].push_back(q);
                            ++nub; nnz += bn;
                        }
                    }
                    for (int jj = 0; jj < wj; ++jj) rukp += ox[ko + 1] - ui[iukp + UB_DESCRIPTOR + jj];
                    iukp += UB_DESCRIPTOR + wj;
                }
            }
            if (nnz > 0x7fffffff) { set_error("U block row too large"); return SLUAMD_ESTRUCT; }
            u[0] = nub; u[1] = (int) nnz; u[2] = (int) u.size();
            if (!nub) u.clear();
        }
    }
    if (xy)   // the block graph must be the same on every rank: all pieces of every successor (a superset of the true refined graph)
        for (int ko = 0; ko < nso; ++ko) {
            if (!hs.present[ko]) continue;
            for (int id = M.first[ko]; id < M.first[ko + 1]; ++id) {
                nsucc[id].clear();
                for (int q = id + 1; q < M.first[ko + 1]; ++q) nsucc[id].push_back(q);
                for (int gk : in.succ[ko]) for (int q = M.first[gk]; q < M.first[gk + 1]; ++q) nsucc[id].push_back(q);
            }
        }
    for (auto &v : nsucc) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    for (auto &l : in.lists) {
        std::vector<int> e;
        for (int ko : l) for (int q = M.first[ko]; q < M.first[ko + 1]; ++q) e.push_back(q);
        l.swap(e);
    }
    in.lidx.swap(nl); in.uidx.swap(nu); in.succ.swap(nsucc);
    hs.nsupers = ns; hs.xsup = nx; hs.present = npresent;
    H.grid.own = M.owner.data();
    return 0;
}

// rows of every block of an Lrowind image into ascending order; perm[internal slot row] = the caller's slot row (left empty when nothing moved)
static void sort_block_rows(std::vector<int> &li, std::vector<int> &perm)
{
    perm.clear();
    const int nb = li[0], nsupr = li[1];
    std::vector<int> order;
    int p = BC_HEADER, r0 = 0;
    for (int b = 0; b < nb; ++b) {
        const int nr = li[p + 1];
        int *rows = li.data() + p + LB_DESCRIPTOR;
        if (!std::is_sorted(rows, rows + nr)) {
            if (perm.empty()) { perm.resize(nsupr); std::iota(perm.begin(), perm.end(), 0); }
            order.resize(nr); std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return rows[a] < rows[c]; });
            std::vector<int> sorted(nr);
            for (int i = 0; i < nr; ++i) { sorted[i] = rows[order[i]]; perm[r0 + i] = r0 + order[i]; }
            std::copy(sorted.begin(), sorted.end(), rows);
        }
        p += LB_DESCRIPTOR + nr; r0 += nr;
    }
}

int slots_from_view(Handle &H, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests, Comm *comm, SlotInput &in)
{
    if (!lu || !lu->xsup || lu->nsupers <= 0) { set_error("invalid LU view"); return SLUAMD_EINVAL; }
    Grid &g = H.grid;
    g = Grid{lu->nprow, lu->npcol, lu->npdep, lu->myrow, lu->mycol, lu->myzlayer};
    if (g.Pr < 1 || g.Pc < 1 || g.Pz < 1 || g.r < 0 || g.r >= g.Pr || g.c < 0 || g.c >= g.Pc || g.z < 0 || g.z >= g.Pz) { set_error("bad grid coordinates in the LU view"); return SLUAMD_EINVAL; }
    if (g.size() > 1 && !comm) { set_error("a process grid with more than one rank needs a communicator: use sluamd_dCreateLUHandleGrid"); return SLUAMD_EINVAL; }
    if (comm) {
        const Grid &cg = comm->grid;
        if (cg.Pr != g.Pr || cg.Pc != g.Pc || cg.Pz != g.Pz || cg.r != g.r || cg.c != g.c || cg.z != g.z) { set_error("communicator grid does not match the LU view's grid"); return SLUAMD_EINVAL; }
    }
    const int ns = lu->nsupers;
    HostStruct &hs = H.hs;
    hs.n = lu->n; hs.nsupers = ns;
    hs.xsup.assign(lu->xsup, lu->xsup + ns + 1);
    int rc = lists_from_forests(forests, ns, g.Pz, in);
    if (rc) return rc;
    hs.present.assign(ns, 0);
    for (auto &l : in.lists) for (int k : l) hs.present[k] = 1;
    in.lidx.assign(ns, {}); in.uidx.assign(ns, {}); in.succ.assign(ns, {});
    // rows inside the L blocks ascending (HostStruct::lrow_perm) -- unless a supernode is wider than 256 columns: the refinement into pieces (SplitMap) gathers
    // from the caller's layout by runs of its own.  xsup is global, so every rank of the grid decides alike and the peers receive sorted index images.
    bool sort_rows = H.env.sort_block_rows;
    for (int k = 0; k < ns && sort_rows; ++k) if (hs.xsup[k + 1] - hs.xsup[k] > 256) sort_rows = false;
    hs.lrow_perm.assign(ns, {});
    std::vector<int> own_l, own_u;
    for (int k = 0; k < ns; ++k) {
        if (!hs.present[k]) continue;
        if (g.kcol(k) == g.c) {
            const int *li = lu->Lrowind_bc_ptr[k / g.Pc];
            if (li) {
                if (li[0] < 0 || li[1] < 0) { set_error("malformed L block column header"); return SLUAMD_ESTRUCT; }
                in.lidx[k].assign(li, li + BC_HEADER + (int64_t) li[0] * LB_DESCRIPTOR + li[1]);
                if (sort_rows) sort_block_rows(in.lidx[k], hs.lrow_perm[k]);
            } else if (g.krow(k) == g.r) { set_error("L panel with the diagonal block missing on its owner"); return SLUAMD_ESTRUCT; }
            own_l.push_back(k);
        }
        if (g.krow(k) == g.r) {
            const int *ui = lu->Ufstnz_br_ptr[k / g.Pr];
            if (ui) {
                if (ui[2] < BR_HEADER) { set_error("malformed U block row header"); return SLUAMD_ESTRUCT; }
                in.uidx[k].assign(ui, ui + ui[2]);
            }
            own_u.push_back(k);
        }
    }
    for (int k = 0; k < ns; ++k) if (hs.present[k]) add_succ_from_slot(in, k, in.lidx[k], in.uidx[k], ns, hs.xsup.data());
    if (g.Pr * g.Pc > 1) {
        // the index arrays of the panels this rank will receive (the reference ships them inside every panel message,
        // dIBcast_LPanel dcommunication_aux.c:32-60) + the block graph of the other parts, once
        std::vector<int> lmsg, umsg, gmsg;
        pack_slots(in.lidx, own_l, lmsg);
        pack_slots(in.uidx, own_u, umsg);
        {   // block graph of my own slots: [count][k, n, gids...]*
            std::vector<std::vector<int>> mine(ns);
            std::vector<int> ks;
            SlotInput tmp; tmp.succ.assign(ns, {});
            for (int k : own_l) add_succ_from_slot(tmp, k, in.lidx[k], {}, ns, hs.xsup.data());
            for (int k : own_u) add_succ_from_slot(tmp, k, {}, in.uidx[k], ns, hs.xsup.data());
            finish_succ(tmp);
            for (int k = 0; k < ns; ++k) if (!tmp.succ[k].empty()) { ks.push_back(k); mine[k] = tmp.succ[k]; }
            pack_slots(mine, ks, gmsg);
        }
        std::vector<int> ps, pr;
        std::vector<const std::vector<int> *> sends;
        std::vector<int> kind;   // of each receive: 0 = L slots, 1 = U slots, 2 = graph
        for (int c2 = 0; c2 < g.Pc; ++c2) if (c2 != g.c) { ps.push_back(g.rank_of(g.r, c2, g.z)); sends.push_back(&lmsg); }
        for (int r2 = 0; r2 < g.Pr; ++r2) if (r2 != g.r) { ps.push_back(g.rank_of(r2, g.c, g.z)); sends.push_back(&umsg); }
        for (int r2 = 0; r2 < g.Pr; ++r2) for (int c2 = 0; c2 < g.Pc; ++c2) if (r2 != g.r || c2 != g.c) { ps.push_back(g.rank_of(r2, c2, g.z)); sends.push_back(&gmsg); }
        for (int c2 = 0; c2 < g.Pc; ++c2) if (c2 != g.c) { pr.push_back(g.rank_of(g.r, c2, g.z)); kind.push_back(0); }
        for (int r2 = 0; r2 < g.Pr; ++r2) if (r2 != g.r) { pr.push_back(g.rank_of(r2, g.c, g.z)); kind.push_back(1); }
        for (int r2 = 0; r2 < g.Pr; ++r2) for (int c2 = 0; c2 < g.Pc; ++c2) if (r2 != g.r || c2 != g.c) { pr.push_back(g.rank_of(r2, c2, g.z)); kind.push_back(2); }
        std::vector<std::vector<int>> recvs;
        if ((rc = exchange_ints(comm, ps, sends, pr, recvs))) return rc;
        std::vector<std::vector<int>> gr(ns);
        for (size_t i = 0; i < recvs.size(); ++i) {
            if (kind[i] == 0) rc = unpack_slots(recvs[i], in.lidx, ns);
            else if (kind[i] == 1) rc = unpack_slots(recvs[i], in.uidx, ns);
            else {
                for (auto &v : gr) v.clear();
                rc = unpack_slots(recvs[i], gr, ns);
                for (int k = 0; k < ns && !rc; ++k) in.succ[k].insert(in.succ[k].end(), gr[k].begin(), gr[k].end());
            }
            if (rc) return rc;
        }
    }
    finish_succ(in);
    return split_wide_supernodes(H, in);
}

// ================================================================================================
//                           slot structures from the library's own symbolic factorisation
// ================================================================================================
int slots_from_symb(Handle &H, const Symb &sy, const Grid &g, const int32_t *sn_tree, SlotInput &in)
{
    const HostStruct &full = sy.hs;
    const int ns = full.nsupers;
    H.grid = g;
    HostStruct &hs = H.hs;
    hs.n = full.n; hs.nsupers = ns; hs.xsup = full.xsup;
    in.lists.clear(); in.z_active.clear();
    if (g.Pz == 1) {
        std::vector<int> v(ns); std::iota(v.begin(), v.end(), 0);
        in.lists.push_back(std::move(v)); in.z_active.push_back(1);
    } else {
        if (!sn_tree) { set_error("npdep > 1 needs sn_tree (sluamd_symb_partition)"); return SLUAMD_EINVAL; }
        // tree ids on the path of layer z (getGridTrees, supernodal_etree.c:840-851): leaf Pz - 1 + z, then parents
        int maxLvl = 1;
        while ((1 << (maxLvl - 1)) < g.Pz) ++maxLvl;
        std::vector<int> trees(maxLvl);
        trees[0] = g.Pz - 1 + g.z;
        for (int i = 1; i < maxLvl; ++i) trees[i] = (trees[i - 1] - 1) / 2;
        in.lists.assign(maxLvl, {});
        for (int k = 0; k < ns; ++k)
            for (int l = 0; l < maxLvl; ++l) if (sn_tree[k] == trees[l]) in.lists[l].push_back(k);
        for (int l = 0; l < maxLvl; ++l) in.z_active.push_back((g.z % (1 << l)) == 0);   // myZeroTrIdxs, supernodal_etree.c:853-869
    }
    hs.present.assign(ns, 0);
    for (auto &l : in.lists) for (int k : l) hs.present[k] = 1;
    in.lidx.assign(ns, {}); in.uidx.assign(ns, {}); in.succ.assign(ns, {});
    // (every supernode fills its own three lists: on the planner's threads)
    parallel_chunks(ns, 64, [&](int64_t kk0, int64_t kk1) {
    for (int k = (int) kk0; k < (int) kk1; ++k) {
        if (!hs.present[k]) continue;
        const int *li = full.lidx.data() + full.lidx_off[k];
        const int64_t ulen = full.uidx_off[k + 1] - full.uidx_off[k];
        const int *ui = ulen ? full.uidx.data() + full.uidx_off[k] : nullptr;
        {   // rows of the block rows ib with ib % Pr == myrow
            auto &o = in.lidx[k];
            o.assign(BC_HEADER, 0);
            int p = BC_HEADER, nb = 0, nr = 0;
            for (int b = 0; b < li[0]; ++b) {
                const int gid = li[p], nbrow = li[p + 1];
                if (g.krow(gid) == g.r) { o.insert(o.end(), li + p, li + p + LB_DESCRIPTOR + nbrow); ++nb; nr += nbrow; }
                if (gid > k) in.succ[k].push_back(gid);
                p += LB_DESCRIPTOR + nbrow;
            }
            o[0] = nb; o[1] = nr;
            if (!nb) o.clear();
        }
        if (ui) {   // blocks jb with jb % Pc == mycol
            auto &o = in.uidx[k];
            o.assign(BR_HEADER, 0);
            int p = BR_HEADER, nb = 0, nnz = 0;
            const int klst = full.xsup[k + 1];
            for (int b = 0; b < ui[0]; ++b) {
                const int jb = ui[p], nsj = full.xsup[jb + 1] - full.xsup[jb];
                if (g.kcol(jb) == g.c) {
                    o.insert(o.end(), ui + p, ui + p + UB_DESCRIPTOR + nsj);
                    int bn = 0;
                    for (int jj = 0; jj < nsj; ++jj) bn += klst - ui[p + UB_DESCRIPTOR + jj];
                    o[o.size() - UB_DESCRIPTOR - nsj + 1] = bn;
                    ++nb; nnz += bn;
                }
                in.succ[k].push_back(jb);
                p += UB_DESCRIPTOR + nsj;
            }
            o[0] = nb; o[1] = nnz; o[2] = (int) o.size();
            if (!nb) o.clear();
        }
        { auto &sc = in.succ[k]; std::sort(sc.begin(), sc.end()); sc.erase(std::unique(sc.begin(), sc.end()), sc.end()); }      // (finish_succ, per supernode)
    }
    });
    return split_wide_supernodes(H, in);   // supernodes of 257..512 columns (maxsup up to MAX_SUPER_SIZE): refined like the view path
}

}  // namespace sluamd
