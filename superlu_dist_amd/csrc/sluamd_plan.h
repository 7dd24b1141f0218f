// sluamd_plan.h -- internal: creation-time planning (slot structures -> device tables, schedules, exchange plans)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include "sluamd_internal.h"

namespace sluamd {

// Host threads of the planner (SLUAMD_PLAN_THREADS; default: the hardware's, at most 16 -- the pre-processing of a handle is integer / index work over
// millions of supernode blocks, memory-bound beyond a dozen cores).
inline int plan_threads()
{
    static const int want = getenv("SLUAMD_PLAN_THREADS") ? std::max(1, atoi(getenv("SLUAMD_PLAN_THREADS"))) : (int) std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return want;
}
// body(b, e) over [0, count) in chunks of `grain` handed out dynamically (the work per supernode spans four orders of magnitude); serial when one chunk or one thread.
template <class F> inline void parallel_chunks(int64_t count, int64_t grain, F &&body)
{
    if (count <= 0) return;
    grain = std::max<int64_t>(grain, 1);
    const int T = (int) std::min<int64_t>(plan_threads(), (count + grain - 1) / grain);
    if (T <= 1) { body((int64_t) 0, count); return; }
    std::atomic<int64_t> next{0};
    auto worker = [&] { for (;;) { const int64_t b = next.fetch_add(grain); if (b >= count) return; body(b, std::min(count, b + grain)); } };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(worker);
    worker();
    for (auto &x : th) x.join();
}

struct Comm;

// What both producers (caller's dLocalLU_t view + structure exchange, or the library's own symbolic factorisation)
// hand to the planner, indexed by GLOBAL supernode id.
struct SlotInput {
    std::vector<std::vector<int>> lidx, uidx;   // slot index arrays in the reference formats (empty = no blocks)
    std::vector<std::vector<int>> succ;         // block graph: gids (> k) of every block of L(:,k) and U(k,:), ALL process rows / columns
    std::vector<std::vector<int>> lists;        // per Z level of this layer's path: ascending supernodes of the forest
    std::vector<uint8_t> z_active;              // per Z level: this layer factors it
};

struct HostTables {
    std::vector<int64_t> sn_lval, sn_uval, sn_lidx, sn_uidx, sn_dinv, sn_dptr, sn_inv;
    int64_t dinv_total = 0, inv_total = 0;
    std::vector<int> sn_nsupr, sn_flags, sn_ldiag, sn_dlda, sn_ldu, sn_ncolu, sn_lb_off, sn_nlb, sn_ub_off, sn_nub, sn_rt_off, sn_nrt, sn_ct_off, sn_nct;
    std::vector<int> lb_gid, lb_nbrow, lb_rowoff, lb_lptr, lbs_gid, lbs_idx;
    std::vector<int> ub_gid, ub_ncols, ub_iukp, ub_stcol;
    std::vector<int> ucolptr, unzcol;
    std::vector<int> lrow, ucol_cp, ucol_ld, ucol_gc;
    std::vector<int64_t> sn_lrow, sn_ucol;
    std::vector<int4> rtile, ctile;
    std::vector<int2> rt_info;     // per row tile: (gid of its block row, offset of its row ids inside the lidx arena)
    std::vector<int4> ct_info;     // per column tile: (gid of its block column, offset of its U block inside the uidx arena, rank of its first column among the non-empty columns of the U row, 0)
    std::vector<uint8_t> sn_big;   // 1: supernode uses the 128x128 Schur tile configuration
    std::vector<double> sn_flops_exact, sn_bytes_alg;   // per supernode: exact-segment Schur flops and algorithmic destination bytes (by-configuration accounting)
    // merged row tiles (round 4): per U block (index sn_ub_off[k] + b) a range of rtile entries that cover ALL slot rows of the L blocks with
    // gid >= the U block's gid -- every one of them updates the same destination panel -- cut into tiles across block boundaries; count 0 = the
    // block pairs keep their own tiles.  (ulist entries of merged tiles carry destination -3.)
    std::vector<int> ub_mrt_off, ub_mrt_cnt;
    // merged column tiles: per L block (index sn_lb_off[k] + b, gid ib) a range of ctile entries that cover ALL non-empty columns of the U blocks
    // with gid > ib -- every one of them updates the same destination U row (ib) -- cut across block boundaries (ulist destination -4)
    std::vector<int> lb_mct_off, lb_mct_cnt;
    std::vector<uint8_t> sn_ucols_sorted;  // 1: the global ids of the non-empty columns of the U slot are strictly ascending
    std::vector<uint8_t> sn_rows_sorted;   // 1: the global row ids of the slot are strictly ascending over the whole slot (blocks ascending, rows ascending)
};

inline size_t &upload_bytes() { static thread_local size_t b = 0; return b; }   // bytes of the tables this thread uploaded (handle creation runs on one thread)
template <class Tv>
static int upload(std::vector<void *> &keep, const std::vector<Tv> &h, Tv **d)
{
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(Tv);
    upload_bytes() += bytes;
    HIPCHK(hipMalloc((void **) d, bytes));
    keep.push_back(*d);
    if (!h.empty()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(Tv), hipMemcpyHostToDevice));
    return 0;
}

int check_device(int dev);
void read_env(Handle::Env &e);
int trsm_rs(const Handle &H, int nsp);

int slots_from_view(Handle &H, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests, Comm *comm, SlotInput &in);
int slots_from_symb(Handle &H, const Symb &sy, const Grid &g, const int32_t *sn_tree, SlotInput &in);

// Planner: H.hs / H.grid / in -> value-arena layout, device tables, schedules, exchange plans; allocates the arena
// (zero-filled) and uploads everything but the values.  `t` is left filled for the value producers.
int plan_and_upload(Handle *H, SlotInput &in, HostTables &t);

// drivers (sluamd_factor.cpp)
int run_factor(Handle *H, double thresh, int *info);
int run_solve_dev(Handle *H, double *d_x, int64_t ldx, int nrhs);
int run_solve_dist(Handle *H, double *d_b, int64_t ldb, int nrhs, int64_t m_loc, int64_t fst_row, const int *perm_in, const int *perm_out);   // B distributed by rows (pdReDistribute3d_B_to_X / X_to_B inside)
int run_solve_local(Handle *H, double *d_x, int64_t ldx, int nrhs);   // single-rank sweep over every schedule (refinement)
int ensure_dinv(Handle *H);
int ensure_inv(Handle *H);

}  // namespace sluamd
