// Internal types shared by the host planner / drivers (sluamd_host.cpp, sluamd_dist.cpp, sluamd_comm*.cpp), the symbolic
// producer (sluamd_symb.cpp) and the gfx950 kernels (sluamd_kernels.hip).  Not part of the C ABI.
//
// Layering: the host files never launch a kernel themselves; every device operation goes through the `eng::` functions
// declared at the bottom of this header and implemented in sluamd_kernels.hip (hand-written HIP).  The CPU test build
// under oracle/emul/ links the SAME host files against a serial restatement of that interface so that the multi-rank
// orchestration (Z forests, XY block-cyclic panel exchange) is covered by `-m "not gpu"` tests; the product library has
// no CPU path.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include <sluamd_rt.h>   // resolved through the include path: HIP runtime API (product) -- oracle/emul/ supplies a host shim of the same name for the test build
#include "superlu_dist_amd.h"

namespace sluamd {

constexpr int BC_HEADER = 2;      // reference superlu_defs.h:169
constexpr int LB_DESCRIPTOR = 2;  // :170
constexpr int BR_HEADER = 3;      // :190
constexpr int UB_DESCRIPTOR = 2;  // :191
constexpr int DB = 32;            // diagonal sub-block size of the blocked panel kernels / solves
constexpr int KC = 16;            // K chunk of the Schur GEMM pipeline

void set_error(const std::string &msg);
const std::string &get_error();

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ::sluamd::set_error(std::string(#expr) + " failed: " + hipGetErrorString(e_));   \
            return SLUAMD_EHIP;                                                              \
        }                                                                                    \
    } while (0)

// 3D process grid of the reference (gridinfo3d_t, superlu_defs.h:385-420): 2-D block-cyclic layers (PROW/PCOL/PNUM,
// superlu_defs.h:270-279) replicated along Z.  World rank = (z * Pr + r) * Pc + c (layer-major, superlu_grid3d.c).
struct Grid {
    int Pr = 1, Pc = 1, Pz = 1, r = 0, c = 0, z = 0;
    int size() const { return Pr * Pc * Pz; }
    int rank() const { return rank_of(r, c, z); }
    int rank_of(int rr, int cc, int zz) const { return (zz * Pr + rr) * Pc + cc; }
    const int *own = nullptr;                  // refined wide supernodes (SplitMap): the caller's supernode an internal one is a piece of
    int krow(int k) const { return (own ? own[k] : k) % Pr; }   // PROW
    int kcol(int k) const { return (own ? own[k] : k) % Pc; }   // PCOL
};

// Host copy of the L/U *structure* this rank works with, indexed by GLOBAL supernode id.
//   Symb (sluamd_symb.cpp): the complete structure of a 1 x 1 grid, value offsets ascending in k, U offsets relative to
//     the U half.
//   Handle: one "L slot" and one "U slot" per supernode of this rank's forests.  The L slot of k is the image of the
//     part of panel k stored by process (myrow, k % Pc) (rows of the block rows ib with ib % Pr == myrow); the U slot
//     is the image of the part of block row k stored by (k % Pr, mycol).  A slot is OWN when that process is this rank
//     (values live in the resident arena) or REMOTE (index array received at creation, values received per level into
//     the scratch region of the arena).  Value offsets are ABSOLUTE offsets into the value arena.
struct HostStruct {
    int64_t n = 0;
    int nsupers = 0;
    std::vector<int> xsup;                    // [nsupers+1]
    std::vector<int64_t> lidx_off, uidx_off;  // [nsupers+1] into lidx/uidx
    std::vector<int64_t> lval_off, uval_off;  // Symb: [nsupers+1] ascending; Handle: [nsupers+1], entry k = absolute arena offset
    std::vector<int64_t> lval_len, uval_len;  // Handle only: [nsupers] doubles (elements) of each slot
    std::vector<int> lidx, uidx;
    int64_t nnzL = 0, nnzU = 0;               // own stored elements
    std::vector<uint8_t> present;             // [nsupers] 0 = not in any forest of this rank's Z layer
    // View path (round 4): the reference leaves the row subscripts INSIDE an L block in discovery order; the handle keeps them ascending (merged Schur tiles and
    // the joined sweeps need that) and remembers, per own L panel that had an unsorted block, the caller's slot row of every internal slot row (empty: identical).
    // Values are permuted on the way through the pinned staging buffer, both ways (copy_values); peers receive the sorted index image.
    std::vector<std::vector<int>> lrow_perm;
};

// Symbolic object behind sluamd_symb_t
struct Symb {
    HostStruct hs;
    std::vector<int> perm_c_final;  // perm_c_out
    std::vector<double> lval, uval; // optional host values (sluamd_ddistribute_host)
    std::vector<int *> lptr, uptr;  // pointer views for sluamd_symb_view
    std::vector<double *> lvptr, uvptr;
    std::vector<int> supno;         // [n] column -> supernode
    std::vector<int64_t> srow_off;  // [nsupers+1] off-diagonal structure (sorted global rows > last column)
    std::vector<int> srows;
    // unsymmetric structures (sluamd_dsymbfact_unsym; empty for the symmetric-pattern producer, whose U mirrors srows with full segments):
    // per block row the columns that hold a segment (ascending), its first nonzero row and its value offset inside the row's skyline;
    // supernodal parent in the etree of A + A^T (forest partition)
    std::vector<int64_t> ucol_off;  // [nsupers+1]
    std::vector<int> ucol_col, ucol_fnz;
    std::vector<int64_t> ucol_voff;
    std::vector<int> sn_parent;
    double flops = 0;
};

// positions of A's entries inside the host store of a Symb (sluamd_ddistribute_host): out_pos[e] into L values if
// is_u[e]==0 else U values; `owned` (may be null) filters by destination supernode
void compute_scatter_positions(const Symb &sy, const HostStruct &hs, int64_t n, const int *rowptr, const int *colind,
                               const int *perm_c_final, const uint8_t *owned, std::vector<int64_t> &pos,
                               std::vector<uint8_t> &is_u);
// tree (heap-numbered, root 0) of every supernode for a 1 x 1 x npdep grid
void partition_forests(const Symb &sy, int npdep, std::vector<int> &sn_tree);

// ------------------------------------------------------------------------------------------------
// Device-side tables (all pointers into HBM).  SoA per supernode / per block.
// ------------------------------------------------------------------------------------------------
enum : int {
    SNF_OWN_DIAG = 1,   // this rank factors the diagonal block of k (k % Pr == myrow && k % Pc == mycol)
    SNF_HAS_DIAG = 2,   // the factored diagonal block is available here (owner, or received: column / row peers)
    SNF_L_OWN = 4,      // the L slot is this rank's own storage (k % Pc == mycol)
    SNF_U_OWN = 8,      // the U slot is this rank's own storage (k % Pr == myrow)
};

struct DevTables {
    double *val;          // value arena: [own L slots | own U slots | remote-slot scratch | diagonal-block scratch]
    const int *lidx;      // Lrowind images
    const int *uidx;      // Ufstnz images
    const int *ucolptr;   // parallel to uidx: value offset (within the row) of each U column
    const int *unzcol;    // parallel to uidx: compact list of non-empty column ids of each U block
    const int *xsup;
    // per supernode
    const int64_t *sn_lval, *sn_uval;  // offsets into val
    const int64_t *sn_lidx, *sn_uidx;  // offsets into lidx / uidx
    const int *sn_nsupr;               // LDA of the L slot
    const int *sn_flags;               // SNF_*
    const int *sn_ldiag;               // rows of the diagonal block at the top of the L slot (nsupc when k % Pr == myrow, else 0)
    const int64_t *sn_dptr;            // offset into val of the nsupc x nsupc diagonal block (inside the L slot or in the scratch)
    const int *sn_dlda;                // ... and its leading dimension
    double *dinv;                      // inverted 32x32 diagonal sub-blocks of U_kk and L_kk^T (workspace)
    double *inv;                       // full inverses Linv | Uinv (ns x ns each, ld = ns) of the diagonal blocks this rank owns (solve)
    const int64_t *sn_inv;             // offset of supernode k's pair in inv
    const int64_t *sn_dinv;            // offset of supernode k's blocks in dinv
    const int *sn_ldu;                 // max U segment height of block row k (this slot)
    const int *sn_ncolu;               // total non-empty U columns of block row k (this slot)
    const int *sn_lb_off, *sn_nlb;     // L block table range
    const int *sn_ub_off, *sn_nub;     // U block table range
    const int2 *rt_info;               // per row tile / column tile: what the tile prologue would otherwise chase through four tables
    const int4 *ct_info;
    const int *sn_rt_off, *sn_nrt;     // row-tile range
    const int *sn_ct_off, *sn_nct;     // col-tile range
    // flat maps that replace dependent index walks in the latency-bound kernels: global row id of every L slot row
    // (lrow[sn_lrow[k] + slot row]) and, per non-empty U column in rank order (ucol_*[sn_ucol[k] + c]): value offset inside the
    // row, leading zeros (ns - segment), global column
    const int *lrow; const int64_t *sn_lrow;
    const int *ucol_cp, *ucol_ld, *ucol_gc; const int64_t *sn_ucol;
    // per L block (stored order) + gid-sorted directory
    const int *lb_gid, *lb_nbrow, *lb_rowoff, *lb_lptr;
    const int *lbs_gid, *lbs_idx;
    // per U block (sorted by gid)
    const int *ub_gid, *ub_ncols, *ub_iukp, *ub_stcol;
    // tiles
    const int4 *rtile;  // (L block idx within slot, row start in block, nrows, slot row offset)
    const int4 *ctile;  // (U block idx within row, first non-empty col rank, ncols, unused)
    // K-fused updates (null when disabled): fuse_prev[3k + j], j = 0..2 = the up to three predecessor supernodes (k-1,
    // k-2, k-3 of the same chain; -1 = none) whose deferred updates k's tiles also accumulate; defer[k] = 1 when k's
    // non-urgent tiles are skipped (a later chain member applies them).  Per (k, j): pair_rowmap[pair_roff[3k+j] + r] =
    // row of that predecessor's L panel holding the same global row as row r of k's panel (-1: absent);
    // pair_colinfo[2*(pair_coff[3k+j] + c)] = (value offset, leading zeros) of k's c-th non-empty U column inside the
    // predecessor's U row (leading zeros = predecessor width when absent)
    const int *fuse_prev, *defer, *pair_roff, *pair_coff, *pair_rowmap, *pair_colinfo;
    // joined sweeps (1 x 1 layers, LevelSched::join): per entry of lrow / ucol_gc, 1 = the row / column belongs to a supernode of the NEXT level of
    // the owner's schedule -- its update is applied inside the joined diagonal units of that level, the regular units skip it (null: not planned)
    const uint8_t *lrow_near, *ucol_near;
};

// One exchange message of the XY panel exchange: a contiguous range of the value arena sent to / received from one peer
// value arenas on a process-level pool of physical device chunks (sluamd_devpool.cpp; the CPU test build: oracle/emul/emul_rt.cpp)
int devpool_alloc(void **p, size_t bytes, int device);
void devpool_free(void *p);
void devpool_trim(int device);            // device < 0: every device
size_t devpool_cached_bytes(int device);

struct XMsg { int peer; int64_t off, len; };   // world rank, arena offset, doubles

struct LevelSched {
    int nlevels = 0;
    std::vector<int> lvl_off;       // [nlevels+1] into nodes
    std::vector<int> nodes;         // supernodes sorted by level (big-tile supernodes first inside a level)
    std::vector<int> tile_prefix;   // per node (aligned with nodes), exclusive prefix WITHIN its level (+1 total slot per level)
    std::vector<int> ltr_prefix;    // L-TRSM strips
    std::vector<int> utr_prefix;    // U-TRSM column chunks
    std::vector<int> inv_prefix;    // diagonal sub-block inversion tasks
    std::vector<int> zltr_prefix;   // complex path: 64-row L strips (the 64-column U chunks reuse bwd_prefix)
    std::vector<int> lvl_poff;      // [nlevels+1] offset of each level's prefix arrays (size nodes_in_level+1)
    std::vector<int> lvl_soff;      // [nlevels+1] offset of each level's Schur prefix arrays (big group | small group)
    std::vector<int> n_big;         // per level: nodes using the 128x128 tile configuration (listed first)
    std::vector<int> fwd_prefix, bwd_prefix;  // solve work units: 64-row L strips / 64-column U chunks
    std::vector<int> zfwd_prefix;             // complex path: 256-row L strips
    std::vector<int> zffu_prefix, zbfu_prefix; // complex path, fused links (eng::zsweep_fused): 256-row strips / 64-column chunks with AT LEAST ONE unit per owned diagonal block
    int *d_zffu_prefix = nullptr, *d_zbfu_prefix = nullptr;
    // the same units as explicit (supernode, strip / chunk) lists, per level [urgent | bulk]: urgent = touches a supernode of the
    // adjacent level (l+1: rows the forward update writes / columns the backward update reads) = what the next diagonal solve of the
    // chain waits for, bulk = levels >= l+2 only (launched together with that diagonal solve, eng::sweep_step)
    std::vector<int2> fwd_units, bwd_units;
    std::vector<int> fu_off, bu_off;          // [2*nlevels+1]: index 2*level + part
    std::vector<int2> diag_units;             // (supernode, 64-row strip) of every owned diagonal block, level by level: the out-of-place diagonal solves of the sweeps
    std::vector<int> du_off;                  // [nlevels+1]
    std::vector<int> finv_prefix;   // per level (lvl_poff layout): 64-row identity strips of the Linv / Uinv computation, 2 * ceil(ns / 64) per owned diagonal block
    std::vector<int> max_nsupc;     // per level
    std::vector<double> lvl_flops_schur, lvl_flops_panel;   // per level, THIS rank's share: exact-segment Schur flops of its tiles; diagonal LU + panel solves it owns (sluamd_plan_table)
    std::vector<uint8_t> lvl_has_group;   // (contracted schedule of the sweeps) the level holds a merged chain group: always run as a joined link
    bool no_join = false;           // the sweeps follow Handle::ssched: no joined tables for this schedule
    std::vector<uint8_t> lvl_defer; // per level: some supernode's non-urgent tiles are deferred to its K-fused partner
    // Split panel solves (look-ahead schedule, 1 x 1 layers, real): per level the 64-row strips of L(:, k) / 64-column chunks of U(k, :) that the level's
    // part-0 tiles (destination: a diagonal block of the next level) read -- "urgent": solved first, on the panel stream, so that the next level's diagonal
    // LU starts after those strips and their tiles alone -- and all the others, solved beside that diagonal LU on the urgent-tile stream.
    // ps_units: per split level [urgent L | urgent U | rest L | rest U]; ps_off[4 l .. 4 l + 4] its boundaries (equal: level not split)
    std::vector<int2> ps_units;
    std::vector<int> ps_off;        // [4*nlevels+1]
    int2 *d_ps_units = nullptr;
    std::vector<int> sn_level;      // [nsupers] level of each supernode in this schedule (-1: not in it)
    std::vector<int4> ulist;        // tile lists (k, absolute row tile, absolute column tile, destination block or -1): per level and tile-size group
                                    // [diagonal blocks of level l+1 | rest of the level-(l+1) panels | level-(l+2) panels | bulk, 8 x 8 bands per supernode]
    std::vector<int> u_off;         // [8*nlevels+1] offsets into ulist: index (2*level + group) * 4 + part
    std::vector<int64_t> m_off;     // [2*nlevels+1] first int of the tile records of (level, group) in d_tmaps (record of tile u of the group: + (u - u_off[group start]) * rec)
    int *d_tmaps = nullptr;         // per-tile records of k_schur, built on the first factorisation (owned by Handle::d_misc)
    // Balanced bulk launches (round 6; balance_bulk): k_schur gives XCD x the tiles [x_off[x], x_off[x + 1]) of the launch -- eight contiguous ranges of EQUAL MODELLED COST
    // (K chunks of all sources x share of the waves that run MFMAs) instead of equal counts; the level's supernodes are listed longest tiles first.
    // x_off: 10 ints per (level, group) at 10 * (2 * level + group): nine range boundaries relative to the launch's first tile + the longest range; [9] == 0: not balanced
    std::vector<int> x_off;
    int *d_x_off = nullptr;
    int maps_state = 0;             // 0: not built yet, 1: built, -1: unavailable (memory / switched off)
    // ---- XY block-cyclic exchange plan (empty on a 1 x 1 layer): per level, in ascending supernode order ----
    std::vector<int> dg_prefix;               // per node (lvl_poff layout): 1024-double chunks of the own diagonal blocks to pack
    std::vector<int64_t> dg_off;              // ... and their offsets inside the level's diagonal staging range
    std::vector<std::vector<XMsg>> x_diag_send, x_diag_recv;     // phase 1: packed diagonal blocks (column + row peers)
    std::vector<std::vector<XMsg>> x_panel_send, x_panel_recv;   // phase 2: L slots along the process row, U slots down the column
    std::vector<int64_t> dg_stage_off;        // [nlevels] arena offset of the level's packed own diagonal blocks
    // solve exchange plan: per level, x segments (as [first row, rows) runs) reduced to / broadcast from the diagonal owners
    struct XSeg { int peer = -1; std::vector<std::pair<int, int>> runs; int64_t total = 0; mutable int *d_runs = nullptr; /* device image, uploaded on first use (owned by Handle::d_misc) */ };
    std::vector<std::vector<XSeg>> xs_red_send, xs_red_recv, xs_bc_send, xs_bc_recv;
    // device copies
    int *d_nodes = nullptr, *d_tile_prefix = nullptr, *d_ltr_prefix = nullptr, *d_utr_prefix = nullptr;
    int *d_fwd_prefix = nullptr, *d_bwd_prefix = nullptr, *d_inv_prefix = nullptr, *d_sn_level = nullptr, *d_zltr_prefix = nullptr;
    int *d_finv_prefix = nullptr, *d_zfwd_prefix = nullptr;
    int4 *d_ulist = nullptr;
    int2 *d_fwd_units = nullptr, *d_bwd_units = nullptr, *d_diag_units = nullptr;
    // unit records of the 1 x 1 layer sweeps, two int4 per unit in the order of the unit lists: what a unit otherwise looks up in six tables
    // behind its list entry (fwd / bwd_update_body, diag_strip_body)
    std::vector<int4> fwd_recs, bwd_recs, diag_recs;
    int4 *d_fwd_recs = nullptr, *d_bwd_recs = nullptr, *d_diag_recs = nullptr;
    int *d_dg_prefix = nullptr; int64_t *d_dg_off = nullptr;
    // ---- joined sweeps (round 4; 1 x 1 layers, real) ----
    // The level-set sweeps above pay TWO dependent launches per level: the urgent updates of level l, then the diagonal solves of level l + 1.  Joined
    // form: the diagonal solve of supernode j (level l + 1) is cut into 64 x 64 blocks (s, c) of its inverse, and the unit of block (s, c) first applies the
    // updates of level l to ITS 64 entries of the right-hand side itself -- the rows of the level-l panels that fall into column block c of j (forward), the
    // columns of level l + 1 in U(k, :) over the rows of block c (backward) -- then multiplies by the inverse block and ADDS into the (zeroed) output rows of
    // strip s.  The near rows / columns are recomputed by every strip that needs them (a 256-column supernode: 10 units, block c read by 4 - c of them) and
    // skipped by the regular units (DevTables::lrow_near / ucol_near), which ALL run beside the next level's joined units: one launch per level.  Chosen per
    // level (SLUAMD_JOIN_MAX_NODES): on levels of many supernodes the recomputation costs bandwidth and they keep the two-launch form.  (One unit per column
    // block c with all its strips -- no recomputation -- was built and measured slower on every level: 16 + 16 loads per thread in flight, spills at the
    // register budgets of the multi-workgroup builds; profiles/r04_ab_solve_join.txt.)
    //   forward unit, 8 int4:  (fst_j, ns_j, s, c) (Linv offset lo, hi, sources, first overflow source in jf_aux) then up to 3 sources of 2 int4:
    //                          (fst_k, ns_k, lda_k, rows) (value offset of the first row lo, hi, lrow index of the first row lo, hi)
    //   backward unit, 4 int4: (fst_k, ns_k, s, c) (Uinv offset lo, hi, first near column in jb_aux, near columns) (U value offset lo, hi, 0, 0) (0)
    //   jb_aux: one int4 (leading zeros, value offset, global column, 0) per near column
    //   regular units: the 2-int4 records of fwd_recs / bwd_recs; bit 16 of .y (width) set = the unit has near rows / columns to skip
    bool join = false;
    std::vector<int4> jf_recs, jf_aux, jb_recs, jb_aux, jfu_recs, jbu_recs;
    std::vector<int> jf_off, jb_off, jfu_off, jbu_off;     // [nlevels + 1] unit ranges per level
    int4 *d_jf_recs = nullptr, *d_jf_aux = nullptr, *d_jb_recs = nullptr, *d_jb_aux = nullptr, *d_jfu_recs = nullptr, *d_jbu_recs = nullptr;
};

struct Comm;   // sluamd_comm.h

// Refinement of the caller's supernode partition: supernodes wider than 256 columns (the reference allows up to
// MAX_SUPER_SIZE = 512, superlu_defs.h:154, sp_ienv.c:95-110) are handled as chains of <= 256-column pieces internally; the
// caller's panels keep the reference layout, the value upload / download gathers / scatters by column.
struct SplitMap {
    bool active = false;
    std::vector<int> oxsup;     // the caller's xsup [original nsupers + 1]
    std::vector<int> first;     // [original nsupers + 1]: internal id of the first piece of each original supernode
    std::vector<int> owner;     // [internal nsupers]: the original supernode of each piece (Grid::own)
    struct Piece { int arr; int ok; int64_t hoff, len; };   // arr 0 / 1: Lnzval_bc_ptr[ok] / Unzval_br_ptr[ok]; element offset and count
    std::vector<std::vector<Piece>> lsrc, usrc;             // per INTERNAL supernode, in slot (column-major) order
};


// merged chain groups of the sweeps (Handle::groups): device descriptors of eng::grp_gather / eng::gemm_batched
struct GrpDesc { int nm, nG; int64_t ginv; int k[4], o[4], w[4]; };                                  // members: supernode, column offset inside the group, width
struct GemmDesc { int64_t a, b, c; int lda, ldb, ldc, M, N, K, abase, bbase, cbase, neg; };        // 64 bytes
constexpr int64_t GRP_SCR = 1024 * 1024;                                                           // doubles per scratch image: [LG | UG | TL | TU]

// wall-clock laps of handle creation (sluamd_setup_times): where the pre-processing of a handle goes
struct SetupTimer {
    std::vector<std::pair<std::string, double>> laps;
    double t_prev = -1.0;
    static double now();
    void start() { t_prev = now(); }
    void lap(const char *what) { const double t = now(); if (t_prev < 0) t_prev = t; laps.emplace_back(what, t - t_prev); t_prev = t; }
};

struct Handle {
    int device = 0;
    SetupTimer setup;
    sluamd_options_t opt{};
    HostStruct hs;
    Grid grid;
    int Pz = 1, myz = 0;
    // environment switches, read ONCE at creation (they may differ per handle)
    struct Env {
        bool no_lookahead = false, no_fuse = false, no_big_tiles = false, schur_4waves = false, trsm_rs32 = false, profile = false, profile_dump = false, trsm_panels = false, diag_v1 = false;
        int fuse_min_pct = 75, fuse_max_prev = 3, reserve_cus = 0;
        int fuse_group_min_nodes = 8;   // SLUAMD_FUSE_GROUP_MIN_NODES: ... and only on levels of at least this many supernodes
        int fuse_tail_guard = 0;    // SLUAMD_FUSE_TAIL_GUARD: groups of more than two K-fused supernodes (fuse_max_prev > 1) only below the last N levels
        int diag_tail = 64;          // SLUAMD_DIAG_TAIL: last N single-supernode levels factor their diagonal block with the whole-register-file build of k_diag_lu2
        int trsm_tail = 64;          // SLUAMD_TRSM_TAIL: last N single-supernode levels of a 1 x 1 layer solve their panels by blocked substitution, full inverses off the chain
        int level_split_min = 2048;  // SLUAMD_LEVEL_SPLIT_MIN: sub-levels never get smaller than this, forests whose largest level has fewer than 4 x this are not cut (tests lower it)
        int level_split_wdiv = 0;    // SLUAMD_LEVEL_SPLIT_WDIV (opt-in, e.g. 128): levels heavier (panel values, upper bound from the block graph) than 1 / this of the forest's total are cut too;
                                     // <= 1: off.  Off by default: cut levels break K-fused pairs (150^3 on 2x2x2 with 128: allocated / values 1.27-1.29 -> 1.17-1.24, tile executions + 20 %)
        double level_split_wmin = 1e9;   // SLUAMD_LEVEL_SPLIT_WMIN: ... in forests of at least this many panel values in total
        int join_max_nodes = 32;     // SLUAMD_JOIN_MAX_NODES: levels of more supernodes than this keep the two-launch links
        bool sort_block_rows = true; // SLUAMD_SORT_BLOCK_ROWS=0: keep the caller's row order inside the L blocks of a view (round 3: no merged tiles / joined sweeps on such panels)
        bool solve_join = true;      // SLUAMD_SOLVE_JOIN=0: the two-launch links of round 3 (urgent updates, then diagonal strips) instead of the joined units
        bool fuse_small = true;      // SLUAMD_FUSE_SMALL=0: K-fused pairs only where the 128 x 128 tile configuration runs (round 3)
        int ksplit = 4;              // SLUAMD_KSPLIT: workgroups per tile (shares of K) for the diagonal-block tiles on the panel chain when a launch has at most 64 of them (1 = off)
        int big_util_pct = 50, big_min_cols = 96;   // SLUAMD_BIG_UTIL_PCT / SLUAMD_BIG_MIN_COLS: a supernode runs 128 x 128 tiles when it is at least this wide and its block pairs fill that share of them
        bool no_merge_tiles = false; // SLUAMD_NO_MERGE_TILES: every (L block, U block) pair keeps its own Schur tiles (round 3)
        bool no_level_split = false; // SLUAMD_NO_LEVEL_SPLIT: XY layers keep whole DAG levels (round 3's exchange scratch: the largest level)
        int panel_split_max_nodes = 1024;   // SLUAMD_PANEL_SPLIT: levels of at most this many supernodes solve their panels in two parts (urgent strips on the chain, the rest beside the next diagonal LU); 0 = off
        bool solve_groups = false;   // SLUAMD_SOLVE_GROUPS=1: merged chain groups of the sweeps (Handle::groups)
        int solve_group_level_nodes = 8;   // SLUAMD_SOLVE_GROUP_LEVEL_NODES: ... only where every member's level holds at most this many supernodes (latency-bound levels)
        int z_fuse_max_nodes = 16;   // SLUAMD_ZFUSE_MAX_NODES: complex16 sweeps run the levels of at most this many supernodes as fused links (one launch per level and sweep); 0 = never
        bool info_last = false;      // SLUAMD_INFO_LAST=1: `info` = the zero pivot met LAST on a rank (largest column; what pdgstrf2.c:568-571 leaves in *info), MIN over the ranks (pdgstrf3d.c:388-392); default: the first column
        bool no_tile_maps = false;   // SLUAMD_NO_TILE_MAPS: the Schur tiles chase their tables instead of reading the per-tile records
        int balance_min_tiles = 1024; // SLUAMD_BALANCE_MIN_TILES: bulk launches of at least this many tiles give the XCDs ranges of equal modelled cost (LevelSched::x_off); 0 = never
        double balance_ovh = 4.0;    // SLUAMD_BALANCE_OVH: fixed cost of a tile (record, prologue, scatter) in K-chunk periods of the cost model
    } env;
    // device arenas
    double *d_val = nullptr;
    int64_t arena_len = 0;          // elements (doubles, or doublecomplex for z) of the whole arena
    int64_t own_len = 0;            // ... of the resident part [own L | own U]
    int *d_lidx = nullptr, *d_uidx = nullptr, *d_ucolptr = nullptr, *d_unzcol = nullptr, *d_xsup = nullptr;
    std::vector<void *> d_misc;  // everything else to free
    DevTables T{};
    std::vector<LevelSched> sched;  // one per Z level (forests) or a single one
    // Merged chain groups of the sweeps (round 5; 1 x 1 x 1 grids, real, SLUAMD_SOLVE_GROUPS): up to four consecutive supernodes of a chain -- the pieces of one
    // separator: every member but the first has the previous member as its ONLY child -- solved as ONE block of <= 1024 columns with the inverse of their block
    // triangle (computed during the factorisation, beside the panel chain: group_inverse), so that four single-supernode levels of the sweeps become one.
    // ssched: the level schedule of the sweeps on the DAG with every group contracted to one node (empty: the sweeps follow `sched`).
    struct SolveGroup { int nm = 0, nG = 0, k[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0}; int64_t ginv = 0; int last_level = 0; int tile_off[7] = {0, 0, 0, 0, 0, 0, 0}; };
    std::vector<SolveGroup> groups;
    std::vector<int> grp_of;                  // [nsupers] group of a supernode, -1: none
    std::vector<std::vector<int>> lvl_groups;  // [factor level] groups whose last member is factored at that level
    std::vector<LevelSched> ssched;
    GrpDesc *d_grpdesc = nullptr; GemmDesc *d_gemmdesc = nullptr; int4 *d_gemmtiles = nullptr; double *d_gscr = nullptr;
    hipStream_t gstream = nullptr;            // group inverses run here, beside the factorisation
    std::vector<std::vector<int>> forest_nodes;   // ascending supernodes of the forest of every Z level on this layer's path (even when not factored here)
    std::vector<uint8_t> z_active;  // [Z levels] this layer factors that level's forest (!myZeroTrIdxs)
    std::vector<int> own_l_order, own_u_order;   // own L / U slots in value-arena order
    SplitMap split;
    hipStream_t stream = nullptr;
    hipStream_t pstream = nullptr;          // high-priority stream for the panel kernels (look-ahead)
    hipStream_t ustream = nullptr;          // high-priority stream for the Schur tiles that feed the next level's panels
    hipStream_t u2stream = nullptr;         // ... and for those that feed the panels of the level after it
    hipStream_t rstream = nullptr;          // Z ancestor reduction: receives + adds run here, beside the factorisation of the ancestor forest
    // pipelined ancestor reduction (reduce_ancestors): chunk events of the reduction that follows the last factored Z level; a DAG level
    // of the next forest may start once the chunks covering its own L and U slots have been added
    struct RedEv { int64_t lend, uend; hipEvent_t ev; };      // own-slot arena offsets (values) covered so far in the L / U range
    std::vector<RedEv> red_events;
    std::vector<hipEvent_t> red_pool; size_t red_pool_used = 0;
    hipEvent_t red_all = nullptr;            // everything queued on rstream so far
    std::vector<hipEvent_t> ev_pool;        // look-ahead dependency events
    size_t ev_pool_used = 0;
    int *d_info = nullptr;      // [0]=first zero pivot column (INT_MAX if none), [1]=tiny pivots, [2]=missing dest blocks, [3]=sink, [4]=last zero pivot column (0 if none)
    double *d_x = nullptr; int64_t x_cap = 0;
    double *d_xtmp = nullptr; int64_t xtmp_cap = 0;   // exchange staging of the distributed solve / ancestor reduction
    double *d_w = nullptr; int64_t w_cap = 0;         // second vector of the 1 x 1-layer sweeps (out-of-place diagonal solves: forward solution, backward accumulators)
    int64_t *d_apos = nullptr; double *d_aval = nullptr; int64_t a_nnz = 0;  // A's entries for device-side (re)distribution
    // iterative refinement (sluamd_dAttachMatrix): the ORIGINAL matrix in CSR + perm_c, and work vectors
    int *d_rfs_rp = nullptr, *d_rfs_ci = nullptr, *d_rfs_pc = nullptr; double *d_rfs_av = nullptr;
    double *d_rfs_work = nullptr; unsigned long long *d_rfs_s = nullptr; int64_t rfs_nnz = 0;
    void *h_pinned = nullptr; size_t pinned_bytes = 0;      // bounded pinned staging buffer (value upload / download)
    bool z = false;                                         // complex16 (doublecomplex) values: 16-byte elements
    bool dinv_ready = false;                                // T.dinv holds the inverses for the current factors
    bool inv_ready = false;                                 // T.inv (Linv / Uinv) too
    bool factored = false;                                  // the resident values are FACTORS (run_factor since the last SetValues / ResetValues): sluamd_dGetDiagInv refuses otherwise
    int *d_ztickets = nullptr;                              // complex fused backward links: one ticket counter per supernode (zero between sweeps)
    bool profile = false;                                   // per-kernel-family HIP-event timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_schur, ev_panel, ev_xchg, ev_red;
    size_t ev_schur_used = 0, ev_panel_used = 0, ev_xchg_used = 0, ev_red_used = 0;
    std::vector<uint8_t> ev_schur_big;     // parallel to ev_schur: the launch ran the 128 x 128 tile configuration
    struct SchurRec { int level, pass, big, ntiles, mx; };
    std::vector<SchurRec> schur_rec;   // SLUAMD_PROFILE_DUMP: one record per profiled Schur launch (parallel to ev_schur)
    sluamd_stats_t st{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // host tables kept for stats / planning
    std::vector<int> h_nsupr, h_ldu, h_ncolu, h_flags, h_ldiag;
    std::vector<int64_t> h_sn_dinv, h_dptr;
    // K-fused updates (see DevTables): host images, built by build_schedule
    std::vector<int> h_fuse_prev, h_defer, h_pair_roff, h_pair_coff, h_pair_rowmap, h_pair_colinfo;
    std::vector<uint8_t> h_lrow_near, h_ucol_near;   // DevTables::lrow_near / ucol_near while the schedules are built
    int fused_pairs = 0;
    int64_t xy_scratch_len = 0;    // values of the received-panel scratch (all copies)
    int xy_scratch_copies = 0;     // XY layers: copies of the received-panel scratch (by level modulo this), 0 on a 1 x 1 layer
    int max_nsupc = 0;
    Comm *comm = nullptr;           // not owned; set by sluamd_dCreateLUHandleGrid / ...FromSymbGrid
    std::map<int, LevelSched::XSeg> xseg_cache;        // x-segment run lists of the grid solve (Z exchanges, owner rows), built once
    std::vector<LevelSched::XSeg> owner_runs; bool owner_ready = false;      // every rank's owner rows (where x is final after the sweeps), indexed by world rank; exchanged once
    // distributed right-hand side at the solve boundary (sluamd_pdgstrs3d_dist): routing of the caller's local rows of B to the ranks
    // that consume them and back, built once per (m_loc, fst_row, perm)
    struct DistRoute {                              // rows of B (original order, local index) <-> rows of x (factored order) for one permutation
        std::vector<int64_t> cnt_b, cnt_x;          // per world rank: my rows of B paired with it / my owner rows of x paired with it
        std::vector<int *> d_bidx, d_xidx;          // device: local row of B per pair (ordered by the row of x); row of x per pair (ascending)
    };
    struct DistPlan {
        bool ready = false, same = false;           // same: perm_out == perm_in (route_out unused)
        int64_t m_loc = -1, fst_row = -1;
        std::vector<int> perm_in, perm_out; bool null_in = true, null_out = true;   // the permutations the routes were built for (compared element by element)
        DistRoute in, out;                          // B -> x through perm_in, x -> B through perm_out
        std::vector<void *> bufs;                   // device allocations of this plan
    } dist;
    double *d_bloc = nullptr; int64_t bloc_cap = 0;   // the caller's local rows of B on the device
};

// ------------------------------------------------------------------------------------------------
// Engine: every device operation of the hot path.  sluamd_kernels.hip implements these as gfx950 kernel launches on
// `s`; all pointer arguments are device pointers.
// ------------------------------------------------------------------------------------------------
namespace eng {
int setup();   // one-time function attributes (dynamic LDS limits)
// flags: bit 0 = ReplaceTinyPivot, bit 1 = round-1 right-looking kernel, bit 2 = k_diag_lu2 built for the whole register file (tail levels).  Also leaves the inverted 32 x 32 diagonal sub-blocks of the
// owned blocks in T.dinv (what diag_inv computes for blocks received from another rank)
void diag_lu(hipStream_t s, const DevTables &T, const int *nodes, int nn, int max_nsupc, int flags, double thresh, int *info);
void diag_inv(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int ntask);
// L strips (work units [0, nl)) and U column strips ([nl, nl + nu)); strip height rs = 32 or 64
void panel_trsm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu,
                int rs, int max_nsupc, const int2 *units = nullptr);
// the same two panel solves as GEMMs with the full inverses T.inv (1 x 1 layers; 64-high work units)
// units (both): explicit (supernode, strip / chunk) list [nl L units | nu U units] instead of the prefix arrays -- one part of a split panel solve
void panel_gemm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int max_nsupc,
                const int2 *units = nullptr);
// cfg: 0 = 128x128 tiles / 8 waves, 1 = 128x128 / 4 waves, 2 = 64x64 / 4 waves
void schur(hipStream_t s, int cfg, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info,
           const int4 *ulist = nullptr, int prio = 0, const int *tmaps = nullptr, int mmode = 0, int ksplit = 1 /* > 1: that many workgroups per tile, each a share of K (128 x 128 tiles) */,
           const int *xoff = nullptr, int xmax = 0 /* balanced bulk launch: device pointer to the launch's nine XCD range boundaries (LevelSched::x_off) and the longest range */);
// per-tile records of the list schedules (k_schur): mmode 1 = build pass (writes the records of the launch's tiles at tmaps, no update),
// mmode 2 = the tiles read their records; ints per record for a tile configuration
inline int schur_rec_ints(int cfg, bool z) { return (cfg <= 1) ? 64 + 128 + 3 * 128 : 64 + 64 + 3 * 64; (void) z; }
// Linv / Uinv of every owned diagonal block of `nodes` from the factored blocks + dinv (pdCompute_Diag_Inv, pdgstrs.c:842)
void full_inv(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, int max_nsupc);
// ---- merged chain groups of the sweeps (round 5): the inverse of the (<= 1024)-column block triangle of up to four consecutive chain supernodes ----
// grp_gather: zero-fills nothing (the caller does), copies the members' Linv / Uinv into the diagonal blocks of the group's pair in T.inv and gathers the
// off-diagonal blocks of the members' panels / skylines into the dense scratch images LG / UG (ld = nG); gemm_batched: C = (-)A B on 64 x 64 tiles (fp64 MFMA),
// operands in T.inv (base 0) or the scratch (base 1)
void grp_gather(hipStream_t s, const DevTables &T, const GrpDesc *d_desc, double *scratch);
void gemm_batched(hipStream_t s, const DevTables &T, const GemmDesc *d_descs, const int4 *d_tiles, int ntiles, double *scratch);
void solve_diag(hipStream_t s, bool lower, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs, int max_nsupc);
// units != null: the launch runs the host-built (supernode, strip / chunk) list `units[0 .. nwork)` instead of the level's prefix arrays.
// Two vectors (they may be the same one: XY layers, profiling): the update reads solved blocks from xsrc / xcols and subtracts from x
void fwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xsrc, double *x, int64_t ldx, int nrhs,
                int max_nsupc, const int2 *units = nullptr, const int4 *recs = nullptr);   // recs: unit records of the same list (LevelSched::fwd_recs)
void bwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xcols, double *x, int64_t ldx, int nrhs,
                int max_nsupc, const int2 *units = nullptr, const int4 *recs = nullptr);
// one link of a sweep on a 1 x 1 layer: the diagonal-solve strips `dunits` (supernode, 64-row strip; OUT OF PLACE: lower xa -> xb, upper
// xb -> xa) + the update units `units` that do not feed them (lower: read xb, subtract from xa; upper: read xa, subtract from xb), one
// launch (max_nsupc over everything in the launch)
void sweep_step(hipStream_t s, bool lower, const DevTables &T, const int2 *dunits, int ndu, const int2 *units, int nunits,
                double *xa, double *xb, int64_t ldx, int nrhs, int max_nsupc, const int4 *drecs = nullptr, const int4 *urecs = nullptr);
// joined link (LevelSched::join): `nj` joined diagonal units (jrecs: 8 int4 each forward, 4 backward; jaux: overflow sources / near columns) + `nunits` regular
// units given by their records (urecs, near rows / columns skipped), one launch; the joined units ADD into xb (lower) / xa (upper): zeroed rows expected
void sweep_join(hipStream_t s, bool lower, const DevTables &T, const int4 *jrecs, int nj, const int4 *jaux, const int4 *urecs, int nunits,
                double *xa, double *xb, int64_t ldx, int nrhs, int max_nsupc);
// x[rows of the supernodes `nodes`] = 0 (all right-hand sides)
void zero_nodes(hipStream_t s, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs);
void scatter_values(hipStream_t s, double *val, const int64_t *pos, const double *a, int64_t nnz);
void rfs_residual(hipStream_t s, int n, const int *rp, const int *ci, const double *av, const double *x, const double *b, const int *pc,
                  double *r_perm, unsigned long long *s_out, double safe1, double safe2);
void rfs_update(hipStream_t s, int n, const int *pc, const double *dx_perm, double *x);
// y[i] += a * x[i]  (ancestor reduction: dzRecvLPanel / dzRecvUPanel's daxpy, pd3dcomm.c:189-331)
void axpy(hipStream_t s, int64_t n, double a, const double *x, double *y);
// y[i] += x[i] with fp64 atomics (safe beside the atomic scatter of Schur tiles into the same panels)
void add_atomic(hipStream_t s, int64_t n, const double *x, double *y);
// XY exchange helpers: own diagonal blocks of a level -> contiguous staging range (ns x ns, lda = ns each)
void pack_diag(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, const int64_t *off, int nn, int nwork, double *stage /* already offset */,
               int vs = 1 /* doubles per value: 2 = complex16 */);
// x segments <-> contiguous buffer; mode 0: buf = x, 1: x = buf, 2: x += buf, 3: buf = x then x = 0
void xseg_copy(hipStream_t s, double *x, int64_t ldx, int nrhs, const int *runs /*device: (row0, nrows, rows before) triples*/, int nruns, int64_t total,
               double *buf, int mode);
// indexed rows <-> contiguous cnt x nrhs buffer; mode 0: buf[j] = v[idx[j]], 1: v[idx[j]] = buf[j]  (pdReDistribute3d_B_to_X / X_to_B)
void rows_copy(hipStream_t s, double *v, int64_t ldv, int nrhs, const int *idx, int64_t cnt, double *buf, int mode, int vs = 1 /* doubles per value; ldv in values */);
int mfma_selftest(const double *A, const double *B, double *D);   // host pointers
// complex16 twins (any grid: the diagonal-block operand comes from sn_dptr / sn_dlda like the double kernels')
void zdiag_lu(hipStream_t s, const DevTables &T, const int *nodes, int nn, int max_nsupc, int replace_tiny, double thresh, int *info);
void zpanel_trsm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int max_nsupc);
void zschur(hipStream_t s, int cfg, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info,
            const int4 *ulist = nullptr, int prio = 0, const int *tmaps = nullptr, int mmode = 0, const int *xoff = nullptr, int xmax = 0);   // cfg 0: tiles of 64 panel rows x 128 columns, else 32 x 64
void zsolve_diag(hipStream_t s, bool lower, const DevTables &T, const int *nodes, int nn, void *x, int64_t ldx, int nrhs, int max_nsupc);
void zfwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, int64_t ldx, int nrhs,
                 int max_nsupc);
void zbwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, int64_t ldx, int nrhs);
// one link of a complex16 sweep in ONE launch (supernodes of <= 64 columns): forward = every strip solves y_k itself (strip 0 stores it to w), then x[rows] -= L y_k;
// backward = w_k -= U(k, chunk) x, the workgroup taking the last ticket of k solves x_k = inv(U_kk) w_k.  prefix: >= 1 unit per supernode (LevelSched::zffu / zbfu)
void zsweep_fused(hipStream_t s, bool lower, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, void *w, int64_t ldx, int nrhs,
                  int max_nsupc, int *tickets);
void zscatter_values(hipStream_t s, void *val, const int64_t *pos, const void *a, int64_t nnz);
}  // namespace eng

}  // namespace sluamd

struct sluamd_lu_handle_s { sluamd::Handle H; };
