// sluamd_core.hip -- MI355X (gfx950) implementation of the 3D supernodal LU hot path.
//
// Host side: flattens the caller's reference-format L/U store (superlu_dist_amd.h), uploads it ONCE to
// HBM (index arena + value arena stay resident for factor and solve), builds device-side block
// directories, tile lists and an elimination-DAG level schedule.
// Device side (hand-written HIP, wave64): per level of the schedule
//     k_diag_lu      unpivoted LU of every diagonal block of the level   (Local_Dgstrf2, pdgstrf2.c:508)
//     k_panel_trsm<0> L(:,k) <- L(:,k) U_kk^-1                           (dLPanelTrSolve, dtrfCommWrapper.c:120)
//     k_panel_trsm<1> U(k,:) <- L_kk^-1 U(k,:) directly on the skyline   (dTrs2_GatherTrsmScatter, pdgstrf2.c:804)
//     k_schur        A(I,J) -= L(I,k) U(k,J): fused gather -> fp64 MFMA GEMM -> scatter, no bigU/bigV
//                    round trip (dRgather_L/U dgather.c:133-398 + dblock_gemm_scatter dscatter3d.c:81-189
//                    + dscatter_l dscatter.c:109 + scatter_u dscatter3d.c:555)
// and for pdgstrs3d the level-set forward/backward block solves (dlsum_fmod_inv / dlsum_bmod_inv,
// pdgstrs_lsum.c:414 / :1362).
//
// There is NO CPU fallback in this file: every entry point fails when no HIP device is present.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>
#include "sluamd_internal.h"

namespace sluamd {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_error(std::string(#expr) + " failed: " + hipGetErrorString(e_));             \
            return SLUAMD_EHIP;                                                              \
        }                                                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Device-side tables (all pointers into HBM).  SoA per supernode / per block.
// ------------------------------------------------------------------------------------------------
struct DevTables {
    double *val;          // value arena: [L panels | U rows]
    const int *lidx;      // Lrowind arena
    const int *uidx;      // Ufstnz arena
    const int *ucolptr;   // parallel to uidx: value offset (within the row) of each U column
    const int *unzcol;    // parallel to uidx: compact list of non-empty column ids of each U block
    const int *xsup;
    // per supernode
    const int64_t *sn_lval, *sn_uval;  // offsets into val
    const int64_t *sn_lidx, *sn_uidx;  // offsets into lidx / uidx
    const int *sn_nsupr;               // LDA of the L panel
    double *dinv;                      // inverted 32x32 diagonal sub-blocks of U_kk and L_kk^T (workspace)
    const int64_t *sn_dinv;            // offset of supernode k's blocks in dinv
    const int *sn_ldu;                 // max U segment height of block row k
    const int *sn_ncolu;               // total non-empty U columns of block row k
    const int *sn_lb_off, *sn_nlb;     // L block table range
    const int *sn_ub_off, *sn_nub;     // U block table range
    const int *sn_rt_off, *sn_nrt;     // row-tile range
    const int *sn_ct_off, *sn_nct;     // col-tile range
    // per L block (stored order) + gid-sorted directory
    const int *lb_gid, *lb_nbrow, *lb_rowoff, *lb_lptr;
    const int *lbs_gid, *lbs_idx;
    // per U block (sorted by gid)
    const int *ub_gid, *ub_ncols, *ub_iukp, *ub_stcol;
    // tiles
    const int4 *rtile;  // (L block idx within panel, row start in block, nrows, panel row offset)
    const int4 *ctile;  // (U block idx within row, first non-empty col rank, ncols, unused)
    // cooperative (owner-computes) mode inside one shared ancestor forest: block column jb belongs to rank jb % own_G
    int own_G, own_g;
    // K-fused updates (null when disabled): fuse_prev[3k + j], j = 0..2 = the up to three predecessor supernodes (k-1,
    // k-2, k-3 of the same chain; -1 = none) whose deferred updates k's tiles also accumulate; defer[k] = 1 when k's
    // non-urgent tiles are skipped (a later chain member applies them).  Per (k, j): pair_rowmap[pair_roff[3k+j] + r] =
    // row of that predecessor's L panel holding the same global row as row r of k's panel (-1: absent);
    // pair_colinfo[2*(pair_coff[3k+j] + c)] = (value offset, leading zeros) of k's c-th non-empty U column inside the
    // predecessor's U row (leading zeros = predecessor width when absent)
    const int *fuse_prev, *defer, *pair_roff, *pair_coff, *pair_rowmap, *pair_colinfo;
};

struct LevelSched {
    int nlevels = 0;
    std::vector<int> lvl_off;       // [nlevels+1] into nodes
    std::vector<int> nodes;         // supernodes sorted by level
    std::vector<int> tile_prefix;   // per node (aligned with nodes), exclusive prefix WITHIN its level (+1 total slot per level)
    std::vector<int> ltr_prefix;    // L-TRSM strips
    std::vector<int> utr_prefix;    // U-TRSM column chunks
    std::vector<int> inv_prefix;    // diagonal sub-block inversion tasks
    std::vector<int> zltr_prefix;   // complex path: 64-row L strips (the 64-column U chunks reuse bwd_prefix)
    std::vector<int> lvl_poff;      // [nlevels+1] offset of each level's prefix arrays (size nodes_in_level+1)
    std::vector<int> lvl_soff;      // [nlevels+1] offset of each level's Schur prefix arrays (big group | small group)
    std::vector<int> n_big;         // per level: nodes using the 128x128 tile configuration (listed first)
    std::vector<int> fwd_prefix, bwd_prefix;  // solve work units
    std::vector<int> max_nsupc;     // per level
    std::vector<size_t> diag_lds;   // per level: dynamic LDS bytes k_diag_lu needs (max over the level's nodes)
    std::vector<int> pk_prefix;     // cooperative mode: 4096-double chunks of each node's (L panel | dinv) payload
    std::vector<int64_t> pk_off;    // ... and its offset (doubles) inside the level's staging buffer
    std::vector<uint8_t> lvl_defer; // per level: some supernode's non-urgent tiles are deferred to its K-fused partner
    std::vector<int> sn_level;      // [nsupers] level of each supernode in this schedule (-1: not in it)
    std::vector<int4> ulist;        // urgent tiles (k, rt, ct, 0): per level [big group | small group]
    std::vector<int> u_off;         // [2*nlevels+1] offsets into ulist (2 groups per level)
    // device copies
    int *d_nodes = nullptr, *d_tile_prefix = nullptr, *d_ltr_prefix = nullptr, *d_utr_prefix = nullptr;
    int *d_fwd_prefix = nullptr, *d_bwd_prefix = nullptr, *d_inv_prefix = nullptr, *d_sn_level = nullptr, *d_zltr_prefix = nullptr;
    int4 *d_ulist = nullptr;
    int *d_pk_prefix = nullptr; int64_t *d_pk_off = nullptr;
};

struct Handle {
    int device = 0;
    sluamd_options_t opt{};
    HostStruct hs;
    int Pz = 1, myz = 0;
    // device arenas
    double *d_val = nullptr;
    int *d_lidx = nullptr, *d_uidx = nullptr, *d_ucolptr = nullptr, *d_unzcol = nullptr, *d_xsup = nullptr;
    std::vector<void *> d_misc;  // everything else to free
    DevTables T{};
    std::vector<LevelSched> sched;  // one per Z level (forests) or a single one
    hipStream_t stream = nullptr;
    hipStream_t user_stream = nullptr; bool has_user_stream = false;   // cooperative mode: the caller's (torch/RCCL) stream
    hipStream_t pstream = nullptr;          // high-priority stream for the panel kernels (look-ahead)
    std::vector<hipEvent_t> ev_pool;        // look-ahead dependency events
    size_t ev_pool_used = 0;
    int *d_info = nullptr;      // [0]=first zero pivot column (INT_MAX if none), [1]=tiny pivots, [2]=missing dest blocks
    double *d_x = nullptr; int64_t x_cap = 0;
    int64_t *d_apos = nullptr; double *d_aval = nullptr; int64_t a_nnz = 0;  // A's entries for device-side (re)distribution
    // iterative refinement (sluamd_dAttachMatrix): the ORIGINAL matrix in CSR + perm_c, and work vectors
    int *d_rfs_rp = nullptr, *d_rfs_ci = nullptr, *d_rfs_pc = nullptr; double *d_rfs_av = nullptr;
    double *d_rfs_work = nullptr; unsigned long long *d_rfs_s = nullptr; int64_t rfs_nnz = 0;
    bool z = false;                                         // complex16 (doublecomplex) values: 16-byte elements
    bool dinv_ready = false;                                // T.dinv holds the inverses for the current factors
    bool profile = false;                                   // per-kernel-family HIP-event timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_schur, ev_panel;
    size_t ev_schur_used = 0, ev_panel_used = 0;
    sluamd_stats_t st{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // host tables kept for stats
    std::vector<int> h_nsupr, h_ldu, h_ncolu;
    std::vector<int64_t> h_sn_dinv;
    // K-fused updates (see DevTables): host images, built by build_schedule
    std::vector<int> h_fuse_prev, h_defer, h_pair_roff, h_pair_coff, h_pair_rowmap, h_pair_colinfo;
    int fused_pairs = 0;
    int max_nsupc = 0;
};

// ================================================================================================
//                                          KERNELS
// ================================================================================================
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KC = 16;  // K chunk of the Schur GEMM pipeline
constexpr int PKC = 4096;  // cooperative mode: doubles per pack/unpack workgroup

__device__ __forceinline__ int find_node(const int *__restrict__ prefix, int nn, int id)
{   // largest i in [0,nn) with prefix[i] <= id   (prefix has nn+1 entries)
    int lo = 0, hi = nn;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= id) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void atomic_sub_f64(double *p, double v)
{
    unsafeAtomicAdd(p, -v);  // global_atomic_add_f64 (hardware fp64 atomic on gfx950)
}

// ---- diagonal block LU ----------------------------------------------------------------------------
// One workgroup per supernode of the level.  Arithmetic = right-looking elimination without pivoting as
// Local_Dgstrf2 (pdgstrf2.c:508-601; tiny-pivot replacement :544-560, zero-pivot info :568-571).
//   ns <= 128 : whole block factored inside LDS (rank-1 updates).
//   ns  > 128 : blocked by 32 columns: LDS-resident column panel, U12 = L11^-1 A12 per thread-column,
//               rank-32 trailing update with the panel rows held in registers.
// Afterwards the workgroup inverts the 32x32 diagonal sub-blocks of U_kk and of L_kk^T (unit) into
// T.dinv; the panel TRSM kernels use them (block TRSM with inverted 32x32 diagonal blocks).
constexpr int DB = 32;  // diagonal sub-block size

__device__ __forceinline__ void pivot_fix(double *p, int col1based, int replace_tiny, double thresh, int *info, double *s_piv)
{
    double v = *p;
    if (replace_tiny && fabs(v) < thresh) { v = (v < 0) ? -thresh : thresh; *p = v; atomicAdd(&info[1], 1); }
    if (v == 0.0) atomicMin(&info[0], col1based);
    *s_piv = v;
}

__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{   // wave-uniform source lane (compile-time after unrolling) -> v_readlane_b32 x2, no LDS
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// Unpivoted LU of the nb x nb (nb <= 32) block at P (LDS, column-major, ld), executed by ONE wave entirely in
// registers: lane r holds row r (identity-padded to 32), pivot rows are broadcast with v_readlane.
// s_rinv[j] receives 1/U(j,j) (1 for a zero pivot, which leaves the column unscaled like pdgstrf2.c:566-575).
__device__ __forceinline__ void wave_lu32(double *P, int ld, int nb, int col1, int replace_tiny, double thresh,
                                          int *info, double *s_rinv)
{
    const int lane = threadIdx.x & 63;
    double a[DB];
#pragma unroll
    for (int c = 0; c < DB; ++c) a[c] = (lane < nb && c < nb) ? P[c * ld + lane] : ((c == lane) ? 1.0 : 0.0);
#pragma unroll
    for (int j = 0; j < DB; ++j) {
        double p = lane_bcast(a[j], j);
        if (j < nb) {
            if (replace_tiny && fabs(p) < thresh) {
                p = (p < 0) ? -thresh : thresh;
                if (lane == j) a[j] = p;
                if (lane == 0) atomicAdd(&info[1], 1);
            }
            if (p == 0.0 && lane == 0) atomicMin(&info[0], col1 + j);
        }
        const double rinv = (p != 0.0) ? 1.0 / p : 1.0;
        if (lane == 0) s_rinv[j] = rinv;
        const bool below = lane > j;
        const double l = a[j] * rinv;
        if (below) a[j] = l;
#pragma unroll
        for (int c = j + 1; c < DB; ++c) {
            const double u = lane_bcast(a[c], j);
            if (below) a[c] -= l * u;
            if ((c & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bound the live range of the broadcast SGPRs
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < DB; ++c) if (lane < nb && c < nb) P[c * ld + lane] = a[c];
}

// Blocked right-looking LU of the diagonal block in place in HBM/L2 (the block is re-read through L2 only):
// per 32 columns: panel -> LDS, 32x32 head factored in registers by one wave, rows below solved one per thread
// in registers, U12 one column per thread in registers, rank-32 trailing update on fp64 MFMA.
// NSMAX (64/128/256) fixes the LDS strides at compile time so that the unrolled substitutions address LDS with
// immediate offsets.
template <int NSMAX>
__global__ __launch_bounds__(256) void k_diag_lu(DevTables T, const int *__restrict__ nodes,
                                                 int replace_tiny, double thresh, int *__restrict__ info)
{
    constexpr int UC = 64;                  // U12 is staged 64 columns at a time: keeps the workgroup at <= 82 KB of LDS
    constexpr int ldp = NSMAX + 1, lus = UC;
    __shared__ double s_a[DB * ldp + DB * lus];
    __shared__ double s_rinv[DB];
    const int k = nodes[blockIdx.x];
    if (T.own_G > 1 && (k % T.own_G) != T.own_g) return;   // cooperative mode: the owner of block column k factors it
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const int lda = T.sn_nsupr[k];
    double *A = T.val + T.sn_lval[k];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *Ps = s_a;                       // column panel: element (r, c) at Ps[c * ldp + r]
    double *Us = s_a + DB * ldp;            // 64-column slice of U12: element (kk, c) at Us[kk * lus + c]
    for (int jb = 0; jb < ns; jb += DB) {
        const int nb = min(DB, ns - jb), m = ns - jb, nc = m - nb;
#pragma unroll 8
        for (int idx = tid; idx < m * nb; idx += 256) { int r = idx % m, c = idx / m; Ps[c * ldp + r] = A[jb + r + (size_t) (jb + c) * lda]; }
        __syncthreads();
        if (wave == 0) wave_lu32(Ps, ldp, nb, fst + jb + 1, replace_tiny, thresh, info, s_rinv);
        __syncthreads();
        if (tid < nc) {   // L21 row: x U11 = a   (nc > 0 implies nb == 32)
            double x[DB];
#pragma unroll
            for (int c = 0; c < DB; ++c) x[c] = Ps[c * ldp + nb + tid];
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                double acc = x[j];
#pragma unroll
                for (int kk = 0; kk < j; ++kk) acc -= x[kk] * Ps[j * ldp + kk];
                x[j] = acc * s_rinv[j];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < DB; ++c) Ps[c * ldp + nb + tid] = x[c];
        }
        __syncthreads();
#pragma unroll 8
        for (int idx = tid; idx < m * nb; idx += 256) { int r = idx % m, c = idx / m; A[jb + r + (size_t) (jb + c) * lda] = Ps[c * ldp + r]; }
        if (nc > 0) {
            // U12 = L11^-1 A12 : one thread per column, forward substitution in registers, result back to HBM/L2
            for (int c = tid; c < nc; c += 256) {
                double *col = A + jb + (size_t) (jb + nb + c) * lda;
                double x[DB];
#pragma unroll
                for (int i2 = 0; i2 < DB; ++i2) x[i2] = col[i2];
#pragma unroll
                for (int i2 = 1; i2 < DB; ++i2) {
                    double acc = x[i2];
#pragma unroll
                    for (int kk = 0; kk < i2; ++kk) acc -= Ps[kk * ldp + i2] * x[kk];
                    x[i2] = acc;
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i2 = 0; i2 < DB; ++i2) col[i2] = x[i2];
            }
            __syncthreads();
            // A22 -= L21 U12 on MFMA, 64 columns of U12 at a time; A := U12^T, B := L21^T so that the 16 fast lanes run
            // along rows (contiguous in the column-major block).  4 output blocks per wave iteration: 4 independent
            // MFMA chains, and the 16 destination values per lane go as 16 loads then 16 stores (one L2 round trip).
            const int ntr = (nc + 15) >> 4;
            for (int c0 = 0; c0 < nc; c0 += UC) {
                const int ncc = min(UC, nc - c0), ntc = (ncc + 15) >> 4;
#pragma unroll 8
                for (int idx = tid; idx < DB * ncc; idx += 256) { int i2 = idx & 31, c = idx >> 5; Us[i2 * lus + c] = A[jb + i2 + (size_t) (jb + nb + c0 + c) * lda]; }
                __syncthreads();
                for (int t0 = wave * 4; t0 < ntr * ntc; t0 += 16) {
                    d4 acc[4];
                    int ti[4], tj[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int t = min(t0 + g, ntr * ntc - 1);
                        ti[g] = t % ntr; tj[g] = t / ntr;
                        acc[g] = (d4){0.0, 0.0, 0.0, 0.0};
                    }
#pragma unroll
                    for (int k4 = 0; k4 < DB; k4 += 4) {
                        const int kk = k4 + (lane >> 4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int rr = min(ti[g] * 16 + (lane & 15), nc - 1), cc = min(tj[g] * 16 + (lane & 15), ncc - 1);
                            acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(Us[kk * lus + cc], Ps[kk * ldp + nb + rr], acc[g], 0, 0, 0);
                        }
                    }
                    double old[4][4];
                    double *dst[4][4];
                    bool ok[4][4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = ti[g] * 16 + (lane & 15);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = tj[g] * 16 + (lane >> 4) + 4 * r;
                            ok[g][r] = (t0 + g < ntr * ntc) && row < nc && col < ncc;
                            dst[g][r] = A + jb + nb + min(row, nc - 1) + (size_t) (jb + nb + c0 + min(col, ncc - 1)) * lda;
                            old[g][r] = ok[g][r] ? *dst[g][r] : 0.0;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ok[g][r]) *dst[g][r] = old[g][r] - acc[g][r];
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// Inverses of the 32x32 diagonal sub-blocks of U_kk (typ 0) and L_kk^T (typ 1, unit), identity-padded past
// ns, written to T.dinv; 4 sub-blocks per 128-thread workgroup, one thread per column of an inverse
// (back substitution with the block and the private solution column staged in LDS).
__global__ __launch_bounds__(128) void k_diag_inv(DevTables T, const int *__restrict__ nodes,
                                                  const int *__restrict__ prefix, int nn)
{
    __shared__ double Bs[4][DB * (DB + 1)];
    __shared__ double Xi[4][DB * (DB + 1)];
    const int g = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int task = blockIdx.x * 4 + g;
    bool valid = task < prefix[nn];
    int k = 0, typ = 0, b = 0, ns = 0, lda = 1, nblk = 1;
    const double *A = nullptr;
    int ni = 0;
    if (valid) {
        ni = find_node(prefix, nn, task);
        k = nodes[ni];
        if (T.own_G > 1 && (k % T.own_G) != T.own_g) valid = false;
    }
    if (valid) {
        ns = T.xsup[k + 1] - T.xsup[k];
        nblk = (ns + DB - 1) / DB;
        const int rem = task - prefix[ni];
        typ = rem / nblk; b = rem - typ * nblk;
        lda = T.sn_nsupr[k];
        A = T.val + T.sn_lval[k];
    }
    const int o = b * DB;
    if (valid) {
        for (int i = 0; i < DB; ++i) {
            double v = (i == c) ? 1.0 : 0.0;
            if (o + i < ns && o + c < ns && i <= c) {
                if (typ == 0) v = A[o + i + (size_t) (o + c) * lda];               // U(i,c)
                else if (i < c) v = A[o + c + (size_t) (o + i) * lda];            // L(c,i) = (L^T)(i,c)
            }
            Bs[g][i * (DB + 1) + c] = v;
        }
    }
    __syncthreads();
    if (valid) {
        for (int i = c; i >= 0; --i) {
            double a = (i == c) ? 1.0 : 0.0;
            for (int jj = i + 1; jj <= c; ++jj) a -= Bs[g][i * (DB + 1) + jj] * Xi[g][jj * (DB + 1) + c];
            Xi[g][i * (DB + 1) + c] = a / Bs[g][i * (DB + 1) + i];
        }
        double *dst = T.dinv + T.sn_dinv[k] + (size_t) (typ * nblk + b) * DB * DB + c * DB;
        for (int i = 0; i < DB; ++i) dst[i] = (i <= c) ? Xi[g][i * (DB + 1) + c] : 0.0;
    }
}

// ---- panel TRSMs: blocked by 32 with inverted diagonal sub-blocks, GEMM parts on fp64 MFMA ---------------
// MODE 0  L(:,k) <- L(:,k) U_kk^-1        (dLPanelTrSolve, dtrfCommWrapper.c:120-223: TRSM R,U,N,N)
//         strip = 32 panel rows; T = U_kk.
// MODE 1  U(k,:) <- L_kk^-1 U(k,:)        (dTrs2_GatherTrsmScatter, pdgstrf2.c:804-840: gather, TRSM L,L,N,U,
//         scatter) solved as X^T L_kk^T = B^T on the skyline in place: strip = 32 non-empty U columns
//         (implicit zero padding above each segment), T = L_kk^T (unit upper).
// The 32 x ns strip lives in LDS for the whole solve: HBM traffic = one read + one write of the panel.
// Strip height RSv = 64 (nsp <= 128) or 32 (nsp <= 256): either way the workgroup needs <= 82 KB of LDS, so a panel
// workgroup fits beside ONE 128x128 Schur workgroup on a CU -- the high-priority look-ahead stream can then take any
// slot a finishing Schur workgroup frees instead of waiting for a whole idle CU.
// LDS images are split in 16-wide groups ([group][k][16]): a 16x4 MFMA fragment read touches 4 k-rows x 16
// consecutive doubles = all 64 banks once.  Xs = strip (RSv x nsp), Tb = double-buffered 32x32 operand block.
static inline int trsm_rs(int nsp)
{
    // 32-row strips (80 KB of LDS, two workgroups per CU or one beside a Schur workgroup) measured ~2 % slower end to
    // end than 64-row strips (144 KB, one per CU) on 100^3; kept selectable
    static const bool rs32 = getenv("SLUAMD_TRSM_RS32") != nullptr;
    return (nsp > 128 && rs32) ? 32 : 64;
}
constexpr int TB_SZ = DB * 48;   // doubles per operand buffer: [32][48] (c-fastest chunks) or [32][34] (k-fastest chunks)
static inline size_t trsm_lds_bytes(int nsp) { return sizeof(double) * ((size_t) trsm_rs(nsp) * nsp + 2 * TB_SZ); }

// The solve is a flat pipeline of 32x32 operand blocks ("chunks"): for every block column jb the off-diagonal
// blocks T(kc, jb), kc = 0, 32, .. jb-32, then the inverted diagonal block inv(T_jj).  Chunks are fetched from L2
// into registers TWO iterations ahead (the chain is latency-bound, not bandwidth-bound); one barrier per chunk guards
// the LDS double buffer.  A wave only ever reads and writes its own 16 strip rows, so the strip needs no barrier.
template <int MODE, int RSv>
__device__ __forceinline__ void panel_trsm_body(const DevTables &T, int k, int strip, double *sm)
{
    constexpr int NT = RSv * 4;            // one wave per 16 strip rows
    constexpr int PQ = DB * DB / NT;       // chunk elements per thread
    constexpr int ES = NT / 32;            // slow-index stride of the chunk loader
    const int klst = T.xsup[k + 1], ns = klst - T.xsup[k];
    const int nsp = (ns + DB - 1) & ~(DB - 1);
    const int lda = T.sn_nsupr[k];
    const int nblk = nsp / DB;
    double *A = T.val + T.sn_lval[k];
    double *Uv = T.val + T.sn_uval[k];
    const double *dinv = T.dinv + T.sn_dinv[k] + (MODE == 0 ? 0 : (size_t) nblk * DB * DB);
    double *Xs = sm;                          // [RSv/16][nsp][16]: element (r, c) at ((r>>4)*nsp + c)*16 + (r&15)
    double *Tb = sm + (size_t) RSv * nsp;     // [2 buffers] x one 32x32 operand block; the element (kk, cc) sits at cc*34 + kk when
                                              // the chunk was fetched k-fastest (U_kk blocks, inverse blocks) and at kk*48 + cc when it
                                              // was fetched c-fastest (L_kk^T blocks): coalesced fetch, conflict-free stash AND fragment reads
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // per-strip-row skyline metadata (MODE 1) lives in the Tb region while the pipeline is not running: the workgroup
    // then needs exactly RSv*nsp + 2048 doubles (80 KB for nsp = 256) and two of them fit on one CU
    int *s_cp = reinterpret_cast<int *>(Tb), *s_ld = s_cp + RSv;
    auto locate = [&]() {
        if (tid < RSv) {
            const int cr = strip * RSv + tid;
            int cp = 0, ld = nsp;
            if (cr < T.sn_ncolu[k]) {
                const int ub0 = T.sn_ub_off[k], nub = T.sn_nub[k];
                int lo = 0, hi = nub;
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (T.ub_stcol[ub0 + mid] <= cr) lo = mid; else hi = mid; }
                const int b = ub0 + lo;
                const int64_t u0 = T.sn_uidx[k] + T.ub_iukp[b];
                const int jj = T.unzcol[u0 + (cr - T.ub_stcol[b])];
                ld = ns - (klst - T.uidx[u0 + jj]);
                cp = T.ucolptr[u0 + jj];
            }
            s_cp[tid] = cp; s_ld[tid] = ld;
        }
        __syncthreads();
    };

    if (MODE == 0) {
        const int row0 = ns + strip * RSv;
#pragma unroll 8
        for (int idx = tid; idx < RSv * nsp; idx += NT) {
            const int r = idx % RSv, c = idx / RSv;
            double v = 0.0;
            if (c < ns && row0 + r < lda) v = A[row0 + r + (size_t) c * lda];
            Xs[((r >> 4) * nsp + c) * 16 + (r & 15)] = v;
        }
    } else {
        locate();
        for (int idx = tid; idx < RSv * nsp; idx += NT) {
            const int c = idx % nsp, r = idx / nsp;
            double v = 0.0;
            const int ld = s_ld[r];
            if (c >= ld && c < ns) v = Uv[s_cp[r] + (c - ld)];
            Xs[((r >> 4) * nsp + c) * 16 + (r & 15)] = v;
        }
        __syncthreads();   // the metadata overlay is about to be overwritten by the first chunk
    }

    // chunk (jb, t): t < jb/32 -> T(32 t, jb) ; t == jb/32 -> inv(T_jj)
    const int e0 = tid & 31, e1 = tid >> 5;   // fast / slow element index of the 32x32 chunk (slow: e1 + ES*q)
    auto fetch = [&](double *pv, int jb, int t) {
        if (t * DB < jb) {
            const int kc = t * DB;
#pragma unroll
            for (int q = 0; q < PQ; ++q) {
                // MODE 0: T(k,c) = U_kk(k,c) = A[k + c*lda], k fastest ; MODE 1: T(k,c) = L_kk(c,k) = A[c + k*lda], c fastest
                const int kg = kc + (MODE == 0 ? e0 : e1 + ES * q), cg = jb + (MODE == 0 ? e1 + ES * q : e0);
                pv[q] = (kg < ns && cg < ns) ? (MODE == 0 ? A[kg + (size_t) cg * lda] : A[cg + (size_t) kg * lda]) : 0.0;
            }
        } else {
            const double *dblk = dinv + (size_t) (jb / DB) * DB * DB;
#pragma unroll
            for (int q = 0; q < PQ; ++q) pv[q] = dblk[(e1 + ES * q) * DB + e0];   // D(kk = e0, cc = e1 + ES*q)
        }
    };
    auto stash = [&](const double *pv, int jb, int t, int buf) {
        double *tb = Tb + buf * TB_SZ;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (t * DB < jb && MODE == 1) tb[(e1 + ES * q) * 48 + e0] = pv[q];   // (kk = e1 + ES q, cc = e0)
            else tb[(e1 + ES * q) * 34 + e0] = pv[q];                            // (kk = e0, cc = e1 + ES q)
        }
    };
    auto advance = [&](int &jb, int &t) { if (++t > jb / DB) { jb += DB; t = 0; } };

    d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
    const double *xa = Xs + ((size_t) wave * nsp + (lane >> 4)) * 16 + (lane & 15);
    auto compute = [&](int jb, int t, int buf) {
        // fragment element (kk = k4 + lane>>4, cc = half*16 + lane&15)
        const bool cfast = (MODE == 1) && (t < jb / DB);
        const int sk = cfast ? 48 : 1, sc = cfast ? 1 : 34;
        const double *tb0 = Tb + buf * TB_SZ + (lane >> 4) * sk + (lane & 15) * sc;
        const double *tb1 = tb0 + 16 * sc;
        const int ks = 4 * sk;   // pointer step per k4
        if (t < jb / DB) {
            const double *a = xa + (size_t) (t * DB) * 16;
#pragma unroll
            for (int k4 = 0; k4 < DB; k4 += 4) {
                const double av = a[k4 * 16];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tb0[(k4 >> 2) * ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tb1[(k4 >> 2) * ks], acc1, 0, 0, 0);
            }
        } else {
            // rhs = X_jb - acc (own 16 rows), then X_jb = rhs * inv(T_jj)
            double *x0 = Xs + ((size_t) wave * nsp + jb + (lane & 15)) * 16 + (lane >> 4);
            double *x1 = x0 + 16 * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) { x0[4 * r] -= acc0[r]; x1[4 * r] -= acc1[r]; }
            acc0 = (d4){0.0, 0.0, 0.0, 0.0}; acc1 = (d4){0.0, 0.0, 0.0, 0.0};
            const double *a = xa + (size_t) jb * 16;
#pragma unroll
            for (int k4 = 0; k4 < DB; k4 += 4) {
                const double av = a[k4 * 16];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tb0[(k4 >> 2) * ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tb1[(k4 >> 2) * ks], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { x0[4 * r] = acc0[r]; x1[4 * r] = acc1[r]; }
            acc0 = (d4){0.0, 0.0, 0.0, 0.0}; acc1 = (d4){0.0, 0.0, 0.0, 0.0};
        }
    };

    // software pipeline, two chunks in flight in registers (pA: chunk i+1, pB: chunk i+2, roles swap every iteration)
    double pA[PQ], pB[PQ];
    int cj = 0, ct = 0;            // chunk being computed
    int nj = 0, nt = 0;            // chunk held in the "next" register set
    int fj = 0, ft = 0;            // chunk held in the "far" register set
    fetch(pA, 0, 0);
    stash(pA, 0, 0, 0);
    advance(nj, nt);
    fj = nj; ft = nt; advance(fj, ft);
    if (nj < nsp) fetch(pA, nj, nt);
    if (fj < nsp) fetch(pB, fj, ft);
    __syncthreads();
    int buf = 0;
    while (cj < nsp) {
        // iteration with roles (next = pA, far = pB)
        compute(cj, ct, buf);
        if (nj < nsp) stash(pA, nj, nt, buf ^ 1);
        cj = nj; ct = nt; nj = fj; nt = ft; advance(fj, ft);
        if (fj < nsp) fetch(pA, fj, ft);           // pA is free again: becomes the new "far" set
        __syncthreads();
        buf ^= 1;
        if (cj >= nsp) break;
        // iteration with roles swapped (next = pB, far = pA)
        compute(cj, ct, buf);
        if (nj < nsp) stash(pB, nj, nt, buf ^ 1);
        cj = nj; ct = nt; nj = fj; nt = ft; advance(fj, ft);
        if (fj < nsp) fetch(pB, fj, ft);
        __syncthreads();
        buf ^= 1;
    }

    if (MODE == 0) {
        const int row0 = ns + strip * RSv;
#pragma unroll 8
        for (int idx = tid; idx < RSv * ns; idx += NT) {
            const int r = idx % RSv, c = idx / RSv;
            if (row0 + r < lda) A[row0 + r + (size_t) c * lda] = Xs[((r >> 4) * nsp + c) * 16 + (r & 15)];
        }
    } else {
        locate();
        for (int idx = tid; idx < RSv * nsp; idx += NT) {
            const int c = idx % nsp, r = idx / nsp;
            const int ld = s_ld[r];
            if (c >= ld && c < ns) Uv[s_cp[r] + (c - ld)] = Xs[((r >> 4) * nsp + c) * 16 + (r & 15)];
        }
    }
}

// L strips (blocks [0, nl)) and U column strips (blocks [nl, nl+nu)) of one level in ONE launch
template <int RSv>
__global__ __launch_bounds__(RSv * 4) void k_panel_trsm(DevTables T, const int *__restrict__ nodes,
                                                        const int *__restrict__ lprefix, const int *__restrict__ uprefix,
                                                        int nn, int nl)
{
    extern __shared__ double sm[];
    if ((int) blockIdx.x < nl) {
        const int ni = find_node(lprefix, nn, blockIdx.x);
        if (T.own_G > 1 && (nodes[ni] % T.own_G) != T.own_g) return;
        panel_trsm_body<0, RSv>(T, nodes[ni], blockIdx.x - lprefix[ni], sm);
    } else {
        const int id = blockIdx.x - nl;
        const int ni = find_node(uprefix, nn, id);
        panel_trsm_body<1, RSv>(T, nodes[ni], id - uprefix[ni], sm);
    }
}

// ---- iterative refinement (pdgsrfs3d, SRC/double/pdgsrfs.c:345-510) --------------------------------
// One pass over the CSR matrix does both of the reference's pdgsmv calls (abs = 0 and abs = 1, pdgsmv.c): residual
// r = b - A x (stored permuted, r_perm[perm_c[i]] = r_i: the right-hand side of the triangular solves on Pc A Pc^T),
// temp = |A||x| + |b|, and the componentwise backward error max_i |r_i| / temp_i with the SAFE1/SAFE2 guards
// (:463-469), reduced per workgroup and combined with an integer atomicMax (non-negative doubles order like
// their bit patterns).  HBM-bound: 12 B per nonzero + 32 B per row.
__global__ __launch_bounds__(256) void k_rfs_residual(int n, const int *__restrict__ rp, const int *__restrict__ ci,
                                                      const double *__restrict__ av, const double *__restrict__ x,
                                                      const double *__restrict__ b, const int *__restrict__ pc,
                                                      double *__restrict__ r_perm, unsigned long long *__restrict__ s_out,
                                                      double safe1, double safe2)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double q = 0.0;
    if (i < n) {
        double ax = 0.0, t = 0.0;
        for (int e = rp[i]; e < rp[i + 1]; ++e) {
            const double a = av[e], xv = x[ci[e]];
            ax += a * xv;
            t += fabs(a) * fabs(xv);
        }
        const double r = b[i] - ax;
        t += fabs(b[i]);
        r_perm[pc[i]] = r;
        if (t > safe2) q = fabs(r) / t;
        else if (t != 0.0) q = (safe1 + fabs(r)) / t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q = fmax(q, __shfl_xor(q, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        q = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(s_out, (unsigned long long) __double_as_longlong(q));
    }
}

__global__ __launch_bounds__(256) void k_rfs_update(int n, const int *__restrict__ pc, const double *__restrict__ dx_perm,
                                                    double *__restrict__ x)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] += dx_perm[pc[i]];
}

// ---- cooperative (owner-computes) mode helpers ----------------------------------------------------
// Pack (unpack = 0): the level's staging buffer receives the factored L panel + inverted diagonal sub-blocks of the
// supernodes this rank owns, zeros elsewhere; after a sum all-reduce over the group every rank holds every panel
// of the level.  Unpack (unpack = 1): copy the panels this rank does NOT own from the staging buffer into its arena.
__global__ __launch_bounds__(256) void k_coop_pack(DevTables T, const int *__restrict__ nodes, const int *__restrict__ cprefix,
                                                   const int64_t *__restrict__ off, int nn, double *__restrict__ stage, int unpack)
{
    const int ni = find_node(cprefix, nn, blockIdx.x);
    const int k = nodes[ni];
    const bool mine = (k % T.own_G) == T.own_g;
    if (unpack && mine) return;
    const int ns = T.xsup[k + 1] - T.xsup[k];
    const int64_t lsz = (int64_t) T.sn_nsupr[k] * ns, tot = lsz + (int64_t) 2 * ((ns + DB - 1) / DB) * DB * DB;
    const int64_t e0 = (int64_t) (blockIdx.x - cprefix[ni]) * PKC;
    const int64_t e1 = (e0 + PKC < tot) ? e0 + PKC : tot;
    double *Lp = T.val + T.sn_lval[k], *Dp = T.dinv + T.sn_dinv[k], *Sp = stage + off[ni];
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        double *p = (e < lsz) ? Lp + e : Dp + (e - lsz);
        if (unpack) *p = Sp[e]; else Sp[e] = mine ? *p : 0.0;
    }
}

// Zero the U blocks (k, jb) this rank does not own (jb % G != g): after a sum all-reduce of the forest's U rows
// every rank holds the complete factor.
__global__ __launch_bounds__(256) void k_coop_mask_u(DevTables T, const int *__restrict__ nodes)
{
    const int k = nodes[blockIdx.x];
    double *U = T.val + T.sn_uval[k];
    const int ub0 = T.sn_ub_off[k], nub = T.sn_nub[k];
    for (int b = 0; b < nub; ++b) {
        const int jb = T.ub_gid[ub0 + b];
        if ((jb % T.own_G) == T.own_g) continue;
        const int64_t ip = T.sn_uidx[k] + T.ub_iukp[ub0 + b];
        const int start = T.ucolptr[ip], nnz = T.uidx[ip - 1];
        for (int e = threadIdx.x; e < nnz; e += 256) U[start + e] = 0.0;
    }
}

// ---- Schur complement update: fused gather -> MFMA fp64 GEMM -> scatter ---------------------------
// One workgroup (4 waves) per 64x64 tile of one (L block, U block) pair of one supernode of the level.
// MFMA v_mfma_f64_16x16x4_f64 computes D[i][j] = sum_k A[i][k] B[k][j] with lane l holding
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[(l>>4)+4r][l&15].  We feed A := U^T (i = tile column) and
// B := L^T (j = tile row) so that the 16 fast lanes of every accumulator register run along tile ROWS:
// the scatter then writes 128-byte runs of a destination column (column-major L panel / U skyline).

// Two tile configurations: 128x128 (wide supernodes, big block pairs: 4x4 MFMA blocks per wave) and 64x64
// (everything else).  Software pipeline: the next K chunk is fetched from HBM/L2 into registers while the
// MFMAs of the current chunk run out of the other LDS buffer (one barrier per chunk).
template <int TMv, int TNv, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : (TMv == 128 ? 2 : 4))) void k_schur(DevTables T, const int *__restrict__ nodes,
                                                                    const int *__restrict__ prefix, int nn, int id_base,
                                                                    int ntiles, int *__restrict__ info,
                                                                    const int4 *__restrict__ ulist, const int *__restrict__ sn_level,
                                                                    int skip_level)
{
    constexpr int LDL = TMv + 16;   // == 16 mod 32 doubles: conflict-free ds_read_b64 fragment reads
    constexpr int LDU = TNv + 17;   // odd: the k-major U stash (16 lanes x stride LDU) spreads over all banks too
    constexpr int NT = NW * 64;                         // threads per workgroup
    constexpr int WR = (NW == 8) ? 4 : 2, WC = 2;       // wave grid (rows x cols): 4 waves = 2x2, 8 waves = 4x2
    constexpr int NBR = TMv / (16 * WR), NBC = TNv / (16 * WC);   // 16x16 MFMA blocks per wave (rows, cols)
    constexpr int LQ = TMv * KC / NT, UQ = TNv * KC / NT;         // prefetch registers per thread
    constexpr int LKS = NT / TMv;                       // k stride of the L loader
    constexpr int UJS = NT / 16;                        // column stride of the U loader
    __shared__ double Ls[2][KC * LDL];
    __shared__ double Us[2][KC * LDU];
    __shared__ int s_ind[256 + 8];
    __shared__ int s_rowmap[TMv];
    __shared__ int s_colmap[TNv];
    __shared__ int s_cptr[TNv];   // value offset of tile column j inside U(k,:)
    __shared__ int s_lead[TNv];   // ns - seg (leading zeros) of tile column j
    __shared__ int s_cptr2[TNv];  // the same two for the fused predecessor supernode (K-fused chain update)
    __shared__ int s_lead2[TNv];
    __shared__ int s_jj[TNv];     // column id inside supernode jb
    __shared__ int64_t s_dbase;
    __shared__ int s_dinfo[4];

    const int tid = threadIdx.x;
    // XCD-aware mapping: workgroup b runs on XCD b%8; give every XCD a contiguous range of tiles so that the
    // row tile (L rows) shared by consecutive tiles stays in ONE XCD's L2
    int bid;
    {
        const int chunk = (ntiles + 7) >> 3;
        bid = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= chunk || bid >= ntiles) return;
        bid += id_base;
    }
    // look-ahead split: `ulist` != null -> explicit (k, row tile, col tile) list of the tiles that update the NEXT
    // level's panels ("urgent"); otherwise the full tile grid, minus those tiles when skip_level >= 0
    int k, rt, ct;
    if (ulist) {
        const int4 u = ulist[bid];
        k = u.x; rt = u.y; ct = u.z;
    } else {
        const int ni = find_node(prefix, nn, bid);
        k = nodes[ni];
        const int local = bid - prefix[ni];
        const int nct = T.sn_nct[k];
        rt = local / nct; ct = local - rt * nct;
    }
    const int4 R = T.rtile[T.sn_rt_off[k] + rt];
    const int4 C = T.ctile[T.sn_ct_off[k] + ct];
    const int lb = T.sn_lb_off[k] + R.x, ub = T.sn_ub_off[k] + C.x;
    const int nr = R.z, nc = C.z;
    const int ib = T.lb_gid[lb], jb = T.ub_gid[ub];
    if (T.own_G > 1 && (jb % T.own_G) != T.own_g) return;   // cooperative mode: owner of destination block column jb
    if (!ulist && skip_level >= 0 && (sn_level[ib] == skip_level || sn_level[jb] == skip_level)) return;  // done by the urgent launch
    if (!ulist && T.defer && T.defer[k]) return;   // K-fused: the partner supernode's tiles apply this update
    const int klst = T.xsup[k + 1], ns = klst - T.xsup[k];
    const int lda = T.sn_nsupr[k];
    const int *lsub = T.lidx + T.sn_lidx[k] + T.lb_lptr[lb] + R.y;  // global row ids of the tile rows
    const int64_t uix0 = T.sn_uidx[k] + T.ub_iukp[ub];
    const double *Lp = T.val + T.sn_lval[k] + R.w;                   // first tile row, column 0 of the panel
    const double *Uv = T.val + T.sn_uval[k];

    // K-fused update: the deferred updates of up to three predecessors of k in its chain (k = parent(k-1) = ..., consecutive
    // levels) are accumulated here in the same registers -> ONE prologue and ONE scatter for K = sum of their widths.  A
    // predecessor's block structure beyond k is a subset of k's: host-built maps give, per panel row / non-empty U
    // column of k, where the same global row / column sits in its panel / U row (or that it is absent = zeros).
    int nprev = 0;
    if (T.fuse_prev) { while (nprev < 3 && T.fuse_prev[3 * k + nprev] >= 0) ++nprev; }
    for (int t = tid; t < TNv; t += NT) {
        int cp = 0, lead = ns, jj = 0;
        if (t < nc) {
            jj = T.unzcol[uix0 + C.y + t];
            lead = ns - (klst - T.uidx[uix0 + jj]);
            cp = T.ucolptr[uix0 + jj];
        }
        s_cptr[t] = cp; s_lead[t] = lead; s_jj[t] = jj;
    }
    // ---- destination lookup (dscatter_l :138-147 / scatter_u :593-602 linear searches -> binary search) ----
    if ((tid >> 6) == NW - 1) {
        // one wave scans the gid directory of the destination panel / row with ONE coalesced load per 64 blocks and a
        // ballot, instead of a binary search whose every step is a dependent L2 round trip
        const int ln = tid & 63;
        const bool ldest = ib >= jb;
        const int o = ldest ? T.sn_lb_off[jb] : T.sn_ub_off[ib];
        const int nb = ldest ? T.sn_nlb[jb] : T.sn_nub[ib];
        const int *dir = ldest ? T.lbs_gid : T.ub_gid;
        const int want = ldest ? ib : jb;
        int pos = -1;
        for (int base = 0; base < nb && pos < 0; base += 64) {
            const int g = (base + ln < nb) ? dir[o + base + ln] : -1;
            const unsigned long long m = __ballot(g == want);
            if (m) pos = base + __ffsll((long long) m) - 1;
        }
        if (ln == 0) {
            if (pos >= 0) {
                if (ldest) {
                    const int d = o + T.lbs_idx[o + pos];
                    s_dinfo[0] = T.lb_rowoff[d]; s_dinfo[1] = T.lb_lptr[d]; s_dinfo[2] = T.lb_nbrow[d];
                    s_dbase = T.sn_lval[jb];
                } else {
                    s_dinfo[0] = T.ub_iukp[o + pos];
                    s_dbase = T.sn_uval[ib];
                }
            } else atomicAdd(&info[2], 1);
            s_dinfo[3] = pos >= 0;
        }
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int rm0 = (wave % WR) * (TMv / WR), cn0 = (wave / WR) * (TNv / WC);
    d4 acc[NBC][NBR];
#pragma unroll
    for (int a = 0; a < NBC; ++a)
#pragma unroll
        for (int b = 0; b < NBR; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

    const int li = tid % TMv, lk = tid / TMv;           // L loader: row li, k = lk + LKS*q
    const int uk = tid & 15, uj = tid >> 4;             // U loader: k = uk, col = uj + UJS*q
    double pl[LQ], pu[UQ];
    int ucp[UQ], uld[UQ];
    bool lrow_ok = li < nr;
    // per-source state of the K loop (source 0 = fused predecessor ka, source 1 = k itself)
    int ns_s = ns, lda_s = lda;
    const double *Lrow = Lp + li, *Uvs = Uv;

    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < LQ; ++q) {
            const int kg = k0 + lk + LKS * q;
            pl[q] = (lrow_ok && kg < ns_s) ? Lrow[(size_t) kg * lda_s] : 0.0;
        }
        const int kg = k0 + uk;
#pragma unroll
        for (int q = 0; q < UQ; ++q) pu[q] = (kg >= uld[q] && kg < ns_s) ? Uvs[ucp[q] + (kg - uld[q])] : 0.0;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LQ; ++q) Ls[buf][(lk + LKS * q) * LDL + li] = pl[q];
#pragma unroll
        for (int q = 0; q < UQ; ++q) Us[buf][uk * LDU + uj + UJS * q] = pu[q];
    };

    int buf = 0;
    for (int src = 0; src <= nprev; ++src) {       // farthest predecessor first, k itself last
        int kbeg;
        if (src < nprev) {
            const int pj = 3 * k + (nprev - 1 - src);
            const int ks = T.fuse_prev[pj];
            const int nss = T.xsup[ks + 1] - T.xsup[ks];
            const int *cinfo = T.pair_colinfo + 2 * (size_t) (T.pair_coff[pj] + T.ub_stcol[ub] + C.y);
            const int ra = (li < nr) ? T.pair_rowmap[T.pair_roff[pj] + R.w + li] : -1;
            for (int t = tid; t < TNv; t += NT) {
                s_cptr2[t] = (t < nc) ? cinfo[2 * t] : 0;
                s_lead2[t] = (t < nc) ? cinfo[2 * t + 1] : nss;
            }
            __syncthreads();
            ns_s = nss; lda_s = T.sn_nsupr[ks]; Lrow = T.val + T.sn_lval[ks] + max(ra, 0); Uvs = T.val + T.sn_uval[ks];
            kbeg = (nss - T.sn_ldu[ks]) & ~3; lrow_ok = ra >= 0;
#pragma unroll
            for (int q = 0; q < UQ; ++q) { ucp[q] = s_cptr2[uj + UJS * q]; uld[q] = s_lead2[uj + UJS * q]; }
        } else {
            ns_s = ns; lda_s = lda; Lrow = Lp + li; Uvs = Uv; lrow_ok = li < nr;
            kbeg = (ns - T.sn_ldu[k]) & ~3;              // U is zero above its tallest segment: skip those k
#pragma unroll
            for (int q = 0; q < UQ; ++q) { ucp[q] = s_cptr[uj + UJS * q]; uld[q] = s_lead[uj + UJS * q]; }
        }
        // (re)start of the software pipeline: every wave is past the last chunk's MFMAs (closing barrier of the loop)
        fetch(kbeg);
        stash(buf);
        __syncthreads();
        for (int k0 = kbeg; k0 < ns_s; k0 += KC) {
            const bool more = k0 + KC < ns_s;
            if (more) fetch(k0 + KC);
            const double *Lb = Ls[buf], *Ub = Us[buf];
#pragma unroll
            for (int k4 = 0; k4 < KC; k4 += 4) {
                const int kr = k4 + (lane >> 4);
                double a[NBC], b[NBR];
#pragma unroll
                for (int c = 0; c < NBC; ++c) a[c] = Ub[kr * LDU + cn0 + 16 * c + (lane & 15)];
#pragma unroll
                for (int r = 0; r < NBR; ++r) b[r] = Lb[kr * LDL + rm0 + 16 * r + (lane & 15)];
#pragma unroll
                for (int c = 0; c < NBC; ++c)
#pragma unroll
                    for (int r = 0; r < NBR; ++r) acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], b[r], acc[c][r], 0, 0, 0);
            }
            if (more) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- scatter (epilogue) ----------------------------------------------------------------------
    if (!s_dinfo[3]) return;
    double *dst = T.val + s_dbase;
    if (ib >= jb) {
        // indirect[rel] = position of global row (xsup[ib]+rel) inside destination block L(ib,jb)
        const int *drows = T.lidx + T.sn_lidx[jb] + s_dinfo[1];
        const int fnz = T.xsup[ib], dn = s_dinfo[2];
        for (int i = tid; i < dn; i += NT) s_ind[drows[i] - fnz] = i;
        __syncthreads();
        for (int t = tid; t < TMv; t += NT) s_rowmap[t] = (t < nr) ? s_dinfo[0] + s_ind[lsub[t] - fnz] : 0;
        const int ldv = T.sn_nsupr[jb];
        for (int t = tid; t < TNv; t += NT) s_colmap[t] = s_jj[t] * ldv;
    } else {
        const int64_t d0 = T.sn_uidx[ib] + s_dinfo[0];
        for (int t = tid; t < TMv; t += NT) s_rowmap[t] = (t < nr) ? lsub[t] : 0;
        for (int t = tid; t < TNv; t += NT) {
            int cm = 0;
            if (t < nc) cm = T.ucolptr[d0 + s_jj[t]] - T.uidx[d0 + s_jj[t]];  // colptr - fstnz
            s_colmap[t] = cm;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < NBC; ++ci)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = cn0 + 16 * ci + (lane >> 4) + 4 * r;
            if (col < nc) {
                double *dcol = dst + s_colmap[col];
#pragma unroll
                for (int ri = 0; ri < NBR; ++ri) {
                    const int row = rm0 + 16 * ri + (lane & 15);
                    if (row < nr) atomic_sub_f64(dcol + s_rowmap[row], acc[ci][ri][r]);
                }
            }
        }
}

// ---- triangular solves --------------------------------------------------------------------------
// x_k <- inv(L_kk) x_k (unit lower) or inv(U_kk) x_k (upper): one workgroup per supernode of the level,
// blocked by 32 with the inverted diagonal sub-blocks left in T.dinv by the factorisation (what the reference's
// DiagInv=YES solve does with Linv/Uinv, pdgstrs_lsum.c:414-520): 2 barriers per 32 columns.
template <bool LOWER>
__global__ __launch_bounds__(256) void k_solve_diag(DevTables T, const int *__restrict__ nodes, double *__restrict__ x,
                                                    int64_t ldx, int nrhs)
{
    extern __shared__ double xs[];  // ns x nrhs, then 32 x nrhs scratch
    const int k = nodes[blockIdx.x];
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const int lda = T.sn_nsupr[k];
    const double *A = T.val + T.sn_lval[k];
    const int nblk = (ns + DB - 1) / DB;
    const double *dinv = T.dinv + T.sn_dinv[k] + (LOWER ? (size_t) nblk * DB * DB : 0);
    double *ys = xs + (size_t) ns * nrhs;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < ns * nrhs; idx += 256) xs[idx] = x[fst + (idx % ns) + (int64_t) (idx / ns) * ldx];
    __syncthreads();
    for (int bb = 0; bb < nblk; ++bb) {
        const int b = LOWER ? bb : nblk - 1 - bb;
        const int o = b * DB, nb = min(DB, ns - o);
        const double *D = dinv + (size_t) b * DB * DB;
        // y = inv(T_bb) x_b : LOWER inv(L_bb)(r,c) = D(c,r) ; UPPER inv(U_bb)(r,c) = D(r,c) ; D(i,j) at D[j*32+i]
        for (int idx = tid; idx < nb * nrhs; idx += 256) {
            const int r = idx % nb, q = idx / nb;
            const double *xb = xs + o + q * ns;
            // fixed trip count + predication: the 32 loads of D are issued together instead of one L2 round trip each
            double dv[DB];
#pragma unroll
            for (int c = 0; c < DB; ++c) dv[c] = (LOWER ? (c <= r) : (c >= r && c < nb)) ? (LOWER ? D[r * DB + c] : D[c * DB + r]) : 0.0;
            double a = 0.0;
#pragma unroll
            for (int c = 0; c < DB; ++c) a += dv[c] * ((c < nb) ? xb[c] : 0.0);
            ys[r + q * DB] = a;
        }
        __syncthreads();
        // x_b = y ; remaining rows -= T(rows, b) y
        const int r0 = LOWER ? o + nb : 0, r1 = LOWER ? ns : o;
        for (int idx = tid; idx < (r1 - r0 + nb) * nrhs; idx += 256) {
            const int rr = idx % (r1 - r0 + nb), q = idx / (r1 - r0 + nb);
            if (rr < nb) { xs[o + rr + q * ns] = ys[rr + q * DB]; continue; }
            const int i = r0 + (rr - nb);
            double av[DB];
#pragma unroll
            for (int c = 0; c < DB; ++c) av[c] = (c < nb) ? A[i + (size_t) (o + c) * lda] : 0.0;
            double a = 0.0;
#pragma unroll
            for (int c = 0; c < DB; ++c) a += av[c] * ys[c + q * DB];
            xs[i + q * ns] -= a;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < ns * nrhs; idx += 256) x[fst + (idx % ns) + (int64_t) (idx / ns) * ldx] = xs[idx];
}

// lsum_i -= L_ik x_k for the off-diagonal rows of panel k (dlsum_fmod_inv, pdgstrs_lsum.c:414):
// one thread per panel row, 256-row strips; x_k staged in LDS.
__global__ __launch_bounds__(256) void k_fwd_update(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix,
                                                    int nn, double *__restrict__ x, int64_t ldx, int nrhs)
{
    extern __shared__ double xk[];  // ns x nrhs
    const int ni = find_node(prefix, nn, blockIdx.x);
    const int k = nodes[ni];
    const int strip = blockIdx.x - prefix[ni];
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const int lda = T.sn_nsupr[k];
    for (int idx = threadIdx.x; idx < ns * nrhs; idx += 256) xk[idx] = x[fst + (idx % ns) + (int64_t) (idx / ns) * ldx];
    __syncthreads();
    const int row = ns + strip * 256 + threadIdx.x;
    if (row >= lda) return;
    const double *L = T.val + T.sn_lval[k] + row;
    // global row id of panel row `row`: rows are listed block after block, 2 descriptor ints per block
    // -> precomputed flat map is not stored; walk the (few) blocks
    const int *lsub = T.lidx + T.sn_lidx[k];
    int p = BC_HEADER, base = 0, grow = -1;
    const int nb = lsub[0];
    for (int b = 0; b < nb; ++b) {
        const int nbrow = lsub[p + 1];
        if (row < base + nbrow) { grow = lsub[p + LB_DESCRIPTOR + (row - base)]; break; }
        base += nbrow; p += LB_DESCRIPTOR + nbrow;
    }
    for (int r = 0; r < nrhs; ++r) {
        double acc = 0.0;
        for (int kk = 0; kk < ns; ++kk) acc += L[(size_t) kk * lda] * xk[kk + r * ns];
        atomic_sub_f64(x + grow + (int64_t) r * ldx, acc);
    }
}

// x_k -= U(k, chunk of 64 non-empty columns) x_cols  (dlsum_bmod_inv, pdgstrs_lsum.c:1362):
// lanes run along the rows of supernode k (coalesced over the skyline segments), the 4 waves split the columns.
__global__ __launch_bounds__(256) void k_bwd_update(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix,
                                                    int nn, double *__restrict__ x, int64_t ldx, int nrhs)
{
    __shared__ int s_cp[64], s_ld[64], s_gc[64];
    __shared__ double s_red[4][64];
    const int ni = find_node(prefix, nn, blockIdx.x);
    const int k = nodes[ni];
    const int chunk = blockIdx.x - prefix[ni];
    const int fst = T.xsup[k], klst = T.xsup[k + 1], ns = klst - fst;
    const int ncol = min(64, T.sn_ncolu[k] - chunk * 64);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 64) {
        int cp = 0, ld = ns, gc = 0;
        if (tid < ncol) {
            const int c = chunk * 64 + tid;
            const int ub0 = T.sn_ub_off[k], nub = T.sn_nub[k];
            int lo = 0, hi = nub;
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (T.ub_stcol[ub0 + mid] <= c) lo = mid; else hi = mid; }
            const int b = ub0 + lo;
            const int64_t u0 = T.sn_uidx[k] + T.ub_iukp[b];
            const int jj = T.unzcol[u0 + (c - T.ub_stcol[b])];
            ld = ns - (klst - T.uidx[u0 + jj]);
            cp = T.ucolptr[u0 + jj];
            gc = T.xsup[T.ub_gid[b]] + jj;
        }
        s_cp[tid] = cp; s_ld[tid] = ld; s_gc[tid] = gc;
    }
    __syncthreads();
    const double *Uv = T.val + T.sn_uval[k];
    for (int r = 0; r < nrhs; ++r) {
        for (int rb = 0; rb < ns; rb += 64) {
            const int i = rb + lane;
            double acc = 0.0;
            for (int c = wave; c < ncol; c += 4) {
                const int ld = s_ld[c];
                if (i < ns && i >= ld) acc += Uv[s_cp[c] + (i - ld)] * x[s_gc[c] + (int64_t) r * ldx];
            }
            s_red[wave][lane] = acc;
            __syncthreads();
            if (wave == 0 && i < ns) {
                const double s = s_red[0][lane] + s_red[1][lane] + s_red[2][lane] + s_red[3][lane];
                if (s != 0.0) atomic_sub_f64(x + fst + i + (int64_t) r * ldx, s);
            }
            __syncthreads();
        }
    }
}

// A's entries -> value arena (device-side pddistribute): val[pos[e]] = a[e]
__global__ void k_scatter_values(double *__restrict__ val, const int64_t *__restrict__ pos, const double *__restrict__ a, int64_t nnz)
{
    int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) val[pos[e]] = a[e];
}

// MFMA layout self-test (used by tests): D = A(16x4) * B(4x16)
__global__ void k_mfma_selftest(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    d4 acc = (d4){0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

#include "sluamd_zkernels.inc"

// ================================================================================================
//                                       HOST: planning
// ================================================================================================
template <class Tv>
static int upload(std::vector<void *> &keep, const std::vector<Tv> &h, Tv **d)
{
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(Tv);
    HIPCHK(hipMalloc((void **) d, bytes));
    keep.push_back(*d);
    if (!h.empty()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(Tv), hipMemcpyHostToDevice));
    return 0;
}

static int flatten_view(const sluamd_dLUview_t *lu, HostStruct &hs, bool want_sizes_only)
{
    (void) want_sizes_only;
    if (!lu || !lu->xsup || lu->nsupers <= 0) { set_error("invalid LU view"); return SLUAMD_EINVAL; }
    if (lu->nprow != 1 || lu->npcol != 1) {
        set_error("round-1 library supports 1 x 1 x Pz process grids only (XY block-cyclic panels: later round)");
        return SLUAMD_EINVAL;
    }
    const int ns = lu->nsupers;
    hs.n = lu->n; hs.nsupers = ns;
    hs.xsup.assign(lu->xsup, lu->xsup + ns + 1);
    hs.lidx_off.assign(ns + 1, 0); hs.uidx_off.assign(ns + 1, 0);
    hs.lval_off.assign(ns + 1, 0); hs.uval_off.assign(ns + 1, 0);
    hs.present.assign(ns, 1);
    for (int k = 0; k < ns; ++k) {
        const int *li = lu->Lrowind_bc_ptr[k];
        if (!li) {   // panel of another Z layer's forest: not stored on this rank
            if (lu->npdep == 1) { set_error("L panel missing on a 1x1x1 grid"); return SLUAMD_ESTRUCT; }
            hs.present[k] = 0;
            hs.lidx_off[k + 1] = hs.lidx_off[k]; hs.lval_off[k + 1] = hs.lval_off[k];
            hs.uidx_off[k + 1] = hs.uidx_off[k]; hs.uval_off[k + 1] = hs.uval_off[k];
            continue;
        }
        const int nsupc = hs.xsup[k + 1] - hs.xsup[k];
        hs.lidx_off[k + 1] = hs.lidx_off[k] + BC_HEADER + (int64_t) li[0] * LB_DESCRIPTOR + li[1];
        hs.lval_off[k + 1] = hs.lval_off[k] + (int64_t) li[1] * nsupc;
        const int *ui = lu->Ufstnz_br_ptr[k];
        hs.uidx_off[k + 1] = hs.uidx_off[k] + (ui ? ui[2] : 0);
        hs.uval_off[k + 1] = hs.uval_off[k] + (ui ? ui[1] : 0);
    }
    hs.nnzL = hs.lval_off[ns]; hs.nnzU = hs.uval_off[ns];
    hs.lidx.resize(hs.lidx_off[ns]); hs.uidx.resize(hs.uidx_off[ns]);
    for (int k = 0; k < ns; ++k) {
        if (!hs.present[k]) continue;
        std::memcpy(hs.lidx.data() + hs.lidx_off[k], lu->Lrowind_bc_ptr[k], sizeof(int) * (hs.lidx_off[k + 1] - hs.lidx_off[k]));
        if (hs.uidx_off[k + 1] > hs.uidx_off[k])
            std::memcpy(hs.uidx.data() + hs.uidx_off[k], lu->Ufstnz_br_ptr[k], sizeof(int) * (hs.uidx_off[k + 1] - hs.uidx_off[k]));
    }
    return 0;
}

struct HostTables {
    std::vector<int64_t> sn_lval, sn_uval, sn_lidx, sn_uidx, sn_dinv;
    int64_t dinv_total = 0;
    std::vector<int> sn_nsupr, sn_ldu, sn_ncolu, sn_lb_off, sn_nlb, sn_ub_off, sn_nub, sn_rt_off, sn_nrt, sn_ct_off, sn_nct;
    std::vector<int> lb_gid, lb_nbrow, lb_rowoff, lb_lptr, lbs_gid, lbs_idx;
    std::vector<int> ub_gid, ub_ncols, ub_iukp, ub_stcol;
    std::vector<int> ucolptr, unzcol;
    std::vector<int4> rtile, ctile;
    std::vector<uint8_t> sn_big;   // 1: supernode uses the 128x128 Schur tile configuration
};

static int build_tables(Handle &H, HostTables &t)
{
    const HostStruct &hs = H.hs;
    const int ns = hs.nsupers;
    t.sn_lval.resize(ns); t.sn_uval.resize(ns); t.sn_lidx.resize(ns); t.sn_uidx.resize(ns); t.sn_dinv.resize(ns);
    t.sn_nsupr.resize(ns); t.sn_ldu.assign(ns, 0); t.sn_ncolu.assign(ns, 0);
    t.sn_lb_off.resize(ns); t.sn_nlb.resize(ns); t.sn_ub_off.resize(ns); t.sn_nub.resize(ns);
    t.sn_rt_off.resize(ns); t.sn_nrt.resize(ns); t.sn_ct_off.resize(ns); t.sn_nct.resize(ns);
    t.ucolptr.assign(hs.uidx.size(), 0); t.unzcol.assign(hs.uidx.size(), 0);
    H.max_nsupc = 0;
    auto &st = H.st;
    st.flops_schur_padded = st.flops_schur_exact = st.flops_panel = 0;
    st.schur_bytes_alg = 0;
    for (int k = 0; k < ns; ++k) {
        const int nsupc = hs.xsup[k + 1] - hs.xsup[k], klst = hs.xsup[k + 1];
        if (!hs.present[k]) {
            t.sn_lval[k] = t.sn_uval[k] = t.sn_lidx[k] = t.sn_uidx[k] = 0; t.sn_dinv[k] = t.dinv_total;
            t.sn_nsupr[k] = 0; t.sn_lb_off[k] = (int) t.lb_gid.size(); t.sn_nlb[k] = 0; t.sn_ub_off[k] = (int) t.ub_gid.size(); t.sn_nub[k] = 0;
            t.sn_rt_off[k] = (int) t.rtile.size(); t.sn_nrt[k] = 0; t.sn_ct_off[k] = (int) t.ctile.size(); t.sn_nct[k] = 0;
            t.sn_big.push_back(0);
            continue;
        }
        H.max_nsupc = std::max(H.max_nsupc, nsupc);
        if (nsupc > 256) { set_error("supernodes wider than 256 columns are not supported yet (set SUPERLU_MAXSUP <= 256)"); return SLUAMD_EINVAL; }
        t.sn_dinv[k] = t.dinv_total;
        t.dinv_total += (int64_t) 2 * ((nsupc + 31) / 32) * 32 * 32;
        const int *li = hs.lidx.data() + hs.lidx_off[k];
        const int nb = li[0], nsupr = li[1];
        t.sn_lval[k] = hs.lval_off[k];
        t.sn_uval[k] = hs.nnzL + hs.uval_off[k];
        t.sn_lidx[k] = hs.lidx_off[k]; t.sn_uidx[k] = hs.uidx_off[k];
        t.sn_nsupr[k] = nsupr;
        t.sn_lb_off[k] = (int) t.lb_gid.size(); t.sn_nlb[k] = nb;
        t.sn_rt_off[k] = (int) t.rtile.size();
        int p = BC_HEADER, rowoff = 0;
        std::vector<std::pair<int, int>> dir;
        for (int b = 0; b < nb; ++b) {
            const int gid = li[p], nbrow = li[p + 1];
            if (gid < k || gid >= ns || nbrow <= 0 || rowoff + nbrow > nsupr) { set_error("malformed L block"); return SLUAMD_ESTRUCT; }
            if (b == 0 && gid != k) { set_error("diagonal block must be the first L block of its panel"); return SLUAMD_ESTRUCT; }
            t.lb_gid.push_back(gid); t.lb_nbrow.push_back(nbrow); t.lb_rowoff.push_back(rowoff); t.lb_lptr.push_back(p + LB_DESCRIPTOR);
            dir.emplace_back(gid, b);
            rowoff += nbrow; p += LB_DESCRIPTOR + nbrow;
        }
        if (rowoff != nsupr) { set_error("L panel row count mismatch"); return SLUAMD_ESTRUCT; }
        std::sort(dir.begin(), dir.end());
        for (auto &d : dir) { t.lbs_gid.push_back(d.first); t.lbs_idx.push_back(d.second); }
        // U block row
        t.sn_ub_off[k] = (int) t.ub_gid.size();
        t.sn_ct_off[k] = (int) t.ctile.size();
        int nub = 0, ldu = 0, ncol_tot = 0;
        double exact = 0;
        if (hs.uidx_off[k + 1] > hs.uidx_off[k]) {
            const int *ui = hs.uidx.data() + hs.uidx_off[k];
            int *cp = t.ucolptr.data() + hs.uidx_off[k];
            int *nz = t.unzcol.data() + hs.uidx_off[k];
            nub = ui[0];
            int iukp = BR_HEADER; int64_t rukp = 0; int prev = k;
            for (int b = 0; b < nub; ++b) {
                const int jb = ui[iukp];
                if (jb <= prev || jb >= ns) { set_error("U blocks must be sorted by block column"); return SLUAMD_ESTRUCT; }
                prev = jb;
                const int nsj = hs.xsup[jb + 1] - hs.xsup[jb];
                int nc = 0;
                for (int jj = 0; jj < nsj; ++jj) {
                    const int seg = klst - ui[iukp + UB_DESCRIPTOR + jj];
                    if (seg < 0 || seg > nsupc) { set_error("bad U segment"); return SLUAMD_ESTRUCT; }
                    cp[iukp + UB_DESCRIPTOR + jj] = (int) rukp;
                    if (seg) { nz[iukp + UB_DESCRIPTOR + nc] = jj; ++nc; rukp += seg; ldu = std::max(ldu, seg); exact += seg; }
                }
                t.ub_gid.push_back(jb); t.ub_ncols.push_back(nc); t.ub_iukp.push_back(iukp + UB_DESCRIPTOR); t.ub_stcol.push_back(ncol_tot);
                ncol_tot += nc;
                iukp += UB_DESCRIPTOR + nsj;
            }
            if (rukp != hs.uval_off[k + 1] - hs.uval_off[k]) { set_error("U value count mismatch"); return SLUAMD_ESTRUCT; }
        }
        t.sn_nub[k] = nub; t.sn_ldu[k] = ldu; t.sn_ncolu[k] = ncol_tot;
        {   // tile configuration + tile lists of supernode k
            const int lb0 = t.sn_lb_off[k], ub0 = t.sn_ub_off[k];
            long t128r = 0, t128c = 0;
            for (int b = 1; b < nb; ++b) t128r += (t.lb_nbrow[lb0 + b] + 127) / 128;
            for (int b = 0; b < nub; ++b) t128c += (t.ub_ncols[ub0 + b] + 127) / 128;
            const double cells = (double) (nsupr - nsupc) * ncol_tot;
            const double util128 = (t128r * t128c) ? cells / ((double) t128r * t128c * 128.0 * 128.0) : 0.0;
            const bool big = !H.z && nsupc >= 96 && util128 >= 0.5 && !getenv("SLUAMD_NO_BIG_TILES");
            t.sn_big.push_back(big);
            const int tm = big ? 128 : 64;
            t.sn_rt_off[k] = (int) t.rtile.size();
            for (int b = 1; b < nb; ++b) {
                const int nbrow = t.lb_nbrow[lb0 + b], ro = t.lb_rowoff[lb0 + b];
                for (int r0 = 0; r0 < nbrow; r0 += tm) t.rtile.push_back(make_int4(b, r0, std::min(tm, nbrow - r0), ro + r0));
            }
            t.sn_nrt[k] = (int) t.rtile.size() - t.sn_rt_off[k];
            t.sn_ct_off[k] = (int) t.ctile.size();
            for (int b = 0; b < nub; ++b) {
                const int nc = t.ub_ncols[ub0 + b];
                for (int c0 = 0; c0 < nc; c0 += tm) t.ctile.push_back(make_int4(b, c0, std::min(tm, nc - c0), 0));
            }
        }
        t.sn_nct[k] = (int) t.ctile.size() - t.sn_ct_off[k];
        const double rrows = nsupr - nsupc;
        st.flops_schur_padded += 2.0 * rrows * ldu * ncol_tot;
        st.schur_bytes_alg += 16.0 * rrows * ncol_tot;   // read-modify-write of every updated destination element
        st.flops_schur_exact += 2.0 * rrows * exact;
        st.flops_panel += (2.0 / 3.0) * nsupc * (double) nsupc * nsupc + (double) nsupc * nsupc * rrows + (double) nsupc * exact;
    }
    if (H.z) { st.flops_schur_padded *= 4; st.flops_schur_exact *= 4; st.flops_panel *= 4; st.schur_bytes_alg *= 2; }   // complex multiply-add = 8 flop
    H.h_nsupr = t.sn_nsupr; H.h_ldu = t.sn_ldu; H.h_ncolu = t.sn_ncolu;
    return 0;
}

static void build_urgent_lists(const HostTables &t, int nsupers, const std::vector<int> &lvl, LevelSched &S);

// level schedule over `list` (a valid elimination order); node k's level = longest path of updates into it
// K-fused source a of supernode b (a < b members of one chain): b's tiles will also accumulate a's deferred update.  Needs
// every row / column of a's structure beyond b to exist in b's structure (true when b is an ancestor of a in the supernodal
// elimination tree); builds
// the row map (per panel row of b: row in a's panel or -1) and the column info (per non-empty U column of row b:
// value offset and leading zeros inside a's U row).  Rejects pairs whose a is much smaller than b (the fused tiles
// would multiply mostly zeros).
static bool build_pair_maps(const HostStruct &hs, const HostTables &t, int a, int b, std::vector<int> &rowmap, std::vector<int> &colinfo)
{
    const int sa = hs.xsup[a + 1] - hs.xsup[a];
    const int nsupr_b = t.sn_nsupr[b], ncolu_b = t.sn_ncolu[b];
    rowmap.assign(nsupr_b, -1);
    colinfo.assign(2 * (size_t) ncolu_b, 0);
    for (int c = 0; c < ncolu_b; ++c) colinfo[2 * c + 1] = sa;
    const int la = t.sn_lb_off[a], lb = t.sn_lb_off[b];
    int rows_a = 0, cols_a = 0;
    for (int x = 1; x < t.sn_nlb[a]; ++x) {
        const int g = t.lb_gid[la + x];
        if (g <= b) continue;                      // a's updates of the chain members up to b: their urgent tiles
        int y = -1;
        for (int q = 1; q < t.sn_nlb[b]; ++q) if (t.lb_gid[lb + q] == g) { y = q; break; }
        if (y < 0) return false;
        const int *ra = hs.lidx.data() + hs.lidx_off[a] + t.lb_lptr[la + x], *rb = hs.lidx.data() + hs.lidx_off[b] + t.lb_lptr[lb + y];
        const int na = t.lb_nbrow[la + x], nb = t.lb_nbrow[lb + y];
        for (int i = 0; i < na; ++i) {
            const int *f = std::find(rb, rb + nb, ra[i]);
            if (f == rb + nb) return false;
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
        for (int i = 0; i < na; ++i) {
            const int *f = std::find(cb, cb + nb, ca[i]);
            if (f == cb + nb) return false;
            const int c = t.ub_stcol[ub + y] + (int) (f - cb), jj = ca[i];
            colinfo[2 * c] = t.ucolptr[pa + jj];
            colinfo[2 * c + 1] = sa - (klst_a - hs.uidx[pa + jj]);
        }
        cols_a += na;
    }
    const int rows_b = nsupr_b - (hs.xsup[b + 1] - hs.xsup[b]);
    static const int pct = getenv("SLUAMD_FUSE_MIN_PCT") ? atoi(getenv("SLUAMD_FUSE_MIN_PCT")) : 75;
    return rows_a > 0 && cols_a > 0 && 100 * (int64_t) rows_a >= (int64_t) pct * rows_b && 100 * (int64_t) cols_a >= (int64_t) pct * ncolu_b;
}

static void build_schedule(Handle &H, const HostTables &t, const std::vector<int> &list, LevelSched &S)
{
    const HostStruct &hs = H.hs;
    const int ns = hs.nsupers;
    std::vector<int> lvl(ns, -1);
    for (int k : list) lvl[k] = 0;
    std::vector<int> sorted = list;
    std::sort(sorted.begin(), sorted.end());
    int maxl = 0;
    for (int k : sorted) {
        const int l1 = lvl[k] + 1;
        for (int b = 1; b < t.sn_nlb[k]; ++b) { int g = t.lb_gid[t.sn_lb_off[k] + b]; if (lvl[g] >= 0 && lvl[g] < l1) lvl[g] = l1; }
        for (int b = 0; b < t.sn_nub[k]; ++b) { int g = t.ub_gid[t.sn_ub_off[k] + b]; if (lvl[g] >= 0 && lvl[g] < l1) lvl[g] = l1; }
        maxl = std::max(maxl, lvl[k]);
    }
    S.nlevels = list.empty() ? 0 : maxl + 1;
    S.lvl_off.assign(S.nlevels + 1, 0);
    for (int k : sorted) S.lvl_off[lvl[k] + 1]++;
    for (int l = 0; l < S.nlevels; ++l) S.lvl_off[l + 1] += S.lvl_off[l];
    S.nodes.resize(sorted.size());
    std::vector<int> fill(S.lvl_off.begin(), S.lvl_off.end() - (S.nlevels ? 1 : 0));
    for (int pass = 1; pass >= 0; --pass)   // big-tile supernodes first inside each level
        for (int k : sorted) if ((int) t.sn_big[k] == pass) S.nodes[fill[lvl[k]]++] = k;
    S.n_big.assign(S.nlevels, 0);
    for (int k : sorted) if (t.sn_big[k]) S.n_big[lvl[k]]++;
    S.lvl_soff.assign(S.nlevels + 1, 0);
    for (int l = 0; l < S.nlevels; ++l) S.lvl_soff[l + 1] = S.lvl_soff[l] + (S.lvl_off[l + 1] - S.lvl_off[l]) + 2;
    S.lvl_poff.assign(S.nlevels + 1, 0);
    for (int l = 0; l < S.nlevels; ++l) S.lvl_poff[l + 1] = S.lvl_poff[l] + (S.lvl_off[l + 1] - S.lvl_off[l]) + 1;
    const int psz = S.lvl_poff[S.nlevels];
    S.tile_prefix.assign(S.lvl_soff[S.nlevels], 0); S.ltr_prefix.assign(psz, 0); S.utr_prefix.assign(psz, 0);
    S.fwd_prefix.assign(psz, 0); S.bwd_prefix.assign(psz, 0); S.inv_prefix.assign(psz, 0); S.zltr_prefix.assign(psz, 0);
    S.pk_prefix.assign(psz, 0); S.pk_off.assign(psz, 0);
    S.max_nsupc.assign(S.nlevels, 0);
    S.diag_lds.assign(S.nlevels, 0);
    for (int l = 0; l < S.nlevels; ++l)
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i)
            S.max_nsupc[l] = std::max(S.max_nsupc[l], hs.xsup[S.nodes[i] + 1] - hs.xsup[S.nodes[i]]);
    for (int l = 0; l < S.nlevels; ++l) {
        int po = S.lvl_poff[l];
        int so = S.lvl_soff[l];
        const int rs = trsm_rs((S.max_nsupc[l] + 31) & ~31);   // strip height of this level's TRSM launch
        for (int i = S.lvl_off[l]; i < S.lvl_off[l + 1]; ++i, ++po, ++so) {
            const int k = S.nodes[i];
            if (i - S.lvl_off[l] == S.n_big[l]) ++so;      // start of the small group: its own prefix, from 0
            S.tile_prefix[so + 1] = S.tile_prefix[so] + t.sn_nrt[k] * t.sn_nct[k];
            const int nsupc = hs.xsup[k + 1] - hs.xsup[k];
            const int rrows = t.sn_nsupr[k] - nsupc;
            S.max_nsupc[l] = std::max(S.max_nsupc[l], nsupc);
            S.diag_lds[l] = std::max(S.diag_lds[l], sizeof(double) * ((size_t) 32 * (nsupc | 1) + (size_t) 32 * nsupc));
            S.ltr_prefix[po + 1] = S.ltr_prefix[po] + (rrows + rs - 1) / rs;
            S.utr_prefix[po + 1] = S.utr_prefix[po] + (t.sn_ncolu[k] + rs - 1) / rs;
            S.inv_prefix[po + 1] = S.inv_prefix[po] + 2 * ((nsupc + 31) / 32);
            {
                const int64_t pay = (int64_t) t.sn_nsupr[k] * nsupc + (int64_t) 2 * ((nsupc + 31) / 32) * 32 * 32;
                S.pk_prefix[po + 1] = S.pk_prefix[po] + (int) ((pay + PKC - 1) / PKC);
                S.pk_off[po + 1] = S.pk_off[po] + pay;
            }
            S.zltr_prefix[po + 1] = S.zltr_prefix[po] + (rrows + 63) / 64;
            S.fwd_prefix[po + 1] = S.fwd_prefix[po] + (rrows + 255) / 256;
            S.bwd_prefix[po + 1] = S.bwd_prefix[po] + (t.sn_ncolu[k] + 63) / 64;
        }
    }
    build_urgent_lists(t, ns, lvl, S);
    // K-fused chain groups of up to four supernodes (a, a+1, a+2, a+3) in consecutive levels: every member but the last
    // runs only its urgent tiles; every member's executed tiles accumulate all earlier members' deferred updates
    S.lvl_defer.assign(S.nlevels, 0);
    if (H.h_fuse_prev.empty()) { H.h_fuse_prev.assign(3 * (size_t) ns, -1); H.h_defer.assign(ns, 0); H.h_pair_roff.assign(3 * (size_t) ns, -1); H.h_pair_coff.assign(3 * (size_t) ns, -1); }
    if (!getenv("SLUAMD_NO_FUSE") && !H.opt.deterministic && !H.z) {
        static const int maxprev = getenv("SLUAMD_FUSE_MAX_PREV") ? std::max(1, std::min(3, atoi(getenv("SLUAMD_FUSE_MAX_PREV")))) : 1;   // measured: pairs beat groups of 3-4 end to end (longer urgent tiles sit on the panel chain)
        std::vector<int> rowmap[3], colinfo[3];
        for (int l = 0; l + 1 < S.nlevels; ++l)
            for (int i = S.lvl_off[l + 1]; i < S.lvl_off[l + 2]; ++i) {
                const int b = S.nodes[i], a = b - 1;
                if (a < 0 || lvl[a] != l || !t.sn_big[a] || !t.sn_big[b]) continue;
                int srcs[3] = {a, -1, -1}, nsrc = 1;
                for (int j = 0; j < 3 && H.h_fuse_prev[3 * (size_t) a + j] >= 0; ++j) {
                    if (nsrc == maxprev) { nsrc = -1; break; }       // a already closes a full group: b starts a new one later
                    srcs[nsrc++] = H.h_fuse_prev[3 * (size_t) a + j];
                }
                if (nsrc < 0) continue;
                bool ok = true;
                for (int j = 0; j < nsrc && ok; ++j) ok = build_pair_maps(hs, t, srcs[j], b, rowmap[j], colinfo[j]);
                if (!ok) {   // the far members do not fit b: fall back to the plain pair when a is not fused itself
                    if (nsrc > 1 || !build_pair_maps(hs, t, a, b, rowmap[0], colinfo[0])) continue;
                }
                for (int j = 0; j < nsrc; ++j) {
                    const size_t pj = 3 * (size_t) b + j;
                    H.h_fuse_prev[pj] = srcs[j];
                    H.h_pair_roff[pj] = (int) H.h_pair_rowmap.size(); H.h_pair_coff[pj] = (int) (H.h_pair_colinfo.size() / 2);
                    H.h_pair_rowmap.insert(H.h_pair_rowmap.end(), rowmap[j].begin(), rowmap[j].end());
                    H.h_pair_colinfo.insert(H.h_pair_colinfo.end(), colinfo[j].begin(), colinfo[j].end());
                }
                H.h_defer[a] = 1;
                S.lvl_defer[l] = 1;
                H.fused_pairs += 1;
            }
    }
}
static void build_urgent_lists(const HostTables &t, int nsupers, const std::vector<int> &lvl, LevelSched &S)
{
    S.sn_level = lvl;
    S.u_off.assign(2 * S.nlevels + 1, 0);
    std::vector<uint8_t> rflag, cflag;
    for (int l = 0; l < S.nlevels; ++l) {
        const int nbig = S.n_big[l];
        for (int g = 0; g < 2; ++g) {
            const int b = S.lvl_off[l] + (g == 0 ? 0 : nbig), e = (g == 0) ? S.lvl_off[l] + nbig : S.lvl_off[l + 1];
            for (int i = b; i < e; ++i) {
                const int k = S.nodes[i];
                const int nrt = t.sn_nrt[k], nct = t.sn_nct[k];
                if (!nrt || !nct) continue;
                rflag.assign(nrt, 0); cflag.assign(nct, 0);
                bool any = false;
                for (int r = 0; r < nrt; ++r) { const int ib = t.lb_gid[t.sn_lb_off[k] + t.rtile[t.sn_rt_off[k] + r].x]; rflag[r] = (lvl[ib] == l + 1); any |= rflag[r]; }
                for (int c = 0; c < nct; ++c) { const int jb = t.ub_gid[t.sn_ub_off[k] + t.ctile[t.sn_ct_off[k] + c].x]; cflag[c] = (lvl[jb] == l + 1); any |= cflag[c]; }
                if (!any) continue;
                for (int r = 0; r < nrt; ++r)
                    for (int c = 0; c < nct; ++c)
                        if (rflag[r] || cflag[c]) S.ulist.push_back(make_int4(k, r, c, 0));
            }
            S.u_off[2 * l + g + 1] = (int) S.ulist.size();
        }
    }
    (void) nsupers;
}

static int upload_schedule(Handle &H, LevelSched &S)
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
    if (upload(H.d_misc, S.pk_prefix, &S.d_pk_prefix)) return SLUAMD_EHIP;
    if (upload(H.d_misc, S.pk_off, &S.d_pk_off)) return SLUAMD_EHIP;
    return 0;
}

static int finish_create(Handle *H, const sluamd_forest_view_t *forests, const std::vector<std::vector<int>> *given_lists = nullptr)
{
    HostTables t;
    int rc = build_tables(*H, t);
    if (rc) return rc;
    // ---- device uploads ----
    HIPCHK(hipStreamCreate(&H->stream));
    {
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&H->pstream, hipStreamNonBlocking, hi));
    }
    HIPCHK(hipEventCreate(&H->ev0)); HIPCHK(hipEventCreate(&H->ev1));
    const HostStruct &hs = H->hs;
    auto &K = H->d_misc;
    DevTables &T = H->T;
    T.val = H->d_val;
    if (upload(K, hs.lidx, &H->d_lidx) || upload(K, hs.uidx, &H->d_uidx) || upload(K, t.ucolptr, &H->d_ucolptr) ||
        upload(K, t.unzcol, &H->d_unzcol) || upload(K, hs.xsup, &H->d_xsup)) return SLUAMD_EHIP;
    T.lidx = H->d_lidx; T.uidx = H->d_uidx; T.ucolptr = H->d_ucolptr; T.unzcol = H->d_unzcol; T.xsup = H->d_xsup;
#define UP(field, vec, type) { type *p_; if (upload(K, vec, &p_)) return SLUAMD_EHIP; T.field = p_; }
    UP(sn_lval, t.sn_lval, int64_t) UP(sn_uval, t.sn_uval, int64_t) UP(sn_lidx, t.sn_lidx, int64_t) UP(sn_uidx, t.sn_uidx, int64_t)
    UP(sn_dinv, t.sn_dinv, int64_t)
    H->h_sn_dinv = t.sn_dinv;
    {
        double *dv;
        if (hipMalloc((void **) &dv, sizeof(double) * std::max<int64_t>(t.dinv_total, 1)) != hipSuccess) { set_error("hipMalloc(dinv) failed"); return SLUAMD_ENOMEM; }
        K.push_back(dv); T.dinv = dv;
    }
    UP(sn_nsupr, t.sn_nsupr, int) UP(sn_ldu, t.sn_ldu, int) UP(sn_ncolu, t.sn_ncolu, int)
    UP(sn_lb_off, t.sn_lb_off, int) UP(sn_nlb, t.sn_nlb, int) UP(sn_ub_off, t.sn_ub_off, int) UP(sn_nub, t.sn_nub, int)
    UP(sn_rt_off, t.sn_rt_off, int) UP(sn_nrt, t.sn_nrt, int) UP(sn_ct_off, t.sn_ct_off, int) UP(sn_nct, t.sn_nct, int)
    UP(lb_gid, t.lb_gid, int) UP(lb_nbrow, t.lb_nbrow, int) UP(lb_rowoff, t.lb_rowoff, int) UP(lb_lptr, t.lb_lptr, int)
    UP(lbs_gid, t.lbs_gid, int) UP(lbs_idx, t.lbs_idx, int)
    UP(ub_gid, t.ub_gid, int) UP(ub_ncols, t.ub_ncols, int) UP(ub_iukp, t.ub_iukp, int) UP(ub_stcol, t.ub_stcol, int)
    UP(rtile, t.rtile, int4) UP(ctile, t.ctile, int4)
#undef UP
    // ---- schedules ----
    std::vector<std::vector<int>> lists;
    if (given_lists) {
        lists = *given_lists;
    } else if (forests && forests->maxLvl > 0 && forests->nodeList) {
        for (int l = 0; l < forests->maxLvl; ++l) {
            std::vector<int> v;
            if (!forests->myZeroTrIdxs[l]) {
                const int f = forests->myTreeIdxs[l];
                if (f >= 0 && f < forests->numForests && forests->nNodes[f] > 0)
                    v.assign(forests->nodeList[f], forests->nodeList[f] + forests->nNodes[f]);
            }
            lists.push_back(std::move(v));
        }
    } else {
        std::vector<int> v;
        for (int k = 0; k < hs.nsupers; ++k) if (hs.present[k]) v.push_back(k);
        lists.push_back(std::move(v));
    }
    for (auto &l : lists)
        for (int k : l)
            if (k < 0 || k >= hs.nsupers || !hs.present[k]) { set_error("forest node list names a supernode that is not stored on this rank"); return SLUAMD_ESTRUCT; }
    H->sched.resize(lists.size());
    int nlev = 0;
    for (size_t i = 0; i < lists.size(); ++i) {
        build_schedule(*H, t, lists[i], H->sched[i]);
        if (upload_schedule(*H, H->sched[i])) return SLUAMD_EHIP;
        nlev += H->sched[i].nlevels;
    }
    H->st.num_levels = nlev;
    if (H->fused_pairs) {
        int *p0, *p1, *p2, *p3, *p4, *p5;
        if (upload(H->d_misc, H->h_fuse_prev, &p0) || upload(H->d_misc, H->h_defer, &p1) || upload(H->d_misc, H->h_pair_roff, &p2) ||
            upload(H->d_misc, H->h_pair_coff, &p3) || upload(H->d_misc, H->h_pair_rowmap, &p4) || upload(H->d_misc, H->h_pair_colinfo, &p5)) return SLUAMD_EHIP;
        T.fuse_prev = p0; T.defer = p1; T.pair_roff = p2; T.pair_coff = p3; T.pair_rowmap = p4; T.pair_colinfo = p5;
    }
    H->st.reserved_i = H->fused_pairs;   // K-fused supernode pairs (diagnostic)
    HIPCHK(hipMalloc((void **) &H->d_info, 4 * sizeof(int)));
    // kernels that keep a whole panel strip / diagonal block in LDS need more than the default 64 KiB
    HIPCHK(hipFuncSetAttribute((const void *) k_panel_trsm<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_panel_trsm<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) kz_diag_lu, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    H->st.nnz_L = hs.nnzL; H->st.nnz_U = hs.nnzU;
    size_t idxb = (hs.lidx.size() + 3 * hs.uidx.size()) * sizeof(int);
    H->st.bytes_device = (int64_t) ((hs.nnzL + hs.nnzU) * sizeof(double) + idxb);
    return 0;
}

static int check_device(int dev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible: this library has no CPU fallback"); return SLUAMD_ENODEVICE; }
    if (dev >= 0) { HIPCHK(hipSetDevice(dev)); }
    return 0;
}

// ================================================================================================
//                                   HOST: factorisation driver
// ================================================================================================
static void ev_begin(Handle *H, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t &used)
{
    if (!H->profile) return;
    if (used == v.size()) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); v.emplace_back(a, b); }
    hipEventRecord(v[used].first, H->stream);
}
static void ev_end(Handle *H, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t &used)
{
    if (!H->profile) return;
    hipEventRecord(v[used].second, H->stream);
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

// One elimination forest, level by level.  Serial mode (profiling / deterministic): everything on one stream.
// Look-ahead mode (default): the Schur update of level l is split into the tiles that feed level l+1's panels
// ("urgent", explicit list) and the rest; the panel kernels of level l+1 run on a high-priority stream as soon as
// the urgent tiles are done and overlap with the rest -- the GPU analogue of the reference's look-ahead pipeline
// (dsparseTreeFactor_ASYNC, dtreeFactorization.c:381-706, num_lookaheads).
static int run_factor_sched(Handle *H, LevelSched &S, double thresh)
{
    const DevTables &T = H->T;
    const bool lookahead = !H->profile && !H->opt.deterministic && getenv("SLUAMD_NO_LOOKAHEAD") == nullptr;
    hipStream_t s = H->stream, ps = lookahead ? H->pstream : H->stream;
    auto schur = [&](hipStream_t st, bool big, int ntile, const int *nodes, const int *prefix, int nn, int id_base,
                     const int4 *ulist, int skip_level) {
        ev_begin(H, H->ev_schur, H->ev_schur_used);
        const int grid = ((ntile + 7) / 8) * 8;
        static const bool w8 = getenv("SLUAMD_SCHUR_4WAVES") == nullptr;
        if (big && w8) hipLaunchKernelGGL((k_schur<128, 128, 8>), dim3(grid), dim3(512), 0, st, T, nodes, prefix, nn, id_base, ntile, H->d_info, ulist, S.d_sn_level, skip_level);
        else if (big) hipLaunchKernelGGL((k_schur<128, 128, 4>), dim3(grid), dim3(256), 0, st, T, nodes, prefix, nn, id_base, ntile, H->d_info, ulist, S.d_sn_level, skip_level);
        else hipLaunchKernelGGL((k_schur<64, 64, 4>), dim3(grid), dim3(256), 0, st, T, nodes, prefix, nn, id_base, ntile, H->d_info, ulist, S.d_sn_level, skip_level);
        ev_end(H, H->ev_schur, H->ev_schur_used);
        H->st.num_launches++; H->st.schur_launches++; H->st.schur_tiles += ntile;
    };
    auto panel = [&](int l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        const int *nodes = S.d_nodes + n0;
        const int mx = S.max_nsupc[l];
        const size_t lds_tr = trsm_lds_bytes((mx + 31) & ~31);
        ev_begin(H, H->ev_panel, H->ev_panel_used);
        if (mx <= 64) hipLaunchKernelGGL(k_diag_lu<64>, dim3(nn), dim3(256), 0, ps, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info);
        else if (mx <= 128) hipLaunchKernelGGL(k_diag_lu<128>, dim3(nn), dim3(256), 0, ps, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info);
        else hipLaunchKernelGGL(k_diag_lu<256>, dim3(nn), dim3(256), 0, ps, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info);
        hipLaunchKernelGGL(k_diag_inv, dim3((S.inv_prefix[po + nn] + 3) / 4), dim3(128), 0, ps, T, nodes, S.d_inv_prefix + po, nn);
        const int nl = S.ltr_prefix[po + nn], nu = S.utr_prefix[po + nn];
        if (nl + nu) {
            if (trsm_rs((mx + 31) & ~31) == 32) hipLaunchKernelGGL(k_panel_trsm<32>, dim3(nl + nu), dim3(128), lds_tr, ps, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, nl);
            else hipLaunchKernelGGL(k_panel_trsm<64>, dim3(nl + nu), dim3(256), lds_tr, ps, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, nl);
        }
        ev_end(H, H->ev_panel, H->ev_panel_used);
        H->st.num_launches += 2 + (nl + nu > 0);
    };
    if (lookahead && S.nlevels) {
        hipEvent_t e = next_event(H);    // the side streams must see everything queued so far on the main stream
        hipEventRecord(e, s); hipStreamWaitEvent(ps, e, 0);
    }
    bool panel_queued = false;           // panel(l) already queued on ps by the previous level's look-ahead
    for (int l = 0; l < S.nlevels; ++l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0;
        const int *nodes = S.d_nodes + n0;
        if (!panel_queued) {
            if (lookahead && l > 0) {   // no look-ahead was done for this level: its panels need ALL of Schur(l-1)
                hipEvent_t e = next_event(H);
                hipEventRecord(e, s); hipStreamWaitEvent(ps, e, 0);
            }
            panel(l);
        }
        panel_queued = false;
        if (lookahead) {
            hipEvent_t e = next_event(H);
            hipEventRecord(e, ps); hipStreamWaitEvent(s, e, 0);     // Schur(l) needs panel(l)
        }
        const int nbig = S.n_big[l];
        // look-ahead split of this level's Schur update (SLUAMD_LOOKAHEAD_MAX_STRIPS bounds the panel work overlapped)
        bool split = false;
        if (lookahead && l + 1 < S.nlevels) {
            const int po1 = S.lvl_poff[l + 1], nn1 = S.lvl_off[l + 2] - S.lvl_off[l + 1];
            static const int max_strips = getenv("SLUAMD_LOOKAHEAD_MAX_STRIPS") ? atoi(getenv("SLUAMD_LOOKAHEAD_MAX_STRIPS")) : (1 << 30);
            split = (S.ltr_prefix[po1 + nn1] + S.utr_prefix[po1 + nn1]) <= max_strips;
            (void) nn1;
        }
        // K-fused pairs: a deferred supernode runs only its urgent tiles (everything the next level's panels need), so the
        // urgent pass is needed even without look-ahead; the rest of its update is accumulated by its partner's tiles
        const bool urgent_pass = split || (T.defer && S.lvl_defer[l]);
        // pass 0: urgent tiles (explicit lists); pass 1: the rest (full grids, urgent tiles skipped)
        for (int pass = urgent_pass ? 0 : 1; pass < 2; ++pass) {
            hipStream_t st = s;   // every Schur launch stays on the main stream (side streams for the bulk tiles measured slower)
            for (int g = 0; g < 2; ++g) {
                const int cnt = g == 0 ? nbig : nn - nbig;
                if (!cnt) continue;
                const int so = S.lvl_soff[l] + (g == 0 ? 0 : nbig + 1);
                const int *gn = nodes + (g == 0 ? 0 : nbig);
                if (pass == 0) {
                    const int u0 = S.u_off[2 * l + g], nu = S.u_off[2 * l + g + 1] - u0;
                    if (nu) schur(st, g == 0, nu, gn, S.d_tile_prefix + so, cnt, 0, S.d_ulist + u0, -1);
                    continue;
                }
                const int nt = S.tile_prefix[so + cnt];
                if (!nt) continue;
                if (T.defer && S.lvl_defer[l]) {   // nothing to launch when every supernode of the group is deferred
                    bool all = true;
                    const int i0 = n0 + (g == 0 ? 0 : nbig);
                    for (int i = 0; i < cnt && all; ++i) all = H->h_defer[S.nodes[i0 + i]] != 0;
                    if (all) continue;
                }
                if (!H->opt.deterministic) {
                    schur(st, g == 0, nt, gn, S.d_tile_prefix + so, cnt, 0, nullptr, urgent_pass ? l + 1 : -1);
                } else {  // one supernode per launch: tiles of one k hit distinct destinations -> fixed summation order
                    for (int i = 0; i < cnt; ++i) {
                        const int c = S.tile_prefix[so + i + 1] - S.tile_prefix[so + i];
                        if (c) schur(st, g == 0, c, gn, S.d_tile_prefix + so, cnt, S.tile_prefix[so + i], nullptr, -1);
                    }
                }
            }
            if (pass == 0 && split) {
                // panel(l+1) may start once the urgent tiles of level l (and, by stream order, the rest of level
                // l-1) are complete; it then overlaps with the rest of level l
                hipEvent_t eu = next_event(H);
                hipEventRecord(eu, s); hipStreamWaitEvent(ps, eu, 0);
                panel(l + 1);
                panel_queued = true;
            }
        }
    }
    if (lookahead && S.nlevels) {
        hipEvent_t e = next_event(H);
        hipEventRecord(e, ps); hipStreamWaitEvent(s, e, 0);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// forward / backward block solves of one Z level (one elimination forest): DAG levels ascending / descending
static void solve_fwd_z(Handle *H, int z, double *d_x, int64_t ldx, int nrhs, size_t lds)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    LevelSched &S = H->sched[z];
    for (int l = 0; l < S.nlevels; ++l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        hipLaunchKernelGGL(k_solve_diag<true>, dim3(nn), dim3(256), lds, s, T, S.d_nodes + n0, d_x, ldx, nrhs);
        const int nf = S.fwd_prefix[po + nn];
        if (nf) hipLaunchKernelGGL(k_fwd_update, dim3(nf), dim3(256), lds, s, T, S.d_nodes + n0, S.d_fwd_prefix + po, nn, d_x, ldx, nrhs);
    }
}
static void solve_bwd_z(Handle *H, int z, double *d_x, int64_t ldx, int nrhs, size_t lds)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    LevelSched &S = H->sched[z];
    for (int l = S.nlevels - 1; l >= 0; --l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
        const int nb = S.bwd_prefix[po + nn];
        if (nb) hipLaunchKernelGGL(k_bwd_update, dim3(nb), dim3(256), 0, s, T, S.d_nodes + n0, S.d_bwd_prefix + po, nn, d_x, ldx, nrhs);
        hipLaunchKernelGGL(k_solve_diag<false>, dim3(nn), dim3(256), lds, s, T, S.d_nodes + n0, d_x, ldx, nrhs);
    }
}

// dir: +1 forward only, -1 backward only, 0 both; zsel: one Z level or -1 = all
static int run_solve(Handle *H, double *d_x, int64_t ldx, int nrhs, int dir = 0, int zsel = -1)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    const size_t lds = (size_t) (H->max_nsupc + 32) * nrhs * sizeof(double);
    if (lds > 64 * 1024) { set_error("nrhs too large for the LDS-staged solve ((max_nsupc+32)*nrhs*8 must be <= 64 KiB)"); return SLUAMD_EINVAL; }
    if (!H->dinv_ready) {   // factors were uploaded already factored: build the diagonal sub-block inverses once
        for (auto &S : H->sched)
            for (int l = 0; l < S.nlevels; ++l) {
                const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
                hipLaunchKernelGGL(k_diag_inv, dim3((S.inv_prefix[po + nn] + 3) / 4), dim3(128), 0, s, T, S.d_nodes + n0, S.d_inv_prefix + po, nn);
            }
        H->dinv_ready = true;
    }
    for (int z = 0; z < (int) H->sched.size(); ++z) if (zsel < 0 || zsel == z) if (dir >= 0) solve_fwd_z(H, z, d_x, ldx, nrhs, lds);
    for (int z = (int) H->sched.size() - 1; z >= 0; --z) if (zsel < 0 || zsel == z) if (dir <= 0) solve_bwd_z(H, z, d_x, ldx, nrhs, lds);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace sluamd

// ================================================================================================
//                                          C ABI
// ================================================================================================
using namespace sluamd;

struct sluamd_lu_handle_s { Handle H; };

extern "C" {

const char *sluamd_last_error(void) { return g_err.c_str(); }

int sluamd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void sluamd_default_options(sluamd_options_t *opt)
{
    std::memset(opt, 0, sizeof(*opt));
    opt->device = -1;
}

static int upload_values(Handle *H, const sluamd_dLUview_t *lu)
{
    // value pointers are read as raw bytes: esz = 8 (double) or 16 (doublecomplex, via the layout-identical zLUview)
    const HostStruct &hs = H->hs;
    const size_t esz = H->z ? 16 : 8;
    const int ns = hs.nsupers;
    std::vector<char> stage((size_t) std::max(hs.nnzL, hs.nnzU) * esz);
    char *dv = reinterpret_cast<char *>(H->d_val);
    for (int k = 0; k < ns; ++k) {
        const int64_t len = hs.lval_off[k + 1] - hs.lval_off[k];
        if (len) std::memcpy(stage.data() + hs.lval_off[k] * esz, lu->Lnzval_bc_ptr[k], esz * len);
    }
    if (hs.nnzL) HIPCHK(hipMemcpy(dv, stage.data(), esz * hs.nnzL, hipMemcpyHostToDevice));
    for (int k = 0; k < ns; ++k) {
        const int64_t len = hs.uval_off[k + 1] - hs.uval_off[k];
        if (len) std::memcpy(stage.data() + hs.uval_off[k] * esz, lu->Unzval_br_ptr[k], esz * len);
    }
    if (hs.nnzU) HIPCHK(hipMemcpy(dv + esz * hs.nnzL, stage.data(), esz * hs.nnzU, hipMemcpyHostToDevice));
    return 0;
}

static int create_from_view(sluamd_handle_t *out, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests,
                            const sluamd_options_t *opt, bool z)
{
    if (!out) { set_error("null handle pointer"); return SLUAMD_EINVAL; }
    *out = nullptr;
    sluamd_options_t o;
    if (opt) o = *opt; else sluamd_default_options(&o);
    int rc = check_device(o.device);
    if (rc) return rc;
    auto *hh = new sluamd_lu_handle_s();
    Handle *H = &hh->H;
    H->opt = o;
    H->z = z;
    HIPCHK(hipGetDevice(&H->device));
    rc = flatten_view(lu, H->hs, false);
    if (rc) { delete hh; return rc; }
    H->Pz = lu->npdep; H->myz = lu->myzlayer;
    const int64_t tot = H->hs.nnzL + H->hs.nnzU;
    if (hipMalloc((void **) &H->d_val, (z ? 16 : 8) * std::max<int64_t>(tot, 1)) != hipSuccess) {
        set_error("hipMalloc of the value arena failed"); delete hh; return SLUAMD_ENOMEM;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    rc = upload_values(H, lu);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); H->st.t_h2d_ms = ms;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (!rc) rc = finish_create(H, forests);
    if (rc) { sluamd_dDestroyLUHandle(hh); return rc; }
    *out = hh;
    return 0;
}

int sluamd_dCreateLUHandle(sluamd_handle_t *out, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests,
                           const sluamd_options_t *opt)
{
    return create_from_view(out, lu, forests, opt, false);
}

// complex16 twin: zCreateLUgpuHandle (SRC/include/superlu_upacked.h, z section)
int sluamd_zCreateLUHandle(sluamd_handle_t *out, const sluamd_zLUview_t *lu, const sluamd_forest_view_t *forests,
                           const sluamd_options_t *opt)
{
    return create_from_view(out, reinterpret_cast<const sluamd_dLUview_t *>(lu), forests, opt, true);
}

int sluamd_dSetValues(sluamd_handle_t h, const sluamd_dLUview_t *lu)
{
    if (!h || !lu) { set_error("null argument"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(h->H.device));
    h->H.dinv_ready = false;
    return upload_values(&h->H, lu);
}

int sluamd_pdgstrf3d(sluamd_handle_t h, double thresh, int *info)
{
    if (!h) { set_error("null handle"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("complex16 handle: call sluamd_pzgstrf3d"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    int init[4] = {0x7fffffff, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(H->d_info, init, sizeof(init), hipMemcpyHostToDevice, H->stream));
    H->st.num_launches = 0; H->st.schur_launches = 0; H->st.schur_tiles = 0;
    H->profile = H->opt.verbose >= 2 || getenv("SLUAMD_PROFILE") != nullptr;
    H->ev_schur_used = H->ev_panel_used = 0;
    H->ev_pool_used = 0;
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    // Z levels in order (pdgstrf3d.c:333-385); the ancestor reduction between levels is the caller's
    // collective (RCCL reduce on the arena slice) in a multi-rank run.
    for (size_t z = 0; z < H->sched.size(); ++z) {
        int rc = run_factor_sched(H, H->sched[z], thresh);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    int res[4];
    HIPCHK(hipMemcpyAsync(res, H->d_info, sizeof(res), hipMemcpyDeviceToHost, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_factor_ms = ms;
    H->dinv_ready = true;
    H->st.t_schur_ms = H->profile ? ev_sum(H->ev_schur, H->ev_schur_used) : 0.0;
    H->st.t_panel_ms = H->profile ? ev_sum(H->ev_panel, H->ev_panel_used) : 0.0;
    H->st.tiny_pivots = res[1];
    if (info) *info = (res[0] == 0x7fffffff) ? 0 : res[0];
    if (res[2]) { set_error("Schur update found no destination block for " + std::to_string(res[2]) + " tiles (structure not closed)"); return SLUAMD_ESTRUCT; }
    return 0;
}

// factor one Z level only (multi-rank orchestration: level, reduce, level, ...)
int sluamd_pdgstrf3d_level(sluamd_handle_t h, int zlevel, double thresh)
{
    if (!h || zlevel < 0 || zlevel >= (int) h->H.sched.size()) { set_error("bad level"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    if (zlevel == 0) {
        H->ev_pool_used = 0;
        int init[4] = {0x7fffffff, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(H->d_info, init, sizeof(init), hipMemcpyHostToDevice, H->stream));
    }
    int rc = run_factor_sched(H, H->sched[zlevel], thresh);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(H->stream));
    H->dinv_ready = true;   // inverses of every level this rank factors are written by that level's kernels
    return 0;
}

// ---- cooperative (owner-computes) factorisation of one shared ancestor forest ----------------------------------
// The G = 2^zlevel ranks that share the forest of Z level `zlevel` all hold its panels (replicated storage).  Block
// column jb (L panel jb and the U blocks (*, jb)) is kept current by rank jb % G only:
//   sluamd_coop_panel  : owners factor the diagonal blocks + L panels of DAG level l, then everybody packs the
//                        level's staging buffer (own panels | zeros) -> caller sum-all-reduces it over the group;
//   sluamd_coop_update : unpack the other owners' panels, U-panel TRSM, Schur update of the destinations I own.
// Everything is queued on the caller's stream (sluamd_set_stream) without host synchronisation, so the RCCL
// all-reduce of the staging buffer orders naturally between the two calls.
int sluamd_set_stream(sluamd_handle_t h, void *stream)
{
    if (!h) { set_error("null handle"); return SLUAMD_EINVAL; }
    h->H.user_stream = reinterpret_cast<hipStream_t>(stream);
    h->H.has_user_stream = true;
    return 0;
}

int sluamd_coop_info(sluamd_handle_t h, int zlevel, int *nlevels, int64_t *max_stage)
{
    if (!h || zlevel < 0 || zlevel >= (int) h->H.sched.size()) { set_error("bad level"); return SLUAMD_EINVAL; }
    const LevelSched &S = h->H.sched[zlevel];
    int64_t mx = 0;
    for (int l = 0; l < S.nlevels; ++l) {
        const int nn = S.lvl_off[l + 1] - S.lvl_off[l];
        mx = std::max(mx, S.pk_off[S.lvl_poff[l] + nn]);
    }
    if (nlevels) *nlevels = S.nlevels;
    if (max_stage) *max_stage = mx;
    return 0;
}

int sluamd_coop_level_size(sluamd_handle_t h, int zlevel, int l, int *nnodes, int64_t *stage_doubles)
{
    if (!h || zlevel < 0 || zlevel >= (int) h->H.sched.size() || l < 0 || l >= h->H.sched[zlevel].nlevels) { set_error("bad level"); return SLUAMD_EINVAL; }
    const LevelSched &S = h->H.sched[zlevel];
    const int nn = S.lvl_off[l + 1] - S.lvl_off[l];
    if (nnodes) *nnodes = nn;
    if (stage_doubles) *stage_doubles = S.pk_off[S.lvl_poff[l] + nn];
    return 0;
}

int sluamd_coop_level_nodes(sluamd_handle_t h, int zlevel, int l, int *nodes_out)
{
    if (!h || !nodes_out || zlevel < 0 || zlevel >= (int) h->H.sched.size() || l < 0 || l >= h->H.sched[zlevel].nlevels) { set_error("bad level"); return SLUAMD_EINVAL; }
    const LevelSched &S = h->H.sched[zlevel];
    std::copy(S.nodes.begin() + S.lvl_off[l], S.nodes.begin() + S.lvl_off[l + 1], nodes_out);
    return 0;
}

// device ranges holding supernode k's factored L panel (diagonal block first) and its inverted diagonal sub-blocks
int sluamd_coop_panel_ptrs(sluamd_handle_t h, int k, double **d_lpanel, int64_t *lpanel_doubles, double **d_dinv, int64_t *dinv_doubles)
{
    if (!h || k < 0 || k >= h->H.hs.nsupers || !h->H.hs.present[k]) { set_error("bad supernode"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    const int ns = H->hs.xsup[k + 1] - H->hs.xsup[k];
    if (d_lpanel) *d_lpanel = H->d_val + H->hs.lval_off[k];
    if (lpanel_doubles) *lpanel_doubles = H->hs.lval_off[k + 1] - H->hs.lval_off[k];
    if (d_dinv) *d_dinv = H->T.dinv + H->h_sn_dinv[k];
    if (dinv_doubles) *dinv_doubles = (int64_t) 2 * ((ns + 31) / 32) * 32 * 32;
    return 0;
}

static int coop_check(sluamd_handle_t h, int zlevel, int l, int G, int g)
{
    if (!h || zlevel < 0 || zlevel >= (int) h->H.sched.size() || G < 1 || g < 0 || g >= G) { set_error("bad cooperative-level arguments"); return SLUAMD_EINVAL; }
    if (l >= h->H.sched[zlevel].nlevels || l < -1) { set_error("bad DAG level"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("cooperative mode is double precision only"); return SLUAMD_EINVAL; }
    return 0;
}

int sluamd_coop_panel(sluamd_handle_t h, int zlevel, int l, int G, int g, double thresh, double *d_stage)
{
    int rc = coop_check(h, zlevel, l, G, g);
    if (rc) return rc;
    if (l < 0) { set_error("bad cooperative-level arguments"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    LevelSched &S = H->sched[zlevel];
    DevTables T = H->T; T.own_G = G; T.own_g = g; T.fuse_prev = nullptr; T.defer = nullptr;
    hipStream_t cs = H->has_user_stream ? H->user_stream : H->stream;
    const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
    const int *nodes = S.d_nodes + n0;
    bool any = false;
    for (int i = 0; i < nn; ++i) any |= (S.nodes[n0 + i] % G) == g;
    if (any) {
        const int mx = S.max_nsupc[l];
        if (mx <= 64) hipLaunchKernelGGL(k_diag_lu<64>, dim3(nn), dim3(256), 0, cs, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info);
        else if (mx <= 128) hipLaunchKernelGGL(k_diag_lu<128>, dim3(nn), dim3(256), 0, cs, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info);
        else hipLaunchKernelGGL(k_diag_lu<256>, dim3(nn), dim3(256), 0, cs, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info);
        hipLaunchKernelGGL(k_diag_inv, dim3((S.inv_prefix[po + nn] + 3) / 4), dim3(128), 0, cs, T, nodes, S.d_inv_prefix + po, nn);
        const int nl = S.ltr_prefix[po + nn];
        if (nl) {
            const size_t lds_tr = trsm_lds_bytes((mx + 31) & ~31);
            if (trsm_rs((mx + 31) & ~31) == 32) hipLaunchKernelGGL(k_panel_trsm<32>, dim3(nl), dim3(128), lds_tr, cs, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, nl);
            else hipLaunchKernelGGL(k_panel_trsm<64>, dim3(nl), dim3(256), lds_tr, cs, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, nl);
        }
    }
    // d_stage == NULL: the caller broadcasts the owners' panels straight out of the arena (sluamd_coop_panel_ptrs)
    if (G > 1 && d_stage) hipLaunchKernelGGL(k_coop_pack, dim3(S.pk_prefix[po + nn]), dim3(256), 0, cs, T, nodes, S.d_pk_prefix + po, S.d_pk_off + po, nn, d_stage, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

int sluamd_coop_update(sluamd_handle_t h, int zlevel, int l, int G, int g, const double *d_stage)
{
    int rc = coop_check(h, zlevel, l, G, g);
    if (rc) return rc;
    if (l < 0) { set_error("bad cooperative-level arguments"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    LevelSched &S = H->sched[zlevel];
    DevTables T = H->T; T.own_G = G; T.own_g = g; T.fuse_prev = nullptr; T.defer = nullptr;
    hipStream_t cs = H->has_user_stream ? H->user_stream : H->stream;
    const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
    const int *nodes = S.d_nodes + n0;
    if (G > 1 && d_stage) {
        hipLaunchKernelGGL(k_coop_pack, dim3(S.pk_prefix[po + nn]), dim3(256), 0, cs, T, nodes, S.d_pk_prefix + po, S.d_pk_off + po, nn, const_cast<double *>(d_stage), 1);
    }
    const int mx = S.max_nsupc[l];
    const int nu = S.utr_prefix[po + nn];
    if (nu) {   // U strips only: nl = 0 sends every workgroup down the U branch
        const size_t lds_tr = trsm_lds_bytes((mx + 31) & ~31);
        if (trsm_rs((mx + 31) & ~31) == 32) hipLaunchKernelGGL(k_panel_trsm<32>, dim3(nu), dim3(128), lds_tr, cs, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, 0);
        else hipLaunchKernelGGL(k_panel_trsm<64>, dim3(nu), dim3(256), lds_tr, cs, T, nodes, S.d_ltr_prefix + po, S.d_utr_prefix + po, nn, 0);
    }
    const int nbig = S.n_big[l];
    for (int grp = 0; grp < 2; ++grp) {
        const int cnt = grp == 0 ? nbig : nn - nbig;
        if (!cnt) continue;
        const int so = S.lvl_soff[l] + (grp == 0 ? 0 : nbig + 1);
        const int *gn = nodes + (grp == 0 ? 0 : nbig);
        const int nt = S.tile_prefix[so + cnt];
        if (!nt) continue;
        const int grid = ((nt + 7) / 8) * 8;
        if (grp == 0) hipLaunchKernelGGL((k_schur<128, 128, 8>), dim3(grid), dim3(512), 0, cs, T, gn, S.d_tile_prefix + so, cnt, 0, nt, H->d_info, (const int4 *) nullptr, S.d_sn_level, -1);
        else hipLaunchKernelGGL((k_schur<64, 64, 4>), dim3(grid), dim3(256), 0, cs, T, gn, S.d_tile_prefix + so, cnt, 0, nt, H->d_info, (const int4 *) nullptr, S.d_sn_level, -1);
    }
    H->dinv_ready = true;
    HIPCHK(hipGetLastError());
    return 0;
}

// zero the U blocks of the forest this rank does not own; the caller then sum-all-reduces the forest's U rows
int sluamd_coop_mask_u(sluamd_handle_t h, int zlevel, int G, int g)
{
    int rc = coop_check(h, zlevel, -1, G, g);
    if (rc) return rc;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    LevelSched &S = H->sched[zlevel];
    if (G == 1 || S.nodes.empty()) return 0;
    DevTables T = H->T; T.own_G = G; T.own_g = g; T.fuse_prev = nullptr; T.defer = nullptr;
    hipStream_t cs = H->has_user_stream ? H->user_stream : H->stream;
    hipLaunchKernelGGL(k_coop_mask_u, dim3((unsigned) S.nodes.size()), dim3(256), 0, cs, T, S.d_nodes);
    HIPCHK(hipGetLastError());
    return 0;
}

int sluamd_factor_info(sluamd_handle_t h, int *info, int *tiny)
{
    if (!h) return SLUAMD_EINVAL;
    int res[4];
    HIPCHK(hipMemcpy(res, h->H.d_info, sizeof(res), hipMemcpyDeviceToHost));
    if (info) *info = (res[0] == 0x7fffffff) ? 0 : res[0];
    if (tiny) *tiny = res[1];
    return 0;
}

int sluamd_dCopyLU2Host(sluamd_handle_t h, const sluamd_dLUview_t *lu)
{
    if (!h || !lu) { set_error("null argument"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    const HostStruct &hs = H->hs;
    const size_t esz = H->z ? 16 : 8;
    HIPCHK(hipSetDevice(H->device));
    std::vector<char> stage((size_t) std::max(hs.nnzL, hs.nnzU) * esz);
    const char *dv = reinterpret_cast<const char *>(H->d_val);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
    if (hs.nnzL) HIPCHK(hipMemcpy(stage.data(), dv, esz * hs.nnzL, hipMemcpyDeviceToHost));
    for (int k = 0; k < hs.nsupers; ++k) {
        const int64_t len = hs.lval_off[k + 1] - hs.lval_off[k];
        if (len) std::memcpy(lu->Lnzval_bc_ptr[k], stage.data() + hs.lval_off[k] * esz, esz * len);
    }
    if (hs.nnzU) HIPCHK(hipMemcpy(stage.data(), dv + esz * hs.nnzL, esz * hs.nnzU, hipMemcpyDeviceToHost));
    for (int k = 0; k < hs.nsupers; ++k) {
        const int64_t len = hs.uval_off[k + 1] - hs.uval_off[k];
        if (len) std::memcpy(lu->Unzval_br_ptr[k], stage.data() + hs.uval_off[k] * esz, esz * len);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); H->st.t_d2h_ms = ms;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

int sluamd_zCopyLU2Host(sluamd_handle_t h, const sluamd_zLUview_t *lu)
{
    if (!h || !h->H.z) { set_error("not a complex16 handle"); return SLUAMD_EINVAL; }
    return sluamd_dCopyLU2Host(h, reinterpret_cast<const sluamd_dLUview_t *>(lu));
}

int sluamd_zSetValues(sluamd_handle_t h, const sluamd_zLUview_t *lu)
{
    if (!h || !lu || !h->H.z) { set_error("not a complex16 handle"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(h->H.device));
    return upload_values(&h->H, reinterpret_cast<const sluamd_dLUview_t *>(lu));
}

// ---- complex16 numeric factorisation and solve (serial level loop; see sluamd_zkernels.inc) ----
static int run_factor_z(Handle *H, LevelSched &S, double thresh)
{
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    for (int l = 0; l < S.nlevels; ++l) {
        const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0;
        const int *nodes = S.d_nodes + n0;
        const int ldp = S.max_nsupc[l] | 1;
        hipLaunchKernelGGL(kz_diag_lu, dim3(nn), dim3(256), zdiag_lds_bytes(S.max_nsupc[l]), s, T, nodes, H->opt.replace_tiny_pivot, thresh, H->d_info, ldp);
        const int po = S.lvl_poff[l];
        const int nl = S.zltr_prefix[po + nn], nu = S.bwd_prefix[po + nn];   // 64-row strips / 64-column chunks
        if (nl + nu) hipLaunchKernelGGL(kz_panel_trsm, dim3(nl + nu), dim3(64), 0, s, T, nodes, S.d_zltr_prefix + po, S.d_bwd_prefix + po, nn, nl);
        H->st.num_launches += 1 + (nl + nu > 0);
        const int so = S.lvl_soff[l] + S.n_big[l] + 1;     // complex handles have no 128-tile group
        const int nt = S.tile_prefix[so + nn];
        if (nt) {
            hipLaunchKernelGGL(kz_schur, dim3(((nt + 7) / 8) * 8), dim3(256), 0, s, T, nodes, S.d_tile_prefix + so, nn, 0, nt, H->d_info);
            H->st.num_launches++; H->st.schur_launches++; H->st.schur_tiles += nt;
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int sluamd_pzgstrf3d(sluamd_handle_t h, double thresh, int *info)
{
    if (!h || !h->H.z) { set_error("not a complex16 handle"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    int init[4] = {0x7fffffff, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(H->d_info, init, sizeof(init), hipMemcpyHostToDevice, H->stream));
    H->st.num_launches = 0; H->st.schur_launches = 0; H->st.schur_tiles = 0;
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    for (size_t zl = 0; zl < H->sched.size(); ++zl) {
        int rc = run_factor_z(H, H->sched[zl], thresh);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    int res[4];
    HIPCHK(hipMemcpyAsync(res, H->d_info, sizeof(res), hipMemcpyDeviceToHost, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_factor_ms = ms;
    H->st.tiny_pivots = res[1];
    if (info) *info = (res[0] == 0x7fffffff) ? 0 : res[0];
    if (res[2]) { set_error("Schur update found no destination block for " + std::to_string(res[2]) + " tiles (structure not closed)"); return SLUAMD_ESTRUCT; }
    return 0;
}

// x: n x nrhs doublecomplex, column-major, host memory; overwritten by the solution of L U y = x
int sluamd_pzgstrs3d(sluamd_handle_t h, sluamd_doublecomplex *x, int64_t ldx, int32_t nrhs)
{
    if (!h || !h->H.z || !x || nrhs < 0 || ldx < h->H.hs.n) { set_error("bad complex solve arguments"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int64_t need = 2 * ldx * nrhs;   // in doubles
    if (need > H->x_cap) {
        if (H->d_x) hipFree(H->d_x);
        HIPCHK(hipMalloc((void **) &H->d_x, sizeof(double) * need));
        H->x_cap = need;
    }
    HIPCHK(hipMemcpy(H->d_x, x, sizeof(double) * need, hipMemcpyHostToDevice));
    const size_t lds = (size_t) H->max_nsupc * nrhs * 16;
    if (lds > 64 * 1024) { set_error("nrhs too large for the LDS-staged complex solve"); return SLUAMD_EINVAL; }
    const DevTables &T = H->T;
    hipStream_t s = H->stream;
    zc *dx = reinterpret_cast<zc *>(H->d_x);
    HIPCHK(hipEventRecord(H->ev0, s));
    for (size_t zl = 0; zl < H->sched.size(); ++zl) {
        LevelSched &S = H->sched[zl];
        for (int l = 0; l < S.nlevels; ++l) {
            const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
            hipLaunchKernelGGL(kz_solve_diag<true>, dim3(nn), dim3(256), lds, s, T, S.d_nodes + n0, dx, ldx, nrhs);
            const int nf = S.fwd_prefix[po + nn];
            if (nf) hipLaunchKernelGGL(kz_fwd_update, dim3(nf), dim3(256), lds, s, T, S.d_nodes + n0, S.d_fwd_prefix + po, nn, dx, ldx, nrhs);
        }
    }
    for (int zl = (int) H->sched.size() - 1; zl >= 0; --zl) {
        LevelSched &S = H->sched[zl];
        for (int l = S.nlevels - 1; l >= 0; --l) {
            const int n0 = S.lvl_off[l], nn = S.lvl_off[l + 1] - n0, po = S.lvl_poff[l];
            const int nb = S.bwd_prefix[po + nn];
            if (nb) hipLaunchKernelGGL(kz_bwd_update, dim3(nb), dim3(256), 0, s, T, S.d_nodes + n0, S.d_bwd_prefix + po, nn, dx, ldx, nrhs);
            hipLaunchKernelGGL(kz_solve_diag<false>, dim3(nn), dim3(256), lds, s, T, S.d_nodes + n0, dx, ldx, nrhs);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(H->ev1, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_solve_ms = ms;
    HIPCHK(hipMemcpy(x, H->d_x, sizeof(double) * need, hipMemcpyDeviceToHost));
    return 0;
}

int sluamd_pdgstrs3d_dev(sluamd_handle_t h, double *d_x, int64_t ldx, int32_t nrhs)
{
    if (!h || !d_x || nrhs < 0 || ldx < h->H.hs.n) { set_error("bad solve arguments"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("complex16 handle: call sluamd_pzgstrs3d"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    int rc = run_solve(H, d_x, ldx, nrhs);
    if (rc) return rc;
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_solve_ms = ms;
    return 0;
}

// ---- iterative refinement: pdgsrfs3d (SRC/double/pdgsrfs.c:345-510), SURVEY 8(f)-2 ----
int sluamd_dAttachMatrix(sluamd_handle_t h, sluamd_int_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                         const double *nzval, const sluamd_int_t *perm_c)
{
    if (!h || !rowptr || !colind || !nzval || !perm_c || n != h->H.hs.n) { set_error("bad matrix arguments"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("iterative refinement is double precision only in this round"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int64_t nnz = rowptr[n];
    for (void *p : {(void *) H->d_rfs_rp, (void *) H->d_rfs_ci, (void *) H->d_rfs_pc, (void *) H->d_rfs_av, (void *) H->d_rfs_work, (void *) H->d_rfs_s})
        if (p) hipFree(p);
    HIPCHK(hipMalloc((void **) &H->d_rfs_rp, sizeof(int) * (n + 1)));
    HIPCHK(hipMalloc((void **) &H->d_rfs_ci, sizeof(int) * std::max<int64_t>(nnz, 1)));
    HIPCHK(hipMalloc((void **) &H->d_rfs_av, sizeof(double) * std::max<int64_t>(nnz, 1)));
    HIPCHK(hipMalloc((void **) &H->d_rfs_pc, sizeof(int) * n));
    HIPCHK(hipMalloc((void **) &H->d_rfs_work, sizeof(double) * 3 * (size_t) n));   // r_perm | b | x
    HIPCHK(hipMalloc((void **) &H->d_rfs_s, sizeof(unsigned long long)));
    HIPCHK(hipMemcpy(H->d_rfs_rp, rowptr, sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(H->d_rfs_ci, colind, sizeof(int) * nnz, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(H->d_rfs_av, nzval, sizeof(double) * nnz, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(H->d_rfs_pc, perm_c, sizeof(int) * n, hipMemcpyHostToDevice));
    H->rfs_nnz = nnz;
    return 0;
}

// d_B, d_X: device-resident, original ordering, column-major; X holds the initial solution and is refined in place
int sluamd_pdgsrfs3d_dev(sluamd_handle_t h, const double *d_B, int64_t ldb, double *d_X, int64_t ldx, int32_t nrhs,
                         double *berr, int32_t *steps)
{
    if (!h || !d_B || !d_X || !berr || nrhs < 0 || ldb < h->H.hs.n || ldx < h->H.hs.n) { set_error("bad refinement arguments"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    if (!H->d_rfs_rp) { set_error("no matrix attached: call sluamd_dAttachMatrix first"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(H->device));
    const int n = H->hs.n;
    const int ITMAX = 20;                                   // pdgsrfs.c:371
    const double eps = 0x1p-53, safmin = 2.2250738585072014e-308;
    const double safe1 = (double) (n + 1) * safmin, safe2 = safe1 / eps;
    double *r_perm = H->d_rfs_work;
    hipStream_t s = H->stream;
    const dim3 grid((n + 255) / 256), blk(256);
    int count = 0;
    for (int j = 0; j < nrhs; ++j) {
        const double *Bc = d_B + (size_t) j * ldb;
        double *Xc = d_X + (size_t) j * ldx;
        double lstres = 3.0;
        count = 0;
        for (;;) {
            HIPCHK(hipMemsetAsync(H->d_rfs_s, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k_rfs_residual, grid, blk, 0, s, n, H->d_rfs_rp, H->d_rfs_ci, H->d_rfs_av, Xc, Bc, H->d_rfs_pc, r_perm, H->d_rfs_s, safe1, safe2);
            double sv = 0.0;
            HIPCHK(hipMemcpyAsync(&sv, H->d_rfs_s, sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            berr[j] = sv;
            if (sv > eps && sv * 2 <= lstres && count < ITMAX) {
                int rc = run_solve(H, r_perm, n, 1);
                if (rc) return rc;
                hipLaunchKernelGGL(k_rfs_update, grid, blk, 0, s, n, H->d_rfs_pc, r_perm, Xc);
                lstres = sv;
                ++count;
            } else break;
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (steps) *steps = count;
    return 0;
}

int sluamd_pdgsrfs3d(sluamd_handle_t h, const double *B, int64_t ldb, double *X, int64_t ldx, int32_t nrhs, double *berr,
                     int32_t *steps)
{
    if (!h || !B || !X || !berr || nrhs < 0 || ldb < h->H.hs.n || ldx < h->H.hs.n) { set_error("bad refinement arguments"); return SLUAMD_EINVAL; }
    if (nrhs == 0) { if (steps) *steps = 0; return 0; }
    Handle *H = &h->H;
    if (!H->d_rfs_rp) { set_error("no matrix attached: call sluamd_dAttachMatrix first"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(H->device));
    const int n = H->hs.n;
    double *d_b = H->d_rfs_work + n, *d_x = H->d_rfs_work + 2 * (size_t) n;
    int last = 0;
    for (int j = 0; j < nrhs; ++j) {   // one column at a time through the two resident work vectors
        HIPCHK(hipMemcpy(d_b, B + (size_t) j * ldb, sizeof(double) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_x, X + (size_t) j * ldx, sizeof(double) * n, hipMemcpyHostToDevice));
        int rc = sluamd_pdgsrfs3d_dev(h, d_b, n, d_x, n, 1, berr + j, &last);
        if (rc) return rc;
        HIPCHK(hipMemcpy(X + (size_t) j * ldx, d_x, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    if (steps) *steps = last;
    return 0;
}

int sluamd_pdgstrs3d(sluamd_handle_t h, double *x, int64_t ldx, int32_t nrhs)
{
    if (!h || !x || nrhs < 0 || ldx < h->H.hs.n) { set_error("bad solve arguments"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int64_t need = ldx * nrhs;
    if (need > H->x_cap) {
        if (H->d_x) hipFree(H->d_x);
        HIPCHK(hipMalloc((void **) &H->d_x, sizeof(double) * need));
        H->x_cap = need;
    }
    HIPCHK(hipMemcpy(H->d_x, x, sizeof(double) * need, hipMemcpyHostToDevice));
    int rc = sluamd_pdgstrs3d_dev(h, H->d_x, ldx, nrhs);
    if (rc) return rc;
    HIPCHK(hipMemcpy(x, H->d_x, sizeof(double) * need, hipMemcpyDeviceToHost));
    return 0;
}

void sluamd_dDestroyLUHandle(sluamd_handle_t h)
{
    if (!h) return;
    Handle *H = &h->H;
    hipSetDevice(H->device);
    if (H->stream) hipStreamSynchronize(H->stream);
    for (void *p : H->d_misc) hipFree(p);
    if (H->d_val) hipFree(H->d_val);
    if (H->d_info) hipFree(H->d_info);
    if (H->d_x) hipFree(H->d_x);
    if (H->d_apos) hipFree(H->d_apos);
    for (void *q : {(void *) H->d_rfs_rp, (void *) H->d_rfs_ci, (void *) H->d_rfs_pc, (void *) H->d_rfs_av, (void *) H->d_rfs_work, (void *) H->d_rfs_s})
        if (q) hipFree(q);
    if (H->d_aval) hipFree(H->d_aval);
    for (auto &e : H->ev_schur) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : H->ev_panel) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (H->ev0) hipEventDestroy(H->ev0);
    if (H->ev1) hipEventDestroy(H->ev1);
    for (auto e : H->ev_pool) hipEventDestroy(e);
    if (H->pstream) hipStreamDestroy(H->pstream);
    if (H->stream) hipStreamDestroy(H->stream);
    delete h;
}

int sluamd_get_stats(sluamd_handle_t h, sluamd_stats_t *out)
{
    if (!h || !out) return SLUAMD_EINVAL;
    *out = h->H.st;
    return 0;
}

// raw access for the multi-rank orchestration (RCCL collectives operate on arena slices)
int sluamd_arena(sluamd_handle_t h, double **d_val, int64_t *nnzL, int64_t *nnzU)
{
    if (!h) return SLUAMD_EINVAL;
    if (d_val) *d_val = h->H.d_val;
    if (nnzL) *nnzL = h->H.hs.nnzL;
    if (nnzU) *nnzU = h->H.hs.nnzU;
    return 0;
}

// tree ids on the path of layer z: tree(ilvl) as getGridTrees (supernodal_etree.c:840-851)
static void path_trees(int npdep, int myz, std::vector<int> &trees)
{
    int maxLvl = 1;
    while ((1 << (maxLvl - 1)) < npdep) ++maxLvl;
    trees.resize(maxLvl);
    trees[0] = npdep - 1 + myz;
    for (int i = 1; i < maxLvl; ++i) trees[i] = (trees[i - 1] - 1) / 2;
}

static int create_from_symb(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr,
                            const sluamd_int_t *colind, const double *nzval, const sluamd_int_t *perm_c_final,
                            const sluamd_options_t *opt, int npdep, int myz, const int32_t *sn_tree, bool z = false)
{
    if (!out || !s) { set_error("null argument"); return SLUAMD_EINVAL; }
    *out = nullptr;
    if (npdep < 1 || (npdep & (npdep - 1)) || myz < 0 || myz >= npdep || (npdep > 1 && !sn_tree)) { set_error("bad 1x1xPz arguments"); return SLUAMD_EINVAL; }
    sluamd_options_t o;
    if (opt) o = *opt; else sluamd_default_options(&o);
    int rc = check_device(o.device);
    if (rc) return rc;
    Symb *sy = reinterpret_cast<Symb *>(s);
    auto *hh = new sluamd_lu_handle_s();
    Handle *H = &hh->H;
    H->opt = o;
    H->z = z;
    H->Pz = npdep; H->myz = myz;
    HIPCHK(hipGetDevice(&H->device));
    const HostStruct &full = sy->hs;
    const int ns = full.nsupers;
    std::vector<uint8_t> owned;
    std::vector<std::vector<int>> lists;
    if (npdep == 1) {
        H->hs = full;  // structure copy (index arrays only)
    } else {
        // local subset: my leaf tree + the ancestor trees on my path; A's entries of an ancestor tree live on the
        // first layer of the group that shares it (the one that will factor it), dinit3DLUstructForest's rule
        std::vector<int> trees;
        path_trees(npdep, myz, trees);
        const int maxLvl = (int) trees.size();
        HostStruct &hs = H->hs;
        hs.n = full.n; hs.nsupers = ns; hs.xsup = full.xsup;
        hs.present.assign(ns, 0); owned.assign(ns, 0);
        lists.assign(maxLvl, {});
        for (int k = 0; k < ns; ++k)
            for (int l = 0; l < maxLvl; ++l)
                if (sn_tree[k] == trees[l]) {
                    hs.present[k] = 1;
                    lists[l].push_back(k);                 // schedule on every rank of the sharing group (cooperative mode)
                    if (myz % (1 << l) == 0) owned[k] = 1;   // A's entries only on the group's first layer
                }
        hs.lidx_off.assign(ns + 1, 0); hs.uidx_off.assign(ns + 1, 0); hs.lval_off.assign(ns + 1, 0); hs.uval_off.assign(ns + 1, 0);
        for (int k = 0; k < ns; ++k) {
            const int p = hs.present[k];
            hs.lidx_off[k + 1] = hs.lidx_off[k] + (p ? full.lidx_off[k + 1] - full.lidx_off[k] : 0);
            hs.uidx_off[k + 1] = hs.uidx_off[k] + (p ? full.uidx_off[k + 1] - full.uidx_off[k] : 0);
            hs.lval_off[k + 1] = hs.lval_off[k] + (p ? full.lval_off[k + 1] - full.lval_off[k] : 0);
            hs.uval_off[k + 1] = hs.uval_off[k] + (p ? full.uval_off[k + 1] - full.uval_off[k] : 0);
        }
        hs.nnzL = hs.lval_off[ns]; hs.nnzU = hs.uval_off[ns];
        hs.lidx.resize(hs.lidx_off[ns]); hs.uidx.resize(hs.uidx_off[ns]);
        for (int k = 0; k < ns; ++k) {
            if (!hs.present[k]) continue;
            std::copy(full.lidx.begin() + full.lidx_off[k], full.lidx.begin() + full.lidx_off[k + 1], hs.lidx.begin() + hs.lidx_off[k]);
            std::copy(full.uidx.begin() + full.uidx_off[k], full.uidx.begin() + full.uidx_off[k + 1], hs.uidx.begin() + hs.uidx_off[k]);
        }
    }
    const HostStruct &hs = H->hs;
    const int64_t tot = hs.nnzL + hs.nnzU;
    const size_t esz = z ? 16 : 8;
    if (hipMalloc((void **) &H->d_val, esz * std::max<int64_t>(tot, 1)) != hipSuccess) {
        set_error("hipMalloc of the value arena failed"); delete hh; return SLUAMD_ENOMEM;
    }
    HIPCHK(hipMemset(H->d_val, 0, esz * tot));
    {   // device-side distribution of A's values
        std::vector<int64_t> pos; std::vector<uint8_t> isu;
        compute_scatter_positions(*sy, hs, hs.n, rowptr, colind, perm_c_final, owned.empty() ? nullptr : owned.data(), pos, isu);
        const int w = z ? 2 : 1;   // doubles per value
        std::vector<int64_t> pos2; std::vector<double> val2;
        pos2.reserve(pos.size()); val2.reserve(pos.size() * w);
        for (size_t e = 0; e < pos.size(); ++e)
            if (pos[e] >= 0) {
                pos2.push_back(pos[e] + (isu[e] ? hs.nnzL : 0));
                for (int q = 0; q < w; ++q) val2.push_back(nzval[e * w + q]);
            }
        const int64_t nnz = (int64_t) pos2.size();
        HIPCHK(hipMalloc((void **) &H->d_apos, sizeof(int64_t) * std::max<int64_t>(nnz, 1)));
        HIPCHK(hipMalloc((void **) &H->d_aval, esz * std::max<int64_t>(nnz, 1)));
        H->a_nnz = nnz;
        HIPCHK(hipMemcpy(H->d_apos, pos2.data(), sizeof(int64_t) * nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(H->d_aval, val2.data(), esz * nnz, hipMemcpyHostToDevice));
        if (nnz && !z) hipLaunchKernelGGL(k_scatter_values, dim3((unsigned) ((nnz + 255) / 256)), dim3(256), 0, 0, H->d_val, H->d_apos, H->d_aval, nnz);
        if (nnz && z) hipLaunchKernelGGL(kz_scatter_values, dim3((unsigned) ((nnz + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<zc *>(H->d_val), H->d_apos, reinterpret_cast<const zc *>(H->d_aval), nnz);
        HIPCHK(hipDeviceSynchronize());
    }
    rc = finish_create(H, nullptr, lists.empty() ? nullptr : &lists);
    if (rc) { sluamd_dDestroyLUHandle(hh); return rc; }
    *out = hh;
    return 0;
}

int sluamd_dCreateLUHandleFromSymb(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr,
                                   const sluamd_int_t *colind, const double *nzval,
                                   const sluamd_int_t *perm_c_final, const sluamd_options_t *opt)
{
    return create_from_symb(out, s, rowptr, colind, nzval, perm_c_final, opt, 1, 0, nullptr);
}

// complex16 values (nzval = doublecomplex[nnz] aligned with colind), 1x1x1 grid
int sluamd_zCreateLUHandleFromSymb(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr,
                                   const sluamd_int_t *colind, const sluamd_doublecomplex *nzval,
                                   const sluamd_int_t *perm_c_final, const sluamd_options_t *opt)
{
    return create_from_symb(out, s, rowptr, colind, reinterpret_cast<const double *>(nzval), perm_c_final, opt, 1, 0, nullptr, true);
}

// 1 x 1 x npdep grid: this rank = Z layer `myz`; sn_tree from sluamd_symb_partition.  Stores only the layer's own
// sub-forest and its ancestors (replicated ancestors start from zero except on their owner layer).
int sluamd_dCreateLUHandleFromSymb3D(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr,
                                     const sluamd_int_t *colind, const double *nzval, const sluamd_int_t *perm_c_final,
                                     const sluamd_options_t *opt, int32_t npdep, int32_t myz, const int32_t *sn_tree)
{
    return create_from_symb(out, s, rowptr, colind, nzval, perm_c_final, opt, npdep, myz, sn_tree);
}

// offsets of every supernode's L panel / U row inside this rank's value arena ([L | U]; U offsets relative to the
// U half) -- the multi-rank orchestration reduces ancestor slices with them
int sluamd_local_offsets(sluamd_handle_t h, int64_t *lval_off, int64_t *uval_off)
{
    if (!h) return SLUAMD_EINVAL;
    const HostStruct &hs = h->H.hs;
    if (lval_off) std::copy(hs.lval_off.begin(), hs.lval_off.end(), lval_off);
    if (uval_off) std::copy(hs.uval_off.begin(), hs.uval_off.end(), uval_off);
    return 0;
}

// forward (dir=+1) or backward (dir=-1) block solve restricted to one Z level, device-resident x
int sluamd_pdgstrs3d_level(sluamd_handle_t h, int zlevel, int dir, double *d_x, int64_t ldx, int32_t nrhs)
{
    if (!h || !d_x || zlevel < 0 || zlevel >= (int) h->H.sched.size() || (dir != 1 && dir != -1)) { set_error("bad level-solve arguments"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    int rc = run_solve(H, d_x, ldx, nrhs, dir, zlevel);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(H->stream));
    return 0;
}

// Device-side re-distribution of A's values into the resident store (handles made by
// sluamd_dCreateLUHandleFromSymb): zero-fill + scatter, asynchronous on the handle's stream.
int sluamd_dResetValues(sluamd_handle_t h)
{
    if (!h || !h->H.d_apos) { set_error("handle has no device-side copy of A"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    H->dinv_ready = false;
    HIPCHK(hipMemsetAsync(H->d_val, 0, (H->z ? 16 : 8) * (H->hs.nnzL + H->hs.nnzU), H->stream));
    if (H->a_nnz && !H->z) hipLaunchKernelGGL(k_scatter_values, dim3((unsigned) ((H->a_nnz + 255) / 256)), dim3(256), 0, H->stream,
                                              H->d_val, H->d_apos, H->d_aval, H->a_nnz);
    if (H->a_nnz && H->z) hipLaunchKernelGGL(kz_scatter_values, dim3((unsigned) ((H->a_nnz + 255) / 256)), dim3(256), 0, H->stream,
                                             reinterpret_cast<zc *>(H->d_val), H->d_apos, reinterpret_cast<const zc *>(H->d_aval), H->a_nnz);
    HIPCHK(hipGetLastError());
    return 0;
}

int sluamd_device_synchronize(void)
{
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

int sluamd_set_profile(sluamd_handle_t h, int on)
{
    if (!h) return SLUAMD_EINVAL;
    h->H.opt.verbose = on ? 2 : 0;
    return 0;
}

// test hook: MFMA fp64 fragment layout check
int sluamd_mfma_selftest(const double *A, const double *B, double *D)
{
    int rc = check_device(-1);
    if (rc) return rc;
    double *dA, *dB, *dD;
    HIPCHK(hipMalloc((void **) &dA, 64 * 8)); HIPCHK(hipMalloc((void **) &dB, 64 * 8)); HIPCHK(hipMalloc((void **) &dD, 256 * 8));
    HIPCHK(hipMemcpy(dA, A, 64 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dB, B, 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    HIPCHK(hipMemcpy(D, dD, 256 * 8, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dB); hipFree(dD);
    return 0;
}

}  // extern "C"
