// sluamd_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 3D supernodal LU hot path + their launchers
// (the `eng::` interface of sluamd_internal.h).  Per level of the elimination-DAG schedule:
//     k_diag_lu      unpivoted LU of every diagonal block of the level   (Local_Dgstrf2, pdgstrf2.c:508)
//     k_panel_trsm<0> L(:,k) <- L(:,k) U_kk^-1                           (dLPanelTrSolve, dtrfCommWrapper.c:120)
//     k_panel_trsm<1> U(k,:) <- L_kk^-1 U(k,:) directly on the skyline   (dTrs2_GatherTrsmScatter, pdgstrf2.c:804)
//     k_schur        A(I,J) -= L(I,k) U(k,J): fused gather -> fp64 MFMA GEMM -> scatter, no bigU/bigV
//                    round trip (dRgather_L/U dgather.c:133-398 + dblock_gemm_scatter dscatter3d.c:81-189
//                    + dscatter_l dscatter.c:109 + scatter_u dscatter3d.c:555)
// and for pdgstrs3d the level-set forward/backward block solves (dlsum_fmod_inv / dlsum_bmod_inv,
// pdgstrs_lsum.c:414 / :1362).
//
// On an XY block-cyclic layer (nprow * npcol > 1) the same kernels run on "slots": the L slot of supernode k is this
// process row's part of panel k (own storage or the image received from process column k % Pc), the U slot this process
// column's part of block row k; the diagonal block may sit at the top of the L slot or in a scratch range
// (DevTables::sn_dptr / sn_dlda / sn_ldiag).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "sluamd_internal.h"

namespace sluamd {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int find_node(const int *__restrict__ prefix, int nn, int id)
{   // largest i in [0,nn) with prefix[i] <= id   (prefix has nn+1 entries)
    int lo = 0, hi = nn;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= id) lo = mid; else hi = mid;
    }
    return lo;
}

// The same search by a whole wave -- id wave-uniform, all 64 lanes active (kernel entry): 64 probes per round trip instead of one.  The kernels of a
// latency-bound level start with this search; a binary search over a level of 16 supernodes is four dependent loads, over a leaf level of 9 000 fourteen.
#ifndef SLUAMD_FIND_WAVE
#define SLUAMD_FIND_WAVE 1
#endif
__device__ __forceinline__ int find_node_wave(const int *__restrict__ prefix, int nn, int id)
{
#if SLUAMD_FIND_WAVE
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = nn;                             // prefix[lo] <= id < prefix[hi]
    while (hi - lo > 1) {
        const int stride = (hi - lo + 62) >> 6;      // candidates lo + 1 .. hi - 1 in at most 64 steps of `stride`
        const int p = lo + (lane + 1) * stride;
        const bool le = p < hi && prefix[p] <= id;   // monotone: the set lanes are a prefix of the wave
        const int cnt = __popcll(__ballot(le));
        lo += cnt * stride;
        hi = min(hi, lo + stride);
    }
    return __builtin_amdgcn_readfirstlane(lo);
#else
    return find_node(prefix, nn, id);
#endif
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

__device__ __forceinline__ void pivot_fix(double *p, int col1based, int replace_tiny, double thresh, int *info, double *s_piv)
{
    double v = *p;
    if (replace_tiny && fabs(v) < thresh) { v = (v < 0) ? -thresh : thresh; *p = v; atomicAdd(&info[1], 1); }
    if (v == 0.0) { atomicMin(&info[0], col1based); atomicMax(&info[4], col1based); }
    *s_piv = v;
}

__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{   // wave-uniform source lane (compile-time after unrolling) -> v_readlane_b32 x2, no LDS
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// Reciprocal of a pivot off the IEEE division sequence: v_rcp_f64 + two Newton steps (5 dependent instructions instead of ~12;
// relative error < 2^-52 for normal pivots -- tiny pivots were replaced, zero pivots never get here).  The pivot chain of the one-wave
// LU kernels is latency: 32 (64) dependent reciprocals per block.
__device__ __forceinline__ double pivot_recip(double p)
{
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0);
    return __builtin_fma(r, e, r);
}

// pivot of column j as the one-wave kernels see it: broadcast, tiny-pivot replacement (pdgstrf2.c:544-560), zero-pivot info (:568-571);
// returns 1 / pivot (1 for a zero pivot, which leaves the column unscaled like pdgstrf2.c:566-575)
__device__ __forceinline__ double wave_pivot(double &acol, int j, int lane, int col1based, int replace_tiny, double thresh, int *info)
{
    double p = lane_bcast(acol, j);
    if (replace_tiny && fabs(p) < thresh) {
        p = (p < 0) ? -thresh : thresh;
        if (lane == j) acol = p;
        if (lane == 0) atomicAdd(&info[1], 1);
    }
    if (p == 0.0 && lane == 0) { atomicMin(&info[0], col1based); atomicMax(&info[4], col1based); }
    return (p != 0.0) ? pivot_recip(p) : 1.0;
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
    // software pipeline of the pivot chain: step j updates column j + 1 FIRST and starts the reciprocal of pivot j + 1 at once, so that
    // its latency hides behind the updates of the columns j + 2 .. 31 instead of heading the next step
    double rinv = wave_pivot(a[0], 0, lane, col1, replace_tiny && 0 < nb, thresh, info);
#pragma unroll
    for (int j = 0; j < DB; ++j) {
        if (lane == 0) s_rinv[j] = rinv;
        const bool below = lane > j;
        const double l = a[j] * rinv;
        if (below) a[j] = l;
        const double lm = below ? l : 0.0;      // rows at and above the pivot take a zero multiplier: one FMA per column, no select
        double rnext = 1.0;
        if (j + 1 < DB) {
            const double u = lane_bcast(a[j + 1], j);
            a[j + 1] -= lm * u;
            rnext = (j + 1 < nb) ? wave_pivot(a[j + 1], j + 1, lane, col1 + j + 1, replace_tiny, thresh, info) : 1.0;   // identity padding past nb
        }
#pragma unroll
        for (int c = j + 2; c < DB; ++c) {
            const double u = lane_bcast(a[c], j);
            a[c] -= lm * u;
            if ((c & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bound the live range of the broadcast SGPRs
        }
        __builtin_amdgcn_sched_barrier(0);
        rinv = rnext;
    }
#pragma unroll
    for (int c = 0; c < DB; ++c) if (lane < nb && c < nb) P[c * ld + lane] = a[c];
}

// Diagonal blocks of at most 64 columns -- the tens of thousands of leaf supernodes at the bottom of the elimination DAG: ONE WAVE per
// block, the whole block in registers (lane r holds row r, 64 columns = 128 VGPRs), pivot rows broadcast with v_readlane: no LDS, no
// barrier, one load and one store of the block.  (The workgroup-per-block kernel below spends ~150 us per block in barriers and L2
// round trips: 1.5 - 2.4 ms per level for the four bottom levels of the 100^3 tree, exposed -- nothing else can run yet.)
// Arithmetic as k_diag_lu / Local_Dgstrf2 (pdgstrf2.c:508-601): unpivoted right-looking elimination, tiny-pivot replacement, zero-pivot info.
__global__ __launch_bounds__(256, 2) void k_diag_lu_wave(DevTables T, const int *__restrict__ nodes, int nn, int replace_tiny, double thresh,
                                                         int *__restrict__ info)
{
    const int bi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per block; the launch chooses how many waves share a workgroup
    if (bi >= nn) return;
    const int k = nodes[bi];
    if (!(T.sn_flags[k] & SNF_OWN_DIAG)) return;
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const int lda = T.sn_dlda[k];
    double *A = T.val + T.sn_dptr[k];
    const int lane = threadIdx.x & 63;
    const bool rok = lane < ns;
    double a[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) a[c] = (rok && c < ns) ? A[lane + (size_t) c * lda] : ((c == lane) ? 1.0 : 0.0);   // identity-padded
    // pivot chain pipelined as in wave_lu32: column j + 1 first, its reciprocal in flight behind the other columns' updates
    double rinv = wave_pivot(a[0], 0, lane, fst + 1, replace_tiny, thresh, info);
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        if (j < ns) {                                  // wave-uniform
            const bool below = lane > j;
            const double l = a[j] * rinv;
            if (below) a[j] = l;
            const double lm = below ? l : 0.0;
            double rnext = 1.0;
            if (j + 1 < 64) {
                const double u = lane_bcast(a[j + 1], j);
                a[j + 1] -= lm * u;
                if (j + 1 < ns) rnext = wave_pivot(a[j + 1], j + 1, lane, fst + j + 2, replace_tiny, thresh, info);
            }
#pragma unroll
            for (int c = j + 2; c < 64; ++c) {
                const double u = lane_bcast(a[c], j);
                a[c] -= lm * u;
                if ((c & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bound the live range of the broadcast SGPRs
            }
            rinv = rnext;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) if (rok && c < ns) A[lane + (size_t) c * lda] = a[c];
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
    if (!(T.sn_flags[k] & SNF_OWN_DIAG)) return;           // XY grid: the owner of the diagonal block factors it
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const int lda = T.sn_dlda[k];
    double *A = T.val + T.sn_dptr[k];
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

// ---- diagonal block LU, Crout form ------------------------------------------------------------------------------------
// Same arithmetic contract as k_diag_lu (Local_Dgstrf2, pdgstrf2.c:508-601: unpivoted, tiny-pivot replacement, zero-pivot
// info), organised so that nothing on the critical path of the factorisation waits on read-modify-write round trips and so
// that the workgroup (66 KB of LDS, <= 256 VGPRs, 4 waves) starts in the slot ONE retiring Schur workgroup frees: the panel
// chain runs inside the previous level's Schur update instead of waiting for a CU to drain.  Per 32-column step jb
//   A  column panel  C = A[jb:, jb:jb+32] - L[jb:, 0:jb] U[0:jb, jb:jb+32]   left-looking fp64 MFMA; U staged through LDS in
//      K halves of 112, the L fragments loaded in register batches; C stays in the ACCUMULATORS (head rows -> LDS)
//   B  32 x 32 head factored in registers by one wave (wave_lu32)
//   C  Uinv11 = inv(U11), LinvT11 = inv(L11^T) by two other waves (also written to T.dinv: k_diag_inv's job for the owner)
//   D  L21 = C21 Uinv11: the accumulator layout D[(l>>4)+4r][l&15] IS the B-operand layout, C21 never leaves registers
//   E  row panel  R = A[jb:jb+32, jb+32:] - L[jb:jb+32, 0:jb] U[0:jb, jb+32:] with the operand roles swapped so that R lands in
//      the A-operand layout of U12 = Linv11 R: no LDS round trip either
// Every element of the block is written exactly once.
constexpr int DKH = 112;                          // K half: 28 MFMA k-steps = one register batch of fragments
constexpr int DST = DKH + 2;                      // stage stride (== 18 mod 32: conflict-free fragment reads)
constexpr int DXS = 4 * DB * (DB + 1) + 2 * 16 * 17;   // staging buffer: >= 32 * DST and the phase-C scratch
constexpr size_t DIAG_LU2_LDS = sizeof(double) * (DXS + DB * 33 + DB * 34 + DB * 48);

// MINB = 2 (default): at most 256 registers per lane, the kernel starts in the slot of ONE retiring Schur workgroup -- the compiler pays with
// ~170 spilled VGPRs (scratch traffic in the phases that hold 2 x 28 operand fragments beside the accumulators).  MINB = 1: the whole register
// file (256 VGPRs + 256 AGPRs, no scratch) for the levels where nothing else competes for the CU -- the single-supernode levels at the top
// of the tree, whose trailing update is shorter than the panel chain (flag bit 2 of eng::diag_lu).
template <int MINB>
__global__ __launch_bounds__(256, MINB) void k_diag_lu2(DevTables T, const int *__restrict__ nodes, int replace_tiny, double thresh,
                                                        int *__restrict__ info)
{
    __builtin_amdgcn_s_setprio(3);   // panel chain: its waves go first when they share a SIMD with Schur tiles
    extern __shared__ double dsm[];
    double *X = dsm;                             // phase A: U(kk, c) at X[c * DST + kk]; phase E: L(row, kk) at X[row * DST + kk]; phase C scratch
    double *Hs = X + DXS;                        // the 32 x 32 head: (r, c) at Hs[c * 33 + r]
    double *Uis = Hs + DB * 33;                  // Uinv11: (kk, n) at Uis[n * 34 + kk]
    double *Lis = Uis + DB * 34;                 // Linv11: (i, kk) at Lis[kk * 48 + i]
    __shared__ double s_rinv[DB];
    const int k = nodes[blockIdx.x];
    if (!(T.sn_flags[k] & SNF_OWN_DIAG)) return;
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const int lda = T.sn_dlda[k];
    double *A = T.val + T.sn_dptr[k];
    const int nblk = (ns + DB - 1) / DB;
    double *dinv = T.dinv + T.sn_dinv[k];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    for (int jb = 0; jb < ns; jb += DB) {
        const int nb = min(DB, ns - jb), m = ns - jb, nc = m - nb;
        const int nrbp = (m + 15) >> 4;          // 16-row blocks of the column panel (blocks 0 and 1 are the head)
        // ---- A: column panel, left-looking; this wave owns blocks wave + 4 s (s = 0..3), columns 0-15 and 16-31 ----
        d4 cp[4][2];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { cp[s4][0] = (d4){0.0, 0.0, 0.0, 0.0}; cp[s4][1] = cp[s4][0]; }
        for (int k0 = 0; k0 < jb; k0 += DKH) {
            const int kh = min(DKH, jb - k0), nq = kh >> 2;
            __syncthreads();                     // X is free
            for (int idx = tid; idx < kh * DB; idx += 256) {     // kk fastest: coalesced
                const int kk = idx % kh, c = idx / kh;
                X[c * DST + kk] = (c < nb) ? A[k0 + kk + (size_t) (jb + c) * lda] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int b0 = wave + 8 * s2, b1 = b0 + 4;
                if (b0 < nrbp) {
                    const bool has1 = b1 < nrbp;
                    const double *L0 = A + jb + min(16 * b0 + li, m - 1) + (size_t) k0 * lda;
                    const double *L1 = A + jb + min(16 * b1 + li, m - 1) + (size_t) k0 * lda;
                    double lf0[28], lf1[28];
#pragma unroll
                    for (int qq = 0; qq < 28; ++qq) {
                        const int kk = 4 * qq + lk;
                        lf0[qq] = (qq < nq) ? L0[(size_t) kk * lda] : 0.0;
                        lf1[qq] = (qq < nq && has1) ? L1[(size_t) kk * lda] : 0.0;
                    }
#pragma unroll
                    for (int qq = 0; qq < 28; ++qq) {
                        if (qq < nq) {
                            const int kk = 4 * qq + lk;
                            const double u0 = X[li * DST + kk], u1 = X[(16 + li) * DST + kk];
                            cp[2 * s2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(u0, lf0[qq], cp[2 * s2][0], 0, 0, 0);
                            cp[2 * s2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(u1, lf0[qq], cp[2 * s2][1], 0, 0, 0);
                            cp[2 * s2 + 1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(u0, lf1[qq], cp[2 * s2 + 1][0], 0, 0, 0);
                            cp[2 * s2 + 1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(u1, lf1[qq], cp[2 * s2 + 1][1], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // C = A - acc, in place in the accumulators (lane: row 16 b + li, columns lk + 4 r and 16 + lk + 4 r); head rows -> Hs
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int b = wave + 4 * s4, row = 16 * b + li;
            if (b < nrbp) {
                double av[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = lk + 4 * r;
                    av[r] = (row < m && c < nb) ? A[jb + row + (size_t) (jb + c) * lda] : 0.0;
                    av[4 + r] = (row < m && 16 + c < nb) ? A[jb + row + (size_t) (jb + 16 + c) * lda] : 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = lk + 4 * r;
                    cp[s4][0][r] = (row < m && c < nb) ? av[r] - cp[s4][0][r] : 0.0;
                    cp[s4][1][r] = (row < m && 16 + c < nb) ? av[4 + r] - cp[s4][1][r] : 0.0;
                }
                if (s4 == 0 && b < 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { Hs[(lk + 4 * r) * 33 + row] = cp[0][0][r]; Hs[(16 + lk + 4 * r) * 33 + row] = cp[0][1][r]; }
                }
            }
        }
        __syncthreads();
        // ---- B: head ----
        if (wave == 0) wave_lu32(Hs, 33, nb, fst + jb + 1, replace_tiny, thresh, info, s_rinv);
        __syncthreads();
        // ---- C: inverses of the head's triangles (identity-padded past nb), kept in LDS and written to T.dinv; head -> memory ----
        if (wave == 1 || wave == 2) {
            const int typ = wave - 1;            // 0: U11 ; 1: L11^T (unit)
            double *Bs = X + typ * DB * (DB + 1);                // B(i, jj) at Bs[i * 33 + jj]
            double *Xs = X + (2 + typ) * DB * (DB + 1);          // X = inv(B): X(i, jj) at Xs[i * 33 + jj]
            double *Tm = X + 4 * DB * (DB + 1) + typ * 16 * 17;
            for (int e = lane; e < DB * DB; e += 64) {
                const int i = e >> 5, jj = e & 31;
                double v = (i == jj) ? 1.0 : 0.0;
                if (i < nb && jj < nb && i <= jj) {
                    if (typ == 0) v = Hs[jj * 33 + i];           // U(i, jj)
                    else if (i < jj) v = Hs[i * 33 + jj];        // L(jj, i) = (L^T)(i, jj)
                }
                Bs[i * (DB + 1) + jj] = v;
                Xs[i * (DB + 1) + jj] = 0.0;
            }
            // (one wave: its LDS operations complete in order)
            // inv of the two 16 x 16 diagonal blocks: lane c < 32 solves column c % 16 of block c / 16 in registers
            if (lane < 32) {
                const int o = lane & 16, cc = lane & 15;
                double xi[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) xi[i] = 0.0;
#pragma unroll
                for (int i = 15; i >= 0; --i) {
                    double a = (i == cc) ? 1.0 : 0.0;
#pragma unroll
                    for (int jj = i + 1; jj < 16; ++jj) a -= Bs[(o + i) * (DB + 1) + o + jj] * xi[jj];
                    xi[i] = (i <= cc) ? a * ((typ == 0) ? s_rinv[o + i] : 1.0) : 0.0;    // 1 / U(i,i) from the head's factorisation; L is unit
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) Xs[(o + i) * (DB + 1) + o + cc] = xi[i];
            }
            // X01 = -X00 (B01 X11): two 16 x 16 x 16 products on MFMA (whole wave)
            {
                d4 t = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q)     // T(i, j) = sum_k B01(i, k) X11(k, j)
                    t = __builtin_amdgcn_mfma_f64_16x16x4f64(Bs[li * (DB + 1) + 16 + 4 * q + lk], Xs[(16 + 4 * q + lk) * (DB + 1) + 16 + li], t, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Tm[(lk + 4 * r) * 17 + li] = t[r];
                d4 x = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q)     // X01(i, j) = -sum_k X00(i, k) T(k, j)
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[li * (DB + 1) + 4 * q + lk], Tm[(4 * q + lk) * 17 + li], x, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Xs[(lk + 4 * r) * (DB + 1) + 16 + li] = -x[r];
            }
            // outputs: T.dinv block D(kk, cc) at [cc * 32 + kk]; Uis[n * 34 + kk] = Uinv(kk, n); Lis[kk * 48 + row] = Linv(row, kk)
            double *dst = dinv + (size_t) (typ * nblk + jb / DB) * DB * DB;
            for (int e = lane; e < DB * DB; e += 64) {
                const int kk = e & 31, c2 = e >> 5;
                const double v = Xs[kk * (DB + 1) + c2];
                dst[c2 * DB + kk] = v;
                if (typ == 0) Uis[c2 * 34 + kk] = v; else Lis[kk * 48 + c2] = v;
            }
        } else {
            // waves 0 and 3: the factored head L11 \ U11 -> memory
            for (int e = lane + 64 * (wave == 3); e < DB * DB; e += 128) {
                const int r = e & 31, c = e >> 5;
                if (r < nb && c < nb) A[jb + r + (size_t) (jb + c) * lda] = Hs[c * 33 + r];
            }
        }
        __syncthreads();
        if (nc > 0) {
            // ---- D: L21 = C21 Uinv11, straight from the accumulators to memory (nc > 0 implies nb == 32) ----
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int b = wave + 4 * s4, row = 16 * b + li;
                if (b >= 2 && b < nrbp) {
                    d4 a0 = (d4){0.0, 0.0, 0.0, 0.0}, a1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const double bfr = cp[s4][q >> 2][q & 3];
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Uis[li * 34 + 4 * q + lk], bfr, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Uis[(16 + li) * 34 + 4 * q + lk], bfr, a1, 0, 0, 0);
                    }
                    if (row < m) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            A[jb + row + (size_t) (jb + lk + 4 * r) * lda] = a0[r];
                            A[jb + row + (size_t) (jb + 16 + lk + 4 * r) * lda] = a1[r];
                        }
                    }
                }
            }
            // ---- E: row panel; pass p covers columns 128 p .. 128 p + 127, this wave's blocks 16 w and 64 + 16 w of the pass ----
            d4 rr[2][2][2];                      // [pass][block A / B][row half]: lane (row lk + 4 r (+16), column li)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                for (int g = 0; g < 2; ++g) { rr[p2][g][0] = (d4){0.0, 0.0, 0.0, 0.0}; rr[p2][g][1] = rr[p2][g][0]; }
            for (int k0 = 0; k0 < jb; k0 += DKH) {
                const int kh = min(DKH, jb - k0), nq = kh >> 2;
                __syncthreads();                 // X is free (phase C scratch / previous half)
                for (int idx = tid; idx < kh * DB; idx += 256) { const int row = idx & 31, kk = idx >> 5; X[row * DST + kk] = A[jb + row + (size_t) (k0 + kk) * lda]; }
                __syncthreads();
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int cA = 128 * p2 + 16 * wave, cB = cA + 64;
                    if (cA < nc) {
                        const bool actB = cB < nc;
                        const double *UcA = A + k0 + (size_t) min(jb + nb + cA + li, ns - 1) * lda;
                        const double *UcB = A + k0 + (size_t) min(jb + nb + cB + li, ns - 1) * lda;
                        double ufA[28], ufB[28];
#pragma unroll
                        for (int qq = 0; qq < 28; ++qq) {
                            const int kk = 4 * qq + lk;
                            ufA[qq] = (qq < nq) ? UcA[kk] : 0.0;
                            ufB[qq] = (qq < nq && actB) ? UcB[kk] : 0.0;
                        }
#pragma unroll
                        for (int qq = 0; qq < 28; ++qq) {
                            if (qq < nq) {
                                const int kk = 4 * qq + lk;
                                const double l0 = X[li * DST + kk], l1 = X[(16 + li) * DST + kk];
                                rr[p2][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l0, ufA[qq], rr[p2][0][0], 0, 0, 0);
                                rr[p2][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, ufA[qq], rr[p2][0][1], 0, 0, 0);
                                rr[p2][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l0, ufB[qq], rr[p2][1][0], 0, 0, 0);
                                rr[p2][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, ufB[qq], rr[p2][1][1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c0 = 128 * p2 + 64 * g + 16 * wave;
                    if (c0 < nc) {
                        const int cg = jb + nb + c0;
                        // R = A12 - acc in the accumulator layout (row lk + 4 r (+16), column li) = the A-operand layout of U12 = Linv11 R
                        const bool cok = cg + li < ns;
                        const double *Ac = A + jb + (size_t) min(cg + li, ns - 1) * lda;
                        double R[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { R[r] = cok ? Ac[lk + 4 * r] : 0.0; R[4 + r] = cok ? Ac[16 + lk + 4 * r] : 0.0; }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { R[r] = cok ? R[r] - rr[p2][g][0][r] : 0.0; R[4 + r] = cok ? R[4 + r] - rr[p2][g][1][r] : 0.0; }
                        d4 o0 = (d4){0.0, 0.0, 0.0, 0.0}, o1 = o0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {   // U12(i', col) = sum_k Linv(i', k) R(k, col)
                            o0 = __builtin_amdgcn_mfma_f64_16x16x4f64(R[q], Lis[(4 * q + lk) * 48 + li], o0, 0, 0, 0);
                            o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(R[q], Lis[(4 * q + lk) * 48 + 16 + li], o1, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = cg + lk + 4 * r;
                            if (col < ns) { A[jb + li + (size_t) col * lda] = o0[r]; A[jb + 16 + li + (size_t) col * lda] = o1[r]; }
                        }
                    }
                }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// Inverses of the 32x32 diagonal sub-blocks of U_kk (typ 0) and L_kk^T (typ 1, unit), identity-padded past
// ns, written to T.dinv; 4 sub-blocks per 128-thread workgroup, one thread per column of an inverse
// (back substitution with the block and the private solution column staged in LDS).
__global__ __launch_bounds__(128) void k_diag_inv(DevTables T, const int *__restrict__ nodes,
                                                  const int *__restrict__ prefix, int nn)
{
    __builtin_amdgcn_s_setprio(3);   // panel chain: its waves go first when they share a SIMD with Schur tiles
    __shared__ double Bs[4][DB * (DB + 1)];     // 34 KB: the workgroup fits beside two Schur workgroups on a CU
    const int g = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int task = blockIdx.x * 4 + g;
    bool valid = task < prefix[nn];
    int k = 0, typ = 0, b = 0, ns = 0, lda = 1, nblk = 1;
    const double *A = nullptr;
    int ni = 0;
    if (valid) {
        ni = find_node(prefix, nn, task);
        k = nodes[ni];
        if (!(T.sn_flags[k] & SNF_HAS_DIAG)) valid = false;
    }
    if (valid) {
        ns = T.xsup[k + 1] - T.xsup[k];
        nblk = (ns + DB - 1) / DB;
        const int rem = task - prefix[ni];
        typ = rem / nblk; b = rem - typ * nblk;
        lda = T.sn_dlda[k];
        A = T.val + T.sn_dptr[k];
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
        // column c of the inverse by back substitution, the private solution column in registers (static indices: rows past
        // c stay zero, so the sums may run over the full width)
        double xi[DB];
#pragma unroll
        for (int i = 0; i < DB; ++i) xi[i] = 0.0;
#pragma unroll
        for (int i = DB - 1; i >= 0; --i) {
            double a = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int jj = i + 1; jj < DB; ++jj) a -= Bs[g][i * (DB + 1) + jj] * xi[jj];
            xi[i] = (i <= c) ? a / Bs[g][i * (DB + 1) + i] : 0.0;
        }
        double *dst = T.dinv + T.sn_dinv[k] + (size_t) (typ * nblk + b) * DB * DB + c * DB;
#pragma unroll
        for (int i = 0; i < DB; ++i) dst[i] = xi[i];
    }
}

// the same inverses for the OWNED diagonal blocks of a node list, one workgroup per supernode (companion of the
// right-looking k_diag_lu; k_diag_lu2 writes them itself)
template <int NG>   // NG groups of 32 threads, one 32 x 32 inverse each: 8 (any width), or 4 for levels of <= 64-wide supernodes (at most 4 inverses)
__global__ __launch_bounds__(NG * 32) void k_diag_inv_all(DevTables T, const int *__restrict__ nodes)
{
    __shared__ double Bs[NG][DB * (DB + 1)];
    const int k = nodes[blockIdx.x];
    if (!(T.sn_flags[k] & SNF_OWN_DIAG)) return;
    const int ns = T.xsup[k + 1] - T.xsup[k], nblk = (ns + DB - 1) / DB, lda = T.sn_dlda[k];
    const double *A = T.val + T.sn_dptr[k];
    const int g = threadIdx.x >> 5, c = threadIdx.x & 31;
    for (int t0 = 0; t0 < 2 * nblk; t0 += NG) {
        const int task = t0 + g;
        const bool valid = task < 2 * nblk;
        const int typ = task / nblk, b = task - typ * nblk, o = b * DB;
        if (valid)
            for (int i = 0; i < DB; ++i) {
                double v = (i == c) ? 1.0 : 0.0;
                if (o + i < ns && o + c < ns && i <= c) {
                    if (typ == 0) v = A[o + i + (size_t) (o + c) * lda];
                    else if (i < c) v = A[o + c + (size_t) (o + i) * lda];
                }
                Bs[g][i * (DB + 1) + c] = v;
            }
        __syncthreads();
        if (valid) {
            double xi[DB];
#pragma unroll
            for (int i = 0; i < DB; ++i) xi[i] = 0.0;
#pragma unroll
            for (int i = DB - 1; i >= 0; --i) {
                double a = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int jj = i + 1; jj < DB; ++jj) a -= Bs[g][i * (DB + 1) + jj] * xi[jj];
                xi[i] = (i <= c) ? a / Bs[g][i * (DB + 1) + i] : 0.0;
            }
            double *dst = T.dinv + T.sn_dinv[k] + (size_t) (typ * nblk + b) * DB * DB + c * DB;
#pragma unroll
            for (int i = 0; i < DB; ++i) dst[i] = xi[i];
        }
        __syncthreads();
    }
}

// ---- panel TRSMs: blocked by 32 with inverted diagonal sub-blocks, GEMM parts on fp64 MFMA ---------------
// MODE 0  L(:,k) <- L(:,k) U_kk^-1        (dLPanelTrSolve, dtrfCommWrapper.c:120-223: TRSM R,U,N,N)
//         strip = 32 panel rows; T = U_kk.
// MODE 3  Uinv = I U_kk^-1 and MODE 2  Linv^T = I (L_kk^T)^-1: the SAME solves on strips of the identity give the full
//         inverses of the diagonal block (pdCompute_Diag_Inv, pdgstrs.c:842: dtrtri), stored at T.inv for the solve.
// MODE 1  U(k,:) <- L_kk^-1 U(k,:)        (dTrs2_GatherTrsmScatter, pdgstrf2.c:804-840: gather, TRSM L,L,N,U,
//         scatter) solved as X^T L_kk^T = B^T on the skyline in place: strip = 32 non-empty U columns
//         (implicit zero padding above each segment), T = L_kk^T (unit upper).
// The 32 x ns strip lives in LDS for the whole solve: HBM traffic = one read + one write of the panel.
// Strip height RSv = 64 (nsp <= 128) or 32 (nsp <= 256): either way the workgroup needs <= 82 KB of LDS, so a panel
// workgroup fits beside ONE 128x128 Schur workgroup on a CU -- the high-priority look-ahead stream can then take any
// slot a finishing Schur workgroup frees instead of waiting for a whole idle CU.
// LDS images are split in 16-wide groups ([group][k][16]): a 16x4 MFMA fragment read touches 4 k-rows x 16
// consecutive doubles = all 64 banks once.  Xs = strip (RSv x nsp), Tb = double-buffered 32x32 operand block.
// (32-row strips -- 80 KB of LDS, two workgroups per CU or one beside a Schur workgroup -- measured ~2 % slower end to end
// than 64-row strips -- 144 KB, one per CU -- on 100^3; the host picks, Handle::Env::trsm_rs32)
constexpr int TB_SZ = DB * 48;   // doubles per operand buffer: [32][48] (c-fastest chunks) or [32][34] (k-fastest chunks)
static inline size_t trsm_lds_bytes(int rs, int nsp) { return sizeof(double) * ((size_t) rs * nsp + 2 * TB_SZ); }

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
    const int lda = T.sn_nsupr[k];            // L slot (MODE 0 strips)
    const int ldd = T.sn_dlda[k];             // diagonal block: top of the L slot on its owner row, scratch image elsewhere
    const int nblk = nsp / DB;
    double *A = T.val + T.sn_lval[k];
    const double *Dg = T.val + T.sn_dptr[k];
    double *Uv = T.val + T.sn_uval[k];
    constexpr bool TU = (MODE == 0 || MODE == 3);     // T = U_kk (k-fastest chunks); otherwise T = L_kk^T (c-fastest chunks)
    const double *dinv = T.dinv + T.sn_dinv[k] + (TU ? 0 : (size_t) nblk * DB * DB);
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
            if (cr < T.sn_ncolu[k]) { ld = T.ucol_ld[T.sn_ucol[k] + cr]; cp = T.ucol_cp[T.sn_ucol[k] + cr]; }
            s_cp[tid] = cp; s_ld[tid] = ld;
        }
        __syncthreads();
    };

    if (MODE >= 2) {   // strip of the identity
        const int row0 = strip * RSv;
        for (int idx = tid; idx < RSv * nsp; idx += NT) {
            const int r = idx % RSv, c = idx / RSv;
            Xs[((r >> 4) * nsp + c) * 16 + (r & 15)] = (row0 + r == c && c < ns) ? 1.0 : 0.0;
        }
    } else if (MODE == 0) {
        const int row0 = T.sn_ldiag[k] + strip * RSv;
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
                const int kg = kc + (TU ? e0 : e1 + ES * q), cg = jb + (TU ? e1 + ES * q : e0);
                pv[q] = (kg < ns && cg < ns) ? (TU ? Dg[kg + (size_t) cg * ldd] : Dg[cg + (size_t) kg * ldd]) : 0.0;
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
            if (t * DB < jb && !TU) tb[(e1 + ES * q) * 48 + e0] = pv[q];   // (kk = e1 + ES q, cc = e0)
            else tb[(e1 + ES * q) * 34 + e0] = pv[q];                            // (kk = e0, cc = e1 + ES q)
        }
    };
    auto advance = [&](int &jb, int &t) { if (++t > jb / DB) { jb += DB; t = 0; } };

    d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
    const double *xa = Xs + ((size_t) wave * nsp + (lane >> 4)) * 16 + (lane & 15);
    auto compute = [&](int jb, int t, int buf) {
        // fragment element (kk = k4 + lane>>4, cc = half*16 + lane&15)
        const bool cfast = !TU && (t < jb / DB);
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

    if (MODE >= 2) {
        double *out = T.inv + T.sn_inv[k] + (MODE == 3 ? (size_t) ns * ns : 0);
        const int row0 = strip * RSv;
        for (int idx = tid; idx < RSv * ns; idx += NT) {
            const int r = idx % RSv, c = idx / RSv;
            // MODE 3: Uinv(row, c) at [row + c ns]; MODE 2: X' = (L^T)^-1 = Linv^T -> Linv(c, row) at [c + row ns]
            if (row0 + r < ns) out[MODE == 3 ? (row0 + r) + (size_t) c * ns : c + (size_t) (row0 + r) * ns] = Xs[((r >> 4) * nsp + c) * 16 + (r & 15)];
        }
    } else if (MODE == 0) {
        const int row0 = T.sn_ldiag[k] + strip * RSv;
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
                                                        int nn, int nl, const int2 *__restrict__ units)
{
    extern __shared__ double sm[];
    if (units) {   // explicit (supernode, strip) list [L strips | U chunks]: the urgent / remaining parts of a split panel solve (run_factor_sched)
        const int2 u = units[blockIdx.x];
        if ((int) blockIdx.x < nl) panel_trsm_body<0, RSv>(T, u.x, u.y, sm);
        else panel_trsm_body<1, RSv>(T, u.x, u.y, sm);
        return;
    }
    if ((int) blockIdx.x < nl) {
        const int ni = find_node_wave(lprefix, nn, blockIdx.x);
        panel_trsm_body<0, RSv>(T, nodes[ni], blockIdx.x - lprefix[ni], sm);
    } else {
        const int id = blockIdx.x - nl;
        const int ni = find_node_wave(uprefix, nn, id);
        panel_trsm_body<1, RSv>(T, nodes[ni], id - uprefix[ni], sm);
    }
}

// ---- panel solves as GEMMs with the full inverse (1 x 1 layers) ------------------------------------------------------------
// L(:,k) <- L(:,k) Uinv  and  U(k,:) <- Linv U(k,:)  (the same dLPanelTrSolve / dTrs2_GatherTrsmScatter results) without the
// dependent chain of a blocked substitution: every 32-column block of the result is an independent product with the
// triangular inverse.  One wave owns 16 panel rows (MODE 0) or 16 skyline columns (MODE 1) and keeps them in registers as
// MFMA operand fragments for the whole solve (64 doubles per lane for ns = 256), so the update is in place without any
// hazard, there is no LDS strip (many waves per SIMD instead of one workgroup per CU) and no barrier.  Operand roles as in
// k_schur: D = Tinv^T-fragment x strip-fragment, so that the 16 fast lanes of every accumulator register run along panel rows
// (128-byte runs of an L column / contiguous pieces of a skyline segment).
constexpr int PGK = 64;                        // K chunk of the inverse staged in LDS
constexpr int PG_LDS = PGK * 48;               // doubles: [64][48] (c-fastest, MODE 1) or [32][66] (k-fastest, MODE 0)
template <int MODE, int NQ>   // NQ = fragment registers per lane = 64 (supernodes up to 256 columns), 32 (<= 128) or 16 (<= 64: the leaf levels,
                              // where the register budget decides how many of the tens of thousands of small panels are in flight)
__device__ __forceinline__ void panel_gemm_wg(const DevTables &T, int k, int unit64, double *Ts)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int unit = unit64 * 4 + wave;
    const int klst = T.xsup[k + 1], ns = klst - T.xsup[k];
    const int nblk = (ns + DB - 1) / DB;
    const int li = lane & 15, lk = lane >> 4;
    double a[NQ];
    // ---- load the wave's 16 rows: a[q] = X(row li, column 4 q + lk) ----
    const int lda = T.sn_nsupr[k];
    double *A = T.val + T.sn_lval[k];
    double *Uv = T.val + T.sn_uval[k];
    int row = 0, cp = 0, ld = ns;
    bool valid;
    if (MODE == 0) {
        row = T.sn_ldiag[k] + unit * 16 + li;
        valid = row < lda;
    } else {
        const int cr = unit * 16 + li;
        valid = cr < T.sn_ncolu[k];
        if (valid) { ld = T.ucol_ld[T.sn_ucol[k] + cr]; cp = T.ucol_cp[T.sn_ucol[k] + cr]; }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = 4 * q + lk;
        double v = 0.0;
        if (q < nblk * 8) {
            if (MODE == 0) { if (valid && c < ns) v = A[row + (size_t) c * lda]; }
            else { if (valid && c >= ld && c < ns) v = Uv[cp + (c - ld)]; }
        }
        a[q] = v;
    }
    // Tinv(kk, n): MODE 0 -> Uinv(kk, n) at Ui[kk + n ns] (kk fastest); MODE 1 -> (Linv^T)(kk, n) = Linv(n, kk) at Li[n + kk ns] (n fastest)
    const double *Ti = T.inv + T.sn_inv[k] + (MODE == 0 ? (size_t) ns * ns : 0);
    for (int jb = 0; jb < nblk; ++jb) {
        d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kc = 0; kc < NQ / 16; ++kc) {    // K chunks of 64 strip columns: [0, 32 (jb + 1)) in all (Tinv is upper triangular)
            if (kc * 2 <= jb) {
                __syncthreads();                   // the previous chunk's fragment reads are done
                const int k0 = kc * PGK;
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < PGK * DB / 256; ++e) {
                        const int idx = tid + 256 * e, kk = idx & (PGK - 1), np = idx >> 6;
                        const int kg = k0 + kk, n = jb * DB + np;
                        Ts[np * (PGK + 2) + kk] = (kg < ns && n < ns) ? Ti[kg + (size_t) n * ns] : 0.0;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < PGK * DB / 256; ++e) {
                        const int idx = tid + 256 * e, np = idx & 31, kk = idx >> 5;
                        const int kg = k0 + kk, n = jb * DB + np;
                        Ts[kk * 48 + np] = (kg < ns && n < ns) ? Ti[n + (size_t) kg * ns] : 0.0;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    const int kl = 4 * qq + lk;
                    const double t0 = MODE == 0 ? Ts[li * (PGK + 2) + kl] : Ts[kl * 48 + li];
                    const double t1 = MODE == 0 ? Ts[(16 + li) * (PGK + 2) + kl] : Ts[kl * 48 + 16 + li];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(t0, a[16 * kc + qq], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(t1, a[16 * kc + qq], acc1, 0, 0, 0);
                }
            }
        }
        // D[(lk + 4 r)][li] = X_new(row li, column jb*32 + 16 h + lk + 4 r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c0 = jb * DB + lk + 4 * r, c1 = c0 + 16;
            if (MODE == 0) {
                if (valid && c0 < ns) A[row + (size_t) c0 * lda] = acc0[r];
                if (valid && c1 < ns) A[row + (size_t) c1 * lda] = acc1[r];
            } else {
                if (valid && c0 >= ld && c0 < ns) Uv[cp + (c0 - ld)] = acc0[r];
                if (valid && c1 >= ld && c1 < ns) Uv[cp + (c1 - ld)] = acc1[r];
            }
        }
    }
}

// L units (workgroups [0, nl)) and U units ([nl, nl + nu)) of one level in ONE launch; workgroup = 4 waves = 64 rows / columns
// (the same 64-high work units as k_panel_trsm<64>)
template <int NQ>
__global__ __launch_bounds__(256) void k_panel_gemm(DevTables T, const int *__restrict__ nodes, const int *__restrict__ lprefix,
                                                    const int *__restrict__ uprefix, int nn, int nl, const int2 *__restrict__ units)
{
    __builtin_amdgcn_s_setprio(3);   // panel chain: its waves go first when they share a SIMD with Schur tiles
    __shared__ double Ts[PG_LDS];
    if (units) {   // explicit (supernode, 64-row / 64-column unit) list [L units | U units]: one part of a split panel solve
        const int2 u = units[blockIdx.x];
        if ((int) blockIdx.x < nl) panel_gemm_wg<0, NQ>(T, u.x, u.y, Ts);
        else panel_gemm_wg<1, NQ>(T, u.x, u.y, Ts);
        return;
    }
    if ((int) blockIdx.x < nl) {
        const int ni = find_node_wave(lprefix, nn, blockIdx.x);
        panel_gemm_wg<0, NQ>(T, nodes[ni], blockIdx.x - lprefix[ni], Ts);
    } else {
        const int id = blockIdx.x - nl;
        const int ni = find_node_wave(uprefix, nn, id);
        panel_gemm_wg<1, NQ>(T, nodes[ni], id - uprefix[ni], Ts);
    }
}

// Linv / Uinv of the owned diagonal blocks of a node list: work unit = (supernode, typ, 16-row strip of the identity) = one
// wave with 56 KB of LDS, so that the workgroups fit beside a resident Schur workgroup (a 64-row strip needs a whole CU and
// would wait for the Schur kernel to drain)
constexpr int FIS = 16;
__global__ __launch_bounds__(FIS * 4) void k_full_inv(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix, int nn)
{
    __builtin_amdgcn_s_setprio(3);   // panel chain: its waves go first when they share a SIMD with Schur tiles
    extern __shared__ double sm[];
    const int ni = find_node_wave(prefix, nn, blockIdx.x);
    const int k = nodes[ni];
    const int ns = T.xsup[k + 1] - T.xsup[k];
    const int per = (ns + FIS - 1) / FIS;
    const int u = blockIdx.x - prefix[ni];
    if (u < per) panel_trsm_body<2, FIS>(T, k, u, sm);
    else panel_trsm_body<3, FIS>(T, k, u - per, sm);
}

// Linv / Uinv of diagonal blocks of AT MOST 64 COLUMNS -- the tens of thousands of leaf supernodes: k_full_inv's identity strips are 8 one-wave
// workgroups of blocked substitution per block (0.7 ms for level 0 of the 100^3 tree, on the panel chain with nothing else to run).  With at most two
// 32 x 32 diagonal sub-blocks the inverse of the triangle B (U_kk, or L_kk^T) is  [X00, -X00 B01 X11; 0, X11]  with X00, X11 the 32 x 32 inverses the
// diagonal kernels already left in T.dinv: ONE wave per (block, triangle), two 32 x 32 x 32 products on fp64 MFMA whose operands come straight from
// memory in fragment layout -- the accumulator layout of T = B01 X11 IS the B-operand layout of X00 T -- no LDS, no barrier.
__global__ __launch_bounds__(256) void k_full_inv64(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix, int nn)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ni = w >> 1, typ = w & 1;            // typ 0: U_kk -> Uinv; 1: L_kk^T -> (Linv)^T
    if (ni >= nn || prefix[ni + 1] == prefix[ni]) return;
    const int k = nodes[ni];
    const int ns = T.xsup[k + 1] - T.xsup[k], nblk = (ns + DB - 1) / DB, lda = T.sn_dlda[k];
    const double *A = T.val + T.sn_dptr[k];
    const double *D0 = T.dinv + T.sn_dinv[k] + (size_t) (typ * nblk) * DB * DB, *D1 = D0 + DB * DB;     // X(i, c) at [c * 32 + i], identity-padded past ns
    double *out = T.inv + T.sn_inv[k] + (typ == 0 ? (size_t) ns * ns : 0);
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    // element (r, c) of the upper-triangular inverse X: Uinv(r, c) at [r + c ns] (typ 0), Linv(c, r) at [c + r ns] (typ 1)
    auto put = [&](int r, int c, double v) { if (r < ns && c < ns) out[typ == 0 ? r + (size_t) c * ns : c + (size_t) r * ns] = v; };
    for (int e = lane; e < DB * DB; e += 64) {
        const int i = e & 31, c = e >> 5;
        put(i, c, D0[c * DB + i]);
        if (nblk == 2) { put(DB + i, DB + c, D1[c * DB + i]); put(DB + i, c, 0.0); }
    }
    if (nblk < 2) return;
    // T = B01 X11:  B01(i, kk) = B(i, 32 + kk)
    d4 t[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = 16 * it + li, kk = 4 * q + lk;
                double a = 0.0;
                if (DB + kk < ns) a = (typ == 0) ? A[i + (size_t) (DB + kk) * lda] : A[(DB + kk) + (size_t) i * lda];
                const double b = D1[(16 * jt + li) * DB + kk];                 // X11(kk, 16 jt + li)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            t[it][jt] = acc;          // lane: T(16 it + lk + 4 r, 16 jt + li)
        }
    // X01 = -X00 T: the B operand of k-step 4 q' + lk of row tile kt is t[kt][jt][q']
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double a = D0[(16 * kt + 4 * q + lk) * DB + 16 * it + li];      // X00(16 it + li, 16 kt + 4 q + lk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, t[kt][jt][q], acc, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) put(16 * it + lk + 4 * r, DB + 16 * jt + li, -acc[r]);
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

// ---- Schur complement update: fused gather -> MFMA fp64 GEMM -> scatter ---------------------------
// One workgroup (4 waves) per 64x64 tile of one (L block, U block) pair of one supernode of the level.
// MFMA v_mfma_f64_16x16x4_f64 computes D[i][j] = sum_k A[i][k] B[k][j] with lane l holding
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[(l>>4)+4r][l&15].  We feed A := U^T (i = tile column) and
// B := L^T (j = tile row) so that the 16 fast lanes of every accumulator register run along tile ROWS:
// the scatter then writes 128-byte runs of a destination column (column-major L panel / U skyline).

// Two tile configurations: 128x128 (wide supernodes, big block pairs: 4x4 MFMA blocks per wave) and 64x64
// (everything else).  Software pipeline: the next K chunk is fetched from HBM/L2 into registers while the
// MFMAs of the current chunk run out of the other LDS buffer (one barrier per chunk).
// the MFMAs of one K chunk for a wave that owns RA x CA 16 x 16 blocks
typedef const volatile double __attribute__((address_space(3))) *lds_vdouble_t;
// USW: the U stage in the swizzled column-major form (see k_schur): `pa` is the lane's index for k4 = 0, column block 0
template <int RA, int CA, int LDL, int LDU, bool USW>
__device__ __forceinline__ void schur_chunk(const double *Lb, const double *Ub, int rm0, int cn0, int lane, d4 (&acc)[CA][RA], int pa)
{
#pragma unroll
    for (int k4 = 0; k4 < KC; k4 += 4) {
        const int kr = k4 + (lane >> 4);
        double a[CA], b[RA];
#pragma unroll
        for (int c = 0; c < CA; ++c) a[c] = USW ? ((lds_vdouble_t) Ub)[(pa ^ k4) + 256 * c] : Ub[kr * LDU + cn0 + 16 * c + (lane & 15)];
        // (volatile: single ds_read_b64 -- two lane groups of 32 over 64 banks, 2 LDS cycles -- instead of the merged ds_read2[st64]_b64, whose 16-lane groups over 32
        //  banks take 8 cycles and see the swizzled columns 2-way conflicted)
#pragma unroll
        for (int r = 0; r < RA; ++r) b[r] = USW ? ((lds_vdouble_t) Lb)[kr * LDL + rm0 + 16 * r + (lane & 15)] : Lb[kr * LDL + rm0 + 16 * r + (lane & 15)];
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int r = 0; r < RA; ++r) acc[c][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], b[r], acc[c][r], 0, 0, 0);
    }
}

// Instrumentation switches of the epilogue decomposition (scripts/ab_epilogue.sh builds the variants; the product build never sets them):
//   SLUAMD_EXP_EPI    2 = the product's epilogue (one fp64 atomic per updated element), 1 = plain load-subtract-store (racy between concurrent
//                     sources of one destination: timing only), 0 = no scatter at all (the accumulators are kept alive by an impossible branch)
//   SLUAMD_EXP_NOLOAD 1 = the K loop's operands are constants: no global loads between the prologue and the epilogue (LDS stores, reads, barriers, MFMAs stay)
#ifndef SLUAMD_EXP_EPI
#define SLUAMD_EXP_EPI 2
#endif
#ifndef SLUAMD_EXP_NOLOAD
#define SLUAMD_EXP_NOLOAD 0
#endif
//   SLUAMD_EXP_VALU   N > 0: N extra dependent-free integer VALU instructions per K chunk and wave beside the loader (is ordinary VALU work hidden behind the other
//                     waves' fp64 MFMAs, or does it add to them?); SLUAMD_EXP_VALU_MID: the same instructions spread between the MFMA groups of the chunk
#ifndef SLUAMD_EXP_VALU
#define SLUAMD_EXP_VALU 0
#endif
#ifndef SLUAMD_SCHUR_FETCH2
#define SLUAMD_SCHUR_FETCH2 1     // 0: the loader of rounds 1-5 (kept for the same-box A/B)
#endif
// a pointer every lane holds alike, moved to SGPRs so that loads take the (SGPR base + 32-bit VGPR offset) form
#ifndef SLUAMD_SCHUR_FETCH3
#define SLUAMD_SCHUR_FETCH3 1     // the loader in 16-byte loads (two tile rows of one panel column / two k of one U column per lane): half the load instructions,
#endif                            // map reads and predicates of SLUAMD_SCHUR_FETCH2; 0: that loader
// raw buffer loads: address = 48-bit base of a wave-uniform resource descriptor (4 SGPRs) + 32-bit per-lane byte offset (VGPR) + 32-bit wave-uniform byte offset
// (SGPR): no 64-bit vector arithmetic in front of the load
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ i32x4 llvm_amdgcn_raw_buffer_load_i32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ __forceinline__ i32x4 buffer_rsrc(const void *p)     // p wave-uniform; unbounded raw buffer (stride 0)
{
    const uint64_t v = (uint64_t) p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int) (uint32_t) v);
    r.y = __builtin_amdgcn_readfirstlane((int) ((uint32_t) (v >> 32) & 0xFFFFu));
    r.z = (int) 0xFFFFFFFFu;
    r.w = 0x00020000;
    return r;
}
// ... and straight into LDS (LDS-DMA): the wave's 64 x 16 bytes land at the wave-uniform LDS address (M0) + lane * 16, no VGPR on the way.  As an asm statement:
// behind the intrinsic (llvm.amdgcn.raw.buffer.load.lds) hipcc waits vmcnt(0) in front of the NEXT LDS read -- the prefetch of the next chunk would be
// drained before the MFMAs of the current one start.  The statement's loads are outside the compiler's vmcnt bookkeeping: lds_dma_wait() before the barrier that
// publishes the stage (the counter is in order: untracked loads can only make the compiler's own waits longer, never too short).
__device__ __forceinline__ void lds_dma_16(i32x4 rsrc, uint32_t lds_byte, int voffset, int soffset)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voffset), "s"(rsrc), "s"(soffset), "s"(lds_byte) : "memory");
}
#ifdef SLUAMD_EXP_NODMAWAIT      // (instrumented build: nobody waits for the LDS-DMA loads -- wrong factors, timing only)
__device__ __forceinline__ void lds_dma_wait() { asm volatile("" ::: "memory"); }
#else
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) { return (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const void *) p; }
#ifndef SLUAMD_SCHUR_TOUCH
#define SLUAMD_SCHUR_TOUCH 0      // 1: one chunk before the last, a tile touches its destination lines (rounds 2-5; at the 128-VGPR cap of the round-6 kernel the two touched
                                  // values are spilled, i.e. waited for at once: 265.6 ms); 2: the touch as loads into an LDS sink (262.2 ms); 0: none (259.6 ms, profiles/r06_ab_dma_loader.txt)
#endif
#ifndef SLUAMD_SCHUR_TIGHT
#define SLUAMD_SCHUR_TIGHT 1      // the full chunks of a clean source of the LDS-DMA configurations run in a loop of their own (0: inside the general pipeline; same-box A/B)
#endif
#ifndef SLUAMD_SCHUR_DMA
#define SLUAMD_SCHUR_DMA 1        // clean sources of the 128-row tile configurations load their chunks straight into the LDS stage; 0: through registers (same-box A/B)
#endif
#ifndef SLUAMD_SCHUR_CLEAN
#define SLUAMD_SCHUR_CLEAN 1      // the predicate-free loader for clean sources (see `clean` in k_schur); 0: fetch3 / stash3 everywhere (same-box A/B)
#endif
typedef double d2 __attribute__((ext_vector_type(2)));
typedef const d2 __attribute__((address_space(1), aligned(8))) *gd2_t;      // (the pairs are 8-byte aligned only: panel leading dimensions and segment offsets are arbitrary)
typedef const char __attribute__((address_space(1))) *gbytes_t;     // (global address space kept through the integer round trip: global_load, not flat_load)
typedef const double __attribute__((address_space(1))) *gdouble_t;
__device__ __forceinline__ gbytes_t uniform_ptr(const double *p)
{
    const uint64_t v = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) v), hi = __builtin_amdgcn_readfirstlane((uint32_t) (v >> 32));
    return (gbytes_t) (((uint64_t) hi << 32) | lo);
}
#ifndef SCHUR64_WGS
#define SCHUR64_WGS 5   // workgroups per CU the 64 x 64 tile configuration is built for (<= 4: two LDS stages, as the 128 x 128 one); the complex
                        // variant fits six without spills (76 VGPRs) and is built for six: zgrid2d 1000 24.8 -> 23.9 ms; the double one spills at 80: +1.6 ms
#endif
// Z = true: the complex16 update through its REAL embedding -- C (m x n complex) -= L U is the real GEMM
//   Creal (2m x n, rows = re / im interleaved: the native complex column-major layout) -= Lexp (2m x 2K) Ureal (2K x n),
//   Lexp[2i + a][2p + b] = (a == b) ? Re L(i,p) : (a ? Im L(i,p) : -Im L(i,p)),   Ureal[2p + b][j] = (b ? Im : Re) U(p,j)
// so that U and the destination are read / written where they lie and only the L loader differs (it picks the part a ^ b with
// a per-thread constant sign): the same tile machinery, every MFMA a useful one (2 (2m)(2K) n = 8 m K n flop).  A tile of
// TMv real rows is TMv / 2 rows of the complex panel.
// SK (split K): the launch has gridDim.y workgroups per tile, each runs its share of every source's K chunks and scatters it (the scatter is fp64 atomics
// anyway).  For the few tiles that sit on the panel chain -- the diagonal-block destinations of the next level, 4 tiles of K = 256 ... 512 on an otherwise
// waiting device, where a lone workgroup is latency-bound at ~3.5 us per chunk -- the chain link gets the tile in a quarter of the time.  A separate
// instantiation: the bulk instantiation's K loop (at the 128-VGPR cap) stays exactly what it was.
template <int TMv, int TNv, int NW, bool Z = false, int MM = 0, bool SK = false>   // MM 0: the tiles chase the tables; 1: and write their per-tile records instead of updating (plan-time build pass); 2: they read the records
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : (TMv == 128 ? 2 : (Z ? 6 : SCHUR64_WGS)))) void k_schur(DevTables T, const int *__restrict__ nodes,
                                                                    const int *__restrict__ prefix, int nn, int id_base,
                                                                    int ntiles, int *__restrict__ info,
                                                                    const int4 *__restrict__ ulist, int prio,
                                                                    const int *__restrict__ tmaps, const int *__restrict__ xoff)
{
    constexpr int LDL = TMv + 16;   // == 16 mod 32 doubles: conflict-free ds_read_b64 fragment reads
    constexpr int LDU = TNv + 17;   // odd: the k-major U stash (16 lanes x stride LDU) spreads over all banks too
    constexpr int NT = NW * 64;                         // threads per workgroup
    constexpr int WR = (NW == 8) ? 4 : 2, WC = 2;       // wave grid (rows x cols): 4 waves = 2x2, 8 waves = 4x2
    constexpr int NBR = TMv / (16 * WR), NBC = TNv / (16 * WC);   // 16x16 MFMA blocks per wave (rows, cols)
    constexpr int LQ = TMv * KC / NT, UQ = TNv * KC / NT;         // prefetch registers per thread
    constexpr int LKS = NT / TMv;                       // k stride of the L loader
    constexpr int UJS = NT / 16;                        // column stride of the U loader
    constexpr int ZS = Z ? 2 : 1;                       // doubles per value: arena offsets and tile rows of the tables are in values
    static_assert(!Z || (LKS % 2 == 0 && KC % 2 == 0), "complex L loader: the parity of a thread's k must be constant");
    // 128 x 128 tiles: two LDS stages (one barrier per chunk).  64 x 64 tiles -- the bottom of the tree, short K loops, the tile's
    // life is dependent index loads -- trade the second stage for occupancy: 24 KB instead of 44 KB per workgroup
    constexpr int NBUF = (TMv == 64 && SCHUR64_WGS > 4) ? 1 : 2;
    __shared__ double Ls[NBUF][KC * LDL];
    // U stage.  USW (the 16-byte loader, real arithmetic): column-major in 16-byte slots -- column c, k pair kp at doubles c * 16 + 2 * (kp ^ ((c >> 1) & 7)) -- so that a
    // thread's loaded pair (two k of one column) is ONE 16-byte store, a wave's 64 pairs (8 columns x 8 pairs) are 1 KiB in lane order (what a load straight into
    // LDS writes), and the MFMA fragment reads (16 lanes = 16 columns of one k) fall on 16 different slots of a 256-byte window (the XOR); otherwise k-major, padded
    constexpr bool USW = SLUAMD_SCHUR_FETCH3 && !Z;
    __shared__ __attribute__((aligned(16))) double Us[NBUF][USW ? KC * TNv : KC * LDU];
    auto uidx = [](int k, int c) { return USW ? c * 16 + ((((k >> 1) ^ ((c >> 1) & 7)) << 1) | (k & 1)) : k * LDU + c; };
    __shared__ int s_ind[256 + 8];
    __shared__ int s_rowmap[TMv];
    __shared__ int s_colmap[TNv];
    __shared__ int s_cptr[TNv];   // value offset of tile column j inside U(k,:)
    __shared__ int s_lead[TNv];   // ns - seg (leading zeros) of tile column j
    __shared__ int s_cptr2[TNv];  // the same two for the fused predecessor supernode (K-fused chain update)
    __shared__ int s_lead2[TNv];
    __shared__ int s_jj[TNv];     // column id inside supernode jb
    __shared__ int s_dinfo[1];
    __shared__ int s_sink[(SLUAMD_SCHUR_TOUCH == 2 && TMv == 128) ? NW * 64 : 1];     // destination of the touch loads of the LDS-DMA configurations
    constexpr int HDR = 64;                          // header ints of a tile record: [0,16) the tile itself, [16] number of K-fused predecessors, [17 + 11 j, 28 + 11 j) predecessor j (nearest first)
    __shared__ int s_hdr[HDR];
    constexpr int REC = HDR + TMv + 3 * TNv;         // ints per tile record (MM 1 / 2)

    const int tid = threadIdx.x;
    // XCD-aware mapping: workgroup b runs on XCD b%8; give every XCD a contiguous range of tiles so that the
    // row tile (L rows) shared by consecutive tiles stays in ONE XCD's L2
    // (xoff, balanced bulk launches: the eight ranges have equal modelled COST instead of equal counts -- boundaries from the planner, balance_bulk)
    int bid;
    if (xoff) {
        const int x = blockIdx.x & 7;
        const int b0 = xoff[x], b1 = xoff[x + 1];
        bid = b0 + (int) (blockIdx.x >> 3);
        if (bid >= b1) return;
        bid += id_base;
    } else {
        const int chunk = (ntiles + 7) >> 3;
        bid = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= chunk || bid >= ntiles) return;
        bid += id_base;
    }
    // Per-tile state of the K loop and the scatter.  mmode 2 (every schedule but the deterministic one, after the plan's build pass):
    // ONE contiguous record per tile -- header + the four maps -- written once per plan by this kernel itself (mmode 1), instead
    // of the chain of dependent table lookups below (tile list -> tile descriptors -> supernode tables -> column maps ->
    // destination block -> its row list -> inverse row map): the record's address depends on the tile id alone.
    int k, nr, nc, ns, lda, stc, Rw, kbeg_own;
    bool has_dst;
    const double *Lp, *Uv;
    double *dst;
    if (prio) __builtin_amdgcn_s_setprio(2);   // urgent tiles sit on the panel chain
    if (MM == 2) {
        const int *rec = tmaps + (size_t) (bid - id_base) * REC;
        for (int t = tid; t < REC; t += NT) {
            const int v = rec[t];
            if (t < HDR) s_hdr[t] = v;
            else if (t < HDR + TMv) s_rowmap[t - HDR] = v;
            else if (t < HDR + TMv + TNv) s_colmap[t - HDR - TMv] = v;
            else if (t < HDR + TMv + 2 * TNv) s_cptr[t - HDR - TMv - TNv] = v;
            else s_lead[t - HDR - TMv - 2 * TNv] = v;
        }
        __syncthreads();
        k = s_hdr[0];
        nr = __builtin_amdgcn_readfirstlane(s_hdr[1]); nc = __builtin_amdgcn_readfirstlane(s_hdr[2]);
        ns = s_hdr[3]; lda = s_hdr[4]; kbeg_own = s_hdr[5];
        Lp = T.val + (((int64_t) s_hdr[7] << 32) | (uint32_t) s_hdr[6]);
        Uv = T.val + (((int64_t) s_hdr[9] << 32) | (uint32_t) s_hdr[8]);
        const int64_t dbase = ((int64_t) s_hdr[11] << 32) | (uint32_t) s_hdr[10];
        has_dst = dbase >= 0;
        dst = T.val + (has_dst ? dbase : 0);
        stc = s_hdr[12]; Rw = s_hdr[13];
        if (!has_dst && tid == 0) atomicAdd(&info[2], 1);
    } else {
        // `ulist` != null (every schedule but the deterministic one): tile list entry (supernode, absolute row tile, absolute column
        // tile, destination block) + per-tile descriptors -- the prologue is three dependent (scalar) loads deep.  Otherwise the full
        // tile grid of the supernodes `nodes` (one supernode per launch: tiles of one k hit distinct destinations).
        int ib, jb, dblk = -2;
        int4 R, C;
        const int *lsub;            // global row ids of the tile rows
        if (ulist) {
            const int4 u = ulist[bid];
            k = u.x; dblk = u.w;
            R = T.rtile[u.y]; C = T.ctile[u.z];
            const int2 ri = T.rt_info[u.y];
            const int4 ci = T.ct_info[u.z];
            ib = ri.x; jb = ci.x; stc = ci.z;
            // merged row tile (destination -3): rows of SEVERAL L blocks, all with gid >= jb -> their global ids come from the slot's flat row map
            lsub = (dblk == -3) ? T.lrow + T.sn_lrow[k] + R.w : T.lidx + ri.y;
        } else {
            const int ni = find_node(prefix, nn, bid);
            k = nodes[ni];
            const int local = bid - prefix[ni];
            const int nct = T.sn_nct[k];
            const int rt = local / nct, ct = local - rt * nct;
            R = T.rtile[T.sn_rt_off[k] + rt];
            C = T.ctile[T.sn_ct_off[k] + ct];
            const int lb = T.sn_lb_off[k] + R.x, ub = T.sn_ub_off[k] + C.x;
            ib = T.lb_gid[lb]; jb = T.ub_gid[ub];
            lsub = T.lidx + T.sn_lidx[k] + T.lb_lptr[lb] + R.y;
            stc = T.ub_stcol[ub] + C.y;
        }
        nr = __builtin_amdgcn_readfirstlane(R.z) * ZS; nc = __builtin_amdgcn_readfirstlane(C.z);   // workgroup-uniform
        ns = T.xsup[k + 1] - T.xsup[k];
        lda = T.sn_nsupr[k];
        Rw = R.w;
        kbeg_own = ZS * ((ns - T.sn_ldu[k]) & ~3);       // U is zero above its tallest segment: skip those k
        Lp = T.val + ZS * (T.sn_lval[k] + R.w);            // first tile row, column 0 of the panel
        Uv = T.val + ZS * T.sn_uval[k];

        {   // per tile column: value offset inside U(k,:), leading zeros, column id inside supernode jb -- flat per-non-empty-column maps
            const int64_t cb = T.sn_ucol[k] + stc;
            const int fstj = T.xsup[jb];
            for (int t = tid; t < TNv; t += NT) {
                int cp = 0, lead = ns, jj = 0;
                if (t < nc) { cp = T.ucol_cp[cb + t]; lead = T.ucol_ld[cb + t]; jj = T.ucol_gc[cb + t] - (dblk == -4 ? 0 : fstj); }    // merged column tile: the GLOBAL column id
                s_cptr[t] = cp; s_lead[t] = lead; s_jj[t] = jj;
            }
        }
        // ---- destination lookup (dscatter_l :138-147 / scatter_u :593-602 linear searches): host-resolved in list mode ----
        if (dblk == -2 && (tid >> 6) == NW - 1) {
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
            if (ln == 0) s_dinfo[0] = pos < 0 ? -1 : (ldest ? o + T.lbs_idx[o + pos] : o + pos);
        }
        const bool merged = dblk == -3 || dblk == -4, mergedU = dblk == -4;      // (complex16: tile rows are REAL rows of the embedding, two per panel row)
        if (merged && tid == 0) s_dinfo[0] = 0;
        __syncthreads();
        if (mergedU) {
            // merged COLUMN tile: rows of ONE L block (ib), columns of several U blocks, all with gid > ib -> every element lands in U row ib.  Row map =
            // global row id (as for any U destination); column map = (value offset - first nonzero row) of the column's segment in row ib, found by
            // the rank of the global column in that row's ascending list of non-empty columns
            const int64_t cbi = T.sn_ucol[ib];
            const int pn = T.sn_ncolu[ib], fib = T.xsup[ib];
            for (int t = tid; t < TMv; t += NT) s_rowmap[t] = (t < nr) ? ZS * lsub[t / ZS] + t % ZS : 0;
            for (int t = tid; t < TNv; t += NT) {
                int cm = 0;
                if (t < nc) {
                    const int gc = s_jj[t];
                    int lo = 0, hi = pn;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (T.ucol_gc[cbi + mid] < gc) lo = mid + 1; else hi = mid; }
                    if (lo < pn && T.ucol_gc[cbi + lo] == gc) cm = T.ucol_cp[cbi + lo] - (fib + T.ucol_ld[cbi + lo]); else s_dinfo[0] = -1;
                }
                s_colmap[t] = ZS * cm;
            }
            __syncthreads();
        } else
        if (merged) {
            // every row of the tile lands in panel jb (block rows ib >= jb): its position there = its rank in the panel's ascending row list
            // (the planner merges only into panels whose rows ascend over the whole slot); a row the panel lacks voids the tile like a missing block
            const int *prow = T.lrow + T.sn_lrow[jb];
            const int pn = T.sn_nsupr[jb];
            for (int t = tid; t < TMv; t += NT) {
                int v = 0;
                if (t < nr) {
                    const int gr = lsub[t / ZS];
                    int lo = 0, hi = pn;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (prow[mid] < gr) lo = mid + 1; else hi = mid; }
                    if (lo < pn && prow[lo] == gr) v = ZS * lo + t % ZS; else s_dinfo[0] = -1;
                }
                s_rowmap[t] = v;
            }
            for (int t = tid; t < TNv; t += NT) s_colmap[t] = ZS * s_jj[t] * pn;
            __syncthreads();
        }
        if (dblk == -2) dblk = s_dinfo[0];
        has_dst = merged ? s_dinfo[0] >= 0 : dblk >= 0;
        if (MM != 1 && !has_dst && tid == 0) atomicAdd(&info[2], 1);     // (the plan-time build pass only writes records: the factorisation's pass counts)
        // destination block: (row offset inside the panel, its row ids, its row count) of L(ib, jb), or the index position of U(ib, jb)
        int di0 = 0, di1 = 0, di2 = 0;
        int64_t dbase = mergedU ? ZS * T.sn_uval[ib] : (merged ? ZS * T.sn_lval[jb] : 0);
        if (has_dst && !merged) {
            if (ib >= jb) { di0 = T.lb_rowoff[dblk]; di1 = T.lb_lptr[dblk]; di2 = T.lb_nbrow[dblk]; dbase = ZS * T.sn_lval[jb]; }
            else { di0 = T.ub_iukp[dblk]; dbase = ZS * T.sn_uval[ib]; }
        }
        // ---- destination maps (tile row / column -> offset inside the destination panel / U row), before the K loop: the first
        // source fetch is in flight behind these index loads, and the lines can be touched ahead of the scatter ----
        dst = T.val + dbase;
        if (has_dst && !merged) {
            if (ib >= jb) {
                // indirect[rel] = position of global row (xsup[ib]+rel) inside destination block L(ib,jb)
                const int *drows = T.lidx + T.sn_lidx[jb] + di1;
                const int fnz = T.xsup[ib], dn = di2;
                for (int i = tid; i < dn; i += NT) s_ind[drows[i] - fnz] = i;
                __syncthreads();
                for (int t = tid; t < TMv; t += NT) s_rowmap[t] = (t < nr) ? ZS * (di0 + s_ind[lsub[t / ZS] - fnz]) + t % ZS : 0;
                const int ldv = T.sn_nsupr[jb];
                for (int t = tid; t < TNv; t += NT) s_colmap[t] = ZS * s_jj[t] * ldv;
            } else {
                const int64_t d0 = T.sn_uidx[ib] + di0;
                for (int t = tid; t < TMv; t += NT) s_rowmap[t] = (t < nr) ? ZS * lsub[t / ZS] + t % ZS : 0;
                for (int t = tid; t < TNv; t += NT) {
                    int cm = 0;
                    if (t < nc) cm = T.ucolptr[d0 + s_jj[t]] - T.uidx[d0 + s_jj[t]];  // colptr - fstnz
                    s_colmap[t] = ZS * cm;
                }
            }
        }
        __syncthreads();
        if (MM == 1) {
            // plan-time build pass: the record this tile reads from now on
            int *rec = const_cast<int *>(tmaps) + (size_t) (bid - id_base) * REC;
            if (tid == 0) {
                const int64_t lo = Lp - T.val, uo = Uv - T.val, db = has_dst ? (int64_t) (dst - T.val) : -1;
                rec[0] = k; rec[1] = nr; rec[2] = nc; rec[3] = ns; rec[4] = lda; rec[5] = kbeg_own;
                rec[6] = (int) (uint32_t) lo; rec[7] = (int) (lo >> 32); rec[8] = (int) (uint32_t) uo; rec[9] = (int) (uo >> 32);
                rec[10] = (int) (uint32_t) db; rec[11] = (int) (db >> 32); rec[12] = stc; rec[13] = Rw; rec[14] = 0; rec[15] = 0;
            }
            if (tid == 64) {
                // the K-fused predecessor (at most one by default: fuse_max_prev): everything its source pass needs but the two maps
                int np = 0;
                if (!Z && T.fuse_prev) { while (np < 3 && T.fuse_prev[3 * k + np] >= 0) ++np; }
                rec[16] = np;
                for (int q = 17; q < HDR; ++q) rec[q] = 0;
                for (int j = 0; j < np; ++j) {
                    const int pj = 3 * k + j, ks = T.fuse_prev[pj];
                    const int nss = T.xsup[ks + 1] - T.xsup[ks];
                    const int64_t lo = T.sn_lval[ks], uo = T.sn_uval[ks];
                    const int64_t co = 2 * (int64_t) (T.pair_coff[pj] + stc), ro = (int64_t) T.pair_roff[pj] + Rw;
                    int *r = rec + 17 + 11 * j;
                    r[0] = nss; r[1] = T.sn_nsupr[ks]; r[2] = (nss - T.sn_ldu[ks]) & ~3;
                    r[3] = (int) (uint32_t) lo; r[4] = (int) (lo >> 32); r[5] = (int) (uint32_t) uo; r[6] = (int) (uo >> 32);
                    r[7] = (int) (uint32_t) co; r[8] = (int) (co >> 32); r[9] = (int) (uint32_t) ro; r[10] = (int) (ro >> 32);
                }
            }
            for (int t = tid; t < TMv; t += NT) rec[HDR + t] = has_dst ? s_rowmap[t] : 0;
            for (int t = tid; t < TNv; t += NT) {
                rec[HDR + TMv + t] = has_dst ? s_colmap[t] : 0;
                rec[HDR + TMv + TNv + t] = s_cptr[t];
                rec[HDR + TMv + 2 * TNv + t] = s_lead[t];
            }
            return;
        }
    }
    // K-fused update: the deferred updates of up to three predecessors of k in its chain (k = parent(k-1) = ..., consecutive
    // levels) are accumulated here in the same registers -> ONE prologue and ONE scatter for K = sum of their widths.  A
    // predecessor's block structure beyond k is a subset of k's: host-built maps give, per panel row / non-empty U
    // column of k, where the same global row / column sits in its panel / U row (or that it is absent = zeros).
    int nprev = 0;
    if (MM == 2) nprev = Z ? 0 : s_hdr[16];
    else if (!Z && T.fuse_prev) { while (nprev < 3 && T.fuse_prev[3 * k + nprev] >= 0) ++nprev; }
    double touch0 = 0.0, touch1 = 0.0;

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // a wave whose 32 x 64 (32 x 32) part lies entirely outside a ragged tile skips its MFMAs: the MFMA pipe is the resource
    // the co-resident workgroups share.  The row part is rotated by the tile id so that the idle waves of short tiles fall on
    // different SIMDs (wave w runs on SIMD w % 4).
    const int rm0 = ((wave + bid) % WR) * (TMv / WR), cn0 = (wave / WR) * (TNv / WC);
    const bool wave_on = rm0 < nr && cn0 < nc;
    // the lane's index into the swizzled U stage for the k group 0 and its first column block (schur_chunk: ^ k4, + 256 per block of 16 columns)
    const int pa = uidx(lane >> 4, cn0 + (lane & 15));
    d4 acc[NBC][NBR];
#pragma unroll
    for (int a = 0; a < NBC; ++a)
#pragma unroll
        for (int b = 0; b < NBR; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

    const int li = tid % TMv, lk = tid / TMv;           // L loader: row li, k = lk + LKS*q
    const int uk = tid & 15, uj = tid >> 4;             // U loader: k = uk, col = uj + UJS*q
    double pl[LQ], pu[UQ];
    const int *cpS = s_cptr, *ldS = s_lead;   // column maps of the current source (LDS): re-read per chunk, registers are scarce
    bool lrow_ok = li < nr;
    // complex L loader: real row li = (panel row li / 2, part a), the thread's k all have parity b = lk & 1: it reads part a ^ b,
    // negated for (a, b) = (0, 1)
    const int zoff = Z ? ((li ^ lk) & 1) : 0;
    const double zsgn = (Z && !(li & 1) && (lk & 1)) ? -1.0 : 1.0;
    const int lrow0 = Z ? (li & ~1) + zoff : li;
    // per-source state of the K loop (source 0 = fused predecessor ka, source 1 = k itself); ns_s counts REAL k (2 per complex one)
    int ns_s = ZS * ns, lda_s = lda;
    const double *Lrow = Lp + lrow0, *Uvs = Uv;
    const double *Lsrc = Lp;                              // SLUAMD_SCHUR_FETCH2: the source panel's base (uniform) and this thread's byte offset from it
    uint32_t lvo = (uint32_t) (lrow0 + lk * lda) << 3;    //   (row of the tile + its k phase; a slot is far below 2^29 values)
    // SLUAMD_SCHUR_FETCH3: thread = (pair of tile rows 2 li2, 2 li2 + 1; panel column lk3 + LKS3 q) and (pair of k 2 uk2, 2 uk2 + 1; tile column uj3 + UJS3 q)
    constexpr bool F3 = SLUAMD_SCHUR_FETCH3 && !Z;
    constexpr int LRP = TMv / 2, LKS3 = NT / LRP, LQ3 = KC / LKS3, UJS3 = NT / 8, UQ3 = TNv / UJS3;
    const int li2 = tid % LRP, lk3 = tid / LRP, uk2 = tid & 7, uj3 = tid >> 3;
    // (Measured and not kept, profiles/r06_ab_*.txt: a SECOND register set in flight -- 0 ms in the 4-wave configuration with 256 VGPRs, spills and +35 ms at this
    //  configuration's 128-VGPR cap; ONE software pipeline across the K-fused sources with all their maps staged in LDS up front -- 0 ms; s_setprio around the MFMA
    //  phase -- 0 ms; the scatter's maps read in one batch -- 0 ms.  What the K loop paid for was the ISSUE cost of its loader.)
    d2 wl[F3 ? LQ3 : 1], wu[F3 ? UQ3 : 1];        // the raw 16-byte loads; zeros / halves are selected at the stash, when they have arrived
    bool un0[F3 ? UQ3 : 1], un1[F3 ? UQ3 : 1];    // per U load: element k / k + 1 is a stored one (at or below the column's leading zeros, inside the source)
    // per source: which of the two rows exist there (own source: inside the tile; a K-fused predecessor: present in its panel), the byte offset of the first
    // existing one, and -- rows that are not neighbours in a predecessor's panel (its blocks may come in another order) -- the second one's own offset
    bool l_has0 = 2 * li2 < nr, l_has1 = 2 * li2 + 1 < nr;
    bool lon[F3 ? LQ3 : 1];                       // per L load: its panel column lies inside the source
    uint32_t lvo3 = (uint32_t) (2 * li2 + lk3 * lda) << 3;
    // (two rows that both exist in a predecessor's panel are neighbours there: the planner only fuses pairs whose row map is monotone, build_pair_maps)
    bool rows_ok3 = true;                         // this thread's pair of tile rows is one 16-byte run of the source panel (or lies outside the tile): see `clean`
    auto l3_source = [&](int ra0, int ra1, int ldas) {
        l_has0 = ra0 >= 0; l_has1 = ra1 >= 0;
        lvo3 = (uint32_t) ((l_has0 ? ra0 : max(ra1, 0)) + lk3 * ldas) << 3;
        const bool p0 = 2 * li2 < nr, p1 = 2 * li2 + 1 < nr;
        rows_ok3 = !p0 || (l_has0 && (!p1 || ra1 == ra0 + 1));
    };
    // Every load is UNCONDITIONAL (a lane with nothing to fetch reads the first 16 bytes of the source instead): no exec-mask branches, no zero initialisation of
    // the destination registers -- which the compiler guards with s_waitcnt vmcnt(0), i.e. with the arrival of the loads issued just before
    auto fetch3 = [&](int k0) {
        const gbytes_t Lb0 = uniform_ptr(Lsrc), Ub0 = uniform_ptr(Uvs);
#pragma unroll
        for (int q = 0; q < LQ3; ++q) {
            const int kbq = k0 + LKS3 * q;                                            // uniform
            lon[q] = kbq + lk3 < ns_s;
            const uint64_t off = lon[q] ? ((uint64_t) kbq * (uint64_t) lda_s << 3) + (uint64_t) lvo3 : 0;
            wl[q] = *(gd2_t) (Lb0 + off);
        }
        int ldq[UQ3], cpq[UQ3];
#pragma unroll
        for (int q = 0; q < UQ3; ++q) { ldq[q] = ldS[uj3 + UJS3 * q]; cpq[q] = cpS[uj3 + UJS3 * q]; }
        const int kg = k0 + 2 * uk2;
#pragma unroll
        for (int q = 0; q < UQ3; ++q) {
            un0[q] = kg >= ldq[q] && kg < ns_s;
            un1[q] = kg + 1 >= ldq[q] && kg + 1 < ns_s;
            const uint32_t uvo = (un0[q] || un1[q]) ? (uint32_t) (cpq[q] + max(kg - ldq[q], 0)) << 3 : 0;     // first stored element of the pair (only k + 1 stored: the segment's first)
            wu[q] = *(gd2_t) (Ub0 + (uint64_t) uvo);
        }
    };
    auto stash3 = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LQ3; ++q) {
            d2 v;
            v.x = (lon[q] && l_has0) ? wl[q].x : 0.0;
            v.y = (lon[q] && l_has1) ? (l_has0 ? wl[q].y : wl[q].x) : 0.0;
            *(d2 *) &Ls[buf][(lk3 + LKS3 * q) * LDL + 2 * li2] = v;
        }
#pragma unroll
        for (int q = 0; q < UQ3; ++q) {
            d2 y;
            y.x = un0[q] ? wu[q].x : 0.0; y.y = un1[q] ? (un0[q] ? wu[q].y : wu[q].x) : 0.0;
            *(d2 *) &Us[buf][uidx(2 * uk2, uj3 + UJS3 * q)] = y;
        }
    };

    // ---- the CLEAN form of the loader (round 6, profiles/r06_ab_valu_additive.txt) ----
    // On this device the fp64 MFMA runs at the vector ALU's rate and SHARES it: every ordinary VALU instruction of a wave adds its cycles to the MFMA time of its
    // SIMD instead of hiding behind the other waves' MFMAs (32 / 96 extra v_add_u32 per chunk and wave: +7.4 / +20.5 ms on the 247 ms of all Schur launches = 0.22 ms
    // per instruction).  The loader above is ~80 VALU instructions per chunk (predicates, selects, 64-bit address arithmetic).  A source is CLEAN for a tile when no
    // column of the tile has leading zeros inside the source's K range and every pair of tile rows is one 16-byte run of its panel (always true for the tile's own
    // supernode; for a K-fused predecessor: all rows present -- chains inside one separator): its FULL chunks then need no predicate at all -- per-thread byte
    // offsets fixed per source, the chunk's advance in the wave-uniform (SGPR) base, plain stores into the stage.  Rows >= nr / columns >= nc of a ragged tile hold
    // whatever their (valid) addresses hold: they only reach accumulators that are never scattered.  Partial chunks and unclean sources take fetch3 / stash3.
    bool clean = false;
    uint32_t uvoc[F3 ? UQ3 : 1];                 // per U load: byte offset of (column, first k of this thread's pair) from the source's U base at k = kbeg0
    int kbeg0 = 0;
    bool cur_fast = false;                       // the chunk in the registers was fetched by fetch_clean
    i32x4 rsL = {0, 0, 0, 0}, rsU = {0, 0, 0, 0};     // resource descriptors of the current source's panel / U row block (SGPRs; set by source_clean)
    auto source_clean = [&](int kb, bool rows_ok) {
        int ok = rows_ok ? 1 : 0;
#pragma unroll
        for (int q = 0; q < UQ3; ++q) {
            const int c = uj3 + UJS3 * q;
            const int ld = ldS[c], cp = cpS[c];
            if (c < nc && (ld > kb || cp > 0x0F000000)) ok = 0;       // (32-bit byte offsets: value offsets inside the source's U row block below 2 GB)
            // (the thread fetches the k pair uk2 ^ swizzle(c) and owns slot uk2 of column c in the stage: its 16 bytes sit at lane * 16 of the wave's 1 KiB)
            uvoc[q] = (c < nc) ? (uint32_t) (cp - ld + kb + 2 * (uk2 ^ ((c >> 1) & 7))) << 3 : (uint32_t) (2 * uk2) << 3;
        }
        kbeg0 = kb;
        if ((uint64_t) (uint32_t) ns_s * (uint64_t) (uint32_t) lda_s > 0x0F000000ull) ok = 0;     // ... and the source panel
        clean = __syncthreads_and(ok) != 0;
        // (32-bit offsets: a panel / a U row block is far below 4 GB -- 300^3: 9e4 rows x 256 columns x 8 bytes)
        if (clean) { rsL = buffer_rsrc(Lsrc); rsU = buffer_rsrc(Uvs); }
    };
    constexpr bool DMA = F3 && TMv == 128 && NBUF == 2 && SLUAMD_SCHUR_DMA && SLUAMD_SCHUR_CLEAN;
    static_assert(!DMA || (LRP == 64 && UJS3 * UQ3 == TNv), "LDS-DMA loader: one panel column / eight tile columns per wave and load");
    auto fetch_clean = [&](int k0, int tb) {
#ifdef SLUAMD_EXP_NOWAIT      // (instrumented build: the loads are issued, their data is never waited for at the stash -- it is "used" here, a whole chunk later; the stage gets constants)
#pragma unroll
        for (int q = 0; q < LQ3; ++q) asm volatile("" :: "v"(wl[q]));
#pragma unroll
        for (int q = 0; q < UQ3; ++q) asm volatile("" :: "v"(wu[q]));
#endif
        if (DMA) {
            // the stage `tb` is free: every wave is past the MFMAs that read it (the barrier that closed the previous chunk / the previous source)
#pragma unroll
            for (int q = 0; q < LQ3; ++q)
                lds_dma_16(rsL, __builtin_amdgcn_readfirstlane(lds_byte_addr(&Ls[tb][(wave + LKS3 * q) * LDL])), (int) lvo3, (int) ((uint32_t) (k0 + LKS3 * q) * (uint32_t) lda_s << 3));
#pragma unroll
            for (int q = 0; q < UQ3; ++q)
                lds_dma_16(rsU, __builtin_amdgcn_readfirstlane(lds_byte_addr(&Us[tb][(8 * wave + UJS3 * q) * 16])), (int) uvoc[q], (int) ((uint32_t) (k0 - kbeg0) << 3));
            return;
        }
#pragma unroll
        for (int q = 0; q < LQ3; ++q)
            wl[q] = __builtin_bit_cast(d2, llvm_amdgcn_raw_buffer_load_i32x4(rsL, (int) lvo3, (int) ((uint32_t) (k0 + LKS3 * q) * (uint32_t) lda_s << 3), 0));
#pragma unroll
        for (int q = 0; q < UQ3; ++q) wu[q] = __builtin_bit_cast(d2, llvm_amdgcn_raw_buffer_load_i32x4(rsU, (int) uvoc[q], (int) ((uint32_t) (k0 - kbeg0) << 3), 0));
    };
    auto stash_clean = [&](int buf) {
        if (DMA) { lds_dma_wait(); return; }          // (the loads wrote the stage themselves)
#ifdef SLUAMD_EXP_NOSTASH     // (instrumented build, with SLUAMD_EXP_NOWAIT: no stores into the stage either -- what the five LDS writes per chunk cost)
        return;
#endif
#ifdef SLUAMD_EXP_NOWAIT
        const d2 cst = {1.0 + 1e-9 * tid, 1.0 - 1e-9 * tid};
#pragma unroll
        for (int q = 0; q < LQ3; ++q) *(d2 *) &Ls[buf][(lk3 + LKS3 * q) * LDL + 2 * li2] = cst;
#pragma unroll
        for (int q = 0; q < UQ3; ++q) {
            *(d2 *) &Us[buf][(uj3 + UJS3 * q) * 16 + 2 * uk2] = cst;
        }
        return;
#endif
#pragma unroll
        for (int q = 0; q < LQ3; ++q) *(d2 *) &Ls[buf][(lk3 + LKS3 * q) * LDL + 2 * li2] = wl[q];
#pragma unroll
        for (int q = 0; q < UQ3; ++q) {
            *(d2 *) &Us[buf][(uj3 + UJS3 * q) * 16 + 2 * uk2] = wu[q];        // slot uk2 of its column = the k pair uk2 ^ swizzle it fetched
        }
    };

    auto fetch_into = [&](double *pl, double *pu, int k0, bool restart = false) {
#if SLUAMD_EXP_NOLOAD
        // 1: no loads at all; 2: none at the (re)start of a source's pipeline only; 3: none in the steady state only
        if (SLUAMD_EXP_NOLOAD == 1 || (SLUAMD_EXP_NOLOAD <= 3 && (SLUAMD_EXP_NOLOAD == 2) == restart)) {
#pragma unroll
            for (int q = 0; q < LQ; ++q) pl[q] = 1.0 + 1e-9 * (k0 + q);
#pragma unroll
            for (int q = 0; q < UQ; ++q) pu[q] = 1.0 - 1e-9 * (k0 + q);
            return;
        }
#endif
#if SLUAMD_SCHUR_FETCH2
        if (!Z) {
            // Round 6: the loader's issue cost, not the latency of its loads, is what the K loop pays for (profiles/r06_ab_schur_epilogue.txt: no loads at all
            // -36 ms in the steady state, the same loads served by L1 / L2 / MALL -6 ... -4 ms, a second register set in flight 0 ms): the compiler's form was four
            // 64-bit multiply-adds for the L addresses and, per U column, TWO dependent LDS round trips (leading zeros -> branch -> value offset) before each load,
            // serialised by exec-mask branches -- ~1000 cycles per chunk in which the wave issues no MFMA.  Here: wave-uniform 64-bit bases in SGPRs (the panel
            // column k0 + LKS q is a scalar multiply), one 32-bit byte offset per thread, all eight map entries read by one batch of LDS loads.
            const gbytes_t Lb0 = uniform_ptr(Lsrc), Ub0 = uniform_ptr(Uvs);
#pragma unroll
            for (int q = 0; q < LQ; ++q) {
                const int kq = k0 + LKS * q;                                   // uniform
                const gbytes_t sb = Lb0 + ((uint64_t) kq * (uint64_t) lda_s << 3);   // uniform: SALU
                pl[q] = (lrow_ok && kq + lk < ns_s) ? *(gdouble_t) (sb + (uint64_t) lvo) : 0.0;
            }
            int ldq[UQ], cpq[UQ];
#pragma unroll
            for (int q = 0; q < UQ; ++q) { ldq[q] = ldS[uj + UJS * q]; cpq[q] = cpS[uj + UJS * q]; }
#pragma unroll
            for (int q = 0; q < UQ; ++q) { asm volatile("" : "+v"(ldq[q]), "+v"(cpq[q])); }      // (keeps the eight reads in ONE batch in front of the branches)
            const int kg = k0 + uk;
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                const uint32_t uvo = (uint32_t) (cpq[q] - ldq[q] + kg) << 3;
                pu[q] = (kg >= ldq[q] && kg < ns_s) ? *(gdouble_t) (Ub0 + (uint64_t) uvo) : 0.0;
            }
            return;
        }
#endif
#pragma unroll
        for (int q = 0; q < LQ; ++q) {
            const int kg = k0 + lk + LKS * q;
            if (Z) pl[q] = (lrow_ok && kg < ns_s) ? zsgn * Lrow[(size_t) (kg >> 1) * 2 * lda_s] : 0.0;
#if SLUAMD_EXP_NOLOAD >= 4      // 4 / 5 / 6: the same load instructions with their addresses wrapped into a 2 MB / 16 KB / 64 MB window of the arena (always L2 / L1 / MALL hits)
            else pl[q] = (lrow_ok && kg < ns_s) ? T.val[((size_t) (Lrow - T.val) + (size_t) kg * lda_s) & (SLUAMD_EXP_NOLOAD == 4 ? 0x3FFFF : (SLUAMD_EXP_NOLOAD == 5 ? 0x7FF : 0x7FFFFF))] : 0.0;
#else
            else pl[q] = (lrow_ok && kg < ns_s) ? Lrow[(size_t) kg * lda_s] : 0.0;
#endif
        }
        const int kg = k0 + uk;
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int ld = ldS[uj + UJS * q];
            if (Z) pu[q] = ((kg >> 1) >= ld && kg < ns_s) ? Uvs[2 * (cpS[uj + UJS * q] + ((kg >> 1) - ld)) + (kg & 1)] : 0.0;
#if SLUAMD_EXP_NOLOAD >= 4
            else pu[q] = (kg >= ld && kg < ns_s) ? T.val[((size_t) (Uvs - T.val) + cpS[uj + UJS * q] + (kg - ld)) & (SLUAMD_EXP_NOLOAD == 4 ? 0x3FFFF : (SLUAMD_EXP_NOLOAD == 5 ? 0x7FF : 0x7FFFFF))] : 0.0;
#else
            else pu[q] = (kg >= ld && kg < ns_s) ? Uvs[cpS[uj + UJS * q] + (kg - ld)] : 0.0;
#endif
        }
    };
    auto stash_from = [&](const double *pl, const double *pu, int buf) {
#pragma unroll
        for (int q = 0; q < LQ; ++q) Ls[buf][(lk + LKS * q) * LDL + li] = pl[q];
#pragma unroll
        for (int q = 0; q < UQ; ++q) Us[buf][uidx(uk, uj + UJS * q)] = pu[q];
    };
    int kend_cur = 0;                            // end of the current source's K range (this workgroup's share of it)
#ifdef SLUAMD_EXP_COUNT
    int cnt_fast = 0, cnt_slow = 0;
#endif
    auto fetch = [&](int k0, int tb, bool restart = false) {       // tb: the stage this chunk goes to (the LDS-DMA loads write it at once, every other form at `stash`)
        if (F3 && !SLUAMD_EXP_NOLOAD) {
            cur_fast = SLUAMD_SCHUR_CLEAN && clean && k0 + KC <= kend_cur;       // workgroup-uniform
#ifdef SLUAMD_EXP_COUNT
            if (cur_fast) ++cnt_fast; else ++cnt_slow;
#endif
            if (cur_fast) fetch_clean(k0, tb); else fetch3(k0);
        } else fetch_into(pl, pu, k0, restart);
    };
    auto stash = [&](int buf) {
        if (F3 && !SLUAMD_EXP_NOLOAD) { if (cur_fast) stash_clean(buf); else stash3(buf); }
        else stash_from(pl, pu, buf);
    };

    int buf = 0;
#if SLUAMD_EXP_VALU
    int dummy0 = tid, dummy1 = tid + 1, dummy2 = tid + 2, dummy3 = tid + 3;
#endif
    for (int src = 0; src <= nprev; ++src) {       // farthest predecessor first, k itself last
        int kbeg;
        if (MM == 2 && src < nprev) {
            // the predecessor's scalars come with the record (up to three of them, nearest first; the farthest runs first): its two maps are the only loads before its first fetch
            const int *ph = s_hdr + 17 + 11 * (nprev - 1 - src);
            const int nss = ph[0];
            const int64_t co = ((int64_t) ph[8] << 32) | (uint32_t) ph[7], ro = ((int64_t) ph[10] << 32) | (uint32_t) ph[9];
            const int *cinfo = T.pair_colinfo + co;
            const int ra = (li < nr) ? T.pair_rowmap[ro + li] : -1;
            for (int t = tid; t < TNv; t += NT) {
                s_cptr2[t] = (t < nc) ? cinfo[2 * t] : 0;
                s_lead2[t] = (t < nc) ? cinfo[2 * t + 1] : nss;
            }
            __syncthreads();
            ns_s = __builtin_amdgcn_readfirstlane(nss); lda_s = __builtin_amdgcn_readfirstlane(ph[1]);
            Lsrc = T.val + (((int64_t) ph[4] << 32) | (uint32_t) ph[3]); lvo = (uint32_t) (max(ra, 0) + lk * lda_s) << 3;
            if (F3) l3_source(2 * li2 < nr ? T.pair_rowmap[ro + 2 * li2] : -1, 2 * li2 + 1 < nr ? T.pair_rowmap[ro + 2 * li2 + 1] : -1, lda_s);
            Lrow = T.val + (((int64_t) ph[4] << 32) | (uint32_t) ph[3]) + max(ra, 0);
            Uvs = T.val + (((int64_t) ph[6] << 32) | (uint32_t) ph[5]);
            kbeg = __builtin_amdgcn_readfirstlane(ph[2]); lrow_ok = ra >= 0;
            cpS = s_cptr2; ldS = s_lead2;
        } else if (src < nprev) {
            const int pj = 3 * k + (nprev - 1 - src);
            const int ks = T.fuse_prev[pj];
            const int nss = T.xsup[ks + 1] - T.xsup[ks];
            const int *cinfo = T.pair_colinfo + 2 * (size_t) (T.pair_coff[pj] + stc);
            const int ra = (li < nr) ? T.pair_rowmap[T.pair_roff[pj] + Rw + li] : -1;
            for (int t = tid; t < TNv; t += NT) {
                s_cptr2[t] = (t < nc) ? cinfo[2 * t] : 0;
                s_lead2[t] = (t < nc) ? cinfo[2 * t + 1] : nss;
            }
            __syncthreads();
            ns_s = __builtin_amdgcn_readfirstlane(nss); lda_s = __builtin_amdgcn_readfirstlane(T.sn_nsupr[ks]); Lrow = T.val + T.sn_lval[ks] + max(ra, 0); Uvs = T.val + T.sn_uval[ks];
            Lsrc = T.val + T.sn_lval[ks]; lvo = (uint32_t) (max(ra, 0) + lk * lda_s) << 3;
            if (F3) { const int64_t ro = (int64_t) T.pair_roff[pj] + Rw; l3_source(2 * li2 < nr ? T.pair_rowmap[ro + 2 * li2] : -1, 2 * li2 + 1 < nr ? T.pair_rowmap[ro + 2 * li2 + 1] : -1, lda_s); }
            kbeg = __builtin_amdgcn_readfirstlane((nss - T.sn_ldu[ks]) & ~3); lrow_ok = ra >= 0;
            cpS = s_cptr2; ldS = s_lead2;
        } else {
            ns_s = __builtin_amdgcn_readfirstlane(ZS * ns); lda_s = __builtin_amdgcn_readfirstlane(lda); Lrow = Lp + lrow0; Uvs = Uv; lrow_ok = li < nr;
            Lsrc = Lp; lvo = (uint32_t) (lrow0 + lk * lda) << 3;
            if (F3) l3_source(2 * li2 < nr ? 2 * li2 : -1, 2 * li2 + 1 < nr ? 2 * li2 + 1 : -1, lda);
            kbeg = __builtin_amdgcn_readfirstlane(kbeg_own);
            cpS = s_cptr; ldS = s_lead;
        }
        if (F3 && SLUAMD_SCHUR_CLEAN) source_clean(kbeg, rows_ok3);
        int kend = ns_s;
        if (SK) {   // this workgroup's share of the source's chunks
            const int nch = (ns_s - kbeg + KC - 1) / KC;
            const int c0 = nch * (int) blockIdx.y / (int) gridDim.y, c1 = nch * ((int) blockIdx.y + 1) / (int) gridDim.y;
            kend = min(ns_s, kbeg + c1 * KC); kbeg += c0 * KC;
            if (kbeg >= kend) continue;
        }
        // chunk at which the destination lines are touched: the one before the last chunk of the last source
        kend_cur = kend;
        const int ktouch = (src == nprev && has_dst) ? max(kbeg, kbeg + ((kend - 1 - kbeg) / KC - 1) * KC) : -1;
#if SLUAMD_SCHUR_TIGHT
        if (DMA && clean && !SLUAMD_EXP_NOLOAD && SLUAMD_SCHUR_TOUCH == 0) {
            // the full chunks of a clean source in a loop of their own: four LDS-DMA loads, the MFMAs, one wait, one barrier -- none of the general loader's state
            // (predicates, registers of the staged loads, which-path flags) is live or merged at its joins
            const int nfull = (kend - kbeg) / KC;
            if (nfull > 0) {
                fetch_clean(kbeg, buf);
                lds_dma_wait();
                __syncthreads();
                for (int c = 0; c < nfull; ++c) {
                    const bool more = c + 1 < nfull;
                    if (more) fetch_clean(kbeg + KC, buf ^ 1);
                    if (wave_on) schur_chunk<NBR, NBC, LDL, LDU, USW>(Ls[buf], Us[buf], rm0, cn0, lane, acc, pa);
                    if (more) lds_dma_wait();
                    buf ^= 1;
                    kbeg += KC;
                    __syncthreads();
                }
#ifdef SLUAMD_EXP_COUNT
                cnt_fast += nfull;
#endif
                if (kbeg >= kend) continue;
            }
        }
#endif
        // (re)start of the software pipeline: every wave is past the last chunk's MFMAs (closing barrier of the loop)
        fetch(kbeg, buf, true);
        stash(buf);
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += KC) {
            const bool more = k0 + KC < kend;
            if (more) fetch(k0 + KC, NBUF == 2 ? buf ^ 1 : 0);
#if SLUAMD_EXP_VALU
#pragma unroll
            for (int e = 0; e < SLUAMD_EXP_VALU / 4; ++e)
                asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(dummy0), "+v"(dummy1), "+v"(dummy2), "+v"(dummy3));
#endif
            if (SLUAMD_SCHUR_TOUCH && k0 == ktouch) {
                // one chunk before the last: touch one element of every destination line (16 rows x 1 column of the tile) so
                // that the fp64 atomics of the epilogue find their lines in L2 instead of each holding an L2 miss slot
                constexpr int RG = TMv / 16;
                const int id0 = tid, id1 = tid + NT;
                const int c0 = id0 / RG, g0 = id0 % RG, c1 = id1 / RG, g1 = id1 % RG;
#if SLUAMD_SCHUR_TOUCH == 2
                if (DMA) {
                    // the touch as two 4-byte loads per lane into a scratch kilobyte of LDS nobody reads: no destination registers (at the 128-VGPR cap the two
                    // touched values were spilled, i.e. waited for at once), drained with the chunk's own loads / before the scatter
                    const double *t0 = (id0 < TNv * RG && c0 < nc && g0 * 16 < nr) ? dst + s_colmap[c0] + s_rowmap[g0 * 16] : dst;
                    const double *t1 = (id1 < TNv * RG && c1 < nc && g1 * 16 < nr) ? dst + s_colmap[c1] + s_rowmap[g1 * 16] : dst;
                    const uint32_t sink = __builtin_amdgcn_readfirstlane(lds_byte_addr(&s_sink[wave * 64]));
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(t0), "v"(t1), "s"(sink) : "memory");
                } else
#endif
                {
                if (id0 < TNv * RG && c0 < nc && g0 * 16 < nr) touch0 = dst[s_colmap[c0] + s_rowmap[g0 * 16]];
                if (id1 < TNv * RG && c1 < nc && g1 * 16 < nr) touch1 = dst[s_colmap[c1] + s_rowmap[g1 * 16]];
                }
            }
            const double *Lb = Ls[buf], *Ub = Us[buf];
            if (wave_on) schur_chunk<NBR, NBC, LDL, LDU, USW>(Lb, Ub, rm0, cn0, lane, acc, pa);
            if (NBUF == 2) { if (more) stash(buf ^ 1); buf ^= 1; }
            else if (more) { __syncthreads(); stash(0); }   // single stage: every wave is done reading before it is overwritten
            __syncthreads();
        }
    }

    // ---- scatter (epilogue) ----------------------------------------------------------------------
    if (DMA && SLUAMD_SCHUR_TOUCH == 2) lds_dma_wait();
#ifdef SLUAMD_EXP_COUNT
    if (tid == 0 && TMv == 128) { atomicAdd(&info[4], cnt_fast); atomicAdd(&info[5], cnt_slow); }
    if (tid == 0 && TMv == 64) { atomicAdd(&info[6], cnt_fast); atomicAdd(&info[7], cnt_slow); }
#endif
#if SLUAMD_EXP_VALU
    if (__builtin_expect((dummy0 ^ dummy1 ^ dummy2 ^ dummy3) == 0x7fffffff && tid == 777, 0)) atomicAdd(&info[3], 1);
#endif
    if (!has_dst) return;
    if (__builtin_expect(touch0 == 1.2345e-300 && touch1 == 1.2345e-300, 0)) atomicAdd(&info[3], 1);   // keeps the touch loads alive
#if SLUAMD_EXP_EPI == 0
    {
        double keep = 0.0;
#pragma unroll
        for (int ci = 0; ci < NBC; ++ci)
#pragma unroll
            for (int ri = 0; ri < NBR; ++ri) keep += (acc[ci][ri][0] + acc[ci][ri][1]) + (acc[ci][ri][2] + acc[ci][ri][3]);
        if (__builtin_expect(keep == 1.2345e-300, 0)) atomicAdd(&info[3], 1);
        return;
    }
#endif
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
#if SLUAMD_EXP_EPI == 1
                    if (row < nr) dcol[s_rowmap[row]] -= acc[ci][ri][r];
#else
                    if (row < nr) atomic_sub_f64(dcol + s_rowmap[row], acc[ci][ri][r]);
#endif
                }
            }
        }
}

// ---- triangular solves --------------------------------------------------------------------------
// x_k <- Linv x_k (unit lower) or Uinv x_k: one workgroup per supernode of the level, ONE dense triangular GEMV with the full
// inverse (what the reference's DiagInv=YES solve does with Linv / Uinv, pdgstrs_lsum.c:414-520) -- no dependent chain
// inside the block.  Used on XY layers, where the exchanges separate the diagonal solve from the updates.
template <bool LOWER, int NT>   // NT = 1024, or 256 for levels whose supernodes are at most 64 wide (one column quarter)
__device__ __forceinline__ void solve_diag_body(const DevTables &T, int k, const double *xin, double *xout, int64_t ldx, int nrhs, double *xs /* ns x nrhs */)
{
    __shared__ double s_part[NT / 256][256];
    if (!(T.sn_flags[k] & SNF_OWN_DIAG)) return;           // x_k is solved by the owner of the diagonal block
    const int fst = T.xsup[k], ns = T.xsup[k + 1] - fst;
    const double *Ti = T.inv + T.sn_inv[k] + (LOWER ? 0 : (size_t) ns * ns);
    const int tid = threadIdx.x;
    for (int idx = tid; idx < ns * nrhs; idx += NT) xs[idx] = xin[fst + (idx % ns) + (int64_t) (idx / ns) * ldx];
    __syncthreads();
    // thread = (row i, quarter of the columns): <= 4 batches of 16 L2 loads; the inverse stores explicit zeros in the other
    // triangle, column blocks entirely outside the wave's rows are skipped
    const int i = tid & 255, part = tid >> 8;
    for (int q = 0; q < nrhs; ++q) {
        const double *xq = xs + q * ns;
        double acc[4] = {0, 0, 0, 0};
        if (i < ns) {
            const double *Tr = Ti + i;
            const int jlo = LOWER ? 0 : (i & ~63), jhi = LOWER ? min(ns, (i | 63) + 1) : ns;
            for (int j0 = jlo + part * 64; j0 < jhi; j0 += 256) {
                const int j1 = min(j0 + 64, jhi);
                int j2 = j0;
                for (; j2 + 16 <= j1; j2 += 16) {
                    double tv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) tv[u] = Tr[(size_t) (j2 + u) * ns];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc[u & 3] += tv[u] * xq[j2 + u];
                }
                for (; j2 < j1; ++j2) acc[0] += Tr[(size_t) j2 * ns] * xq[j2];
            }
        }
        s_part[part][i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (tid < ns) {
            double a = s_part[0][tid];
            if (NT == 1024) a = (a + s_part[1][tid]) + (s_part[NT == 1024 ? 2 : 0][tid] + s_part[NT == 1024 ? 3 : 0][tid]);
            xout[fst + tid + (int64_t) q * ldx] = a;
        }
        __syncthreads();
    }
}

template <bool LOWER, int NT>
__global__ __launch_bounds__(NT) void k_solve_diag(DevTables T, const int *__restrict__ nodes, double *__restrict__ x,
                                                   int64_t ldx, int nrhs)
{
    extern __shared__ double xs[];  // ns x nrhs
    solve_diag_body<LOWER, NT>(T, nodes[blockIdx.x], x, x, ldx, nrhs, xs);
}

// The sweeps are bound by load latency and per-CU bandwidth, not by HBM (one dependent launch per level of the elimination
// DAG), so the update kernels are shaped for parallelism: small work units (64 panel rows / 64 skyline columns -> several
// hundred workgroups for a top-level supernode) of 1024 threads, every thread issuing ONE batch of <= 16 independent loads.
//
// lsum_i -= L_ik x_k for the off-diagonal rows of panel k (dlsum_fmod_inv, pdgstrs_lsum.c:414): workgroup = (supernode,
// 64-row strip); thread = (row, one of 16 column slices); x_k staged in LDS; fp64 atomics into x.
// RK (round 6): right-hand sides per pass over the values.  1 = the loop of rounds 1-5 (every right-hand side re-reads the unit's factor entries: right for nrhs = 1);
// 4 = a block of four right-hand sides rides along each batch of loads -- the reference's nrhs > 1 path is a GEMM there (pdgstrs_lsum.c:414-960) -- chosen by the
// launch wrappers when nrhs >= 2: nrhs = 16 reads the factors 4 x instead of 16 x
template <int NT, int NBT = 16, int RK = 1>   // NBT: loads per thread and batch (16: one batch covers a 256-column supernode with 1024 threads; 8: the builds for 8 waves per SIMD)
__device__ __forceinline__ void fwd_update_body(const DevTables &T, int k, int strip, const double *xsrc /* solved x_k */, double *xdst /* lsum accumulators */,
                                                int64_t ldx, int nrhs, double *xk /* ns x nrhs */, const int4 *rec = nullptr)
{
    constexpr int NP = NT / 64;     // column slices
    __shared__ double s_red[NP][64 + 1];
    // `rec` (1 x 1 layer sweeps): the unit's scalars in one 32-byte record written at plan time -- (first column, width, panel height, first
    // row of the strip) + (offset of the strip's first L value, offset of its first entry in the flat row map) -- instead of six table
    // lookups behind the unit list entry: the values and x_k go in flight one round trip earlier
    int fst, ns, lda, row0;
    int64_t loff, roff;
    int chk = 0;                // 1: joined links, the strip holds rows of the NEXT level's supernodes (DevTables::lrow_near != 0) -- not this unit's to update;
                                // 2: only the rows of a later member of the unit's own merged group (flag 2) are nobody's
    if (rec) {
        const int4 a = rec[0], b = rec[1];
        fst = a.x; ns = a.y & 0xffff; lda = a.z; row0 = a.w; chk = a.y >> 16;
        loff = ((int64_t) b.y << 32) | (uint32_t) b.x; roff = ((int64_t) b.w << 32) | (uint32_t) b.z;
    } else {
        fst = T.xsup[k]; ns = T.xsup[k + 1] - fst; lda = T.sn_nsupr[k]; row0 = T.sn_ldiag[k] + strip * 64;
        loff = T.sn_lval[k] + row0; roff = T.sn_lrow[k] + row0;
    }
    const int tid = threadIdx.x;
    const int r = tid & 63, part = tid >> 6;
    const int row = row0 + r;
    const bool rvalid = row < lda;
    const double *L = T.val + loff + r;
    const int grow = (rvalid && part == 0) ? T.lrow[roff + r] : 0;   // flat map: no walk over the slot's block descriptors
    const bool mine = !(chk && rvalid && part == 0 && T.lrow_near[roff + r] >= chk);
    const int cpp = (ns + NP - 1) / NP;           // columns per slice
    const int ka = min(ns, part * cpp), kb = min(ns, ka + cpp);
    // the thread's first batch of L (all of it for supernodes of <= 16 NP columns) goes in flight BEFORE x_k is staged: it does not depend
    // on x, and on the levels where a workgroup's life is a chain of round trips this removes one of them
    double lv0[NBT];
#pragma unroll
    for (int u = 0; u < NBT; ++u) lv0[u] = (rvalid && ka + u < kb) ? __builtin_nontemporal_load(L + (size_t) (ka + u) * lda) : 0.0;
    for (int idx = tid; idx < ns * nrhs; idx += NT) xk[idx] = xsrc[fst + (idx % ns) + (int64_t) (idx / ns) * ldx];
    __syncthreads();
    if (RK > 1) {
        for (int q0 = 0; q0 < nrhs; q0 += RK) {
            const double *xq[RK];
#pragma unroll
            for (int j = 0; j < RK; ++j) xq[j] = xk + min(q0 + j, nrhs - 1) * ns;      // (a block past the last right-hand side repeats it; its sums are dropped)
            double acc[RK];
#pragma unroll
            for (int j = 0; j < RK; ++j) acc[j] = 0.0;
            if (rvalid) {
#pragma unroll
                for (int u = 0; u < NBT; ++u) {
                    const int c = min(ka + u, ns - 1);
#pragma unroll
                    for (int j = 0; j < RK; ++j) acc[j] += lv0[u] * xq[j][c];           // lv0 is zero past kb
                }
                int kk = ka + NBT;
                for (; kk + NBT <= kb; kk += NBT) {
                    double lv[NBT];
#pragma unroll
                    for (int u = 0; u < NBT; ++u) lv[u] = __builtin_nontemporal_load(L + (size_t) (kk + u) * lda);
#pragma unroll
                    for (int u = 0; u < NBT; ++u)
#pragma unroll
                        for (int j = 0; j < RK; ++j) acc[j] += lv[u] * xq[j][kk + u];
                }
                for (; kk < kb; ++kk) {
                    const double l1 = __builtin_nontemporal_load(L + (size_t) kk * lda);
#pragma unroll
                    for (int j = 0; j < RK; ++j) acc[j] += l1 * xq[j][kk];
                }
            }
#pragma unroll
            for (int j = 0; j < RK; ++j) {
                if (q0 + j >= nrhs) break;
                s_red[part][r] = acc[j];
                __syncthreads();
                if (part == 0 && rvalid && mine) {
                    double a = 0.0;
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) a += s_red[p2][r];
                    atomic_sub_f64(xdst + grow + (int64_t) (q0 + j) * ldx, a);
                }
                __syncthreads();
            }
        }
        return;
    }
    for (int q = 0; q < nrhs; ++q) {
        const double *xq = xk + q * ns;
        double acc[4] = {0, 0, 0, 0};
        if (rvalid) {
#pragma unroll
            for (int u = 0; u < NBT; ++u) acc[u & 3] += lv0[u] * xq[min(ka + u, ns - 1)];     // lv0 is zero past kb
            int kk = ka + NBT;
            for (; kk + NBT <= kb; kk += NBT) {
                double lv[NBT];
#pragma unroll
                for (int u = 0; u < NBT; ++u) lv[u] = __builtin_nontemporal_load(L + (size_t) (kk + u) * lda);
#pragma unroll
                for (int u = 0; u < NBT; ++u) acc[u & 3] += lv[u] * xq[kk + u];
            }
            for (; kk < kb; ++kk) acc[0] += __builtin_nontemporal_load(L + (size_t) kk * lda) * xq[kk];
        }
        s_red[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (part == 0 && rvalid && mine) {
            double a = 0.0;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) a += s_red[p2][r];
            atomic_sub_f64(xdst + grow + (int64_t) q * ldx, a);
        }
        __syncthreads();
    }
}

template <int NT, int NBT = 16, int MINW = NT / 256, int RK = 1>     // MINW: waves per SIMD the build is for (__launch_bounds__' second argument)
__global__ __launch_bounds__(NT, MINW) void k_fwd_update(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix,
                                                   int nn, const double *xsrc, double *xdst, int64_t ldx, int nrhs, const int2 *__restrict__ units,
                                                   const int4 *__restrict__ recs)
{
    extern __shared__ double xk[];  // ns x nrhs
    if (recs) { fwd_update_body<NT, NBT, RK>(T, 0, 0, xsrc, xdst, ldx, nrhs, xk, recs + 2 * (size_t) blockIdx.x); return; }   // unit records of the same list
    int k, strip;
    if (units) { const int2 u = units[blockIdx.x]; k = u.x; strip = u.y; }   // host-built (supernode, strip) list of one launch
    else { const int ni = find_node_wave(prefix, nn, blockIdx.x); k = nodes[ni]; strip = blockIdx.x - prefix[ni]; }
    fwd_update_body<NT, NBT, RK>(T, k, strip, xsrc, xdst, ldx, nrhs, xk);
}

// x_k -= U(k, chunk of 64 non-empty columns) x_cols  (dlsum_bmod_inv, pdgstrs_lsum.c:1362): workgroup = (supernode, chunk);
// lanes run along the rows of supernode k (coalesced over the skyline segments), wave w takes the chunk's columns 4 w .. 4 w + 3
// (one batch of 16 loads per lane); the 16 partial sums are combined in LDS and subtracted from x_k with fp64 atomics.
template <int NT, int RBv = (NT == 1024 ? 4 : 1), int CBT = 4, bool UNR = true, int RK = 1>
// NT = 1024 (16 waves x 4 columns), or 256 for levels whose supernodes are at most 64 wide (4 waves x 16 columns, one row block);
// RBv = 4 with NT = 512 / 256: supernodes of up to 256 columns in smaller workgroups (more of them resident per CU: levels of MANY units, where overlapping the phases
// of a workgroup's life -- record, maps, values, reduction, atomics -- across workgroups counts for more than the length of one life);
// CBT = columns per batch of loads (CBT x RBv loads per lane in flight), UNR = false: one batch in flight at a time (the builds for 8 waves per SIMD)
__device__ __forceinline__ void bwd_update_body(const DevTables &T, int k, int chunk, const double *xcols /* solved x of the chunk's columns */,
                                                double *xrows /* accumulators of x_k */, int64_t ldx, int nrhs, const int4 *rec = nullptr)
{
    constexpr int NWV = NT / 64, CPW = 64 / NWV, RB = RBv, UF = UNR ? 16 : 1;
    static_assert(CPW % CBT == 0, "columns per wave must be a multiple of the batch");
    __shared__ int s_cp[64], s_ld[64], s_gc[64];
    __shared__ double s_xc[RK][64];
    __shared__ double s_red[NWV][64 * RB];
    int fst, ns, ncol;          // `rec`: (first column, width, columns of this chunk) + (first entry in the flat column maps, offset of U(k,:)) as in fwd_update_body
    int64_t ci0, uoff;
    int chk = 0;                // 1: joined links, columns of the NEXT level's supernodes (DevTables::ucol_near != 0) belong to that level's joined units: treated as empty here;
                                // 2: only the columns of a later member of the row's own merged group (flag 2)
    if (rec) {
        const int4 a = rec[0], b = rec[1];
        fst = a.x; ns = a.y & 0xffff; ncol = a.z; chk = a.y >> 16;
        ci0 = ((int64_t) b.y << 32) | (uint32_t) b.x; uoff = ((int64_t) b.w << 32) | (uint32_t) b.z;
    } else {
        fst = T.xsup[k]; ns = T.xsup[k + 1] - fst; ncol = min(64, T.sn_ncolu[k] - chunk * 64);
        ci0 = T.sn_ucol[k] + chunk * 64; uoff = T.sn_uval[k];
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < ncol) {
        const int64_t ci = ci0 + tid;
        s_ld[tid] = (chk && T.ucol_near[ci] >= chk) ? ns : T.ucol_ld[ci]; s_cp[tid] = T.ucol_cp[ci]; s_gc[tid] = T.ucol_gc[ci];
    }
    __syncthreads();
    const double *Uv = T.val + uoff;
    auto load_batch = [&](double (&uv)[CBT][RB], int cb) {     // one batch: CBT columns of the wave x RB row blocks
#pragma unroll
        for (int cc = 0; cc < CBT; ++cc) {
            const int c = wave * CPW + cb + cc;
            const bool cok = c < ncol;
            const int ld = cok ? s_ld[c] : ns;
            const double *col = Uv + (cok ? s_cp[c] : 0) - ld;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int i = lane + 64 * q;
                uv[cc][q] = (i >= ld && i < ns) ? __builtin_nontemporal_load(col + i) : 0.0;
            }
        }
    };
    // the wave's first batch goes in flight BEFORE the gather of x (it depends on the column maps only): one round trip less in the
    // life of a workgroup on the levels where that is what a workgroup's life consists of
    double uv0[CBT][RB];
    load_batch(uv0, 0);
    if (RK > 1) {
        for (int r0 = 0; r0 < nrhs; r0 += RK) {
            if (tid < ncol)
#pragma unroll
                for (int j = 0; j < RK; ++j) s_xc[j][tid] = xcols[s_gc[tid] + (int64_t) min(r0 + j, nrhs - 1) * ldx];
            __syncthreads();
            double a[RK][RB];
#pragma unroll
            for (int j = 0; j < RK; ++j)
#pragma unroll
                for (int q = 0; q < RB; ++q) a[j][q] = 0.0;
            auto accumulate = [&](const double (&uv)[CBT][RB], int cb) {
#pragma unroll
                for (int cc = 0; cc < CBT; ++cc) {
                    const int c = wave * CPW + cb + cc;
#pragma unroll
                    for (int j = 0; j < RK; ++j) {
                        const double xv = (c < ncol) ? s_xc[j][c] : 0.0;
#pragma unroll
                        for (int q = 0; q < RB; ++q) a[j][q] += uv[cc][q] * xv;
                    }
                }
            };
            if (r0 == 0) accumulate(uv0, 0);
            else { double uv[CBT][RB]; load_batch(uv, 0); accumulate(uv, 0); }
#pragma unroll UF
            for (int cb = CBT; cb < CPW; cb += CBT) {
                double uv[CBT][RB];
                load_batch(uv, cb);
                accumulate(uv, cb);
            }
#pragma unroll
            for (int j = 0; j < RK; ++j) {
                if (r0 + j >= nrhs) break;
#pragma unroll
                for (int q = 0; q < RB; ++q) s_red[wave][lane + 64 * q] = a[j][q];
                __syncthreads();
                if (tid < ns) {
                    double sv = 0.0;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) sv += s_red[w][tid];
                    if (sv != 0.0) atomic_sub_f64(xrows + fst + tid + (int64_t) (r0 + j) * ldx, sv);
                }
                __syncthreads();
            }
        }
        return;
    }
    for (int r = 0; r < nrhs; ++r) {
        if (tid < ncol) s_xc[0][tid] = xcols[s_gc[tid] + (int64_t) r * ldx];      // solved x of this chunk's columns: one gather
        __syncthreads();
        {
            double a[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) a[q] = 0.0;
            auto accumulate = [&](const double (&uv)[CBT][RB], int cb) {
#pragma unroll
                for (int cc = 0; cc < CBT; ++cc) {
                    const int c = wave * CPW + cb + cc;
                    const double xv = (c < ncol) ? s_xc[0][c] : 0.0;
#pragma unroll
                    for (int q = 0; q < RB; ++q) a[q] += uv[cc][q] * xv;
                }
            };
            accumulate(uv0, 0);
#pragma unroll UF
            for (int cb = CBT; cb < CPW; cb += CBT) {
                double uv[CBT][RB];
                load_batch(uv, cb);
                accumulate(uv, cb);
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) s_red[wave][lane + 64 * q] = a[q];
        }
        __syncthreads();
        if (tid < ns) {
            double sv = 0.0;
#pragma unroll
            for (int w = 0; w < NWV; ++w) sv += s_red[w][tid];
            if (sv != 0.0) atomic_sub_f64(xrows + fst + tid + (int64_t) r * ldx, sv);
        }
        __syncthreads();
    }
}

template <int NT, int RBv = (NT == 1024 ? 4 : 1), int CBT = 4, bool UNR = true, int MINW = NT / 256, int RK = 1>
__global__ __launch_bounds__(NT, MINW) void k_bwd_update(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix,
                                                   int nn, const double *xcols, double *xrows, int64_t ldx, int nrhs, const int2 *__restrict__ units,
                                                   const int4 *__restrict__ recs)
{
    if (recs) { bwd_update_body<NT, RBv, CBT, UNR, RK>(T, 0, 0, xcols, xrows, ldx, nrhs, recs + 2 * (size_t) blockIdx.x); return; }
    int k, chunk;
    if (units) { const int2 u = units[blockIdx.x]; k = u.x; chunk = u.y; }
    else { const int ni = find_node_wave(prefix, nn, blockIdx.x); k = nodes[ni]; chunk = blockIdx.x - prefix[ni]; }
    bwd_update_body<NT, RBv, CBT, UNR, RK>(T, k, chunk, xcols, xrows, ldx, nrhs);
}

// One 64-row strip of a diagonal solve, OUT OF PLACE: xout_k[strip rows] = (Linv or Uinv)[strip rows, :] xin_k.  The diagonal solve of a
// chain supernode sits on the critical path of the sweeps (one dependent launch per level): as ONE workgroup it streams the 512 KB
// inverse through one CU (~10 us); as ns / 64 independent strips of the same GEMV shape as the panel update it takes what a launch
// takes.  Independent only because input and output are different vectors (LevelSched sweeps ping-pong between x and a work vector).
template <bool LOWER, int NT, int NBT = 16, int RK = 1>
__device__ __forceinline__ void diag_strip_body(const DevTables &T, int k, int strip, const double *xin, double *xout, int64_t ldx, int nrhs,
                                                double *xk /* ns x nrhs */, const int4 *rec = nullptr)
{
    constexpr int NP = NT / 64;     // column slices
    __shared__ double s_dred[NP][64 + 1];
    int fst, ns;                // `rec`: (first column, width, strip) + (offset of Linv, offset of Uinv)
    int64_t ioff;
    if (rec) {
        const int4 a = rec[0], b = rec[1];
        fst = a.x; ns = a.y; strip = a.z;
        ioff = LOWER ? (((int64_t) b.y << 32) | (uint32_t) b.x) : (((int64_t) b.w << 32) | (uint32_t) b.z);
    } else {
        fst = T.xsup[k]; ns = T.xsup[k + 1] - fst;
        ioff = T.sn_inv[k] + (LOWER ? 0 : (int64_t) ns * ns);
    }
    const double *Ti = T.inv + ioff;
    const int tid = threadIdx.x;
    const int r = tid & 63, part = tid >> 6;
    const int row = strip * 64 + r;
    const bool rvalid = row < ns;
    // the inverse stores explicit zeros in the other triangle: only the column blocks up to (from) the strip's own are read
    const int c0 = LOWER ? 0 : strip * 64, c1 = LOWER ? min(ns, strip * 64 + 64) : ns;
    const int cpp = (c1 - c0 + NP - 1) / NP;
    const int ka = min(c1, c0 + part * cpp), kb = min(c1, ka + cpp);
    const double *Tr = Ti + row;
    double tv0[NBT];                               // first batch of the inverse in flight before x_k is staged (as fwd_update_body)
#pragma unroll
    for (int u = 0; u < NBT; ++u) tv0[u] = (rvalid && ka + u < kb) ? Tr[(size_t) (ka + u) * ns] : 0.0;
    for (int idx = tid; idx < ns * nrhs; idx += NT) xk[idx] = xin[fst + (idx % ns) + (int64_t) (idx / ns) * ldx];
    __syncthreads();
    if (RK > 1) {
        for (int q0 = 0; q0 < nrhs; q0 += RK) {
            const double *xq[RK];
#pragma unroll
            for (int j = 0; j < RK; ++j) xq[j] = xk + min(q0 + j, nrhs - 1) * ns;
            double acc[RK];
#pragma unroll
            for (int j = 0; j < RK; ++j) acc[j] = 0.0;
            if (rvalid) {
#pragma unroll
                for (int u = 0; u < NBT; ++u) {
                    const int c = min(ka + u, ns - 1);
#pragma unroll
                    for (int j = 0; j < RK; ++j) acc[j] += tv0[u] * xq[j][c];
                }
                int kk = ka + NBT;
                for (; kk + NBT <= kb; kk += NBT) {
                    double tv[NBT];
#pragma unroll
                    for (int u = 0; u < NBT; ++u) tv[u] = Tr[(size_t) (kk + u) * ns];
#pragma unroll
                    for (int u = 0; u < NBT; ++u)
#pragma unroll
                        for (int j = 0; j < RK; ++j) acc[j] += tv[u] * xq[j][kk + u];
                }
                for (; kk < kb; ++kk) {
                    const double t1 = Tr[(size_t) kk * ns];
#pragma unroll
                    for (int j = 0; j < RK; ++j) acc[j] += t1 * xq[j][kk];
                }
            }
#pragma unroll
            for (int j = 0; j < RK; ++j) {
                if (q0 + j >= nrhs) break;
                s_dred[part][r] = acc[j];
                __syncthreads();
                if (part == 0 && rvalid) {
                    double a = 0.0;
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) a += s_dred[p2][r];
                    xout[fst + row + (int64_t) (q0 + j) * ldx] = a;
                }
                __syncthreads();
            }
        }
        return;
    }
    for (int q = 0; q < nrhs; ++q) {
        const double *xq = xk + q * ns;
        double acc[4] = {0, 0, 0, 0};
        if (rvalid) {
#pragma unroll
            for (int u = 0; u < NBT; ++u) acc[u & 3] += tv0[u] * xq[min(ka + u, ns - 1)];
            int kk = ka + NBT;
            for (; kk + NBT <= kb; kk += NBT) {
                double tv[NBT];
#pragma unroll
                for (int u = 0; u < NBT; ++u) tv[u] = Tr[(size_t) (kk + u) * ns];
#pragma unroll
                for (int u = 0; u < NBT; ++u) acc[u & 3] += tv[u] * xq[kk + u];
            }
            for (; kk < kb; ++kk) acc[0] += Tr[(size_t) kk * ns] * xq[kk];
        }
        s_dred[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (part == 0 && rvalid) {
            double a = 0.0;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) a += s_dred[p2][r];
            xout[fst + row + (int64_t) q * ldx] = a;
        }
        __syncthreads();
    }
}

// One link of a sweep on a 1 x 1 layer in ONE launch: workgroups [0, ndu) run the diagonal-solve strips `dunits` of the next level of
// the chain, the others run update units that do not feed those diagonal solves (LevelSched::fwd_units / bwd_units, bulk part) -- the
// diagonal solve of a chain supernode hides behind the far updates of its predecessor.  Two vectors: forward, the accumulated
// right-hand side lives in xa and the solved blocks go to xb (updates read xb, subtract from xa); backward, the accumulators are
// xb (= the forward solution minus the updates) and the final x_k goes to xa (updates read xa, subtract from xb).
template <bool LOWER, int NT, int RBv = (NT == 1024 ? 4 : 1), int NBT = 16, int CBT = 4, bool UNR = true, int MINW = NT / 256, int RK = 1>
__global__ __launch_bounds__(NT, MINW) void k_sweep(DevTables T, const int2 *__restrict__ dunits, int ndu, const int2 *__restrict__ units,
                                              double *xa, double *xb, int64_t ldx, int nrhs, const int4 *__restrict__ drecs, const int4 *__restrict__ urecs)
{
    extern __shared__ double dyn[];  // max_nsupc x nrhs
    const int bid = blockIdx.x;
    if (bid < ndu) {
        if (drecs) {     // unit records (same order as dunits)
            if (LOWER) diag_strip_body<true, NT, NBT, RK>(T, 0, 0, xa, xb, ldx, nrhs, dyn, drecs + 2 * (size_t) bid);
            else diag_strip_body<false, NT, NBT, RK>(T, 0, 0, xb, xa, ldx, nrhs, dyn, drecs + 2 * (size_t) bid);
            return;
        }
        const int2 d = dunits[bid];
        if (LOWER) diag_strip_body<true, NT, NBT, RK>(T, d.x, d.y, xa, xb, ldx, nrhs, dyn);
        else diag_strip_body<false, NT, NBT, RK>(T, d.x, d.y, xb, xa, ldx, nrhs, dyn);
        return;
    }
    if (urecs) {
        if (LOWER) fwd_update_body<NT, NBT, RK>(T, 0, 0, xb, xa, ldx, nrhs, dyn, urecs + 2 * (size_t) (bid - ndu));
        else bwd_update_body<NT, RBv, CBT, UNR, RK>(T, 0, 0, xa, xb, ldx, nrhs, urecs + 2 * (size_t) (bid - ndu));
        return;
    }
    const int2 u = units[bid - ndu];
    if (LOWER) fwd_update_body<NT, NBT, RK>(T, u.x, u.y, xb, xa, ldx, nrhs, dyn);
    else bwd_update_body<NT, RBv, CBT, UNR, RK>(T, u.x, u.y, xa, xb, ldx, nrhs);
}

// ---- joined links (LevelSched::join) -----------------------------------------------------------------------------------------------------------------
// Forward unit (s, c) of supernode j (level l + 1):  t = b_j[block c] - sum over the level-l panels k of (rows of panel k inside block c) x_k, then
// y_j[strip s] += Linv_j[s, c] t.  One round trip after the record: the inverse block, the panel rows, their positions, x_k and b_j go in flight together.
template <int NT, int NBT>
__device__ __forceinline__ void join_fwd_body(const DevTables &T, const int4 *rec, const int4 *jaux, const double *xa, double *xb, int64_t ldx, int nrhs, double *xk)
{
    constexpr int NP = NT / 64, CPB = 64 / NP;
    __shared__ double s_t[64];
    __shared__ double s_jred[NP][64 + 1];
    __shared__ int s_pos[64];
    const int4 a = rec[0], b = rec[1];
    const int fst = a.x, ns = a.y, st = a.z, c = a.w, nsrc = b.z, ovf = b.w;
    const double *Ti = T.inv + (((int64_t) b.y << 32) | (uint32_t) b.x);
    const int tid = threadIdx.x, r = tid & 63, part = tid >> 6;
    const int nc = min(64, ns - 64 * c), row = 64 * st + r;
    const bool rv = row < ns;
    double tv[CPB];
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
        const int col = 64 * c + part * CPB + u;
        tv[u] = (rv && col < ns) ? Ti[(size_t) col * ns + row] : 0.0;
    }
    for (int q = 0; q < nrhs; ++q) {
        if (tid < 64) s_t[tid] = tid < nc ? xa[fst + 64 * c + tid + (int64_t) q * ldx] : 0.0;
        for (int si = 0; si < nsrc; ++si) {
            const int4 *sr = si < 3 ? rec + 2 + 2 * si : jaux + 2 * (size_t) (ovf + si - 3);
            const int4 sa = sr[0], sb = sr[1];
            const int fk = sa.x, nk = sa.y, lda = sa.z, nr = sa.w;
            const double *L = T.val + (((int64_t) sb.y << 32) | (uint32_t) sb.x) + r;
            const int64_t roff = ((int64_t) sb.w << 32) | (uint32_t) sb.z;
            const int cpp = (nk + NP - 1) / NP;
            const int ka = min(nk, part * cpp), kb = min(nk, ka + cpp);
            const bool rok = r < nr;
            double lv0[NBT];
#pragma unroll
            for (int u = 0; u < NBT; ++u) lv0[u] = (rok && ka + u < kb) ? __builtin_nontemporal_load(L + (size_t) (ka + u) * lda) : 0.0;
            __syncthreads();      // the previous source's x_k / positions are consumed; s_t initialised
            for (int idx = tid; idx < nk; idx += NT) xk[idx] = xb[fk + idx + (int64_t) q * ldx];
            if (tid < nr) s_pos[tid] = T.lrow[roff + tid] - fst - 64 * c;
            __syncthreads();
            double acc[4] = {0, 0, 0, 0};
            if (rok) {
#pragma unroll
                for (int u = 0; u < NBT; ++u) acc[u & 3] += lv0[u] * xk[min(ka + u, nk - 1)];
                int kk = ka + NBT;
                for (; kk + NBT <= kb; kk += NBT) {
                    double lv[NBT];
#pragma unroll
                    for (int u = 0; u < NBT; ++u) lv[u] = __builtin_nontemporal_load(L + (size_t) (kk + u) * lda);
#pragma unroll
                    for (int u = 0; u < NBT; ++u) acc[u & 3] += lv[u] * xk[kk + u];
                }
                for (; kk < kb; ++kk) acc[0] += __builtin_nontemporal_load(L + (size_t) kk * lda) * xk[kk];
            }
            s_jred[part][r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __syncthreads();
            if (part == 0 && rok) {
                double sum = 0.0;
#pragma unroll
                for (int p2 = 0; p2 < NP; ++p2) sum += s_jred[p2][r];
                s_t[s_pos[r]] -= sum;       // the rows of one panel block are distinct rows of j
            }
        }
        __syncthreads();
        double a2 = 0.0;
#pragma unroll
        for (int u = 0; u < CPB; ++u) a2 += tv[u] * s_t[part * CPB + u];
        s_jred[part][r] = a2;
        __syncthreads();
        if (part == 0 && rv) {
            double sum = 0.0;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) sum += s_jred[p2][r];
            atomic_sub_f64(xb + fst + row + (int64_t) q * ldx, -sum);
        }
        __syncthreads();
    }
}

// Backward unit (s, c) of supernode k (level l):  t = w_k[block c] - U(k rows of block c, columns of level l + 1) x, then x_k[strip s] += Uinv_k[s, c] t.
// Lanes run along the 64 rows of block c, waves take NBT near columns each per pass (their descriptors and x staged in LDS first).
template <int NT, int NBT>
__device__ __forceinline__ void join_bwd_body(const DevTables &T, const int4 *rec, const int4 *jaux, double *xa, const double *xb, int64_t ldx, int nrhs)
{
    constexpr int NWV = NT / 64, CPB = 64 / NWV, NCH = NWV * NBT;
    __shared__ double s_bt[64];
    __shared__ double s_bred[NWV][64 + 1];
    __shared__ int s_nld[NCH], s_ncp[NCH];
    __shared__ double s_nx[NCH];
    const int4 a = rec[0], b = rec[1], c2 = rec[2];
    const int fst = a.x, ns = a.y, st = a.z, c = a.w, noff = b.z, ncnt = b.w;
    const double *Ti = T.inv + (((int64_t) b.y << 32) | (uint32_t) b.x);
    const double *Uv = T.val + (((int64_t) c2.y << 32) | (uint32_t) c2.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = min(64, ns - 64 * c), row = 64 * st + lane;
    const int i = 64 * c + lane - c2.z, nsm = c2.w;     // row inside the supernode whose U row holds block c (a member of a merged group: offset c2.z, width c2.w)
    const bool rv = row < ns;
    double tv[CPB];
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
        const int col = 64 * c + wave * CPB + u;
        tv[u] = (rv && col < ns) ? Ti[(size_t) col * ns + row] : 0.0;
    }
    for (int q = 0; q < nrhs; ++q) {
        if (tid < 64) s_bt[tid] = tid < nc ? xb[fst + 64 * c + tid + (int64_t) q * ldx] : 0.0;
        double acc = 0.0;
        for (int e0 = 0; e0 < ncnt; e0 += NCH) {
            const int cnt = min(NCH, ncnt - e0);
            __syncthreads();
            if (tid < cnt) {
                const int4 col = jaux[noff + e0 + tid];
                s_nld[tid] = col.x; s_ncp[tid] = col.y;
                s_nx[tid] = xa[col.z + (int64_t) q * ldx];
            }
            __syncthreads();
            double uv[NBT];
#pragma unroll
            for (int u = 0; u < NBT; ++u) {
                const int e = wave * NBT + u;
                const int ld = e < cnt ? s_nld[e] : nsm;
                uv[u] = (i >= ld && i < nsm) ? __builtin_nontemporal_load(Uv + s_ncp[min(e, cnt - 1)] - ld + i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NBT; ++u) { const int e = wave * NBT + u; acc += uv[u] * (e < cnt ? s_nx[e] : 0.0); }
        }
        s_bred[wave][lane] = acc;
        __syncthreads();
        if (wave == 0) {
            double sum = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < NWV; ++w2) sum += s_bred[w2][lane];
            s_bt[lane] -= sum;
        }
        __syncthreads();
        double a2 = 0.0;
#pragma unroll
        for (int u = 0; u < CPB; ++u) a2 += tv[u] * s_bt[wave * CPB + u];
        s_bred[wave][lane] = a2;
        __syncthreads();
        if (wave == 0 && rv) {
            double sum = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < NWV; ++w2) sum += s_bred[w2][lane];
            atomic_sub_f64(xa + fst + row + (int64_t) q * ldx, -sum);
        }
        __syncthreads();
    }
}

// One joined link in ONE launch: workgroups [0, nj) run the joined diagonal units of the next level, the others the regular units of this one (near rows /
// columns skipped).  Vectors as k_sweep: forward reads xa (right-hand side), reads / adds xb (solved blocks); backward reads xb, reads / adds xa.
template <bool LOWER, int NT, int RBv, int NBT, int CBT, bool UNR, int MINW>
__global__ __launch_bounds__(NT, MINW) void k_sweep_join(DevTables T, const int4 *__restrict__ jrecs, int nj, const int4 *__restrict__ jaux, const int4 *__restrict__ urecs,
                                                         double *xa, double *xb, int64_t ldx, int nrhs)
{
    extern __shared__ double dyn[];  // max_nsupc x nrhs
    const int bid = blockIdx.x;
    if (bid < nj) {
        if (LOWER) join_fwd_body<NT, NBT>(T, jrecs + 8 * (size_t) bid, jaux, xa, xb, ldx, nrhs, dyn);
        else join_bwd_body<NT, NBT>(T, jrecs + 4 * (size_t) bid, jaux, xa, xb, ldx, nrhs);
        return;
    }
    if (LOWER) fwd_update_body<NT, NBT>(T, 0, 0, xb, xa, ldx, nrhs, dyn, urecs + 2 * (size_t) (bid - nj));
    else bwd_update_body<NT, RBv, CBT, UNR>(T, 0, 0, xa, xb, ldx, nrhs, urecs + 2 * (size_t) (bid - nj));
}

__global__ __launch_bounds__(256) void k_zero_nodes(const int *__restrict__ xsup, const int *__restrict__ nodes, int nn, double *__restrict__ x, int64_t ldx, int nrhs)
{
    const int ni = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ni >= nn) return;
    const int k = nodes[ni], f = xsup[k], l = xsup[k + 1];
    for (int q = 0; q < nrhs; ++q)
        for (int rr = f + lane; rr < l; rr += 64) x[rr + (int64_t) q * ldx] = 0.0;
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

// ---- exchange helpers (XY block-cyclic layers, Z ancestor reduction, distributed solve) ---------------------------
// y += a x : the daxpy of dzRecvLPanel / dzRecvUPanel (pd3dcomm.c:189-331) on a whole forest slice; HBM-bound, 24 B/element
__global__ __launch_bounds__(256) void k_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) y[i] += a * x[i];
}

// ... the same with fp64 atomics: the pipelined ancestor reduction adds the partner layer's partial sums while the Schur tiles of the
// ancestor forest's first levels already scatter (atomically) into the same panels
__global__ __launch_bounds__(256) void k_axpy_atomic(int64_t n, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) { const double v = x[i]; if (v != 0.0) unsafeAtomicAdd(y + i, v); }
}

// Own diagonal blocks of one level -> contiguous staging range (ns x ns, lda = ns each): the payload of dDiagFactIBCast
// (dtrfCommWrapper.c:32-118).  One workgroup per 1024-element chunk.
constexpr int DGC = 1024;
template <int VS>    // VS doubles per value: 1 (double), 2 (complex16)
__global__ __launch_bounds__(256) void k_pack_diag(DevTables T, const int *__restrict__ nodes, const int *__restrict__ prefix,
                                                   const int64_t *__restrict__ off, int nn, double *__restrict__ stage)
{
    const int ni = find_node_wave(prefix, nn, blockIdx.x);
    const int k = nodes[ni];
    const int ns = T.xsup[k + 1] - T.xsup[k];
    const int e0 = (blockIdx.x - prefix[ni]) * DGC, e1 = min(e0 + DGC, ns * ns);
    const double *A = T.val + T.sn_dptr[k] * VS;
    const int lda = T.sn_dlda[k];
    double *S = stage + off[ni] * VS;
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        const size_t a = ((e % ns) + (size_t) (e / ns) * lda) * VS;
#pragma unroll
        for (int v = 0; v < VS; ++v) S[(size_t) e * VS + v] = A[a + v];
    }
}

// x segments (runs of rows, all nrhs columns) <-> one contiguous total x nrhs column-major buffer (runs concatenated).
// mode 0: buf = x ; 1: x = buf ; 2: x += buf ; 3: buf = x, x = 0
__global__ __launch_bounds__(256) void k_xseg_copy(double *__restrict__ x, int64_t ldx, int nrhs, const int *__restrict__ runs, int nruns,
                                                   int64_t total, double *__restrict__ buf, int mode)
{
    // runs: (row0, nrows, prefix) triples; prefix = rows before this run
    for (int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x; e < total * nrhs; e += (int64_t) gridDim.x * 256) {
        const int64_t row = e % total; const int q = (int) (e / total);
        int lo = 0, hi = nruns;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (runs[3 * mid + 2] <= row) lo = mid; else hi = mid; }
        double *xp = x + runs[3 * lo] + (row - runs[3 * lo + 2]) + (int64_t) q * ldx;
        double *bp = buf + row + (int64_t) q * total;
        if (mode == 0) *bp = *xp;
        else if (mode == 1) *xp = *bp;
        else if (mode == 2) *xp += *bp;
        else { *bp = *xp; *xp = 0.0; }
    }
}

// indexed rows (all nrhs columns) <-> contiguous cnt x nrhs buffer: the device side of pdReDistribute3d_B_to_X / X_to_B
// (pdgstrs3d.c:6265, :6404).  mode 0: buf = v[idx], 1: v[idx] = buf
template <int VS>    // doubles per value: 1 (double), 2 (complex16); ldv in values
__global__ __launch_bounds__(256) void k_rows_copy(double *__restrict__ v, int64_t ldv, int nrhs, const int *__restrict__ idx, int64_t cnt,
                                                   double *__restrict__ buf, int mode)
{
    for (int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x; e < cnt * nrhs; e += (int64_t) gridDim.x * 256) {
        const int64_t j = e % cnt; const int q = (int) (e / cnt);
        double *vp = v + (idx[j] + (int64_t) q * ldv) * VS;
#pragma unroll
        for (int t = 0; t < VS; ++t) { if (mode == 0) buf[e * VS + t] = vp[t]; else vp[t] = buf[e * VS + t]; }
    }
}

// ================================================================================================
//                              eng:: launchers (sluamd_internal.h)
// ================================================================================================
namespace eng {

static int g_num_cus = 256;
int sweep_attrs();
int setup()
{
    // kernels that keep a whole panel strip / diagonal block in LDS need more than the default 64 KiB
    HIPCHK(hipFuncSetAttribute((const void *) k_panel_trsm<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_panel_trsm<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_full_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_diag_lu2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) DIAG_LU2_LDS));
    HIPCHK(hipFuncSetAttribute((const void *) k_diag_lu2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) DIAG_LU2_LDS));
    HIPCHK(hipFuncSetAttribute((const void *) kz_diag_lu, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_solve_diag<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_solve_diag<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    { hipDeviceProp_t pr; int dev = 0; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) g_num_cus = pr.multiProcessorCount; }
    // the 256-thread variants stage max_nsupc (<= 64) x nrhs values: above 64 KiB when a matrix of narrow supernodes is solved for many right-hand sides
    if (sweep_attrs()) return 1;
    HIPCHK(hipFuncSetAttribute((const void *) k_solve_diag<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *) k_solve_diag<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
    return 0;
}

void diag_lu(hipStream_t s, const DevTables &T, const int *nodes, int nn, int mx, int replace_tiny, double thresh, int *info)
{
    if (nn <= 0) return;
    // bit 1 of the flag selects the round-1 right-looking kernel (SLUAMD_DIAG_V1).  Levels of narrow supernodes (<= 64: the
    // bottom of the tree, thousands of blocks per launch) are throughput-bound, not latency-bound: the right-looking kernel with
    // its smaller footprint is faster there (0.85 vs 2.2 ms per launch at 100^3)
    if (!(replace_tiny & 2) && mx > 64) {
        if (replace_tiny & 4) hipLaunchKernelGGL(k_diag_lu2<1>, dim3(nn), dim3(256), DIAG_LU2_LDS, s, T, nodes, replace_tiny & 1, thresh, info);
        else hipLaunchKernelGGL(k_diag_lu2<2>, dim3(nn), dim3(256), DIAG_LU2_LDS, s, T, nodes, replace_tiny & 1, thresh, info);
        return;
    }
    replace_tiny &= 1;
    // (four blocks per 256-thread workgroup; one or two per workgroup measured the same beside the bulk tiles: 290.1-290.9 / 290.5-291.1 / 290.5 ms, gpurun call 9)
    if (mx <= 64) hipLaunchKernelGGL(k_diag_lu_wave, dim3((nn + 3) / 4), dim3(256), 0, s, T, nodes, nn, replace_tiny, thresh, info);
    else if (mx <= 128) hipLaunchKernelGGL(k_diag_lu<128>, dim3(nn), dim3(256), 0, s, T, nodes, replace_tiny, thresh, info);
    else hipLaunchKernelGGL(k_diag_lu<256>, dim3(nn), dim3(256), 0, s, T, nodes, replace_tiny, thresh, info);
    if (mx <= 64) hipLaunchKernelGGL(k_diag_inv_all<4>, dim3(nn), dim3(128), 0, s, T, nodes);     // the contract: dinv of the owned blocks
    else hipLaunchKernelGGL(k_diag_inv_all<8>, dim3(nn), dim3(256), 0, s, T, nodes);
}

void diag_inv(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int ntask)
{
    if (ntask > 0) hipLaunchKernelGGL(k_diag_inv, dim3((ntask + 3) / 4), dim3(128), 0, s, T, nodes, prefix, nn);
}

void panel_trsm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int rs,
                int mx, const int2 *units)
{
    if (nl + nu <= 0) return;
    const size_t lds = trsm_lds_bytes(rs, (mx + 31) & ~31);
    if (rs == 32) hipLaunchKernelGGL(k_panel_trsm<32>, dim3(nl + nu), dim3(128), lds, s, T, nodes, lprefix, uprefix, nn, nl, units);
    else hipLaunchKernelGGL(k_panel_trsm<64>, dim3(nl + nu), dim3(256), lds, s, T, nodes, lprefix, uprefix, nn, nl, units);
}

#define SCHUR_LAUNCH(TM, TN, NWV, ZV, THREADS) \
    do { \
        if (mmode == 2) hipLaunchKernelGGL((k_schur<TM, TN, NWV, ZV, 2>), dim3(grid), dim3(THREADS), 0, s, T, nodes, prefix, nn, id_base, ntiles, info, ulist, prio, tmaps, xoff); \
        else if (mmode == 1) hipLaunchKernelGGL((k_schur<TM, TN, NWV, ZV, 1>), dim3(grid), dim3(THREADS), 0, s, T, nodes, prefix, nn, id_base, ntiles, info, ulist, prio, tmaps, xoff); \
        else hipLaunchKernelGGL((k_schur<TM, TN, NWV, ZV, 0>), dim3(grid), dim3(THREADS), 0, s, T, nodes, prefix, nn, id_base, ntiles, info, ulist, prio, tmaps, xoff); \
    } while (0)
void schur(hipStream_t s, int cfg, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info,
           const int4 *ulist, int prio, const int *tmaps, int mmode, int ksplit, const int *xoff, int xmax)
{
    if (ntiles <= 0) return;
    if (!xoff || xmax <= 0 || ksplit > 1) xoff = nullptr;
    const int grid = xoff ? 8 * xmax : ((ntiles + 7) / 8) * 8;
    if (ksplit > 1 && cfg == 0 && mmode != 1) {   // split-K form of the 128 x 128 configuration (chain tiles)
        if (mmode == 2) hipLaunchKernelGGL((k_schur<128, 128, 8, false, 2, true>), dim3(grid, ksplit), dim3(512), 0, s, T, nodes, prefix, nn, id_base, ntiles, info, ulist, prio, tmaps, nullptr);
        else hipLaunchKernelGGL((k_schur<128, 128, 8, false, 0, true>), dim3(grid, ksplit), dim3(512), 0, s, T, nodes, prefix, nn, id_base, ntiles, info, ulist, prio, tmaps, nullptr);
        return;
    }
    if (cfg == 0) SCHUR_LAUNCH(128, 128, 8, false, 512);
    else if (cfg == 1) SCHUR_LAUNCH(128, 128, 4, false, 256);
    else SCHUR_LAUNCH(64, 64, 4, false, 256);
}

void panel_gemm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int mx, const int2 *units)
{
    if (nl + nu <= 0) return;
    if (mx <= 64) hipLaunchKernelGGL(k_panel_gemm<16>, dim3(nl + nu), dim3(256), 0, s, T, nodes, lprefix, uprefix, nn, nl, units);
    else if (mx <= 128) hipLaunchKernelGGL(k_panel_gemm<32>, dim3(nl + nu), dim3(256), 0, s, T, nodes, lprefix, uprefix, nn, nl, units);
    else hipLaunchKernelGGL(k_panel_gemm<64>, dim3(nl + nu), dim3(256), 0, s, T, nodes, lprefix, uprefix, nn, nl, units);
}

// ---- merged chain groups (sluamd_internal.h): gather of the group's block triangle, batched dense products ----------------------------------------------------------
// One workgroup per task of ONE group: task t < nm: diagonal member t (its Linv / Uinv into the group's pair); then the member pairs (i > k): L block of member i in
// panel k -> LG, and (i < k): U block of column member k in row i -> UG.  Rows / columns absent from a block stay zero (the caller zero-fills the images).
__global__ __launch_bounds__(256) void k_grp_gather(DevTables T, const GrpDesc *__restrict__ gd, double *__restrict__ scr)
{
    // grid (task, slice): every task's elements dealt over 64 workgroups (a task is up to 256 x 256 elements behind two dependent index loads: as ONE workgroup it
    // is a 180 us chain of round trips on the side stream of the factorisation)
    const GrpDesc g = *gd;
    const int nG = g.nG, tid = threadIdx.x + 256 * blockIdx.y, nth = 256 * gridDim.y;
    double *LinvG = T.inv + g.ginv, *UinvG = LinvG + (int64_t) nG * nG;
    double *LG = scr, *UG = scr + GRP_SCR;
    int t = blockIdx.x;
    if (t < g.nm) {
        const int k = g.k[t], o = g.o[t], w = g.w[t];
        const double *Li = T.inv + T.sn_inv[k], *Ui = Li + (int64_t) w * w;
        for (int idx = tid; idx < w * w; idx += nth) {
            const int r = idx % w, c = idx / w;
            LinvG[(o + r) + (int64_t) (o + c) * nG] = Li[idx];
            UinvG[(o + r) + (int64_t) (o + c) * nG] = Ui[idx];
        }
        return;
    }
    t -= g.nm;
    const int npair = g.nm * (g.nm - 1) / 2;
    const bool upper = t >= npair;
    if (upper) t -= npair;
    int i = 1, kk = 0;                                        // pair index t -> (i, kk), kk < i
    for (int q = 0; q < t; ++q) { if (++kk == i) { ++i; kk = 0; } }
    if (i >= g.nm) return;
    if (!upper) {      // rows of member i inside the panel of member kk
        const int mi = g.k[i], mk = g.k[kk], oi = g.o[i], ok = g.o[kk], wk = g.w[kk];
        const int lb0 = T.sn_lb_off[mk], nb = T.sn_nlb[mk];
        int b = -1;
        for (int q = 0; q < nb; ++q) if (T.lb_gid[lb0 + q] == mi) { b = lb0 + q; break; }
        if (b < 0) return;
        const int nbrow = T.lb_nbrow[b], ro = T.lb_rowoff[b], lda = T.sn_nsupr[mk], f = T.xsup[mi];
        const double *Lp = T.val + T.sn_lval[mk] + ro;
        const int *rows = T.lrow + T.sn_lrow[mk] + ro;
        for (int idx = tid; idx < nbrow * wk; idx += nth) {
            const int r = idx % nbrow, c = idx / nbrow;
            LG[(oi + rows[r] - f) + (int64_t) (ok + c) * nG] = Lp[r + (size_t) c * lda];
        }
    } else {           // columns of member i (as the LATER member) inside the U row of member kk: block (kk, i), kk < i
        const int mr = g.k[kk], mc = g.k[i], orr = g.o[kk], oc = g.o[i], wr = g.w[kk];
        const int ub0 = T.sn_ub_off[mr], nub = T.sn_nub[mr];
        int b = -1;
        for (int q = 0; q < nub; ++q) if (T.ub_gid[ub0 + q] == mc) { b = ub0 + q; break; }
        if (b < 0) return;
        const int ncol = T.ub_ncols[b], st = T.ub_stcol[b], f = T.xsup[mc];
        const double *Uv = T.val + T.sn_uval[mr];
        const int64_t c0 = T.sn_ucol[mr] + st;
        for (int idx = tid; idx < ncol * wr; idx += nth) {
            const int r = idx % wr, q = idx / wr;
            const int ld = T.ucol_ld[c0 + q];
            if (r >= ld) UG[(orr + r) + (int64_t) (oc + T.ucol_gc[c0 + q] - f) * nG] = Uv[T.ucol_cp[c0 + q] + (r - ld)];
        }
    }
}

// C (M x N) = (-) A (M x K) B (K x N), column-major.  One workgroup per 32 x 32 tile of C, its four waves split K (the products of a group are a chain of six
// dependent launches of few tiles each: the length of ONE wave's K loop is what a launch lasts), partial tiles summed through LDS; operand fragments straight
// from memory (L2-resident images of <= 8 MB), four k-steps of loads in flight ahead of the MFMAs.  D[(l >> 4) + 4 r][l & 15] = sum_k A[l & 15][l >> 4] B[l >> 4][l & 15].
// M, N multiples of 16, K of 16 (the widths of grouped members are).
__global__ __launch_bounds__(256) void k_gemm_batched(DevTables T, const GemmDesc *__restrict__ descs, const int4 *__restrict__ tiles, double *__restrict__ scr)
{
    __shared__ double s_part[4][32 * 33];
    const int4 tl = tiles[blockIdx.x];
    const GemmDesc d = descs[tl.x];
    const double *A = (d.abase ? scr : T.inv) + d.a, *B = (d.bbase ? scr : T.inv) + d.b;
    double *C = (d.cbase ? scr : T.inv) + d.c;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int row0 = 32 * tl.y, col0 = 32 * tl.z;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ra0 = min(row0 + li, d.M - 1), ra1 = min(row0 + 16 + li, d.M - 1), cb0 = min(col0 + li, d.N - 1), cb1 = min(col0 + 16 + li, d.N - 1);
    const int kq4 = (d.K >> 2) >> 2;                          // k-steps of four per wave
    const int kbeg = 4 * kq4 * wave, kend = (wave == 3) ? d.K : kbeg + 4 * kq4;
    const double *Ap0 = A + ra0, *Ap1 = A + ra1, *Bp0 = B + (size_t) cb0 * d.ldb, *Bp1 = B + (size_t) cb1 * d.ldb;
    constexpr int U = 4;
    double a0[U], a1[U], b0[U], b1[U];
    auto load = [&](int k4, double (&x0)[U], double (&x1)[U], double (&y0)[U], double (&y1)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kq = min(k4 + 4 * u + lk, d.K - 1);
            x0[u] = Ap0[(size_t) kq * d.lda]; x1[u] = Ap1[(size_t) kq * d.lda]; y0[u] = Bp0[kq]; y1[u] = Bp1[kq];
        }
    };
    if (kbeg < kend) load(kbeg, a0, a1, b0, b1);
    for (int k4 = kbeg; k4 < kend; k4 += 4 * U) {
        double n0[U], n1[U], m0[U], m1[U];
        const bool more = k4 + 4 * U < kend;
        if (more) load(k4 + 4 * U, n0, n1, m0, m1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k4 + 4 * u < kend) {                          // wave-uniform
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) { a0[u] = n0[u]; a1[u] = n1[u]; b0[u] = m0[u]; b1[u] = m1[u]; }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_part[wave][(16 * b + li) * 33 + 16 * a + lk + 4 * r] = acc[a][b][r];
    __syncthreads();
    const double sg = d.neg ? -1.0 : 1.0;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int r = e & 31, c = e >> 5;
        if (row0 + r < d.M && col0 + c < d.N)
            C[row0 + r + (size_t) (col0 + c) * d.ldc] = sg * ((s_part[0][c * 33 + r] + s_part[1][c * 33 + r]) + (s_part[2][c * 33 + r] + s_part[3][c * 33 + r]));
    }
}

static const bool g_full_inv64 = getenv("SLUAMD_NO_FULL_INV64") == nullptr;
void grp_gather(hipStream_t s, const DevTables &T, const GrpDesc *d_desc, double *scratch)
{
    hipLaunchKernelGGL(k_grp_gather, dim3(4 + 6 + 6, 64), dim3(256), 0, s, T, d_desc, scratch);
}
void gemm_batched(hipStream_t s, const DevTables &T, const GemmDesc *d_descs, const int4 *d_tiles, int ntiles, double *scratch)
{
    if (ntiles > 0) hipLaunchKernelGGL(k_gemm_batched, dim3(ntiles), dim3(256), 0, s, T, d_descs, d_tiles, scratch);
}
void full_inv(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, int mx)
{
    if (nwork > 0 && mx <= 64 && g_full_inv64) { hipLaunchKernelGGL(k_full_inv64, dim3((2 * nn + 3) / 4), dim3(256), 0, s, T, nodes, prefix, nn); return; }   // levels of narrow supernodes
    if (nwork > 0) hipLaunchKernelGGL(k_full_inv, dim3(nwork), dim3(FIS * 4), trsm_lds_bytes(FIS, (mx + 31) & ~31), s, T, nodes, prefix, nn);
}

void solve_diag(hipStream_t s, bool lower, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs, int mx)
{
    if (nn <= 0) return;
    const size_t lds = (size_t) mx * nrhs * sizeof(double);
    if (mx <= 64) {   // levels of narrow supernodes (the bottom of the elimination DAG: thousands of them): small workgroups
        if (lower) hipLaunchKernelGGL((k_solve_diag<true, 256>), dim3(nn), dim3(256), lds, s, T, nodes, x, ldx, nrhs);
        else hipLaunchKernelGGL((k_solve_diag<false, 256>), dim3(nn), dim3(256), lds, s, T, nodes, x, ldx, nrhs);
    } else {
        if (lower) hipLaunchKernelGGL((k_solve_diag<true, 1024>), dim3(nn), dim3(1024), lds, s, T, nodes, x, ldx, nrhs);
        else hipLaunchKernelGGL((k_solve_diag<false, 1024>), dim3(nn), dim3(1024), lds, s, T, nodes, x, ldx, nrhs);
    }
}

// Builds of the sweep kernels.  The work unit is the same in all of them (64 panel rows / 64 skyline columns of one supernode, or one 64 x 64 block of a
// diagonal inverse in the joined links); what differs is how many threads share it and how many workgroups a CU holds at a time:
//   wide levels (supernodes of 65 .. 256 columns)
//     0  1024 threads, one batch of 16 loads per thread, 1 workgroup per CU: the shortest life of a unit -- launches of a FEW units (the chain at the top)
//     1   512 threads, two batches of 16, 2 workgroups per CU
//     5   512 threads, batches of 8, built for 6 waves per SIMD (80 VGPRs, no spills): 3 workgroups per CU -- the default for launches of >= 256 units
//   narrow levels (<= 64 columns): 256 threads, batch of 16 (4 workgroups per CU)
// A launch of many units is bound by how much of a workgroup's life (record -> maps -> values -> reduction -> atomics: a chain of round trips) overlaps with
// other workgroups' loads, not by the length of one life (profiles/r04_ab_solve_join.txt: 6.16 ms with build 0 everywhere, 5.53 with 1, 5.47 with 5; builds for
// 8 waves per SIMD spill and lose, more workgroups of 256 threads are no better, the narrow levels do not react).  SLUAMD_SWEEP_WIDE_V / SLUAMD_SWEEP_WIDE_MIN.
static const int g_sweep_wide_v = getenv("SLUAMD_SWEEP_WIDE_V") ? atoi(getenv("SLUAMD_SWEEP_WIDE_V")) : 5;
static const int g_sweep_wide_min = getenv("SLUAMD_SWEEP_WIDE_MIN") ? atoi(getenv("SLUAMD_SWEEP_WIDE_MIN")) : 256;
static inline int sweep_variant(int nwork, int mx, int nrhs = 1)    // 0 / 1 / 5 wide, 10 narrow
{
    if (mx <= 64) return 10;
    if (mx > 256 || nwork < g_sweep_wide_min || (g_sweep_wide_v != 1 && g_sweep_wide_v != 5)) return 0;
    if (nrhs >= 2 && g_sweep_wide_v == 5) return 1;      // the blocked right-hand-side units need more than build 5's 80 registers (they spill there)
    return g_sweep_wide_v;
}
//                 threads  row blocks  loads/batch (fwd, diag)  columns/batch (bwd)  all batches in flight  waves per SIMD
#define SWEEP_V0   1024,    4,          16,                      4,                   true,                  4
#define SWEEP_V1   512,     4,          16,                      4,                   true,                  2
#define SWEEP_V5   512,     4,          8,                       2,                   false,                 6
#define SWEEP_N0   256,     1,          16,                      4,                   true,                  1
#ifndef SLUAMD_SWEEP_RK
#define SLUAMD_SWEEP_RK 4
#endif
constexpr int SWEEP_RK = SLUAMD_SWEEP_RK;      // right-hand sides per pass over the factor entries in the nrhs >= 2 builds of the update / diagonal-strip units
template <int NT, int RBv, int NBT, int CBT, bool UNR, int MINW> struct SweepCfg {
    static void fwd(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xsrc, double *x, int64_t ldx, int nrhs, int mx,
                    const int2 *units, const int4 *recs)
    {
        if (nrhs >= 2) hipLaunchKernelGGL((k_fwd_update<NT, NBT, MINW, SWEEP_RK>), dim3(nwork), dim3(NT), (size_t) mx * nrhs * sizeof(double), s, T, nodes, prefix, nn, xsrc, x, ldx, nrhs, units, recs);
        else hipLaunchKernelGGL((k_fwd_update<NT, NBT, MINW>), dim3(nwork), dim3(NT), (size_t) mx * nrhs * sizeof(double), s, T, nodes, prefix, nn, xsrc, x, ldx, nrhs, units, recs);
    }
    static void bwd(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xcols, double *x, int64_t ldx, int nrhs,
                    const int2 *units, const int4 *recs)
    {
        if (nrhs >= 2) hipLaunchKernelGGL((k_bwd_update<NT, RBv, CBT, UNR, MINW, SWEEP_RK>), dim3(nwork), dim3(NT), 0, s, T, nodes, prefix, nn, xcols, x, ldx, nrhs, units, recs);
        else hipLaunchKernelGGL((k_bwd_update<NT, RBv, CBT, UNR, MINW>), dim3(nwork), dim3(NT), 0, s, T, nodes, prefix, nn, xcols, x, ldx, nrhs, units, recs);
    }
    static void sweep(hipStream_t s, bool lower, const DevTables &T, const int2 *dunits, int ndu, const int2 *units, int nunits, double *xa, double *xb, int64_t ldx, int nrhs,
                      int mx, const int4 *drecs, const int4 *urecs)
    {
        const size_t lds = (size_t) mx * nrhs * sizeof(double);
        if (nrhs >= 2) {
            if (lower) hipLaunchKernelGGL((k_sweep<true, NT, RBv, NBT, CBT, UNR, MINW, SWEEP_RK>), dim3(ndu + nunits), dim3(NT), lds, s, T, dunits, ndu, units, xa, xb, ldx, nrhs, drecs, urecs);
            else hipLaunchKernelGGL((k_sweep<false, NT, RBv, NBT, CBT, UNR, MINW, SWEEP_RK>), dim3(ndu + nunits), dim3(NT), lds, s, T, dunits, ndu, units, xa, xb, ldx, nrhs, drecs, urecs);
            return;
        }
        if (lower) hipLaunchKernelGGL((k_sweep<true, NT, RBv, NBT, CBT, UNR, MINW>), dim3(ndu + nunits), dim3(NT), lds, s, T, dunits, ndu, units, xa, xb, ldx, nrhs, drecs, urecs);
        else hipLaunchKernelGGL((k_sweep<false, NT, RBv, NBT, CBT, UNR, MINW>), dim3(ndu + nunits), dim3(NT), lds, s, T, dunits, ndu, units, xa, xb, ldx, nrhs, drecs, urecs);
    }
    static void join(hipStream_t s, bool lower, const DevTables &T, const int4 *jrecs, int nj, const int4 *jaux, const int4 *urecs, int nunits, double *xa, double *xb, int64_t ldx,
                     int nrhs, int mx)
    {
        const size_t lds = (size_t) mx * nrhs * sizeof(double);
        if (lower) hipLaunchKernelGGL((k_sweep_join<true, NT, RBv, NBT, CBT, UNR, MINW>), dim3(nj + nunits), dim3(NT), lds, s, T, jrecs, nj, jaux, urecs, xa, xb, ldx, nrhs);
        else hipLaunchKernelGGL((k_sweep_join<false, NT, RBv, NBT, CBT, UNR, MINW>), dim3(nj + nunits), dim3(NT), 0, s, T, jrecs, nj, jaux, urecs, xa, xb, ldx, nrhs);   // the backward units stage nothing in the dynamic segment
    }
    static int attrs()
    {
        HIPCHK(hipFuncSetAttribute((const void *) k_sweep_join<true, NT, RBv, NBT, CBT, UNR, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *) k_fwd_update<NT, NBT, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *) k_sweep<true, NT, RBv, NBT, CBT, UNR, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *) k_sweep<false, NT, RBv, NBT, CBT, UNR, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *) k_fwd_update<NT, NBT, MINW, SWEEP_RK>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *) k_sweep<true, NT, RBv, NBT, CBT, UNR, MINW, SWEEP_RK>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *) k_sweep<false, NT, RBv, NBT, CBT, UNR, MINW, SWEEP_RK>, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
        return 0;
    }
};
#define SWEEP_DISPATCH(v, CALL) \
    switch (v) { \
    case 1: SweepCfg<SWEEP_V1>::CALL; break; \
    case 5: SweepCfg<SWEEP_V5>::CALL; break; \
    case 10: SweepCfg<SWEEP_N0>::CALL; break; \
    default: SweepCfg<SWEEP_V0>::CALL; break; \
    }
int sweep_attrs()
{

    return SweepCfg<SWEEP_V0>::attrs() | SweepCfg<SWEEP_V1>::attrs() | SweepCfg<SWEEP_V5>::attrs() | SweepCfg<SWEEP_N0>::attrs();
}

void fwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xsrc, double *x, int64_t ldx, int nrhs, int mx,
                const int2 *units, const int4 *recs)
{
    if (nwork <= 0) return;
    SWEEP_DISPATCH(sweep_variant(nwork, mx, nrhs), fwd(s, T, nodes, prefix, nn, nwork, xsrc, x, ldx, nrhs, mx, units, recs))
}

void bwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, const double *xcols, double *x, int64_t ldx, int nrhs, int mx,
                const int2 *units, const int4 *recs)
{
    if (nwork <= 0) return;
    SWEEP_DISPATCH(sweep_variant(nwork, mx, nrhs), bwd(s, T, nodes, prefix, nn, nwork, xcols, x, ldx, nrhs, units, recs))
}

void sweep_step(hipStream_t s, bool lower, const DevTables &T, const int2 *dunits, int ndu, const int2 *units, int nunits,
                double *xa, double *xb, int64_t ldx, int nrhs, int mx, const int4 *drecs, const int4 *urecs)
{
    if (ndu + nunits <= 0) return;
    SWEEP_DISPATCH(sweep_variant(ndu + nunits, mx, nrhs), sweep(s, lower, T, dunits, ndu, units, nunits, xa, xb, ldx, nrhs, mx, drecs, urecs))
}

void sweep_join(hipStream_t s, bool lower, const DevTables &T, const int4 *jrecs, int nj, const int4 *jaux, const int4 *urecs, int nunits,
                double *xa, double *xb, int64_t ldx, int nrhs, int mx)
{
    if (nj + nunits <= 0) return;
    SWEEP_DISPATCH(sweep_variant(nj + nunits, mx), join(s, lower, T, jrecs, nj, jaux, urecs, nunits, xa, xb, ldx, nrhs, mx))
}

void zero_nodes(hipStream_t s, const DevTables &T, const int *nodes, int nn, double *x, int64_t ldx, int nrhs)
{
    if (nn > 0) hipLaunchKernelGGL(k_zero_nodes, dim3((nn + 3) / 4), dim3(256), 0, s, T.xsup, nodes, nn, x, ldx, nrhs);
}

void scatter_values(hipStream_t s, double *val, const int64_t *pos, const double *a, int64_t nnz)
{
    if (nnz > 0) hipLaunchKernelGGL(k_scatter_values, dim3((unsigned) ((nnz + 255) / 256)), dim3(256), 0, s, val, pos, a, nnz);
}

void rfs_residual(hipStream_t s, int n, const int *rp, const int *ci, const double *av, const double *x, const double *b, const int *pc,
                  double *r_perm, unsigned long long *s_out, double safe1, double safe2)
{
    hipLaunchKernelGGL(k_rfs_residual, dim3((n + 255) / 256), dim3(256), 0, s, n, rp, ci, av, x, b, pc, r_perm, s_out, safe1, safe2);
}

void rfs_update(hipStream_t s, int n, const int *pc, const double *dx_perm, double *x)
{
    hipLaunchKernelGGL(k_rfs_update, dim3((n + 255) / 256), dim3(256), 0, s, n, pc, dx_perm, x);
}

void add_atomic(hipStream_t s, int64_t n, const double *x, double *y)
{
    if (n <= 0) return;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_axpy_atomic, dim3((unsigned) (nb < 8192 ? nb : 8192)), dim3(256), 0, s, n, x, y);
}

void axpy(hipStream_t s, int64_t n, double a, const double *x, double *y)
{
    if (n <= 0) return;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_axpy, dim3((unsigned) (nb < 8192 ? nb : 8192)), dim3(256), 0, s, n, a, x, y);
}

void pack_diag(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, const int64_t *off, int nn, int nwork, double *stage, int vs)
{
    if (nwork <= 0) return;
    if (vs == 2) hipLaunchKernelGGL(k_pack_diag<2>, dim3(nwork), dim3(256), 0, s, T, nodes, prefix, off, nn, stage);
    else hipLaunchKernelGGL(k_pack_diag<1>, dim3(nwork), dim3(256), 0, s, T, nodes, prefix, off, nn, stage);
}

void xseg_copy(hipStream_t s, double *x, int64_t ldx, int nrhs, const int *runs, int nruns, int64_t total, double *buf, int mode)
{
    if (total <= 0 || nruns <= 0) return;
    const int64_t nb = (total * nrhs + 255) / 256;
    hipLaunchKernelGGL(k_xseg_copy, dim3((unsigned) (nb < 4096 ? nb : 4096)), dim3(256), 0, s, x, ldx, nrhs, runs, nruns, total, buf, mode);
}

void rows_copy(hipStream_t s, double *v, int64_t ldv, int nrhs, const int *idx, int64_t cnt, double *buf, int mode, int vs)
{
    if (cnt <= 0) return;
    const int64_t nb = (cnt * nrhs + 255) / 256;
    if (vs == 2) hipLaunchKernelGGL(k_rows_copy<2>, dim3((unsigned) (nb < 4096 ? nb : 4096)), dim3(256), 0, s, v, ldv, nrhs, idx, cnt, buf, mode);
    else hipLaunchKernelGGL(k_rows_copy<1>, dim3((unsigned) (nb < 4096 ? nb : 4096)), dim3(256), 0, s, v, ldv, nrhs, idx, cnt, buf, mode);
}

int mfma_selftest(const double *A, const double *B, double *D)
{
    double *dA, *dB, *dD;
    HIPCHK(hipMalloc((void **) &dA, 64 * 8)); HIPCHK(hipMalloc((void **) &dB, 64 * 8)); HIPCHK(hipMalloc((void **) &dD, 256 * 8));
    HIPCHK(hipMemcpy(dA, A, 64 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dB, B, 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_selftest, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    HIPCHK(hipMemcpy(D, dD, 256 * 8, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dB); hipFree(dD);
    return 0;
}

// ---- complex16 ----
// SLUAMD_ZLU4_MAX_NODES: levels of at most this many 33 .. 64-column blocks factor them with kz_diag_lu_wave4 (four waves per block).  OFF by default: measured
// (profiles/r05_ab_zlu4.txt) a lone block takes 53-55 us there against 48-52 us in the one-wave kernel -- an elimination step is bound by the v_readlane
// broadcasts of the owner's columns and by the pivot hand-off, not by the column updates the extra waves take over -- and pzgstrf3d 17.5 against 16.9 ms
static const int g_zlu4_max_nodes = getenv("SLUAMD_ZLU4_MAX_NODES") ? atoi(getenv("SLUAMD_ZLU4_MAX_NODES")) : 0;
void zdiag_lu(hipStream_t s, const DevTables &T, const int *nodes, int nn, int mx, int replace_tiny, double thresh, int *info)
{
    if (nn <= 0) return;
    // levels of narrow supernodes (the leaves and the small separators: tens of thousands of blocks): one wave per block in registers
    if (mx <= 8) hipLaunchKernelGGL(kz_diag_lu_wave_small<8>, dim3(nn), dim3(64), 0, s, T, nodes, nn, replace_tiny, thresh, info);
    else if (mx <= 16) hipLaunchKernelGGL(kz_diag_lu_wave_small<16>, dim3(nn), dim3(64), 0, s, T, nodes, nn, replace_tiny, thresh, info);
    else if (mx <= 32) hipLaunchKernelGGL(kz_diag_lu_wave_small<32>, dim3(nn), dim3(64), 0, s, T, nodes, nn, replace_tiny, thresh, info);
    else if (mx <= 64 && nn <= g_zlu4_max_nodes) hipLaunchKernelGGL(kz_diag_lu_wave4, dim3(nn), dim3(256), 0, s, T, nodes, nn, replace_tiny, thresh, info);   // few blocks: four waves per block (the panel chain)
    else if (mx <= 64) hipLaunchKernelGGL(kz_diag_lu_wave, dim3((nn + 3) / 4), dim3(256), 0, s, T, nodes, nn, replace_tiny, thresh, info);
    else hipLaunchKernelGGL(kz_diag_lu, dim3(nn), dim3(256), zdiag_lds_bytes(mx), s, T, nodes, replace_tiny, thresh, info, mx | 1);
}
static const bool g_ztrsm_quad = getenv("SLUAMD_NO_ZTRSM_QUAD") == nullptr;
void zpanel_trsm(hipStream_t s, const DevTables &T, const int *nodes, const int *lprefix, const int *uprefix, int nn, int nl, int nu, int mx)
{
    if (nl + nu <= 0) return;
    if (mx <= 16 && g_ztrsm_quad) hipLaunchKernelGGL(kz_panel_trsm_quad<4>, dim3(nl + nu), dim3(256), 0, s, T, nodes, lprefix, uprefix, nn, nl);       // four lanes per row / column, 16 x 17 triangle
    else if (mx <= 32 && g_ztrsm_quad) hipLaunchKernelGGL(kz_panel_trsm_quad<8>, dim3(nl + nu), dim3(256), 0, s, T, nodes, lprefix, uprefix, nn, nl);  // 32 x 33
    else if (mx <= 64 && g_ztrsm_quad) hipLaunchKernelGGL(kz_panel_trsm_quad<16>, dim3(nl + nu), dim3(256), 0, s, T, nodes, lprefix, uprefix, nn, nl); // 64 x 65
    else hipLaunchKernelGGL(kz_panel_trsm, dim3(nl + nu), dim3(64), 0, s, T, nodes, lprefix, uprefix, nn, nl);
}
void zschur(hipStream_t s, int cfg, const DevTables &T, const int *nodes, const int *prefix, int nn, int id_base, int ntiles, int *info,
            const int4 *ulist, int prio, const int *tmaps, int mmode, const int *xoff, int xmax)
{
    // the real kernel on the real embedding: tiles of 64 panel rows x 128 columns (cfg 0) / 32 x 64
    if (ntiles <= 0) return;
    if (!xoff || xmax <= 0) xoff = nullptr;
    const int grid = xoff ? 8 * xmax : ((ntiles + 7) / 8) * 8;
    if (cfg == 0) SCHUR_LAUNCH(128, 128, 8, true, 512);
    else SCHUR_LAUNCH(64, 64, 4, true, 256);
}
void zsolve_diag(hipStream_t s, bool lower, const DevTables &T, const int *nodes, int nn, void *x, int64_t ldx, int nrhs, int mx)
{
    if (nn <= 0) return;
    const size_t lds = (size_t) mx * nrhs * 16;
    if (mx <= 64) {     // one wave per supernode in registers
        if (lower) hipLaunchKernelGGL(kz_solve_diag_wave<true>, dim3((nn + 3) / 4), dim3(256), 0, s, T, nodes, nn, reinterpret_cast<zc *>(x), ldx, nrhs);
        else hipLaunchKernelGGL(kz_solve_diag_wave<false>, dim3((nn + 3) / 4), dim3(256), 0, s, T, nodes, nn, reinterpret_cast<zc *>(x), ldx, nrhs);
        return;
    }
    if (lower) hipLaunchKernelGGL(kz_solve_diag<true>, dim3(nn), dim3(256), lds, s, T, nodes, reinterpret_cast<zc *>(x), ldx, nrhs);
    else hipLaunchKernelGGL(kz_solve_diag<false>, dim3(nn), dim3(256), lds, s, T, nodes, reinterpret_cast<zc *>(x), ldx, nrhs);
}
void zfwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, int64_t ldx, int nrhs, int mx)
{
    if (nwork > 0) hipLaunchKernelGGL(kz_fwd_update, dim3(nwork), dim3(256), (size_t) mx * nrhs * 16, s, T, nodes, prefix, nn, reinterpret_cast<zc *>(x), ldx, nrhs);
}
void zsweep_fused(hipStream_t s, bool lower, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, void *w, int64_t ldx, int nrhs, int mx, int *tickets)
{
    if (nwork <= 0) return;
    if (lower) hipLaunchKernelGGL(kz_fwd_fused, dim3(nwork), dim3(256), (size_t) mx * nrhs * 16, s, T, nodes, prefix, nn, reinterpret_cast<zc *>(x), reinterpret_cast<zc *>(w), ldx, nrhs);
    else hipLaunchKernelGGL(kz_bwd_fused, dim3(nwork), dim3(256), 0, s, T, nodes, prefix, nn, reinterpret_cast<zc *>(x), reinterpret_cast<zc *>(w), ldx, nrhs, tickets);
}
void zbwd_update(hipStream_t s, const DevTables &T, const int *nodes, const int *prefix, int nn, int nwork, void *x, int64_t ldx, int nrhs)
{
    if (nwork > 0) hipLaunchKernelGGL(kz_bwd_update, dim3(nwork), dim3(256), 0, s, T, nodes, prefix, nn, reinterpret_cast<zc *>(x), ldx, nrhs);
}
void zscatter_values(hipStream_t s, void *val, const int64_t *pos, const void *a, int64_t nnz)
{
    if (nnz > 0) hipLaunchKernelGGL(kz_scatter_values, dim3((unsigned) ((nnz + 255) / 256)), dim3(256), 0, s, reinterpret_cast<zc *>(val), pos, reinterpret_cast<const zc *>(a), nnz);
}

}  // namespace eng
}  // namespace sluamd
