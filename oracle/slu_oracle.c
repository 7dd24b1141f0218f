/*
 * slu_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, written from scratch) of the arithmetic of the reference's
 * 3D supernodal LU hot path, used only as the checker for the HIP kernels (tests/,
 * __graft_entry__.smoke) and as bench.py's `cpu_baseline` ("port") leg.  Nothing in the
 * product library (superlu_dist_amd/csrc) includes, links or calls this file.
 *
 * Parity of this restatement is PINNED: tests/test_oracle_golden.py checks every function
 * below against tests/golden/*.npz, which were recorded from the real reference
 * (xiaoyeli/superlu_dist v9.2.1) run in the build container (oracle/ref/slu_ref_dump.c).
 *
 * Data layout = the reference's own L/U store for a 1x1x1 process grid (SURVEY.md App. A,
 * SRC/include/superlu_defs.h:156-198), passed as flat arrays + offsets:
 *   L block column k : idx = Lrowind + Lrowind_off[k] = [nblocks, nsupr]{[gid, nbrow] rows[nbrow]}*
 *                      val = Lnzval + Lnzval_off[k], column-major nsupr x nsupc, diag block first
 *   U block row   k  : idx = Ufstnz + Ufstnz_off[k] = [nblocks, nnz, len]{[gid, nnzblk] fstnz[nsupc(gid)]}*
 *                      val = Unzval + Unzval_off[k], concatenated column segments (skyline)
 *
 * Each function cites the reference file:line it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BC_HEADER 2      /* superlu_defs.h:169 */
#define LB_DESCRIPTOR 2  /* superlu_defs.h:170 */
#define BR_HEADER 3      /* superlu_defs.h:190 */
#define UB_DESCRIPTOR 2  /* superlu_defs.h:191 */

typedef struct {
    int n, nsupers;
    const int *xsup;
    const int64_t *Lrowind_off; const int *Lrowind;
    const int64_t *Lnzval_off;  double *Lnzval;
    const int64_t *Ufstnz_off;  const int *Ufstnz;
    const int64_t *Unzval_off;  double *Unzval;
} lu_t;

/* ---------------- dense kernels: loop order of the reference's vendored f2c CBLAS ------------- */

/* C = A*B, beta = 0, alpha = 1 ("N","N").  CBLAS/dgemm.c (Form C := alpha*A*B + beta*C). */
static void o_dgemm_nn(int m, int n, int k, const double *A, int lda, const double *B, int ldb,
                       double *C, int ldc)
{
    for (int j = 0; j < n; ++j) {
        double *c = C + (size_t) j * ldc;
        for (int i = 0; i < m; ++i) c[i] = 0.0;
        for (int l = 0; l < k; ++l) {
            double t = B[l + (size_t) j * ldb];
            if (t != 0.0) {
                const double *a = A + (size_t) l * lda;
                for (int i = 0; i < m; ++i) c[i] += t * a[i];
            }
        }
    }
}

/* B := B * inv(A), A upper, non-unit ("R","U","N","N").  CBLAS/dtrsm.c. */
static void o_dtrsm_runn(int m, int n, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < n; ++j) {
        double *bj = B + (size_t) j * ldb;
        for (int k = 0; k < j; ++k) {
            double a = A[k + (size_t) j * lda];
            if (a != 0.0) {
                const double *bk = B + (size_t) k * ldb;
                for (int i = 0; i < m; ++i) bj[i] -= a * bk[i];
            }
        }
        double t = 1.0 / A[j + (size_t) j * lda];
        for (int i = 0; i < m; ++i) bj[i] *= t;
    }
}

/* B := inv(A) * B, A lower, unit ("L","L","N","U").  CBLAS/dtrsm.c. */
static void o_dtrsm_llnu(int m, int n, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < n; ++j) {
        double *b = B + (size_t) j * ldb;
        for (int k = 0; k < m; ++k) {
            if (b[k] != 0.0) {
                const double *a = A + (size_t) k * lda;
                for (int i = k + 1; i < m; ++i) b[i] -= b[k] * a[i];
            }
        }
    }
}

/* B := inv(A) * B, A upper, non-unit ("L","U","N","N").  CBLAS/dtrsm.c. */
static void o_dtrsm_lunn(int m, int n, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < n; ++j) {
        double *b = B + (size_t) j * ldb;
        for (int k = m - 1; k >= 0; --k) {
            if (b[k] != 0.0) {
                const double *a = A + (size_t) k * lda;
                b[k] /= a[k];
                for (int i = 0; i < k; ++i) b[i] -= b[k] * a[i];
            }
        }
    }
}

/* -------------------------------- panel factorisation ----------------------------------------- */

/* Unpivoted right-looking LU of the nsupc x nsupc diagonal block that heads L(:,k);
 * restates Local_Dgstrf2, SRC/double/pdgstrf2.c:508-601 (dger rank-1 updates, tiny-pivot
 * replacement :544-560, zero-pivot info :568-571).  Returns #tiny pivots replaced. */
static int o_dgstrf2(const lu_t *lu, int k, int replace_tiny, double thresh, int *info)
{
    const int *idx = lu->Lrowind + lu->Lrowind_off[k];
    int nsupr = idx[1];
    int jfst = lu->xsup[k], nsupc = lu->xsup[k + 1] - jfst;
    double *lusup = lu->Lnzval + lu->Lnzval_off[k];
    int tiny = 0;
    for (int j = 0; j < nsupc; ++j) {
        double *col = lusup + (size_t) j * nsupr;      /* column j, diag entry at col[j] */
        if (replace_tiny && fabs(col[j]) < thresh) {
            col[j] = (col[j] < 0) ? -thresh : thresh;
            ++tiny;
        }
        if (col[j] == 0.0) {
            *info = j + jfst + 1;
        } else {
            double t = 1.0 / col[j];
            for (int i = j + 1; i < nsupc; ++i) col[i] *= t;
        }
        /* rank-1 update of the trailing block (superlu_dger, alpha = -1) */
        for (int c = j + 1; c < nsupc; ++c) {
            double *cc = lusup + (size_t) c * nsupr;
            double y = cc[j];                           /* U(j,c) */
            if (y != 0.0) {
                double t = -y;
                for (int i = j + 1; i < nsupc; ++i) cc[i] += col[i] * t;
            }
        }
    }
    return tiny;
}

/* L(offdiag rows, k) <- L * inv(U_kk); restates dLPanelTrSolve (iam == pkk branch),
 * SRC/double/dtrfCommWrapper.c:183-219 (TRSM R,U,N,N in 32-row strips). */
static void o_lpanel_trsm(const lu_t *lu, int k)
{
    const int *idx = lu->Lrowind + lu->Lrowind_off[k];
    int nsupr = idx[1];
    int nsupc = lu->xsup[k + 1] - lu->xsup[k];
    double *lusup = lu->Lnzval + lu->Lnzval_off[k];
    int l = nsupr - nsupc;
    for (int off = 0; off < l; off += 32) {
        int len = (l - off < 32) ? (l - off) : 32;
        o_dtrsm_runn(len, nsupc, lusup, nsupr, lusup + nsupc + off, nsupr);
    }
}

static int row_ldu(const lu_t *lu, int k)
{   /* max segment height over U(k,:) : SRC/double/dgather.c:312-321 */
    if (lu->Ufstnz_off[k + 1] == lu->Ufstnz_off[k]) return 0;
    const int *usub = lu->Ufstnz + lu->Ufstnz_off[k];
    int nb = usub[0], klst = lu->xsup[k + 1], ldu = 0;
    int p = BR_HEADER;
    for (int b = 0; b < nb; ++b) {
        int jb = usub[p], ns = lu->xsup[jb + 1] - lu->xsup[jb];
        for (int jj = 0; jj < ns; ++jj) {
            int seg = klst - usub[p + UB_DESCRIPTOR + jj];
            if (seg > ldu) ldu = seg;
        }
        p += UB_DESCRIPTOR + ns;
    }
    return ldu;
}

/* U(k,:) <- inv(L_kk) * U(k,:), block by block: gather skyline -> dense (zero-padded to the
 * block's own ldu), TRSM(L,L,N,U), scatter back; restates dTrs2_GatherTrsmScatter,
 * SRC/double/pdgstrf2.c:804-840 (+ dTrs2_GatherU :757-779, dTrs2_ScatterU :781-802),
 * driven like dUPanelTrSolve, SRC/double/dtrfCommWrapper.c:305-353. */
static void o_upanel_trsm(const lu_t *lu, int k, double *tempv)
{
    if (lu->Ufstnz_off[k + 1] == lu->Ufstnz_off[k]) return;
    const int *usub = lu->Ufstnz + lu->Ufstnz_off[k];
    double *uval = lu->Unzval + lu->Unzval_off[k];
    const int *lidx = lu->Lrowind + lu->Lrowind_off[k];
    int nsupr = lidx[1];
    const double *lusup = lu->Lnzval + lu->Lnzval_off[k];
    int klst = lu->xsup[k + 1], knsupc = klst - lu->xsup[k];
    int nb = usub[0], iukp = BR_HEADER;
    int64_t rukp = 0;
    for (int b = 0; b < nb; ++b) {
        int gb = usub[iukp], nsupc = lu->xsup[gb + 1] - lu->xsup[gb];
        const int *fst = usub + iukp + UB_DESCRIPTOR;
        int ldu = 0, ncols = 0;
        for (int jj = 0; jj < nsupc; ++jj) { int s = klst - fst[jj]; if (s > ldu) ldu = s; }
        /* gather */
        double *t = tempv; int64_t r = rukp;
        for (int jj = 0; jj < nsupc; ++jj) {
            int seg = klst - fst[jj];
            if (seg) {
                int lead = ldu - seg;
                for (int i = 0; i < lead; ++i) t[i] = 0.0;
                for (int i = 0; i < seg; ++i) t[lead + i] = uval[r + i];
                r += seg; t += ldu; ++ncols;
            }
        }
        o_dtrsm_llnu(ldu, ncols, lusup + (size_t) (knsupc - ldu) * (nsupr + 1), nsupr, tempv, ldu);
        /* scatter back */
        t = tempv; r = rukp;
        for (int jj = 0; jj < nsupc; ++jj) {
            int seg = klst - fst[jj];
            if (seg) {
                int lead = ldu - seg;
                for (int i = 0; i < seg; ++i) uval[r + i] = t[lead + i];
                r += seg; t += ldu;
            }
        }
        rukp = r;
        iukp += UB_DESCRIPTOR + nsupc;
    }
}

/* ------------------------------------ Schur update -------------------------------------------- */

/* subtract tile into L(ib, jb): restates dscatter_l, SRC/double/dscatter.c:109-194. */
static void o_scatter_l(const lu_t *lu, int ib, int jb, const int *fst_src, int klst, int nbrow_ld,
                        const int *lrows, int temp_nbrow, const double *tempv, int *indirect, int *indirect2)
{
    const int *index = lu->Lrowind + lu->Lrowind_off[jb];
    int ldv = index[1], nblk = index[0];
    int lptrj = BC_HEADER; int64_t luptrj = 0;
    int b = 0;
    while (index[lptrj] != ib) {
        if (++b == nblk) return;
        luptrj += index[lptrj + 1];
        lptrj += LB_DESCRIPTOR + index[lptrj + 1];
    }
    int fnz = lu->xsup[ib];
    int dest_nbrow = index[lptrj + 1];
    lptrj += LB_DESCRIPTOR;
    for (int i = 0; i < dest_nbrow; ++i) indirect[index[lptrj + i] - fnz] = i;
    for (int i = 0; i < temp_nbrow; ++i) indirect2[i] = indirect[lrows[i] - fnz];
    double *nzval = lu->Lnzval + lu->Lnzval_off[jb] + luptrj;
    int nsupc = lu->xsup[jb + 1] - lu->xsup[jb];
    for (int jj = 0; jj < nsupc; ++jj) {
        if (klst - fst_src[jj]) {
            for (int i = 0; i < temp_nbrow; ++i) nzval[indirect2[i]] -= tempv[i];
            tempv += nbrow_ld;
        }
        nzval += ldv;
    }
}

/* subtract tile into U(ib, jb): restates scatter_u, SRC/double/dscatter3d.c:555-631
 * (== dscatter_u, SRC/double/dscatter.c:197-292). */
static void o_scatter_u(const lu_t *lu, int ib, int jb, const int *fst_src, int klst, int nbrow_ld,
                        const int *lrows, int temp_nbrow, const double *tempv)
{
    if (lu->Ufstnz_off[ib + 1] == lu->Ufstnz_off[ib]) return;
    const int *index = lu->Ufstnz + lu->Ufstnz_off[ib];
    int nblk = index[0], ilst = lu->xsup[ib + 1];
    int iuip = BR_HEADER; int64_t ruip = 0;
    int b = 0;
    while (index[iuip] < jb) {
        if (++b == nblk) return;
        ruip += index[iuip + 1];
        iuip += UB_DESCRIPTOR + (lu->xsup[index[iuip] + 1] - lu->xsup[index[iuip]]);
    }
    iuip += UB_DESCRIPTOR;
    int nsupc = lu->xsup[jb + 1] - lu->xsup[jb];
    double *uv = lu->Unzval + lu->Unzval_off[ib];
    for (int jj = 0; jj < nsupc; ++jj) {
        int fnz = index[iuip++];
        if (klst - fst_src[jj]) {
            double *ucol = uv + ruip;
            for (int i = 0; i < temp_nbrow; ++i) ucol[lrows[i] - fnz] -= tempv[i];
            tempv += nbrow_ld;
        }
        ruip += ilst - fnz;
    }
}

typedef struct { int lptr, ib, nbrow, luptr; } lblk_t;
typedef struct { int iukp, jb, ncols, st_col; int64_t rukp; } ublk_t;

/* A(I,J) -= L(I,k) U(k,J) for every (L block, U block) pair of supernode k:
 * gather (dRgather_L/U, SRC/double/dgather.c:133-398: here L is used in place, U is packed to
 * bigU[ldu x ncols] zero-padded), then per pair GEMM + scatter (dblock_gemm_scatter,
 * SRC/double/dscatter3d.c:81-189) under `omp parallel for schedule(dynamic,2)` like
 * SRC/double/dtreeFactorization.c:497-587.  Returns the reference's flop tally for this k
 * (scuStatUpdate, SRC/prec-independent/sec_structs.c:692-693). */
static double o_schur_update(const lu_t *lu, int k, int ldt, double **bigU_p, size_t *bigU_cap,
                             double *bigV, int *indirect, int *indirect2)
{
    const int *lsub = lu->Lrowind + lu->Lrowind_off[k];
    int nlb = lsub[0], nsupr = lsub[1];
    int klst = lu->xsup[k + 1], knsupc = klst - lu->xsup[k];
    const double *lusup = lu->Lnzval + lu->Lnzval_off[k];
    if (lu->Ufstnz_off[k + 1] == lu->Ufstnz_off[k] || nlb <= 1) return 0.0;
    const int *usub = lu->Ufstnz + lu->Ufstnz_off[k];
    const double *uval = lu->Unzval + lu->Unzval_off[k];
    int nub = usub[0];
    lblk_t *LB = (lblk_t *) malloc(sizeof(lblk_t) * (size_t) nlb);
    ublk_t *UB = (ublk_t *) malloc(sizeof(ublk_t) * (size_t) nub);
    int nl = 0, p = BC_HEADER, luptr = 0, Rnbrow = 0;
    for (int b = 0; b < nlb; ++b) {
        int ib = lsub[p], nbrow = lsub[p + 1];
        if (ib != k) { LB[nl].lptr = p + LB_DESCRIPTOR; LB[nl].ib = ib; LB[nl].nbrow = nbrow; LB[nl].luptr = luptr; ++nl; Rnbrow += nbrow; }
        luptr += nbrow; p += LB_DESCRIPTOR + nbrow;
    }
    int ldu = row_ldu(lu, k), ncols_tot = 0;
    int iukp = BR_HEADER; int64_t rukp = 0;
    for (int b = 0; b < nub; ++b) {
        int jb = usub[iukp], ns = lu->xsup[jb + 1] - lu->xsup[jb], nc = 0; int64_t r = rukp;
        for (int jj = 0; jj < ns; ++jj) { int s = klst - usub[iukp + UB_DESCRIPTOR + jj]; if (s) { ++nc; r += s; } }
        UB[b].iukp = iukp + UB_DESCRIPTOR; UB[b].jb = jb; UB[b].ncols = nc; UB[b].st_col = ncols_tot; UB[b].rukp = rukp;
        ncols_tot += nc; rukp = r; iukp += UB_DESCRIPTOR + ns;
    }
    if ((size_t) ldu * ncols_tot > *bigU_cap) {
        *bigU_cap = (size_t) ldu * ncols_tot * 2 + 1024;
        *bigU_p = (double *) realloc(*bigU_p, sizeof(double) * *bigU_cap);
    }
    double *bigU = *bigU_p;
    /* dgather_u, SRC/double/dgather.c:45-89 */
    for (int b = 0; b < nub; ++b) {
        double *t = bigU + (size_t) ldu * UB[b].st_col; int64_t r = UB[b].rukp;
        int ns = lu->xsup[UB[b].jb + 1] - lu->xsup[UB[b].jb];
        for (int jj = 0; jj < ns; ++jj) {
            int seg = klst - usub[UB[b].iukp + jj];
            if (seg) {
                int lead = ldu - seg;
                for (int i = 0; i < lead; ++i) t[i] = 0.0;
                for (int i = 0; i < seg; ++i) t[lead + i] = uval[r + i];
                r += seg; t += ldu;
            }
        }
    }
    long npairs = (long) nl * nub;
#pragma omp parallel for schedule(dynamic, 2)
    for (long pr = 0; pr < npairs; ++pr) {
        int lb = (int) (pr / nub), j = (int) (pr % nub);
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        double *tempv = bigV + (size_t) tid * ldt * ldt;
        int *ind = indirect + (size_t) tid * ldt, *ind2 = indirect2 + (size_t) tid * ldt;
        int nbrow = LB[lb].nbrow, nc = UB[j].ncols;
        if (!nc) continue;
        o_dgemm_nn(nbrow, nc, ldu, lusup + (size_t) (knsupc - ldu) * nsupr + LB[lb].luptr, nsupr,
                   bigU + (size_t) ldu * UB[j].st_col, ldu, tempv, nbrow);
        if (LB[lb].ib < UB[j].jb)
            o_scatter_u(lu, LB[lb].ib, UB[j].jb, usub + UB[j].iukp, klst, nbrow, lsub + LB[lb].lptr, nbrow, tempv);
        else
            o_scatter_l(lu, LB[lb].ib, UB[j].jb, usub + UB[j].iukp, klst, nbrow, lsub + LB[lb].lptr, nbrow, tempv, ind, ind2);
    }
    free(LB); free(UB);
    return 2.0 * (double) Rnbrow * ldu * ncols_tot;
}

/* ------------------------------------- drivers ------------------------------------------------- */

/* Numeric factorisation of the supernodes in `order[0..norder)` on a 1x1x1 grid: per k the sequence
 * dDiagFactIBCast -> dLPanelUpdate -> dUPanelUpdate -> Schur update of
 * dsparseTreeFactor_ASYNC, SRC/double/dtreeFactorization.c:295-716 (look-ahead only reorders
 * independent work), called per forest by pdgstrf3d, SRC/double/pdgstrf3d.c:333-385.
 * `order` must be a valid elimination order (e.g. sForest_t.nodeList).  flops_out[0] += reference-style
 * padded Schur flops, flops_out[1] += diag+panel flops.  Returns #tiny pivots. */
int slu_oracle_dfactor(int n, int nsupers, const int *xsup,
                       const int64_t *Lrowind_off, const int *Lrowind, const int64_t *Lnzval_off, double *Lnzval,
                       const int64_t *Ufstnz_off, const int *Ufstnz, const int64_t *Unzval_off, double *Unzval,
                       const int *order, int norder, int replace_tiny, double thresh, int *info, double *flops_out)
{
    lu_t lu = { n, nsupers, xsup, Lrowind_off, Lrowind, Lnzval_off, Lnzval, Ufstnz_off, Ufstnz, Unzval_off, Unzval };
    int ldt = 1;
    for (int k = 0; k < nsupers; ++k) if (xsup[k + 1] - xsup[k] > ldt) ldt = xsup[k + 1] - xsup[k];
    int maxrow = ldt;
    for (int k = 0; k < nsupers; ++k) {
        const int *idx = Lrowind + Lrowind_off[k];
        int p = BC_HEADER;
        for (int b = 0; b < idx[0]; ++b) { if (idx[p + 1] > maxrow) maxrow = idx[p + 1]; p += LB_DESCRIPTOR + idx[p + 1]; }
    }
    int tdim = maxrow > ldt ? maxrow : ldt;
#ifdef _OPENMP
    int nth = omp_get_max_threads();
#else
    int nth = 1;
#endif
    double *bigV = (double *) malloc(sizeof(double) * (size_t) tdim * tdim * nth);
    int *indirect = (int *) malloc(sizeof(int) * (size_t) tdim * nth * 2);
    int *indirect2 = indirect + (size_t) tdim * nth;
    double *bigU = NULL; size_t bigU_cap = 0;
    int tiny = 0;
    for (int t = 0; t < norder; ++t) {
        int k = order[t];
        int nsupc = xsup[k + 1] - xsup[k];
        const int *idx = Lrowind + Lrowind_off[k];
        int nsupr = idx[1];
        tiny += o_dgstrf2(&lu, k, replace_tiny, thresh, info);
        o_lpanel_trsm(&lu, k);
        o_upanel_trsm(&lu, k, bigV);
        if (flops_out) {
            flops_out[1] += (2.0 / 3.0) * nsupc * (double) nsupc * nsupc + (double) nsupc * nsupc * (nsupr - nsupc);
            if (Unzval_off[k + 1] > Unzval_off[k]) flops_out[1] += (double) nsupc * (Unzval_off[k + 1] - Unzval_off[k]);
        }
        double f = o_schur_update(&lu, k, tdim, &bigU, &bigU_cap, bigV, indirect, indirect2);
        if (flops_out) flops_out[0] += f;
    }
    free(bigV); free(indirect); free(bigU);
    return tiny;
}

/* Triangular solves L y = b, U x = y on the permuted system, x overwritten (n x nrhs, ld ldx).
 * Forward step per supernode: x_k <- inv(L_kk) x_k (dtrsm L,L,N,U) then lsum_i -= L_ik x_k
 * (dlsum_fmod_inv, SRC/double/pdgstrs_lsum.c:414-700; leaf/non-leaf forward solves
 * SRC/double/pdgstrs3d.c:1819-2179).  Backward step: x_k <- inv(U_kk)(x_k - sum_j U_kj x_j)
 * (dlsum_bmod_inv, SRC/double/pdgstrs_lsum.c:1362-1700; dlsumBmod SRC/double/pdgstrs3d.c:3987).
 * Message-driven scheduling in the reference only reorders independent updates.
 * `nodes` (ascending, may be NULL = all supernodes) restricts the sweep to one elimination forest, which is how
 * the 3D solve walks the Z levels (pdgsTrForwardSolve3d / pdgsTrBackSolve3d, pdgstrs3d.c:7312 / :7564). */
void slu_oracle_dsolve_fwd(int n, int nsupers, const int *xsup,
                           const int64_t *Lrowind_off, const int *Lrowind, const int64_t *Lnzval_off, const double *Lnzval,
                           const int64_t *Ufstnz_off, const int *Ufstnz, const int64_t *Unzval_off, const double *Unzval,
                           double *x, int ldx, int nrhs, const int *nodes, int nnodes)
{
    (void) n; (void) Ufstnz_off; (void) Ufstnz; (void) Unzval_off; (void) Unzval;
    int ldt = 1, maxr = 1;
    for (int k = 0; k < nsupers; ++k) {
        if (xsup[k + 1] - xsup[k] > ldt) ldt = xsup[k + 1] - xsup[k];
        if (Lrowind_off[k + 1] > Lrowind_off[k]) { int r = (Lrowind + Lrowind_off[k])[1]; if (r > maxr) maxr = r; }
    }
    double *rtemp = (double *) malloc(sizeof(double) * (size_t) maxr * nrhs);
    double *xk = (double *) malloc(sizeof(double) * (size_t) ldt * nrhs);
    const int cnt = nodes ? nnodes : nsupers;
    for (int t = 0; t < cnt; ++t) {
        const int k = nodes ? nodes[t] : t;
        const int *lsub = Lrowind + Lrowind_off[k];
        const double *lusup = Lnzval + Lnzval_off[k];
        int nb = lsub[0], nsupr = lsub[1], fst = xsup[k], nsupc = xsup[k + 1] - fst;
        o_dtrsm_llnu(nsupc, nrhs, lusup, nsupr, x + fst, ldx);
        for (int j = 0; j < nrhs; ++j) memcpy(xk + (size_t) j * nsupc, x + fst + (size_t) j * ldx, sizeof(double) * nsupc);
        int p = BC_HEADER, luptr = 0;
        for (int b = 0; b < nb; ++b) {
            int ib = lsub[p], nbrow = lsub[p + 1];
            if (ib != k) {
                o_dgemm_nn(nbrow, nrhs, nsupc, lusup + luptr, nsupr, xk, nsupc, rtemp, nbrow);
                for (int j = 0; j < nrhs; ++j)
                    for (int i = 0; i < nbrow; ++i)
                        x[lsub[p + LB_DESCRIPTOR + i] + (size_t) j * ldx] -= rtemp[i + (size_t) j * nbrow];
            }
            luptr += nbrow; p += LB_DESCRIPTOR + nbrow;
        }
    }
    free(rtemp); free(xk);
}

void slu_oracle_dsolve_bwd(int n, int nsupers, const int *xsup,
                           const int64_t *Lrowind_off, const int *Lrowind, const int64_t *Lnzval_off, const double *Lnzval,
                           const int64_t *Ufstnz_off, const int *Ufstnz, const int64_t *Unzval_off, const double *Unzval,
                           double *x, int ldx, int nrhs, const int *nodes, int nnodes)
{
    (void) n;
    const int cnt = nodes ? nnodes : nsupers;
    for (int t = cnt - 1; t >= 0; --t) {
        const int k = nodes ? nodes[t] : t;
        int fst = xsup[k], klst = xsup[k + 1], nsupc = klst - fst;
        if (Ufstnz_off[k + 1] > Ufstnz_off[k]) {
            const int *usub = Ufstnz + Ufstnz_off[k];
            const double *uval = Unzval + Unzval_off[k];
            int nb = usub[0], iukp = BR_HEADER; int64_t rukp = 0;
            for (int b = 0; b < nb; ++b) {
                int jb = usub[iukp], ns = xsup[jb + 1] - xsup[jb];
                for (int jj = 0; jj < ns; ++jj) {
                    int fnz = usub[iukp + UB_DESCRIPTOR + jj];
                    int seg = klst - fnz;
                    if (seg) {
                        for (int j = 0; j < nrhs; ++j) {
                            double xj = x[xsup[jb] + jj + (size_t) j * ldx];
                            double *dst = x + (size_t) j * ldx;
                            for (int i = 0; i < seg; ++i) dst[fnz + i] -= uval[rukp + i] * xj;
                        }
                        rukp += seg;
                    }
                }
                iukp += UB_DESCRIPTOR + ns;
            }
        }
        const int *lsub = Lrowind + Lrowind_off[k];
        o_dtrsm_lunn(nsupc, nrhs, Lnzval + Lnzval_off[k], lsub[1], x + fst, ldx);
    }
}

void slu_oracle_dsolve(int n, int nsupers, const int *xsup,
                       const int64_t *Lrowind_off, const int *Lrowind, const int64_t *Lnzval_off, const double *Lnzval,
                       const int64_t *Ufstnz_off, const int *Ufstnz, const int64_t *Unzval_off, const double *Unzval,
                       double *x, int ldx, int nrhs)
{
    slu_oracle_dsolve_fwd(n, nsupers, xsup, Lrowind_off, Lrowind, Lnzval_off, Lnzval, Ufstnz_off, Ufstnz, Unzval_off, Unzval, x, ldx, nrhs, NULL, 0);
    slu_oracle_dsolve_bwd(n, nsupers, xsup, Lrowind_off, Lrowind, Lnzval_off, Lnzval, Ufstnz_off, Ufstnz, Unzval_off, Unzval, x, ldx, nrhs, NULL, 0);
}

int slu_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
