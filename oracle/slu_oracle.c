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

/* ---- double precision: SRC/double ---- */
#define T double
#define FN(x) slu_oracle_d##x
#define NM(x) x##_d
#define RECIP(x) (1.0 / (x))
#define DIVS(a, b) ((a) / (b))
#define IS_TINY(x, th) (fabs(x) < (th))                       /* pdgstrf2.c:544-560 */
#define TINY_REPLACEMENT(x, th) (((x) < 0) ? -(th) : (th))
#define ABS1(x) fabs(x)
#include "slu_oracle_body.inc"
#undef T
#undef FN
#undef NM
#undef RECIP
#undef DIVS
#undef IS_TINY
#undef TINY_REPLACEMENT
#undef ABS1

/* ---- complex16: SRC/complex16 (doublecomplex {r,i} == C99 double _Complex in memory) ---- */
#include <complex.h>
typedef double _Complex zc_t;
/* slud_z_div, SRC/complex16/dcomplex_dist.c:29-58 (Smith's algorithm) */
static zc_t z_div(zc_t a, zc_t b)
{
    double ar = creal(a), ai = cimag(a), br = creal(b), bi = cimag(b), ratio, den, cr, ci;
    if (fabs(br) <= fabs(bi)) {
        ratio = br / bi; den = bi * (1 + ratio * ratio);
        cr = (ar * ratio + ai) / den; ci = (ai * ratio - ar) / den;
    } else {
        ratio = bi / br; den = br * (1 + ratio * ratio);
        cr = (ar + ai * ratio) / den; ci = (ai - ar * ratio) / den;
    }
    return cr + ci * I;
}
#define T zc_t
#define FN(x) slu_oracle_z##x
#define NM(x) x##_z
#define RECIP(x) z_div(1.0, (x))
#define DIVS(a, b) z_div((a), (b))
/* pzgstrf2.c Local_Zgstrf2: |re|+|im| < thresh and both parts non-zero; replacement keeps the sign of the real part */
#define IS_TINY(x, th) ((fabs(creal(x)) + fabs(cimag(x))) < (th) && creal(x) != 0.0 && cimag(x) != 0.0)
#define TINY_REPLACEMENT(x, th) ((creal(x) < 0) ? -(th) : (th))
#define ABS1(x) (fabs(creal(x)) + fabs(cimag(x)))   /* slud_z_abs1, dcomplex_dist.c */
#include "slu_oracle_body.inc"

int slu_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
