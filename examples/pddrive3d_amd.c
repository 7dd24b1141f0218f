/*
 * pddrive3d_amd.c -- plain C driver over the C ABI of libsluamd.so (include/superlu_dist_amd.h), no Python involved:
 * the counterpart of the reference's EXAMPLE/pddrive3d.c for the part of the pipeline this library covers
 * (Equil = NO, RowPerm = NOROWPERM, ColPerm = NATURAL or MY_PERMC, 1 x 1 x 1 grid).
 *
 *   pddrive3d_amd N            7-point Poisson on an N^3 grid (what EXAMPLE/pddrive3d reads as a generated .dat), natural order
 *   pddrive3d_amd file.dat     triplet file, the format dreadtriple.c reads: "m n nnz" then "row col value", 0- or 1-based
 *
 * Pipeline: symbolic factorisation (sluamd_dsymbfact) -> device-resident distribution + handle
 * (sluamd_dCreateLUHandleFromSymb) -> sluamd_pdgstrf3d -> sluamd_pdgstrs3d -> sluamd_pdgsrfs3d (IterRefine = SLU_DOUBLE),
 * with xtrue_i = +-1 and b = A xtrue like dGenXtrue_dist / dFillRHS_dist (dutil_dist.c:598).  Prints what pddrive3d
 * prints at the end: ||X - Xtrue||_inf / ||X||_inf, plus the residual and the timings.  Exit code 0 iff residual < 1e-10.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "superlu_dist_amd.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_) { fprintf(stderr, "%s failed (rc=%d): %s\n", #call, rc_, sluamd_last_error()); return 2; } \
    } while (0)

static int poisson3d(int N, int64_t *n_out, int **rp_out, int **ci_out, double **v_out)
{
    const int64_t n = (int64_t) N * N * N;
    int *rp = (int *) malloc(sizeof(int) * (n + 1)), *ci = (int *) malloc(sizeof(int) * 7 * n);
    double *v = (double *) malloc(sizeof(double) * 7 * n);
    if (!rp || !ci || !v) return 1;
    int64_t p = 0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            for (int k = 0; k < N; ++k) {
                const int64_t r = ((int64_t) i * N + j) * N + k;
                rp[r] = (int) p;
                if (i > 0) { ci[p] = (int) (r - (int64_t) N * N); v[p++] = -1.0; }
                if (j > 0) { ci[p] = (int) (r - N); v[p++] = -1.0; }
                if (k > 0) { ci[p] = (int) (r - 1); v[p++] = -1.0; }
                ci[p] = (int) r; v[p++] = 6.0;
                if (k < N - 1) { ci[p] = (int) (r + 1); v[p++] = -1.0; }
                if (j < N - 1) { ci[p] = (int) (r + N); v[p++] = -1.0; }
                if (i < N - 1) { ci[p] = (int) (r + (int64_t) N * N); v[p++] = -1.0; }
            }
    rp[n] = (int) p;
    *n_out = n; *rp_out = rp; *ci_out = ci; *v_out = v;
    return 0;
}

typedef struct { int r, c; double v; } trip_t;
static int cmp_trip(const void *a, const void *b)
{
    const trip_t *x = (const trip_t *) a, *y = (const trip_t *) b;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return x->c < y->c ? -1 : (x->c > y->c);
}

static int read_triplets(const char *path, int64_t *n_out, int **rp_out, int **ci_out, double **v_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    long m, n, nnz;
    if (fscanf(f, "%ld %ld %ld", &m, &n, &nnz) != 3 || m != n) { fclose(f); return 1; }
    trip_t *t = (trip_t *) malloc(sizeof(trip_t) * (size_t) nnz);
    int minidx = 1 << 30;
    for (long e = 0; e < nnz; ++e) {
        if (fscanf(f, "%d %d %lf", &t[e].r, &t[e].c, &t[e].v) != 3) { fclose(f); free(t); return 1; }
        if (t[e].r < minidx) minidx = t[e].r;
        if (t[e].c < minidx) minidx = t[e].c;
    }
    fclose(f);
    for (long e = 0; e < nnz; ++e) { t[e].r -= minidx; t[e].c -= minidx; }   /* 0/1-based auto-detect, dreadtriple.c:85-92 */
    qsort(t, (size_t) nnz, sizeof(trip_t), cmp_trip);
    int *rp = (int *) calloc((size_t) n + 1, sizeof(int)), *ci = (int *) malloc(sizeof(int) * (size_t) nnz);
    double *v = (double *) malloc(sizeof(double) * (size_t) nnz);
    for (long e = 0; e < nnz; ++e) { rp[t[e].r + 1]++; ci[e] = t[e].c; v[e] = t[e].v; }
    for (long i = 0; i < n; ++i) rp[i + 1] += rp[i];
    free(t);
    *n_out = n; *rp_out = rp; *ci_out = ci; *v_out = v;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s N | file.dat\n", argv[0]); return 2; }
    if (sluamd_device_count() < 1) { fprintf(stderr, "no HIP device: this library has no CPU fallback\n"); return 3; }
    int64_t n; int *rp, *ci; double *v;
    char *end;
    const long N = strtol(argv[1], &end, 10);
    if (*end == '\0' && N > 0) { if (poisson3d((int) N, &n, &rp, &ci, &v)) return 2; }
    else if (read_triplets(argv[1], &n, &rp, &ci, &v)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

    /* xtrue, b = A xtrue, anorm (infinity norm, what pdgssvx3d passes to pdgstrf3d) */
    double *xt = (double *) malloc(sizeof(double) * n), *b = (double *) malloc(sizeof(double) * n), anorm = 0.0;
    for (int64_t i = 0; i < n; ++i) xt[i] = (i % 2) ? 1.0 : -1.0;
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0, rs = 0.0;
        for (int e = rp[i]; e < rp[i + 1]; ++e) { s += v[e] * xt[ci[e]]; rs += fabs(v[e]); }
        b[i] = s; if (rs > anorm) anorm = rs;
    }

    /* symbolic factorisation: ColPerm = NATURAL (identity perm_c in; the etree postorder comes back in perm_c) */
    int *perm_c = (int *) malloc(sizeof(int) * n), *perm_c_in = (int *) malloc(sizeof(int) * n);
    for (int64_t i = 0; i < n; ++i) perm_c_in[i] = (int) i;
    sluamd_symb_t symb;
    CHECK(sluamd_dsymbfact(&symb, n, rp, ci, perm_c_in, 32, 256, perm_c));
    int32_t nsupers; int64_t nnzL, nnzU; double flops;
    CHECK(sluamd_symb_info(symb, &nsupers, &nnzL, &nnzU, NULL, NULL, &flops));

    sluamd_options_t opt; sluamd_default_options(&opt);
    sluamd_handle_t h;
    CHECK(sluamd_dCreateLUHandleFromSymb(&h, symb, rp, ci, v, perm_c, &opt));

    int info = 0;
    const double thresh = 1.1920928955078125e-07 * anorm;   /* smach_dist("Epsilon") * anorm, pdgstrf3d.c:132 */
    CHECK(sluamd_pdgstrf3d(h, thresh, &info));
    if (info) { printf("INFO = %d returned from pdgstrf3d (zero pivot)\n", info); return 1; }

    /* solve: y = Pc b ; L U z = y ; x = Pc^T z */
    double *y = (double *) malloc(sizeof(double) * n), *x = (double *) malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; ++i) y[perm_c[i]] = b[i];
    CHECK(sluamd_pdgstrs3d(h, y, n, 1));
    for (int64_t i = 0; i < n; ++i) x[i] = y[perm_c[i]];

    /* IterRefine = SLU_DOUBLE */
    double berr = 0.0; int32_t steps = 0;
    CHECK(sluamd_dAttachMatrix(h, (sluamd_int_t) n, rp, ci, v, perm_c));
    CHECK(sluamd_pdgsrfs3d(h, b, n, x, n, 1, &berr, &steps));

    double err = 0.0, xn = 0.0, rn = 0.0, bn = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double s = b[i];
        for (int e = rp[i]; e < rp[i + 1]; ++e) s -= v[e] * x[ci[e]];
        rn += s * s; bn += b[i] * b[i];
        if (fabs(x[i] - xt[i]) > err) err = fabs(x[i] - xt[i]);
        if (fabs(x[i]) > xn) xn = fabs(x[i]);
    }
    sluamd_stats_t st;
    sluamd_get_stats(h, &st);
    printf("n = %lld  nnz(A) = %d  nsupers = %d  nnz(L+U) = %lld  flops = %.3e\n", (long long) n, rp[n], nsupers,
           (long long) (nnzL + nnzU), flops);
    printf("FACTOR time %.3f ms  (%.1f GFLOP/s)   SOLVE time %.3f ms   refinement steps %d  berr %.2e\n", st.t_factor_ms,
           flops / (st.t_factor_ms * 1e-3) / 1e9, st.t_solve_ms, steps, berr);
    printf("||X-Xtrue||/||X|| = %e   ||b-Ax||_2/||b||_2 = %e\n", err / xn, sqrt(rn / bn));
    sluamd_dDestroyLUHandle(h);
    sluamd_symb_free(symb);
    return sqrt(rn / bn) < 1e-10 ? 0 : 1;
}
