/*
 * pddrive3d_amd.c -- plain C driver over the C ABI of libsluamd.so (include/superlu_dist_amd.h), no Python involved:
 * the counterpart of the reference's EXAMPLE/pddrive3d.c for the part of the pipeline this library covers
 * (Equil = NO, RowPerm = NOROWPERM, ColPerm = NATURAL or MY_PERMC, 1 x 1 x 1 grid).
 *
 *   pddrive3d_amd N            7-point Poisson on an N^3 grid (what EXAMPLE/pddrive3d reads as a generated .dat), natural order
 *   pddrive3d_amd file.dat     triplet file, the format dreadtriple.c reads: "m n nnz" then "row col value", 0- or 1-based
 *   pddrive3d_amd file.mtx     MatrixMarket coordinate file (real / integer / pattern; general / symmetric / skew-symmetric), what
 *                              dreadMM.c reads (SRC/double/dreadMM.c: symmetric storage is expanded to the full pattern) -- e.g.
 *                              SuiteSparse audikw_1.mtx (BASELINE.json configs[3]; not shipped: no network)
 * Files are ordered with sluamd_order_nd (nested dissection of the pattern of A + A^T; the generated grid keeps its natural order
 * unless -nd is given).
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

/* MatrixMarket coordinate format (what dreadMM_dist reads, SRC/double/dreadMM.c:40-230): banner "%%MatrixMarket matrix coordinate
 * <real|integer|pattern> <general|symmetric|skew-symmetric>", comment lines, "m n nnz", then 1-based "row col [value]" lines;
 * symmetric / skew-symmetric files store one triangle and are expanded; pattern files get value 1 (diagonal: row degree + 1, so
 * that the unpivoted factorisation is defined).  Duplicates are summed. */
static int read_matrix_market(const char *path, int64_t *n_out, int **rp_out, int **ci_out, double **v_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    char line[1024], obj[64], fmt[64], field[64], sym[64];
    if (!fgets(line, sizeof line, f) || sscanf(line, "%%%%MatrixMarket %63s %63s %63s %63s", obj, fmt, field, sym) != 4) { fclose(f); return 1; }
    for (char *c = field; *c; ++c) if (*c >= 'A' && *c <= 'Z') *c += 32;
    for (char *c = sym; *c; ++c) if (*c >= 'A' && *c <= 'Z') *c += 32;
    if (strcmp(fmt, "coordinate") || (strcmp(field, "real") && strcmp(field, "integer") && strcmp(field, "pattern"))) { fclose(f); return 1; }
    const int pattern = !strcmp(field, "pattern");
    const int symm = !strcmp(sym, "symmetric"), skew = !strcmp(sym, "skew-symmetric");
    if (!symm && !skew && strcmp(sym, "general")) { fclose(f); return 1; }
    do { if (!fgets(line, sizeof line, f)) { fclose(f); return 1; } } while (line[0] == '%');
    long m, n, nz;
    if (sscanf(line, "%ld %ld %ld", &m, &n, &nz) != 3 || m != n) { fclose(f); return 1; }
    trip_t *t = (trip_t *) malloc(sizeof(trip_t) * (size_t) (2 * nz + 1));
    if (!t) { fclose(f); return 1; }
    long k = 0;
    for (long e = 0; e < nz; ++e) {
        int r, c; double val = 1.0;
        if (pattern ? fscanf(f, "%d %d", &r, &c) != 2 : fscanf(f, "%d %d %lf", &r, &c, &val) != 3) { fclose(f); free(t); return 1; }
        if (r < 1 || r > n || c < 1 || c > n) { fclose(f); free(t); return 1; }
        t[k].r = r - 1; t[k].c = c - 1; t[k].v = val; ++k;
        if ((symm || skew) && r != c) { t[k].r = c - 1; t[k].c = r - 1; t[k].v = skew ? -val : val; ++k; }
    }
    fclose(f);
    qsort(t, (size_t) k, sizeof(trip_t), cmp_trip);
    long u = 0;                                          /* sum duplicates */
    for (long e = 0; e < k; ++e) { if (u && t[u - 1].r == t[e].r && t[u - 1].c == t[e].c) t[u - 1].v += t[e].v; else t[u++] = t[e]; }
    int *rp = (int *) calloc((size_t) n + 1, sizeof(int)), *ci = (int *) malloc(sizeof(int) * (size_t) (u ? u : 1));
    double *v = (double *) malloc(sizeof(double) * (size_t) (u ? u : 1));
    for (long e = 0; e < u; ++e) { rp[t[e].r + 1]++; ci[e] = t[e].c; v[e] = t[e].v; }
    for (long i = 0; i < n; ++i) rp[i + 1] += rp[i];
    if (pattern) for (long i = 0; i < n; ++i) for (int e = rp[i]; e < rp[i + 1]; ++e) if (ci[e] == i) v[e] = (double) (rp[i + 1] - rp[i]) + 1.0;
    free(t);
    *n_out = n; *rp_out = rp; *ci_out = ci; *v_out = v;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s N [-nd] | file.dat | file.mtx\n", argv[0]); return 2; }
    if (sluamd_device_count() < 1) { fprintf(stderr, "no HIP device: this library has no CPU fallback\n"); return 3; }
    int64_t n; int *rp, *ci; double *v;
    char *end;
    const long N = strtol(argv[1], &end, 10);
    int use_nd = argc > 2 && !strcmp(argv[2], "-nd");
    const size_t len = strlen(argv[1]);
    if (*end == '\0' && N > 0) { if (poisson3d((int) N, &n, &rp, &ci, &v)) return 2; }
    else if (len > 4 && !strcmp(argv[1] + len - 4, ".mtx")) {
        if (read_matrix_market(argv[1], &n, &rp, &ci, &v)) { fprintf(stderr, "cannot read MatrixMarket file %s\n", argv[1]); return 2; }
        use_nd = 1;
    } else {
        if (read_triplets(argv[1], &n, &rp, &ci, &v)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
        use_nd = 1;
    }

    /* xtrue, b = A xtrue, anorm (infinity norm, what pdgssvx3d passes to pdgstrf3d) */
    double *xt = (double *) malloc(sizeof(double) * n), *b = (double *) malloc(sizeof(double) * n), anorm = 0.0;
    for (int64_t i = 0; i < n; ++i) xt[i] = (i % 2) ? 1.0 : -1.0;
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0, rs = 0.0;
        for (int e = rp[i]; e < rp[i + 1]; ++e) { s += v[e] * xt[ci[e]]; rs += fabs(v[e]); }
        b[i] = s; if (rs > anorm) anorm = rs;
    }

    /* ColPerm: NATURAL (generated grid) or our nested dissection (files), MY_PERMC in the reference's terms; the etree postorder of
     * the symbolic factorisation is composed into perm_c */
    int *perm_c = (int *) malloc(sizeof(int) * n), *perm_c_in = (int *) malloc(sizeof(int) * n);
    if (use_nd) CHECK(sluamd_order_nd(n, rp, ci, 64, perm_c_in));
    else for (int64_t i = 0; i < n; ++i) perm_c_in[i] = (int) i;
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

    /* solve through pdgstrs3d's own boundary: b in the ORIGINAL row order in, x in the original order out (B_to_X / X_to_B with
     * perm_c run inside the library; a single rank holds all n rows) */
    double *x = (double *) malloc(sizeof(double) * n);
    memcpy(x, b, sizeof(double) * n);
    CHECK(sluamd_pdgstrs3d_dist(h, x, n, 1, n, 0, perm_c, perm_c));

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
