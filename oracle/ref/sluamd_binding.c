/*
 * sluamd_binding.c -- the reference-side binding of the MI355X hot path (this is the code INTEGRATION.md asks a
 * SuperLU_DIST maintainer to add next to SRC/double/pdgssvx3d.c:1013-1021).  It is OUR code; it includes the
 * reference's public headers only to unpack dLUstruct_t / gridinfo3d_t / dtrf3Dpartition_t into the plain-pointer
 * views of include/superlu_dist_amd.h.
 *
 * Built into oracle/_ref/slu_ref_amd (oracle/ref/Makefile) where `ld --wrap=pdgstrf3d` routes the reference's own
 * pdgssvx3d to it: the reference's pre-processing, distribution, triangular solve and refinement run unchanged on
 * the factors our library writes back in the reference's panel / skyline formats.  Test infrastructure only.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <dlfcn.h>
#include <unistd.h>
#ifdef Z_PREC   /* complex16 twin: binds pzgstrf3d (SRC/complex16/pzgstrf3d.c) to the sluamd_z* entry points */
#include "superlu_zdefs.h"
#define xLUstruct_t zLUstruct_t
#define xLocalLU_t zLocalLU_t
#define xtrf3Dpartition_t ztrf3Dpartition_t
#define BIND_NAME sluamd_bind_pzgstrf3d
#define LUVIEW_T sluamd_zLUview_t
#define VALPP(p) ((sluamd_doublecomplex **) (p))
#define SYM_CREATE "sluamd_zCreateLUHandle"
#define SYM_FACTOR "sluamd_pzgstrf3d"
#define SYM_COPY "sluamd_zCopyLU2Host"
#else
#include "superlu_ddefs.h"
#define xLUstruct_t dLUstruct_t
#define xLocalLU_t dLocalLU_t
#define xtrf3Dpartition_t dtrf3Dpartition_t
#define BIND_NAME sluamd_bind_pdgstrf3d
#define LUVIEW_T sluamd_dLUview_t
#define VALPP(p) (p)
#define SYM_CREATE "sluamd_dCreateLUHandle"
#define SYM_FACTOR "sluamd_pdgstrf3d"
#define SYM_COPY "sluamd_dCopyLU2Host"
#endif
#include "superlu_dist_amd.h"

/* The library is C++/HIP; a C/MPI application binds it at run time (dlopen) so that the application's own
 * link line (here: conda's MPICH toolchain with an older libstdc++) does not have to resolve the HIP runtime.
 * A maintainer linking with the ROCm toolchain can call the sluamd_* functions directly instead. */
static struct {
    void *so;
    void (*default_options)(sluamd_options_t *);
    int (*create)(sluamd_handle_t *, const LUVIEW_T *, const sluamd_forest_view_t *, const sluamd_options_t *);
    int (*factor)(sluamd_handle_t, double, int *);
    int (*copy2host)(sluamd_handle_t, const LUVIEW_T *);
    int (*stats)(sluamd_handle_t, sluamd_stats_t *);
    void (*destroy)(sluamd_handle_t);
    const char *(*last_error)(void);
} S;

static void sluamd_load(void)
{
    if (S.so) return;
    char path[4096];
    const char *env = getenv("SLUAMD_LIB");
    if (env) snprintf(path, sizeof path, "%s", env);
    else {   /* <repo>/oracle/_ref/<exe>  ->  <repo>/superlu_dist_amd/libsluamd.so */
        ssize_t k = readlink("/proc/self/exe", path, sizeof path - 64);
        if (k < 0) ABORT("readlink(/proc/self/exe) failed");
        path[k] = 0;
        char *sl = strrchr(path, '/'); if (sl) *sl = 0;
        strcat(path, "/../../superlu_dist_amd/libsluamd.so");
    }
    S.so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!S.so) { fprintf(stderr, "dlopen(%s): %s\n", path, dlerror()); ABORT("cannot load libsluamd.so"); }
    S.default_options = (void (*)(sluamd_options_t *)) dlsym(S.so, "sluamd_default_options");
    S.create = (int (*)(sluamd_handle_t *, const LUVIEW_T *, const sluamd_forest_view_t *, const sluamd_options_t *)) dlsym(S.so, SYM_CREATE);
    S.factor = (int (*)(sluamd_handle_t, double, int *)) dlsym(S.so, SYM_FACTOR);
    S.copy2host = (int (*)(sluamd_handle_t, const LUVIEW_T *)) dlsym(S.so, SYM_COPY);
    S.stats = (int (*)(sluamd_handle_t, sluamd_stats_t *)) dlsym(S.so, "sluamd_get_stats");
    S.destroy = (void (*)(sluamd_handle_t)) dlsym(S.so, "sluamd_dDestroyLUHandle");
    S.last_error = (const char *(*)(void)) dlsym(S.so, "sluamd_last_error");
    if (!S.create || !S.factor || !S.copy2host || !S.destroy) ABORT("libsluamd.so lacks a required symbol");
}

int_t BIND_NAME(superlu_dist_options_t *options, int m, int n, double anorm,
                            xtrf3Dpartition_t *trf3Dpartition, SCT_t *SCT, xLUstruct_t *LUstruct,
                            gridinfo3d_t *grid3d, SuperLUStat_t *stat, int *info)
{
    gridinfo_t *grid = &grid3d->grid2d;
    Glu_persist_t *Glu = LUstruct->Glu_persist;
    xLocalLU_t *Llu = LUstruct->Llu;
    int_t nsupers = Glu->supno[n - 1] + 1;

    LUVIEW_T v;
    v.n = n; v.nsupers = (int32_t) nsupers; v.xsup = Glu->xsup;
    v.nprow = grid->nprow; v.npcol = grid->npcol; v.npdep = grid3d->npdep;
    v.myrow = MYROW(grid->iam, grid); v.mycol = MYCOL(grid->iam, grid); v.myzlayer = grid3d->zscp.Iam;
    v.Lrowind_bc_ptr = Llu->Lrowind_bc_ptr; v.Lnzval_bc_ptr = VALPP(Llu->Lnzval_bc_ptr);
    v.Ufstnz_br_ptr = Llu->Ufstnz_br_ptr;   v.Unzval_br_ptr = VALPP(Llu->Unzval_br_ptr);

    /* elimination forests of this layer (dtrf3Dpartition_t, superlu_ddefs.h:317-337) */
    int maxLvl = log2i(grid3d->zscp.Np) + 1, nf = (1 << maxLvl) - 1;
    int32_t *nNodes = (int32_t *) calloc(nf, sizeof(int32_t));
    const sluamd_int_t **lists = (const sluamd_int_t **) calloc(nf, sizeof(*lists));
    for (int f = 0; f < nf; ++f)
        if (trf3Dpartition->sForests[f]) { nNodes[f] = trf3Dpartition->sForests[f]->nNodes; lists[f] = trf3Dpartition->sForests[f]->nodeList; }
    sluamd_forest_view_t fv = { maxLvl, trf3Dpartition->myTreeIdxs, trf3Dpartition->myZeroTrIdxs, nf, nNodes, lists };

    sluamd_load();
    sluamd_options_t o;
    S.default_options(&o);
    o.replace_tiny_pivot = (options->ReplaceTinyPivot == YES);

    sluamd_handle_t h = NULL;
    int rc = S.create(&h, &v, &fv, &o);                                    /* was dCreateLUgpuHandle    */
    if (rc) ABORT(S.last_error());
    double thresh = smach_dist("Epsilon") * anorm;                           /* pdgstrf3d.c:132-133       */
    rc = S.factor(h, thresh, info);                                          /* was pdgstrf3d_LUv1        */
    if (rc) ABORT(S.last_error());
    rc = S.copy2host(h, &v);                                                 /* was dCopyLUGPU2Host       */
    if (rc) ABORT(S.last_error());
    sluamd_stats_t st;
    S.stats(h, &st);
    if (getenv("SLUAMD_BIND_DEBUG")) {
        double sl = 0, su = 0; long zl = 0;
        for (int_t k = 0; k < nsupers; ++k) {
            int_t *li = Llu->Lrowind_bc_ptr[k];
            if (!li) continue;
            int nsupr = li[1], ns = Glu->xsup[k + 1] - Glu->xsup[k];
            double *lv = (double *) Llu->Lnzval_bc_ptr[k];
            int w = (int) (sizeof(*Llu->Lnzval_bc_ptr[k]) / sizeof(double));
            for (int j = 0; j < ns; ++j) { double a = 0; for (int q = 0; q < w; ++q) a += fabs(lv[((size_t) j * nsupr + j) * w + q]); if (a == 0) ++zl; sl += a; }
            if (Llu->Ufstnz_br_ptr[k]) { double *uv = (double *) Llu->Unzval_br_ptr[k]; for (int_t e = 0; e < Llu->Ufstnz_br_ptr[k][1] * w; ++e) su += fabs(uv[e]); }
        }
        fprintf(stderr, "[sluamd_bind] info %d nnzL %lld nnzU %lld factor_ms %.3f sum|diag| %.6e zero_diag %ld sum|U| %.6e launches %d\n",
                *info, (long long) st.nnz_L, (long long) st.nnz_U, st.t_factor_ms, sl, zl, su, st.num_launches);
    }
    stat->ops[FACT] += (flops_t) (st.flops_schur_padded + st.flops_panel);   /* scuStatUpdate's tally     */
    stat->TinyPivots += st.tiny_pivots;
    S.destroy(h);                                                            /* was dDestroyLUgpuHandle   */
    free(nNodes); free(lists);
    /* *info: minimum over the 3D grid, as pdgstrf3d.c:388-392 */
    if (*info == 0) *info = n + 1;
    int g; MPI_Allreduce(info, &g, 1, MPI_INT, MPI_MIN, grid3d->comm);
    *info = (g == n + 1) ? 0 : g;
    (void) m; (void) SCT;
    return 0;
}
