/*
 * slu_ref_dump.c -- TEST INFRASTRUCTURE (golden-fixture generator), not product code.
 *
 * Our own driver around the REAL reference (xiaoyeli/superlu_dist v9.2.1, linked from
 * /root/reference by oracle/ref/Makefile).  It runs the reference's expert driver
 * pdgssvx3d (SRC/double/pdgssvx3d.c:519) exactly like EXAMPLE/pddrive3d.c:101 does and,
 * by `ld --wrap`, records what crosses the hot-path boundary:
 *
 *   WRAP(gstrf3d)           : LU store + 3D partition BEFORE and AFTER the real
 *                                pdgstrf3d (SRC/double/pdgstrf3d.c:121)
 *   WRAP(gstrs3d_newsolve)  : B before/after the real pdgstrs3d_newsolve
 *   WRAP(gstrs3d)             (SRC/double/pdgstrs3d.c:6935 / :6604), plus perm_r/perm_c
 *
 * Output: one container file per rank, "<out>.r<rank>.slud": a sequence of records
 *   int32 name_len | name bytes | int32 dtype (0=int32,1=int64,2=float64,3=complex128) | int64 count | data
 * read by tests/golden/slud.py.
 *
 * usage: mpiexec -n R*C*D slu_ref_dump -r R -c C -d D [-q colperm] [-p rowperm] [-e equil]
 *                 [-i iterrefine] [-s nrhs] [-P perm_c.txt] [-T replace_tiny] -o OUT matrixfile
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef Z_PREC   /* complex16 twin: pzgssvx3d / pzgstrf3d / pzgstrs3d (SRC/complex16) */
#include "superlu_zdefs.h"
typedef doublecomplex scalar_t;
#define VAL_DTYPE 3
#define xLUstruct_t zLUstruct_t
#define xLocalLU_t zLocalLU_t
#define xtrf3Dpartition_t ztrf3Dpartition_t
#define xScalePermstruct_t zScalePermstruct_t
#define xSOLVEstruct_t zSOLVEstruct_t
#define pxgstrf3d pzgstrf3d
#define pxgstrs3d pzgstrs3d
#define pxgstrs3d_newsolve pzgstrs3d_newsolve
#define pxgssvx3d pzgssvx3d
#define xcreate_matrix_postfix3d zcreate_matrix_postfix3d
#define xScalePermstructInit zScalePermstructInit
#define xLUstructInit zLUstructInit
#define xMalloc_dist doublecomplexMalloc_dist
#define pxinf_norm_error pzinf_norm_error
#define WRAP(name) __wrap_pz##name
#define REAL(name) __real_pz##name
#else
#include "superlu_ddefs.h"
typedef double scalar_t;
#define VAL_DTYPE 2
#define xLUstruct_t dLUstruct_t
#define xLocalLU_t dLocalLU_t
#define xtrf3Dpartition_t dtrf3Dpartition_t
#define xScalePermstruct_t dScalePermstruct_t
#define xSOLVEstruct_t dSOLVEstruct_t
#define pxgstrf3d pdgstrf3d
#define pxgstrs3d pdgstrs3d
#define pxgstrs3d_newsolve pdgstrs3d_newsolve
#define pxgssvx3d pdgssvx3d
#define xcreate_matrix_postfix3d dcreate_matrix_postfix3d
#define xScalePermstructInit dScalePermstructInit
#define xLUstructInit dLUstructInit
#define xMalloc_dist doubleMalloc_dist
#define pxinf_norm_error pdinf_norm_error
#define WRAP(name) __wrap_pd##name
#define REAL(name) __real_pd##name
#endif

static FILE *g_out = NULL;
static scalar_t *g_b0 = NULL;   /* copy of the first right-hand side before the solve */
static int g_solve_count = 0;

static void put(const char *name, int dtype, long long count, const void *data)
{
    if (!g_out) return;            /* -o none: timing run, nothing recorded */
    int nl = (int) strlen(name);
    size_t esz = dtype == 0 ? 4 : (dtype == 3 ? 16 : 8);
    fwrite(&nl, 4, 1, g_out);
    fwrite(name, 1, nl, g_out);
    fwrite(&dtype, 4, 1, g_out);
    fwrite(&count, 8, 1, g_out);
    if (count) fwrite(data, esz, (size_t) count, g_out);
}
static void put_i(const char *name, long long v) { put(name, 1, 1, &v); }
static void put_d(const char *name, double v) { put(name, 2, 1, &v); }
static void put_intt(const char *name, long long count, const int_t *p)
{   /* int_t is 32-bit in this build (no _LONGINT) */
    put(name, sizeof(int_t) == 4 ? 0 : 1, count, p);
}

/* Walk the L/U store of this rank (formats: SURVEY.md Appendix A,
 * SRC/include/superlu_defs.h:156-198) and emit flat copies + offsets. */
static void dump_lu(const char *tag, xLUstruct_t *LUstruct, gridinfo_t *grid, int values_only)
{
    Glu_persist_t *Glu = LUstruct->Glu_persist;
    xLocalLU_t *Llu = LUstruct->Llu;
    int_t *xsup = Glu->xsup;

    char nm[128];

    /* nsupers = supno[n-1]+1; n is xsup[nsupers]; caller stored it in g_nsupers */
    extern int_t g_nsupers_dump;
    int_t ns = g_nsupers_dump;
    int_t Pr = grid->nprow, Pc = grid->npcol;
    int_t nlbc = CEILING(ns, Pc), nlbr = CEILING(ns, Pr);
    long long *loff = (long long *) calloc(nlbc + 1, 8), *lvoff = (long long *) calloc(nlbc + 1, 8);
    long long *uoff = (long long *) calloc(nlbr + 1, 8), *uvoff = (long long *) calloc(nlbr + 1, 8);
    int mycol = MYCOL(grid->iam, grid), myrow = MYROW(grid->iam, grid);
    for (int_t lk = 0; lk < nlbc; ++lk) {
        int_t *idx = Llu->Lrowind_bc_ptr[lk];
        long long li = 0, lv = 0;
        int_t k = lk * Pc + mycol; /* global block col */
        if (idx && k < ns) {
            li = BC_HEADER + (long long) idx[0] * LB_DESCRIPTOR + idx[1];
            lv = (long long) idx[1] * (xsup[k + 1] - xsup[k]);
        }
        loff[lk + 1] = loff[lk] + li;
        lvoff[lk + 1] = lvoff[lk] + lv;
    }
    for (int_t lb = 0; lb < nlbr; ++lb) {
        int_t *idx = Llu->Ufstnz_br_ptr[lb];
        long long ui = 0, uv = 0;
        if (idx) { ui = idx[2]; uv = idx[1]; }
        uoff[lb + 1] = uoff[lb] + ui;
        uvoff[lb + 1] = uvoff[lb] + uv;
    }
    if (!values_only) {
        int_t *lidx = (int_t *) malloc(sizeof(int_t) * (loff[nlbc] + 1));
        int_t *uidx = (int_t *) malloc(sizeof(int_t) * (uoff[nlbr] + 1));
        for (int_t lk = 0; lk < nlbc; ++lk)
            if (loff[lk + 1] > loff[lk])
                memcpy(lidx + loff[lk], Llu->Lrowind_bc_ptr[lk], sizeof(int_t) * (loff[lk + 1] - loff[lk]));
        for (int_t lb = 0; lb < nlbr; ++lb)
            if (uoff[lb + 1] > uoff[lb])
                memcpy(uidx + uoff[lb], Llu->Ufstnz_br_ptr[lb], sizeof(int_t) * (uoff[lb + 1] - uoff[lb]));
        put("Lrowind_off", 1, nlbc + 1, loff);
        put("Lnzval_off", 1, nlbc + 1, lvoff);
        put("Ufstnz_off", 1, nlbr + 1, uoff);
        put("Unzval_off", 1, nlbr + 1, uvoff);
        put_intt("Lrowind", loff[nlbc], lidx);
        put_intt("Ufstnz", uoff[nlbr], uidx);
        free(lidx); free(uidx);
    }
    scalar_t *lval = (scalar_t *) malloc(sizeof(scalar_t) * (lvoff[nlbc] + 1));
    scalar_t *uval = (scalar_t *) malloc(sizeof(scalar_t) * (uvoff[nlbr] + 1));
    for (int_t lk = 0; lk < nlbc; ++lk)
        if (lvoff[lk + 1] > lvoff[lk])
            memcpy(lval + lvoff[lk], Llu->Lnzval_bc_ptr[lk], sizeof(scalar_t) * (lvoff[lk + 1] - lvoff[lk]));
    for (int_t lb = 0; lb < nlbr; ++lb)
        if (uvoff[lb + 1] > uvoff[lb])
            memcpy(uval + uvoff[lb], Llu->Unzval_br_ptr[lb], sizeof(scalar_t) * (uvoff[lb + 1] - uvoff[lb]));
    snprintf(nm, sizeof nm, "Lnzval_%s", tag); put(nm, VAL_DTYPE, lvoff[nlbc], lval);
    snprintf(nm, sizeof nm, "Unzval_%s", tag); put(nm, VAL_DTYPE, uvoff[nlbr], uval);
    free(lval); free(uval); free(loff); free(lvoff); free(uoff); free(uvoff);
    (void) myrow;
}
int_t g_nsupers_dump = 0;

int_t REAL(gstrf3d)(superlu_dist_options_t *, int, int, double, xtrf3Dpartition_t *, SCT_t *,
                       xLUstruct_t *, gridinfo3d_t *, SuperLUStat_t *, int *);

int_t WRAP(gstrf3d)(superlu_dist_options_t *options, int m, int n, double anorm,
                       xtrf3Dpartition_t *part, SCT_t *SCT, xLUstruct_t *LUstruct,
                       gridinfo3d_t *grid3d, SuperLUStat_t *stat, int *info)
{
    gridinfo_t *grid = &grid3d->grid2d;
    Glu_persist_t *Glu = LUstruct->Glu_persist;
    int_t nsupers = Glu->supno[n - 1] + 1;
    g_nsupers_dump = nsupers;
    put_i("n", n); put_i("nsupers", nsupers);
    put_i("Pr", grid->nprow); put_i("Pc", grid->npcol); put_i("Pz", grid3d->npdep);
    put_i("myrow", MYROW(grid->iam, grid)); put_i("mycol", MYCOL(grid->iam, grid));
    put_i("myz", grid3d->zscp.Iam); put_i("iam3d", grid3d->iam);
    put_d("anorm", anorm);
    put_d("thresh", (double) smach_dist("Epsilon") * anorm); /* pdgstrf3d.c:132-133 */
    put_i("ReplaceTinyPivot", options->ReplaceTinyPivot == YES);
    put_i("ldt", sp_ienv_dist(3, options));
    put_intt("xsup", nsupers + 1, Glu->xsup);
    /* 3D partition (superlu_ddefs.h:317-337, superlu_defs.h:926-966) */
    int maxLvl = log2i(grid3d->zscp.Np) + 1;
    int numForests = (1 << maxLvl) - 1;
    put_i("maxLvl", maxLvl);
    put_intt("myTreeIdxs", maxLvl, part->myTreeIdxs);
    put_intt("myZeroTrIdxs", maxLvl, part->myZeroTrIdxs);
    put_intt("myNodeCount", maxLvl, part->myNodeCount);
    put_intt("iperm_c_supno", nsupers, part->iperm_c_supno);
    put_intt("setree", nsupers, part->gEtreeInfo.setree);
    if (part->supernode2treeMap) put_intt("supernode2treeMap", nsupers, part->supernode2treeMap);
    for (int f = 0; f < numForests; ++f) {
        char nm[64];
        sForest_t *sf = part->sForests[f];
        snprintf(nm, sizeof nm, "forest%d_nNodes", f);
        put_i(nm, sf ? sf->nNodes : 0);
        if (!sf) continue;
        snprintf(nm, sizeof nm, "forest%d_nodeList", f);
        put_intt(nm, sf->nNodes, sf->nodeList);
        snprintf(nm, sizeof nm, "forest%d_eTreeTopLims", f);
        put_intt(nm, sf->topoInfo.numLvl + 1, sf->topoInfo.eTreeTopLims);
    }
    if (g_out) dump_lu("pre", LUstruct, grid, 0);
#ifdef USE_SLUAMD   /* slu_ref_amd: the reference pipeline with OUR numeric factorisation (bindings/superlu_dist/sluamd_binding.c) */
#ifdef Z_PREC
#define BIND_NAME sluamd_bind_pzgstrf3d
#else
#define BIND_NAME sluamd_bind_pdgstrf3d
#endif
    extern int_t BIND_NAME(superlu_dist_options_t *, int, int, double, xtrf3Dpartition_t *, SCT_t *,
                           xLUstruct_t *, gridinfo3d_t *, SuperLUStat_t *, int *);
#ifdef Z_PREC
#define BIND_SYNC_HOST sluamd_bind_zsync_host_for
#define BIND_SOLVES_BOUND sluamd_bind_zsolves_bound
#else
#define BIND_SYNC_HOST sluamd_bind_dsync_host_for
#define BIND_SOLVES_BOUND sluamd_bind_dsolves_bound
#endif
    extern int BIND_SYNC_HOST(xLUstruct_t *, gridinfo3d_t *);
    extern void BIND_SOLVES_BOUND(int);
    /* this driver KNOWS at link time that its solve wrappers below forward to the library unless SLUAMD_BIND_SOLVE=0: it declares so, which is
     * what lets the binding defer the copy-back (its default is the reference's eager copy) */
    if (!getenv("SLUAMD_REFDUMP_NO_DECLARE"))    /* test hook: an integrator who wrapped the solves but never told the binding */
        BIND_SOLVES_BOUND(!getenv("SLUAMD_BIND_SOLVE") || atoi(getenv("SLUAMD_BIND_SOLVE")));
    int_t r = BIND_NAME(options, m, n, anorm, part, SCT, LUstruct, grid3d, stat, info);
    /* lazy copy-back (SLUAMD_BIND_COPYBACK): this driver is a host consumer of the factors when it records them, and when the CPU
     * solves of a Z-replicated grid follow (dbroadcastAncestor3d reads the host panels right after this call) */
    if (g_out || (getenv("SLUAMD_BIND_SOLVE") && !atoi(getenv("SLUAMD_BIND_SOLVE")))) BIND_SYNC_HOST(LUstruct, grid3d);
#else
    int_t r = REAL(gstrf3d)(options, m, n, anorm, part, SCT, LUstruct, grid3d, stat, info);
#endif
    if (g_out) dump_lu("post", LUstruct, grid, 1);
    put_i("info", *info);
    put_i("TinyPivots", stat->TinyPivots);
    put_d("ops_fact_float", (double) stat->ops[FACT]);
    return r;
}

static void dump_solve(const char *which, int_t n, xScalePermstruct_t *SP, scalar_t *B,
                       int_t m_loc, int_t fst_row, int_t ldb, int nrhs, int after)
{
    char nm[96];
    if (!after) {
        snprintf(nm, sizeof nm, "solve%d_kind_%s", g_solve_count, which); put_i(nm, 1);
        snprintf(nm, sizeof nm, "solve%d_m_loc", g_solve_count); put_i(nm, m_loc);
        snprintf(nm, sizeof nm, "solve%d_fst_row", g_solve_count); put_i(nm, fst_row);
        snprintf(nm, sizeof nm, "solve%d_nrhs", g_solve_count); put_i(nm, nrhs);
        if (g_solve_count == 0) {
            put("perm_r", 0, n, SP->perm_r);
            put("perm_c", 0, n, SP->perm_c);
        }
    }
    if (!g_out) return;
    scalar_t *buf = (scalar_t *) malloc(sizeof(scalar_t) * (size_t) (m_loc * nrhs + 1));
    for (int j = 0; j < nrhs; ++j)
        for (int_t i = 0; i < m_loc; ++i) buf[i + j * m_loc] = B[i + j * ldb];
    snprintf(nm, sizeof nm, "solve%d_B_%s", g_solve_count, after ? "out" : "in");
    put(nm, VAL_DTYPE, (long long) m_loc * nrhs, buf);
    free(buf);
}

void REAL(gstrs3d_newsolve)(superlu_dist_options_t *, int_t, xLUstruct_t *, xScalePermstruct_t *,
                               xtrf3Dpartition_t *, gridinfo3d_t *, scalar_t *, int_t, int_t, int_t, int,
                               xSOLVEstruct_t *, SuperLUStat_t *, int *);
void WRAP(gstrs3d_newsolve)(superlu_dist_options_t *options, int_t n, xLUstruct_t *LUstruct,
                               xScalePermstruct_t *SP, xtrf3Dpartition_t *part, gridinfo3d_t *grid3d,
                               scalar_t *B, int_t m_loc, int_t fst_row, int_t ldb, int nrhs,
                               xSOLVEstruct_t *SOLVEstruct, SuperLUStat_t *stat, int *info)
{
    dump_solve("newsolve", n, SP, B, m_loc, fst_row, ldb, nrhs, 0);
#ifndef Z_PREC
    if (g_out && g_solve_count == 0 && options->DiagInv == YES && LUstruct->Llu->inv == 1) {
        /* pdCompute_Diag_Inv's result (pdgstrs.c:842-959, dtrtri on the factored diagonal blocks): Linv / Uinv of every diagonal block this rank owns,
         * column-major nsupc x nsupc each, concatenated in supernode order with their offsets -- the direct fixture of SURVEY 8(f)-3 */
        gridinfo_t *grid = &grid3d->grid2d;
        Glu_persist_t *Glu = LUstruct->Glu_persist;
        xLocalLU_t *Llu = LUstruct->Llu;
        int_t ns = g_nsupers_dump, *xsup = Glu->xsup;
        int myrow = MYROW(grid->iam, grid), mycol = MYCOL(grid->iam, grid);
        long long *off = (long long *) calloc(ns + 1, 8);
        for (int_t k = 0; k < ns; ++k) {
            long long w = xsup[k + 1] - xsup[k];
            int own = PROW(k, grid) == myrow && PCOL(k, grid) == mycol && Llu->Lrowind_bc_ptr[LBj(k, grid)];
            off[k + 1] = off[k] + (own ? w * w : 0);
        }
        double *li = (double *) malloc(8 * (off[ns] + 1)), *ui = (double *) malloc(8 * (off[ns] + 1));
        for (int_t k = 0; k < ns; ++k)
            if (off[k + 1] > off[k]) {
                memcpy(li + off[k], Llu->Linv_bc_ptr[LBj(k, grid)], 8 * (off[k + 1] - off[k]));
                memcpy(ui + off[k], Llu->Uinv_bc_ptr[LBj(k, grid)], 8 * (off[k + 1] - off[k]));
            }
        put("diaginv_off", 1, ns + 1, off);
        put("Linv", 2, off[ns], li); put("Uinv", 2, off[ns], ui);
        free(off); free(li); free(ui);
    }
#endif
#if defined(USE_SLUAMD)   /* slu_ref_amd / slu_ref_zamd: OUR triangular solves on the device-resident factors (SLUAMD_BIND_SOLVE=0: the reference's) */
#ifdef Z_PREC
#define BIND_SOLVE_NEW sluamd_bind_pzgstrs3d_newsolve
#define BIND_SOLVE_OLD sluamd_bind_pzgstrs3d
#else
#define BIND_SOLVE_NEW sluamd_bind_pdgstrs3d_newsolve
#define BIND_SOLVE_OLD sluamd_bind_pdgstrs3d
#endif
    extern void BIND_SOLVE_NEW(superlu_dist_options_t *, int_t, xLUstruct_t *, xScalePermstruct_t *, xtrf3Dpartition_t *,
                               gridinfo3d_t *, scalar_t *, int_t, int_t, int_t, int, xSOLVEstruct_t *, SuperLUStat_t *, int *);
    if (!getenv("SLUAMD_BIND_SOLVE") || atoi(getenv("SLUAMD_BIND_SOLVE")))
        BIND_SOLVE_NEW(options, n, LUstruct, SP, part, grid3d, B, m_loc, fst_row, ldb, nrhs, SOLVEstruct, stat, info);
    else
#endif
    REAL(gstrs3d_newsolve)(options, n, LUstruct, SP, part, grid3d, B, m_loc, fst_row, ldb, nrhs,
                              SOLVEstruct, stat, info);
    dump_solve("newsolve", n, SP, B, m_loc, fst_row, ldb, nrhs, 1);
    ++g_solve_count;
}
void REAL(gstrs3d)(superlu_dist_options_t *, int_t, xLUstruct_t *, xScalePermstruct_t *,
                      xtrf3Dpartition_t *, gridinfo3d_t *, scalar_t *, int_t, int_t, int_t, int,
                      xSOLVEstruct_t *, SuperLUStat_t *, int *);
void WRAP(gstrs3d)(superlu_dist_options_t *options, int_t n, xLUstruct_t *LUstruct,
                      xScalePermstruct_t *SP, xtrf3Dpartition_t *part, gridinfo3d_t *grid3d,
                      scalar_t *B, int_t m_loc, int_t fst_row, int_t ldb, int nrhs,
                      xSOLVEstruct_t *SOLVEstruct, SuperLUStat_t *stat, int *info)
{
    dump_solve("legacy", n, SP, B, m_loc, fst_row, ldb, nrhs, 0);
#if defined(USE_SLUAMD)
    extern void BIND_SOLVE_OLD(superlu_dist_options_t *, int_t, xLUstruct_t *, xScalePermstruct_t *, xtrf3Dpartition_t *,
                               gridinfo3d_t *, scalar_t *, int_t, int_t, int_t, int, xSOLVEstruct_t *, SuperLUStat_t *, int *);
    if (!getenv("SLUAMD_BIND_SOLVE") || atoi(getenv("SLUAMD_BIND_SOLVE")))
        BIND_SOLVE_OLD(options, n, LUstruct, SP, part, grid3d, B, m_loc, fst_row, ldb, nrhs, SOLVEstruct, stat, info);
    else
#endif
    REAL(gstrs3d)(options, n, LUstruct, SP, part, grid3d, B, m_loc, fst_row, ldb, nrhs,
                     SOLVEstruct, stat, info);
    dump_solve("legacy", n, SP, B, m_loc, fst_row, ldb, nrhs, 1);
    ++g_solve_count;
}

int main(int argc, char *argv[])
{
    superlu_dist_options_t options;
    SuperLUStat_t stat;
    SuperMatrix A;
    xScalePermstruct_t ScalePermstruct;
    xLUstruct_t LUstruct;
    xSOLVEstruct_t SOLVEstruct;
    gridinfo3d_t grid;
    double *berr;
    scalar_t *b, *xtrue;
    int nprow = 1, npcol = 1, npdep = 1, nrhs = 1;
    int equil = -1, colperm = -1, rowperm = -1, ir = -1, tiny = -1, quiet = 0; int diaginv = -1;
    int info, ldb, ldx;
    const char *outp = "slu_dump", *permfile = NULL, *matfile = NULL;
    FILE *fp = NULL;

    int provided;
    MPI_Init_thread(&argc, &argv, MPI_THREAD_MULTIPLE, &provided);
    for (int a = 1; a < argc; ++a) {
        if (argv[a][0] == '-' && a + 1 < argc) {
            char c = argv[a][1];
            const char *v = argv[++a];
            switch (c) {
            case 'r': nprow = atoi(v); break;
            case 'c': npcol = atoi(v); break;
            case 'd': npdep = atoi(v); break;
            case 'e': equil = atoi(v); break;
            case 'p': rowperm = atoi(v); break;
            case 'q': colperm = atoi(v); break;
            case 'i': ir = atoi(v); break;
            case 's': nrhs = atoi(v); break;
            case 'T': tiny = atoi(v); break;
            case 'D': diaginv = atoi(v); break;
            case 'P': permfile = v; break;
            case 'o': outp = v; break;
            case 'Q': quiet = atoi(v); break;
            default: fprintf(stderr, "unknown flag -%c\n", c); exit(2);
            }
        } else matfile = argv[a];
    }
    if (!matfile || !(fp = fopen(matfile, "r"))) { fprintf(stderr, "cannot open matrix file\n"); exit(2); }
    const char *dot = strrchr(matfile, '.');
    char suffix[16]; snprintf(suffix, sizeof suffix, "%s", dot ? dot + 1 : "rua");

    set_default_options_dist(&options);
    if (equil != -1) options.Equil = equil;
    if (rowperm != -1) options.RowPerm = rowperm;
    if (colperm != -1) options.ColPerm = colperm;
    if (ir != -1) options.IterRefine = ir;
    if (tiny != -1) options.ReplaceTinyPivot = tiny ? YES : NO;
    if (diaginv != -1) options.DiagInv = diaginv ? YES : NO;      /* needs a LAPACK build (oracle/_ref_mkl): pdCompute_Diag_Inv is empty otherwise (pdgstrs.c:845) */
    if (permfile) options.ColPerm = MY_PERMC;
    options.PrintStat = quiet ? NO : YES;

    superlu_gridinit3d(MPI_COMM_WORLD, nprow, npcol, npdep, &grid);
    if (grid.iam == -1) goto out;
    {
        char fn[512];
        snprintf(fn, sizeof fn, "%s.r%d.slud", outp, grid.iam);
        if (strcmp(outp, "none") != 0) {
            g_out = fopen(fn, "wb");
            if (!g_out) { fprintf(stderr, "cannot open %s\n", fn); exit(2); }
        }
    }
    xcreate_matrix_postfix3d(&A, nrhs, &b, &ldb, &xtrue, &ldx, fp, suffix, &grid);
    if (!(berr = doubleMalloc_dist(nrhs))) ABORT("Malloc fails for berr[].");
    int_t m = A.nrow, n = A.ncol;
    xScalePermstructInit(m, n, &ScalePermstruct);
    if (permfile) { /* ColPerm = MY_PERMC: pdgssvx3d.c:749-791 uses ScalePermstruct->perm_c as given */
        FILE *pf = fopen(permfile, "r");
        if (!pf) { fprintf(stderr, "cannot open perm file\n"); exit(2); }
        for (int_t i = 0; i < n; ++i) { long v; if (fscanf(pf, "%ld", &v) != 1) ABORT("perm file short"); ScalePermstruct.perm_c[i] = (int) v; }
        fclose(pf);
    }
    xLUstructInit(n, &LUstruct);
    PStatInit(&stat);

    /* record the local slice of A, b, xtrue (NRformat_loc: supermatrix.h) */
    {
        NRformat_loc *As = (NRformat_loc *) A.Store;
        put_i("A_n", n); put_i("A_m_loc", As->m_loc); put_i("A_fst_row", As->fst_row); put_i("A_nnz_loc", As->nnz_loc);
        put_intt("A_rowptr", As->m_loc + 1, As->rowptr);
        put_intt("A_colind", As->nnz_loc, As->colind);
        put("A_nzval", VAL_DTYPE, As->nnz_loc, As->nzval);
        scalar_t *tmp = (scalar_t *) malloc(sizeof(scalar_t) * (size_t) (As->m_loc * nrhs + 1));
        for (int j = 0; j < nrhs; ++j) for (int_t i = 0; i < As->m_loc; ++i) tmp[i + j * As->m_loc] = b[i + j * ldb];
        put("b", VAL_DTYPE, (long long) As->m_loc * nrhs, tmp);
        for (int j = 0; j < nrhs; ++j) for (int_t i = 0; i < As->m_loc; ++i) tmp[i + j * As->m_loc] = xtrue[i + j * ldx];
        put("xtrue", VAL_DTYPE, (long long) As->m_loc * nrhs, tmp);
        free(tmp);
        put_i("nrhs", nrhs);
        put_i("opt_Equil", options.Equil); put_i("opt_RowPerm", options.RowPerm);
        put_i("opt_ColPerm", options.ColPerm); put_i("opt_IterRefine", options.IterRefine);
    }

    {
        NRformat_loc *As = (NRformat_loc *) A.Store;
        g_b0 = (scalar_t *) malloc(sizeof(scalar_t) * (size_t) (As->m_loc + 1));
        for (int_t i = 0; i < As->m_loc; ++i) g_b0[i] = b[i];
    }
    pxgssvx3d(&options, &A, &ScalePermstruct, b, ldb, nrhs, &grid, &LUstruct, &SOLVEstruct, berr, &stat, &info);

    {
        NRformat_loc *As = (NRformat_loc *) A.Store;
        scalar_t *tmp = (scalar_t *) malloc(sizeof(scalar_t) * (size_t) (As->m_loc * nrhs + 1));
        for (int j = 0; j < nrhs; ++j) for (int_t i = 0; i < As->m_loc; ++i) tmp[i + j * As->m_loc] = b[i + j * ldb];
        put("x", VAL_DTYPE, (long long) As->m_loc * nrhs, tmp);
        free(tmp);
        put("berr", 2, nrhs, berr);
        put_i("final_info", info);
        put_i("RefineSteps", stat.RefineSteps);
        put_d("utime_FACT", stat.utime[FACT]); put_d("utime_SOLVE", stat.utime[SOLVE]);
        put_d("ops_FACT", stat.ops[FACT]); put_d("ops_SOLVE", stat.ops[SOLVE]);
        if (ScalePermstruct.DiagScale == ROW || ScalePermstruct.DiagScale == BOTH) put("R", 2, m, ScalePermstruct.R);
        if (ScalePermstruct.DiagScale == COL || ScalePermstruct.DiagScale == BOTH) put("C", 2, n, ScalePermstruct.C);
        put_i("DiagScale", ScalePermstruct.DiagScale);
    }
    if (info) { if (!grid.iam) printf("ERROR: INFO = %d returned from pdgssvx3d()\n", info); }
    else if (!quiet) pxinf_norm_error(grid.iam, ((NRformat_loc *) A.Store)->m_loc, nrhs, b, ldb, xtrue, ldx, grid.comm);
    if (grid.zscp.Iam == 0 && !quiet) PStatPrint(&options, &stat, &(grid.grid2d));
    {   /* residual on the original system, computed here for the drop-in tests: ||b - A x||_2 / ||b||_2 (A, b, x are
         * distributed by rows over all ranks of the 3D grid: gather x, reduce the two sums) */
        NRformat_loc *As = (NRformat_loc *) A.Store;
        int P; MPI_Comm_size(grid.comm, &P);
        int *cnt = (int *) malloc(sizeof(int) * 2 * P), *dsp = cnt + P;
        int mine = (int) As->m_loc;
        MPI_Allgather(&mine, 1, MPI_INT, cnt, 1, MPI_INT, grid.comm);
        int tot = 0;
        for (int q = 0; q < P; ++q) { dsp[q] = tot; tot += cnt[q]; }
        if (tot == n && g_b0) {
            const int w = (int) (sizeof(scalar_t) / sizeof(double));
            for (int q = 0; q < P; ++q) { cnt[q] *= w; dsp[q] *= w; }
            scalar_t *xf = (scalar_t *) malloc(sizeof(scalar_t) * (size_t) (n + 1));
            MPI_Allgatherv(b, mine * w, MPI_DOUBLE, xf, cnt, dsp, MPI_DOUBLE, grid.comm);
            double loc[2] = {0, 0}, glob[2];
            for (int_t i = 0; i < As->m_loc; ++i) {
#ifdef Z_PREC
                double sr = g_b0[i].r, si = g_b0[i].i;
                for (int_t e = As->rowptr[i]; e < As->rowptr[i + 1]; ++e) {
                    doublecomplex a = ((doublecomplex *) As->nzval)[e], xx = xf[As->colind[e]];
                    sr -= a.r * xx.r - a.i * xx.i; si -= a.r * xx.i + a.i * xx.r;
                }
                loc[0] += sr * sr + si * si; loc[1] += g_b0[i].r * g_b0[i].r + g_b0[i].i * g_b0[i].i;
#else
                double s = g_b0[i];
                for (int_t e = As->rowptr[i]; e < As->rowptr[i + 1]; ++e) s -= ((double *) As->nzval)[e] * xf[As->colind[e]];
                loc[0] += s * s; loc[1] += g_b0[i] * g_b0[i];
#endif
            }
            MPI_Allreduce(loc, glob, 2, MPI_DOUBLE, MPI_SUM, grid.comm);
            if (!grid.iam) printf("RESIDUAL %.6e INFO %d\n", sqrt(glob[0] / glob[1]), info);
            free(xf);
        }
        free(cnt);
    }
    if (!grid.iam) printf("REFTIMES n %ld FACT %.6f s SOLVE %.6f s ops_FACT %.6e\n", (long) n, stat.utime[FACT], stat.utime[SOLVE], (double) stat.ops[FACT]);
    if (g_out) fclose(g_out);
    fclose(fp);
out:
    superlu_gridexit3d(&grid);
    MPI_Finalize();
    return 0;
}
