/*
 * bindings/superlu_dist/sluamd_binding.c -- the reference-side binding of the MI355X hot path: the file a SuperLU_DIST
 * maintainer adds next to SRC/double/pdgssvx3d.c:1013-1021.  It is OUR code; it includes the reference's
 * public headers only to unpack dLUstruct_t / gridinfo3d_t / dtrf3Dpartition_t into the plain-pointer views of
 * include/superlu_dist_amd.h.  Compiled once per precision (-DZ_PREC for complex16):
 *
 *   sluamd_bind_pdgstrf3d            replaces pdgstrf3d            (SRC/double/pdgstrf3d.c:121)
 *   sluamd_bind_pdgstrs3d[_newsolve] replace  pdgstrs3d[_newsolve] (SRC/double/pdgstrs3d.c:6604 / :6935)
 *   sluamd_bind_pzgstrf3d, sluamd_bind_pzgstrs3d[_newsolve]        (SRC/complex16/pzgstrf3d.c, pzgstrs3d.c)
 *
 * on ANY nprow x npcol x npdep grid: the library runs the whole 3D algorithm (XY panel exchange, Z ancestor reduction,
 * distributed triangular solves) itself over a transport, chosen at run time:
 *   SLUAMD_BIND_TRANSPORT=mpi  (default) the application's MPI (grid3d->comm) through the sluamd_comm_callbacks_t hooks: host-staged,
 *                              several ranks may share one GPU;
 *   SLUAMD_BIND_TRANSPORT=rccl one rank per GPU: RCCL called directly by the library (ncclSend / ncclRecv on its HIP streams); the
 *                              ncclUniqueId travels by MPI_Bcast, the device is the rank's index on its node.
 * The right-hand side stays distributed at the solve boundary (sluamd_pdgstrs3d_dist: each rank hands over its m_loc rows of B,
 * pdReDistribute3d_B_to_X / X_to_B run inside the library on the device).
 *
 * `make -C bindings/superlu_dist REF=<superlu_dist tree>` compiles it against a reference tree; oracle/ref/Makefile links it into
 * oracle/_ref/slu_ref_amd / slu_ref_zamd, where `ld --wrap=pdgstrf3d --wrap=pdgstrs3d_newsolve --wrap=pdgstrs3d` routes the
 * reference's own pdgssvx3d to it: the reference's pre-processing, distribution and refinement loop run unchanged around our
 * factorisation and our solves (tests/test_gpu_dropin.py).
 */
#define _GNU_SOURCE   /* readlink, dladdr-free path lookup */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <dlfcn.h>
#include <unistd.h>
#ifdef Z_PREC   /* complex16 twin: binds pzgstrf3d (SRC/complex16/pzgstrf3d.c) to the sluamd_z* entry points (1 x 1 x npdep grids) */
#include "superlu_zdefs.h"
#define xLUstruct_t zLUstruct_t
#define xLocalLU_t zLocalLU_t
#define xtrf3Dpartition_t ztrf3Dpartition_t
#define xScalePermstruct_t zScalePermstruct_t
#define xSOLVEstruct_t zSOLVEstruct_t
#define BIND_NAME sluamd_bind_pzgstrf3d
#define LUVIEW_T sluamd_zLUview_t
#define VALPP(p) ((sluamd_doublecomplex **) (p))
#define SYM_CREATE "sluamd_zCreateLUHandle"
#define SYM_CREATE_GRID "sluamd_zCreateLUHandleGrid"
#define SYM_FACTOR "sluamd_pzgstrf3d"
#define SYM_COPY "sluamd_zCopyLU2Host"
#define SYM_SOLVE_DIST "sluamd_pzgstrs3d_dist"
#else
#include "superlu_ddefs.h"
#define xLUstruct_t dLUstruct_t
#define xLocalLU_t dLocalLU_t
#define xtrf3Dpartition_t dtrf3Dpartition_t
#define xScalePermstruct_t dScalePermstruct_t
#define xSOLVEstruct_t dSOLVEstruct_t
#define BIND_NAME sluamd_bind_pdgstrf3d
#define LUVIEW_T sluamd_dLUview_t
#define VALPP(p) (p)
#define SYM_CREATE "sluamd_dCreateLUHandle"
#define SYM_CREATE_GRID "sluamd_dCreateLUHandleGrid"
#define SYM_FACTOR "sluamd_pdgstrf3d"
#define SYM_COPY "sluamd_dCopyLU2Host"
#define SYM_SOLVE_DIST "sluamd_pdgstrs3d_dist"
#endif
#include "superlu_dist_amd.h"

/* The library's ABI carries 32-bit indices (sluamd_int_t).  An application built with 64-bit int_t (XSDK_INDEX_SIZE=64 / _LONGINT,
 * superlu_defs.h:121-129) hands over narrowed copies of its index arrays -- the values are never copied -- and every index must fit
 * (checked): one rank's panels are far below 2^31 rows and columns, what needs 64 bits in such builds are global counts. */
#if defined(_LONGINT)
static void **narrowed = NULL; static size_t n_narrowed = 0, cap_narrowed = 0;     /* one entry per local block column / block row / forest list: grows */
static sluamd_int_t *sluamd_narrow(const int_t *src, size_t n)
{
    if (!src) return NULL;
    sluamd_int_t *d = (sluamd_int_t *) malloc(sizeof(sluamd_int_t) * (n ? n : 1));
    if (!d) ABORT("sluamd binding: out of memory (narrowed index array)");
    for (size_t i = 0; i < n; ++i) {
        if (src[i] > INT_MAX || src[i] < INT_MIN) ABORT("sluamd binding: an index does not fit the library's 32-bit ABI");
        d[i] = (sluamd_int_t) src[i];
    }
    if (n_narrowed == cap_narrowed) {
        cap_narrowed = cap_narrowed ? 2 * cap_narrowed : 4096;
        narrowed = (void **) realloc(narrowed, sizeof(void *) * cap_narrowed);
        if (!narrowed) ABORT("sluamd binding: out of memory (narrowed index table)");
    }
    narrowed[n_narrowed++] = d;
    return d;
}
static void sluamd_narrow_release(void) { for (size_t i = 0; i < n_narrowed; ++i) free(narrowed[i]); n_narrowed = 0; }
#define NARROW(p, n) sluamd_narrow((p), (size_t) (n))
#else
#define NARROW(p, n) (p)
static void sluamd_narrow_release(void) {}
#endif

/* The library is C++/HIP; a C/MPI application binds it at run time (dlopen) so that the application's own
 * link line (here: conda's MPICH toolchain with an older libstdc++) does not have to resolve the HIP runtime.
 * A maintainer linking with the ROCm toolchain can call the sluamd_* functions directly instead. */
static struct {
    void *so;
    void (*default_options)(sluamd_options_t *);
    int (*create)(sluamd_handle_t *, const LUVIEW_T *, const sluamd_forest_view_t *, const sluamd_options_t *);
    int (*create_grid)(sluamd_handle_t *, const LUVIEW_T *, const sluamd_forest_view_t *, const sluamd_options_t *, sluamd_comm_t);
    int (*factor)(sluamd_handle_t, double, int *);
    int (*copy2host)(sluamd_handle_t, const LUVIEW_T *);
    int (*solve_dist)(sluamd_handle_t, void *, int64_t, int32_t, int64_t, int64_t, const sluamd_int_t *, const sluamd_int_t *);   /* double or doublecomplex rows */
    int (*stats)(sluamd_handle_t, sluamd_stats_t *);
    int (*rccl_id)(void *);
    int (*comm_create_rccl)(sluamd_comm_t *, const void *, int, int, int, int, int, int, int);
    void (*destroy)(sluamd_handle_t);
    int (*comm_create)(sluamd_comm_t *, const sluamd_comm_callbacks_t *, int, int, int, int, int, int);
    void (*comm_destroy)(sluamd_comm_t);
    const char *(*last_error)(void);
} S;

static void sluamd_load(void)
{
    if (S.so) return;
    char path[4096];
    const char *env = getenv("SLUAMD_LIB");
    if (env) snprintf(path, sizeof path, "%s", env);
    else {   /* this repository's test binaries: <repo>/oracle/_ref/<exe>  ->  <repo>/superlu_dist_amd/libsluamd.so */
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
    S.create_grid = (int (*)(sluamd_handle_t *, const LUVIEW_T *, const sluamd_forest_view_t *, const sluamd_options_t *, sluamd_comm_t)) dlsym(S.so, SYM_CREATE_GRID);
    S.factor = (int (*)(sluamd_handle_t, double, int *)) dlsym(S.so, SYM_FACTOR);
    S.copy2host = (int (*)(sluamd_handle_t, const LUVIEW_T *)) dlsym(S.so, SYM_COPY);
    S.solve_dist = (int (*)(sluamd_handle_t, void *, int64_t, int32_t, int64_t, int64_t, const sluamd_int_t *, const sluamd_int_t *)) dlsym(S.so, SYM_SOLVE_DIST);
    S.rccl_id = (int (*)(void *)) dlsym(S.so, "sluamd_comm_rccl_unique_id");
    S.comm_create_rccl = (int (*)(sluamd_comm_t *, const void *, int, int, int, int, int, int, int)) dlsym(S.so, "sluamd_comm_create_rccl");
    S.stats = (int (*)(sluamd_handle_t, sluamd_stats_t *)) dlsym(S.so, "sluamd_get_stats");
    S.destroy = (void (*)(sluamd_handle_t)) dlsym(S.so, "sluamd_dDestroyLUHandle");
    S.comm_create = (int (*)(sluamd_comm_t *, const sluamd_comm_callbacks_t *, int, int, int, int, int, int)) dlsym(S.so, "sluamd_comm_create_callbacks");
    S.comm_destroy = (void (*)(sluamd_comm_t)) dlsym(S.so, "sluamd_comm_destroy");
    S.last_error = (const char *(*)(void)) dlsym(S.so, "sluamd_last_error");
    if (!S.create || !S.create_grid || !S.factor || !S.copy2host || !S.solve_dist || !S.destroy || !S.comm_create || !S.rccl_id || !S.comm_create_rccl)
        ABORT("libsluamd.so lacks a required symbol");
}

/* ---- MPI transport for sluamd_comm_callbacks_t (host buffers; the library stages device ranges) ---- */
static struct {
    MPI_Comm comm;          /* grid3d->comm */
    int *mpi_rank_of;       /* library world rank (z * Pr + r) * Pc + c  ->  rank in grid3d->comm */
    MPI_Request *req; int nreq, cap;
} M;

static int m_push(MPI_Request r)
{
    if (M.nreq == M.cap) { M.cap = M.cap ? 2 * M.cap : 64; M.req = (MPI_Request *) realloc(M.req, sizeof(MPI_Request) * M.cap); }
    M.req[M.nreq++] = r;
    return 0;
}
static int m_isend(void *ctx, const void *buf, int64_t bytes, int peer)
{
    (void) ctx;
    for (int64_t o = 0; o < bytes; o += INT_MAX) {   /* MPI counts are int */
        MPI_Request r; int n = (int) (bytes - o < INT_MAX ? bytes - o : INT_MAX);
        if (MPI_Isend((const char *) buf + o, n, MPI_BYTE, M.mpi_rank_of[peer], 4711, M.comm, &r) != MPI_SUCCESS) return 1;
        m_push(r);
    }
    return 0;
}
static int m_irecv(void *ctx, void *buf, int64_t bytes, int peer)
{
    (void) ctx;
    for (int64_t o = 0; o < bytes; o += INT_MAX) {
        MPI_Request r; int n = (int) (bytes - o < INT_MAX ? bytes - o : INT_MAX);
        if (MPI_Irecv((char *) buf + o, n, MPI_BYTE, M.mpi_rank_of[peer], 4711, M.comm, &r) != MPI_SUCCESS) return 1;
        m_push(r);
    }
    return 0;
}
static int m_waitall(void *ctx)
{
    (void) ctx;
    int rc = M.nreq ? MPI_Waitall(M.nreq, M.req, MPI_STATUSES_IGNORE) : MPI_SUCCESS;
    M.nreq = 0;
    return rc != MPI_SUCCESS;
}
static int m_allmin(void *ctx, int32_t *v)
{
    (void) ctx;
    int in = *v, out = 0;
    if (MPI_Allreduce(&in, &out, 1, MPI_INT, MPI_MIN, M.comm) != MPI_SUCCESS) return 1;
    *v = out;
    return 0;
}

/* the communicator the library runs its exchanges over: MPI callbacks (host-staged) or RCCL (one rank per GPU; the unique id
 * is shipped with MPI_Bcast like any NCCL application does, the device is the rank's index among the ranks of its node) */
static struct {
    sluamd_handle_t h; sluamd_comm_t comm; int n;
    /* lazy copy-back: the factors of the last factorisation live on the device only until a host consumer asks for them */
    int host_stale; xLUstruct_t *LUstruct; gridinfo3d_t *grid3d; int_t nsupers;
} G;
static int bind_comm_create(MPI_Comm comm, int Pr, int Pc, int Pz, int myrow, int mycol, int myz, int use_rccl)
{
    if (use_rccl) {
        int rank; MPI_Comm_rank(comm, &rank);
        int mine = (myz * Pr + myrow) * Pc + mycol, root_is_me = (mine == 0), root = 0, cand = root_is_me ? rank : 0;
        MPI_Allreduce(&cand, &root, 1, MPI_INT, MPI_MAX, comm);          /* the MPI rank that is library world rank 0 creates the id */
        char id[SLUAMD_UNIQUE_ID_BYTES];
        memset(id, 0, sizeof id);
        if (root_is_me && S.rccl_id(id)) return 1;
        MPI_Bcast(id, SLUAMD_UNIQUE_ID_BYTES, MPI_BYTE, root, comm);
        MPI_Comm node; int local = 0;
        MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
        MPI_Comm_rank(node, &local);
        MPI_Comm_free(&node);
        const char *dv = getenv("SLUAMD_BIND_DEVICE");                  /* override, e.g. a launcher that already set HIP_VISIBLE_DEVICES per rank */
        return S.comm_create_rccl(&G.comm, id, Pr, Pc, Pz, myrow, mycol, myz, dv ? atoi(dv) : local);
    }
    sluamd_comm_callbacks_t cb = { NULL, m_isend, m_irecv, m_waitall, m_allmin };
    return S.comm_create(&G.comm, &cb, Pr, Pc, Pz, myrow, mycol, myz);
}

/* the device-resident factors stay alive between pdgstrf3d and the solves (pdgssvx3d calls pdgstrs3d once, pdgsrfs3d
 * once per refinement step); released when the next factorisation starts or at exit */
static void sluamd_bind_release(void)
{
    if (G.h) { S.destroy(G.h); G.h = NULL; }
    G.host_stale = 0; G.LUstruct = NULL; G.grid3d = NULL;
    if (G.comm) { S.comm_destroy(G.comm); G.comm = NULL; }
}

/* the reference's dLocalLU_t of this rank as the library's view (pointer arrays as they are; a _LONGINT build hands over narrowed
 * copies of the index arrays) */
static void bind_view_build(LUVIEW_T *v, int n, xLUstruct_t *LUstruct, gridinfo3d_t *grid3d)
{
    gridinfo_t *grid = &grid3d->grid2d;
    Glu_persist_t *Glu = LUstruct->Glu_persist;
    xLocalLU_t *Llu = LUstruct->Llu;
    int_t nsupers = Glu->supno[n - 1] + 1;
    const int Pr = grid->nprow, Pc = grid->npcol, Pz = grid3d->npdep;
    v->n = n; v->nsupers = (int32_t) nsupers; v->xsup = NARROW(Glu->xsup, nsupers + 1);
    v->nprow = Pr; v->npcol = Pc; v->npdep = Pz;
    v->myrow = MYROW(grid->iam, grid); v->mycol = MYCOL(grid->iam, grid); v->myzlayer = grid3d->zscp.Iam;
    v->Lnzval_bc_ptr = VALPP(Llu->Lnzval_bc_ptr); v->Unzval_br_ptr = VALPP(Llu->Unzval_br_ptr);
#if defined(_LONGINT)
    {   /* index arrays of the local block columns / block rows: lengths from their headers (superlu_defs.h:156-198) */
        const int_t nbc = CEILING(nsupers, Pc), nbr = CEILING(nsupers, Pr);
        v->Lrowind_bc_ptr = (sluamd_int_t **) calloc(nbc ? nbc : 1, sizeof(sluamd_int_t *));
        v->Ufstnz_br_ptr = (sluamd_int_t **) calloc(nbr ? nbr : 1, sizeof(sluamd_int_t *));
        for (int_t lk = 0; lk < nbc; ++lk) {
            const int_t *li = Llu->Lrowind_bc_ptr[lk];
            if (li) v->Lrowind_bc_ptr[lk] = NARROW(li, BC_HEADER + li[0] * LB_DESCRIPTOR + li[1]);
        }
        for (int_t lb = 0; lb < nbr; ++lb) {
            const int_t *ui = Llu->Ufstnz_br_ptr[lb];
            if (ui) v->Ufstnz_br_ptr[lb] = NARROW(ui, ui[2]);
        }
    }
#else
    v->Lrowind_bc_ptr = Llu->Lrowind_bc_ptr; v->Ufstnz_br_ptr = Llu->Ufstnz_br_ptr;
#endif
}
static void bind_view_free(LUVIEW_T *v)
{
#if defined(_LONGINT)
    free(v->Lrowind_bc_ptr); free(v->Ufstnz_br_ptr);
#endif
    (void) v;
    sluamd_narrow_release();
}

/* Copy-back policy.  The reference's GPU path copies the factors to the host after every factorisation (dCopyLUGPU2Host,
 * pdgssvx3d.c:1013-1021): 17 GB at 100^3, longer than the factorisation itself -- and nothing reads them when both triangular solves
 * are bound to the library, which keeps the factors on the device.  But the binding cannot see at link time whether the solves were
 * wrapped: an integration that replaces pdgstrf3d only (the reference's own GPU flow: copy back, then the CPU pdgstrs3d) must find the
 * factored values in LUstruct->Llu.  So the DEFAULT IS EAGER, like the reference, and the copy is deferred only when somebody said so:
 *   SLUAMD_BIND_COPYBACK=eager  copy after every factorisation (the reference's behaviour; the default)
 *   SLUAMD_BIND_COPYBACK=lazy   leave them on the device; BIND_SYNC_HOST[_FOR]() copies on the first host consumer -- call it before
 *                               anything reads LUstruct->Llu's values on the host (the CPU solves pdgstrs3d[_newsolve] when they are NOT
 *                               bound, pdCompute_Diag_Inv, dbroadcastAncestor3d + the CPU solve on a Z-replicated grid,
 *                               dgatherAllFactoredLU, dwriteLUtoDisk / dDumpLblocks3D); dDestroy_LU needs none
 *   unset                       lazy only if (a) the integrator declared the solves bound -- sluamd_bind_[dz]solves_bound(1), one call
 *                               next to the place where the --wrap / call-site replacement of pdgstrs3d[_newsolve] is made -- or (b) a
 *                               bound solve has already run in this process (the binding then KNOWS the solves land in the library);
 *                               otherwise eager.  The first deferred copy prints one line on stderr (SLUAMD_BIND_QUIET=1 silences it). */
static int g_solves_declared = 0, g_solve_seen = 0;
#ifndef Z_PREC
#define BIND_SYNC_HOST sluamd_bind_dsync_host
#define BIND_SYNC_HOST_FOR sluamd_bind_dsync_host_for
#define BIND_SOLVES_BOUND sluamd_bind_dsolves_bound
#define BIND_INVALIDATE sluamd_bind_dinvalidate
#else
#define BIND_SYNC_HOST sluamd_bind_zsync_host
#define BIND_SYNC_HOST_FOR sluamd_bind_zsync_host_for
#define BIND_SOLVES_BOUND sluamd_bind_zsolves_bound
#define BIND_INVALIDATE sluamd_bind_zinvalidate
#endif
void BIND_SOLVES_BOUND(int yes) { g_solves_declared = yes != 0; }
static int bind_copyback_lazy(void)
{
    const char *cb = getenv("SLUAMD_BIND_COPYBACK");
    if (cb) return !strcmp(cb, "lazy");
    return g_solves_declared || g_solve_seen;
}
/* The factors of the last factorisation -> the host arrays of (LUstruct, grid3d), if they are not there yet; returns 1 when a copy was
 * made, 0 when there was nothing to do, -1 when LUstruct does not describe the factored matrix any more (other supernode partition).
 * The caller passes the structures it is about to read: nothing cached is dereferenced. */
int BIND_SYNC_HOST_FOR(xLUstruct_t *LUstruct, gridinfo3d_t *grid3d)
{
    if (!G.h || !G.host_stale) return 0;
    if (!LUstruct || !grid3d || !LUstruct->Glu_persist || !LUstruct->Llu || !LUstruct->Glu_persist->supno || !LUstruct->Glu_persist->xsup) return -1;
    if (G.n <= 0 || (int_t) LUstruct->Glu_persist->supno[G.n - 1] + 1 != G.nsupers || (int_t) LUstruct->Glu_persist->xsup[G.nsupers] != (int_t) G.n) return -1;
    LUVIEW_T v;
    bind_view_build(&v, G.n, LUstruct, grid3d);
    const double t0 = SuperLU_timer_();
    int rc = S.copy2host(G.h, &v);                                           /* was dCopyLUGPU2Host       */
    bind_view_free(&v);
    if (rc) ABORT(S.last_error());
    G.host_stale = 0;
    if (getenv("SLUAMD_BIND_DEBUG")) fprintf(stderr, "[sluamd_bind] copyback on demand %.3f ms\n", 1e3 * (SuperLU_timer_() - t0));
    return 1;
}
/* the same on the structures pdgstrf3d was called with.  Only valid while they are alive: call BIND_INVALIDATE() before dDestroy_LU /
 * dLUstructFree / superlu_gridexit3d or a re-distribution that reallocates Llu's arrays -- or use BIND_SYNC_HOST_FOR, which caches nothing */
int BIND_SYNC_HOST(void) { return BIND_SYNC_HOST_FOR(G.LUstruct, G.grid3d); }
/* forget the structures of the last factorisation (a pending deferred copy is dropped: the host arrays are going away) */
void BIND_INVALIDATE(void) { G.LUstruct = NULL; G.grid3d = NULL; G.host_stale = 0; }

int_t BIND_NAME(superlu_dist_options_t *options, int m, int n, double anorm,
                            xtrf3Dpartition_t *trf3Dpartition, SCT_t *SCT, xLUstruct_t *LUstruct,
                            gridinfo3d_t *grid3d, SuperLUStat_t *stat, int *info)
{
    gridinfo_t *grid = &grid3d->grid2d;
    const int Pr = grid->nprow, Pc = grid->npcol, Pz = grid3d->npdep;
    const int myrow = MYROW(grid->iam, grid), mycol = MYCOL(grid->iam, grid), myz = grid3d->zscp.Iam;

    LUVIEW_T v;
    bind_view_build(&v, n, LUstruct, grid3d);

    /* elimination forests of this layer (dtrf3Dpartition_t, superlu_ddefs.h:317-337) */
    int maxLvl = log2i(grid3d->zscp.Np) + 1, nf = (1 << maxLvl) - 1;
    int32_t *nNodes = (int32_t *) calloc(nf, sizeof(int32_t));
    const sluamd_int_t **lists = (const sluamd_int_t **) calloc(nf, sizeof(*lists));
    for (int f = 0; f < nf; ++f)
        if (trf3Dpartition->sForests[f]) { nNodes[f] = (int32_t) trf3Dpartition->sForests[f]->nNodes; lists[f] = NARROW(trf3Dpartition->sForests[f]->nodeList, nNodes[f]); }
    sluamd_forest_view_t fv = { maxLvl, NARROW(trf3Dpartition->myTreeIdxs, maxLvl), NARROW(trf3Dpartition->myZeroTrIdxs, maxLvl), nf, nNodes, lists };

    sluamd_load();
    static int registered = 0;
    if (!registered) { atexit(sluamd_bind_release); registered = 1; }
    sluamd_bind_release();
    sluamd_options_t o;
    S.default_options(&o);
    o.replace_tiny_pivot = (options->ReplaceTinyPivot == YES);
    o.info_rule = SLUAMD_INFO_REFERENCE;       /* a drop-in user gets the zero-pivot rule of the reference's code (pdgstrf2.c:568-571, MIN over ranks pdgstrf3d.c:388-392) */

    int rc;
    const char *tr = getenv("SLUAMD_BIND_TRANSPORT");
    const int use_rccl = tr && !strcmp(tr, "rccl");
    const double t_start = SuperLU_timer_();
    if (Pr * Pc * Pz > 1 || use_rccl) {
        /* library world rank of every MPI rank of grid3d->comm */
        int P; MPI_Comm_size(grid3d->comm, &P);
        int mine = (myz * Pr + myrow) * Pc + mycol;
        int *all = (int *) malloc(sizeof(int) * P);
        MPI_Allgather(&mine, 1, MPI_INT, all, 1, MPI_INT, grid3d->comm);
        M.comm = grid3d->comm;
        M.mpi_rank_of = (int *) realloc(M.mpi_rank_of, sizeof(int) * P);
        for (int q = 0; q < P; ++q) M.mpi_rank_of[all[q]] = q;
        free(all);
        rc = bind_comm_create(grid3d->comm, Pr, Pc, Pz, myrow, mycol, myz, use_rccl);
        if (rc) ABORT(S.last_error());
        rc = S.create_grid(&G.h, &v, &fv, &o, G.comm);                       /* was dCreateLUgpuHandle    */
    } else {
        rc = S.create(&G.h, &v, &fv, &o);
    }
    if (rc) ABORT(S.last_error());
    G.n = n; G.nsupers = (int_t) v.nsupers;
    double thresh = smach_dist("Epsilon") * anorm;                           /* pdgstrf3d.c:132-133       */
    rc = S.factor(G.h, thresh, info);                                        /* was pdgstrf3d_LUv1: collective, info already MIN over the grid */
    if (rc) ABORT(S.last_error());
    G.LUstruct = LUstruct; G.grid3d = grid3d; G.host_stale = 1;
    if (!bind_copyback_lazy()) {
        const double t0 = SuperLU_timer_();
        rc = S.copy2host(G.h, &v);                                           /* was dCopyLUGPU2Host       */
        if (rc) ABORT(S.last_error());
        G.host_stale = 0;
        if (getenv("SLUAMD_BIND_DEBUG")) fprintf(stderr, "[sluamd_bind] copyback eager %.3f ms\n", 1e3 * (SuperLU_timer_() - t0));
    } else {
        static int warned = 0;
        if (!warned && !getenv("SLUAMD_BIND_QUIET") && myrow + mycol + myz == 0) {
            warned = 1;
            fprintf(stderr, "[sluamd_bind] the factors stay on the device (deferred copy-back): LUstruct->Llu holds UNFACTORED values until "
                            "sluamd_bind_%csync_host[_for]() is called; SLUAMD_BIND_COPYBACK=eager restores the reference's copy\n",
#ifdef Z_PREC
                    'z'
#else
                    'd'
#endif
                    );
        }
        if (getenv("SLUAMD_BIND_DEBUG")) fprintf(stderr, "[sluamd_bind] copyback deferred (factors stay on the device)\n");
    }
    sluamd_stats_t st;
    S.stats(G.h, &st);
    if (getenv("SLUAMD_BIND_DEBUG")) {
        int wmax = 0;
        for (int k = 0; k < v.nsupers; ++k) if (v.xsup[k + 1] - v.xsup[k] > wmax) wmax = v.xsup[k + 1] - v.xsup[k];
        fprintf(stderr, "[sluamd_bind] rank (%d,%d,%d) info %d nnzL %lld nnzU %lld factor_ms %.3f launches %d widest_supernode %d\n", myrow,
                mycol, myz, *info, (long long) st.nnz_L, (long long) st.nnz_U, st.t_factor_ms, st.num_launches, wmax);
    }
    stat->ops[FACT] += (flops_t) (st.flops_schur_padded + st.flops_panel);   /* scuStatUpdate's tally     */
    stat->TinyPivots += st.tiny_pivots;
    /* the reference brackets its level loop with SCT->pdgstrfTimer (pdgstrf3d.c:331, :395) and pdgssvx3d reports stat->utime[FACT]
     * around the call: the numeric phase on the device (HIP events, upload and download of the panels excluded) and the whole call */
    if (SCT) SCT->pdgstrfTimer = 1e-3 * st.t_factor_ms;
    stat->utime[FACT] = SuperLU_timer_() - t_start;
    free(nNodes); free(lists);
    bind_view_free(&v);
    (void) m;
    return 0;
}

/* pdgstrs3d / pdgstrs3d_newsolve (pdgstrs3d.c:6604 / :6935): B holds this rank's m_loc rows (from fst_row) of the right-hand side
 * in the ORIGINAL row order on the layer-0 grid (the other layers' copies are not read: refinement-step right-hand sides exist on
 * layer 0 only, pdgsrfs3d); on return the same rows of the solution of the PERMUTED system (pdgssvx3d applies Pc^T itself). */
/* B stays distributed -- row i of B goes to row perm_c[perm_r[i]] of the factored system inside the library
 * (pdReDistribute3d_B_to_X, :6265), the solution rows come back the same way (pdReDistribute3d_X_to_B, :6404) */
#ifndef Z_PREC
typedef double bind_scalar_t;
#define BIND_SOLVE_NEW sluamd_bind_pdgstrs3d_newsolve
#define BIND_SOLVE_OLD sluamd_bind_pdgstrs3d
#else
typedef doublecomplex bind_scalar_t;
#define BIND_SOLVE_NEW sluamd_bind_pzgstrs3d_newsolve
#define BIND_SOLVE_OLD sluamd_bind_pzgstrs3d
#endif
static void bind_solve(int_t n, xScalePermstruct_t *SP, gridinfo3d_t *grid3d, bind_scalar_t *B, int_t m_loc, int_t fst_row, int_t ldb, int nrhs,
                       SuperLUStat_t *stat, int *info)
{
    *info = 0;
    if (n < 0) { *info = -1; return; }
    if (nrhs < 0) { *info = -9; return; }
    if (!G.h || G.n != n) ABORT("sluamd binding: the triangular solve was called without a factorisation on the device");
    g_solve_seen = 1;                                                        /* the solves DO land here: later factorisations may defer their copy-back */
    if (nrhs == 0) return;
    static sluamd_int_t *perm = NULL; static int_t perm_n = -1;
    if (perm_n != n) { free(perm); perm = (sluamd_int_t *) malloc(sizeof(sluamd_int_t) * (size_t) (n ? n : 1)); perm_n = n; }
    if (!perm) ABORT("sluamd binding: out of memory");
    for (int_t i = 0; i < n; ++i) perm[i] = (sluamd_int_t) SP->perm_c[SP->perm_r[i]];
    const int layer0 = grid3d->zscp.Iam == 0;
    double t0 = SuperLU_timer_();
    if (S.solve_dist(G.h, (void *) B, ldb, nrhs, layer0 ? (int64_t) m_loc : 0, layer0 ? (int64_t) fst_row : 0, perm, NULL)) ABORT(S.last_error());
    stat->utime[SOLVE] = SuperLU_timer_() - t0;
}

void BIND_SOLVE_NEW(superlu_dist_options_t *options, int_t n, xLUstruct_t *LUstruct, xScalePermstruct_t *SP,
                    xtrf3Dpartition_t *part, gridinfo3d_t *grid3d, bind_scalar_t *B, int_t m_loc, int_t fst_row, int_t ldb,
                    int nrhs, xSOLVEstruct_t *SOLVEstruct, SuperLUStat_t *stat, int *info)
{
    (void) options; (void) LUstruct; (void) part; (void) SOLVEstruct;
    bind_solve(n, SP, grid3d, B, m_loc, fst_row, ldb, nrhs, stat, info);
}
void BIND_SOLVE_OLD(superlu_dist_options_t *options, int_t n, xLUstruct_t *LUstruct, xScalePermstruct_t *SP,
                    xtrf3Dpartition_t *part, gridinfo3d_t *grid3d, bind_scalar_t *B, int_t m_loc, int_t fst_row, int_t ldb,
                    int nrhs, xSOLVEstruct_t *SOLVEstruct, SuperLUStat_t *stat, int *info)
{
    (void) options; (void) LUstruct; (void) part; (void) SOLVEstruct;
    bind_solve(n, SP, grid3d, B, m_loc, fst_row, ldb, nrhs, stat, info);
}
