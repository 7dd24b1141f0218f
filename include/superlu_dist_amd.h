/*
 * superlu_dist_amd.h -- C ABI of the MI355X (gfx950) implementation of SuperLU_DIST's 3D supernodal
 * LU hot path: numeric factorisation (pdgstrf3d) and block triangular solve (pdgstrs3d).
 *
 * Drop-in boundary.  The reference is plain C; its GPU factorisation is
 * reached through an opaque-handle hook set called from pdgssvx3d
 * (/root/reference/SRC/double/pdgssvx3d.c:1013-1021):
 *
 *     dCreateLUgpuHandle / pdgstrf3d_LUv1 / dCopyLUGPU2Host / dDestroyLUgpuHandle
 *     (/root/reference/SRC/include/superlu_upacked.h:16-29,
 *      /root/reference/SRC/CplusplusFactor/LUgpuCHandle_interface_impl.cu:11-73)
 *
 * The entry points below replace exactly that set (plus the solve hooks), with the reference's
 * struct arguments flattened to plain pointers and sizes so that the library has no dependency on the
 * reference headers.  INTEGRATION.md shows the ~40-line stub a maintainer adds to pdgssvx3d.c to bind
 * dLUstruct_t / gridinfo3d_t / dtrf3Dpartition_t to these calls.
 *
 * All functions return 0 on success, a negative SLUAMD_E* code on failure; none of them falls back to
 * a CPU path: without a HIP device they fail with SLUAMD_ENODEVICE.
 */
#ifndef SUPERLU_DIST_AMD_H
#define SUPERLU_DIST_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference int_t (superlu_defs.h:121-129): 32-bit in the default build, 64-bit with _LONGINT.
 * The ABI carries 32-bit indices (one rank's panels are far below 2^31 rows / columns); value offsets are 64-bit internally.
 * An application built with 64-bit int_t passes narrowed copies of its index arrays -- oracle/ref/sluamd_binding.c does it under
 * #if defined(_LONGINT), range-checked; the value arrays are used in place. */
typedef int32_t sluamd_int_t;

#define SLUAMD_OK 0
#define SLUAMD_EINVAL (-1)
#define SLUAMD_ENODEVICE (-2)
#define SLUAMD_EHIP (-3)
#define SLUAMD_ENOMEM (-4)
#define SLUAMD_ESTRUCT (-5) /* malformed L/U index structure */

/* One rank's view of dLUstruct_t->Llu + Glu_persist + gridinfo3d_t
 * (superlu_ddefs.h:97-172, superlu_defs.h:454-457, :385-420).  Index formats are the reference's own
 * (superlu_defs.h:156-198).  Pointer arrays have ceil(nsupers/npcol) and
 * ceil(nsupers/nprow) entries; absent panels are NULL. */
typedef struct sluamd_dLUview {
    int64_t n;                      /* order of the matrix                                  */
    int32_t nsupers;                /* Glu_persist->supno[n-1]+1                             */
    const sluamd_int_t *xsup;       /* Glu_persist->xsup[nsupers+1]                          */
    int32_t nprow, npcol, npdep;    /* gridinfo3d_t: grid2d.nprow, grid2d.npcol, npdep       */
    int32_t myrow, mycol, myzlayer; /* MYROW(iam), MYCOL(iam), zscp.Iam                      */
    sluamd_int_t **Lrowind_bc_ptr;  /* Llu->Lrowind_bc_ptr                                   */
    double **Lnzval_bc_ptr;         /* Llu->Lnzval_bc_ptr                                    */
    sluamd_int_t **Ufstnz_br_ptr;   /* Llu->Ufstnz_br_ptr                                    */
    double **Unzval_br_ptr;         /* Llu->Unzval_br_ptr                                    */
} sluamd_dLUview_t;

/* complex16 twin (zLUstruct_t->Llu, superlu_zdefs.h): same layout, values are doublecomplex {r, i} (dcomplex.h) */
typedef struct { double r, i; } sluamd_doublecomplex;
typedef struct sluamd_zLUview {
    int64_t n;
    int32_t nsupers;
    const sluamd_int_t *xsup;
    int32_t nprow, npcol, npdep;
    int32_t myrow, mycol, myzlayer;
    sluamd_int_t **Lrowind_bc_ptr;
    sluamd_doublecomplex **Lnzval_bc_ptr;
    sluamd_int_t **Ufstnz_br_ptr;
    sluamd_doublecomplex **Unzval_br_ptr;
} sluamd_zLUview_t;

/* dtrf3Dpartition_t (superlu_ddefs.h:317-337) flattened: elimination forests of this rank's Z layer.
 * May be NULL for a 1x1x1 grid: the library then derives a level schedule from the block structure. */
typedef struct sluamd_forest_view {
    int32_t maxLvl;                   /* log2(npdep)+1                                        */
    const sluamd_int_t *myTreeIdxs;   /* [maxLvl] forest id handled at each Z level           */
    const sluamd_int_t *myZeroTrIdxs; /* [maxLvl] 1 = this layer does not factor that level   */
    int32_t numForests;               /* 2^maxLvl - 1                                         */
    const int32_t *nNodes;            /* [numForests] sForest_t.nNodes                        */
    const sluamd_int_t *const *nodeList; /* [numForests] sForest_t.nodeList (elimination order) */
} sluamd_forest_view_t;

typedef struct sluamd_options {
    int32_t device;             /* HIP device ordinal (-1 = current device)                      */
    int32_t replace_tiny_pivot; /* options->ReplaceTinyPivot == YES (superlu_defs.h:697)         */
    int32_t deterministic;      /* 1: one supernode per Schur launch -> fixed summation order of the factors (the solve
                                 *    still accumulates lsum with fp64 atomics)                 */
    int32_t verbose;
    int32_t info_rule;          /* which zero pivot `info` names on an exactly singular matrix (round 6; a former reserved slot -- the struct's size and the
                                 * offsets of the fields above are unchanged):
                                 *   SLUAMD_INFO_FIRST (0, sluamd_default_options): the smallest zero-pivot column -- the DOCUMENTED meaning (pdgstrf2.c:493-497)
                                 *   SLUAMD_INFO_REFERENCE (1): what the reference's CODE leaves -- Local_Dgstrf2 overwrites *info at every zero pivot
                                 *     (pdgstrf2.c:568-571), so a rank keeps the one it met LAST; pdgstrf3d takes the MIN over the ranks (pdgstrf3d.c:388-392).
                                 *     The reference-side binding (bindings/superlu_dist/sluamd_binding.c) selects this one. */
    int32_t reserved_i;
    double  reserved[3];
} sluamd_options_t;
#define SLUAMD_INFO_FIRST 0
#define SLUAMD_INFO_REFERENCE 1

typedef struct sluamd_stats {
    double flops_schur_padded; /* reference tally 2*nbrow*ldu*ncols (sec_structs.c:692-693), in double */
    double flops_schur_exact;  /* 2*nbrow*segsize per U column                                        */
    double flops_panel;        /* diag LU + the two panel TRSMs                                       */
    double t_factor_ms;        /* last sluamd_pdgstrf3d: HIP-event time of the numeric phase          */
    double t_schur_ms, t_panel_ms; /* split by kernel family (events on the compute stream)           */
    double t_solve_ms;         /* last sluamd_pdgstrs3d                                               */
    double t_h2d_ms, t_d2h_ms;
    int64_t nnz_L, nnz_U;      /* stored entries (incl. explicit zeros of the supernodal format)      */
    int64_t bytes_device;
    int32_t num_levels, num_launches;
    int32_t tiny_pivots;
    int32_t reserved_i;
    int64_t schur_launches, schur_tiles;
    double  schur_bytes_alg;   /* algorithmic HBM bytes of all Schur launches             */
    int64_t chain_units;       /* always 0 (the opt-in persistent dataflow sweeps of rounds 3-5 were removed: slower, NOTEBOOK.md); kept for layout */
    int32_t chain_levels;      /* always 0, as above                                                  */
    int32_t solve_launches;    /* kernel launches of the last sluamd_pdgstrs3d (one right-hand-side chunk) */
    double  t_exchange_ms;     /* grid handles, profiling on: time of the XY panel-exchange phases ... */
    double  t_reduce_ms;       /* ... and of the Z ancestor reduction inside the last sluamd_pdgstrf3d */
    /* the Schur update by tile configuration (round 4): k_schur<128,128,8> -- supernodes of >= 96 columns whose block pairs fill 128 x 128
     * tiles, the MFMA-bound part -- against the 64 x 64 configuration of the narrow supernodes at the bottom of the tree (HBM / latency-bound,
     * "MFMA-bound only when s_k >~ 120") */
    double  t_schur_big_ms;        /* profiling on: HIP-event time of the 128 x 128 launches (part of t_schur_ms) */
    double  flops_schur_exact_big; /* exact-segment flops of the supernodes that use that configuration (part of flops_schur_exact) */
    double  schur_bytes_alg_big;   /* ... and their algorithmic destination bytes (part of schur_bytes_alg) */
} sluamd_stats_t;

typedef struct sluamd_lu_handle_s *sluamd_handle_t;

void sluamd_default_options(sluamd_options_t *opt);

/* Replaces dCreateLUgpuHandle (LUgpuCHandle_interface_impl.cu:11): uploads this rank's L/U index and
 * value arrays to HBM (kept resident for factor + solve), builds the device-side block directories and
 * the level schedule.  `forests` may be NULL (see above). */
int sluamd_dCreateLUHandle(sluamd_handle_t *h, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests,
                           const sluamd_options_t *opt);

/* Re-upload numeric values only (same structure): the SamePattern_SameRowPerm refactor path
 * (superlu_defs.h:545-566). */
int sluamd_dSetValues(sluamd_handle_t h, const sluamd_dLUview_t *lu);

/* Replaces pdgstrf3d_LUv1 (LUgpuCHandle_interface_impl.cu:66) == the numeric work of pdgstrf3d
 * (pdgstrf3d.c:121-439): factors L\U in place in HBM.  thresh = smach("Epsilon")*anorm
 * (pdgstrf3d.c:132-133).  *info: 0, or 1-based column of a zero pivot (pdgstrf2.c:568-571). */
int sluamd_pdgstrf3d(sluamd_handle_t h, double thresh, int *info);

/* Replaces dCopyLUGPU2Host (LUgpuCHandle_interface_impl.cu:43): writes the factored values back into
 * the caller's Lnzval_bc_ptr / Unzval_br_ptr arrays in the reference panel / skyline formats. */
int sluamd_dCopyLU2Host(sluamd_handle_t h, const sluamd_dLUview_t *lu);

/* Replaces the L- and U-solves of pdgstrs3d / pdgstrs3d_newsolve (pdgstrs3d.c:6604, :6935) between
 * pdReDistribute3d_B_to_X and pdReDistribute3d_X_to_B: x is the right-hand side already permuted by
 * Pc*Pr (row index = global row of the factored matrix), n x nrhs column-major with leading dimension
 * ldx, overwritten by the solution of L U y = x.  Host-pointer and device-pointer variants. */
int sluamd_pdgstrs3d(sluamd_handle_t h, double *x, int64_t ldx, int32_t nrhs);
int sluamd_pdgstrs3d_dev(sluamd_handle_t h, double *d_x, int64_t ldx, int32_t nrhs);

/* complex16 twins of the four entry points above and of the solve (pzgstrf3d, SRC/complex16/pzgstrf3d.c; pzgstrs3d,
 * SRC/complex16/pzgstrs3d.c): same semantics on doublecomplex values.  Handles are shared with the double API for
 * sluamd_dDestroyLUHandle / sluamd_get_stats. */
int sluamd_zCreateLUHandle(sluamd_handle_t *h, const sluamd_zLUview_t *lu, const sluamd_forest_view_t *forests,
                           const sluamd_options_t *opt);
int sluamd_zSetValues(sluamd_handle_t h, const sluamd_zLUview_t *lu);
int sluamd_pzgstrf3d(sluamd_handle_t h, double thresh, int *info);
int sluamd_zCopyLU2Host(sluamd_handle_t h, const sluamd_zLUview_t *lu);
int sluamd_pzgstrs3d(sluamd_handle_t h, sluamd_doublecomplex *x, int64_t ldx, int32_t nrhs);

/* Replaces dDestroyLUgpuHandle (LUgpuCHandle_interface_impl.cu:30). */
void sluamd_dDestroyLUHandle(sluamd_handle_t h);

int sluamd_get_stats(sluamd_handle_t h, sluamd_stats_t *out);
/* wall-clock breakdown of the handle's creation as "phase=seconds;phase=seconds;..." (host planner phases, arena allocation, table uploads,
 * distribution of A): the pre-processing a caller pays once per sparsity structure (pddistribute3d + the GPU handle set-up of the reference) */
int sluamd_setup_times(sluamd_handle_t h, char *buf, int32_t cap);
/* The plan of this rank, one row of SLUAMD_PLAN_COLS doubles per (Z level, DAG level): supernodes, Schur / panel flops of THIS rank, bytes and messages of the two XY
 * exchange phases, bytes of the Z reduction that follows the Z level (column list at the definition, sluamd_api.cpp).  buf may be NULL: *rows = rows needed.
 * Host data only (no device work): what a scaling model needs from the library (scripts/scale_model.py; the reference prints the same quantities as its
 * SCT counters commVolFactor / commVolRed, sec_structs.c). */
#define SLUAMD_PLAN_COLS 20
int sluamd_plan_table(sluamd_handle_t h, double *buf, int64_t cap_rows, int64_t *rows);
const char *sluamd_last_error(void);
/* number of visible HIP devices (0 when none) -- lets callers fail loudly instead of falling back */
int sluamd_device_count(void);
/* The value arenas of 1 x 1 x 1 handles come from a process-level pool of physical device chunks (destroyed handles return their chunks to it, later
 * handles re-map them: no driver-side clearing of re-used memory).  This call returns the chunks no handle uses to the driver (dev < 0: every device);
 * result = the bytes the pool held.  The library trims by itself when one of its allocations fails.  SLUAMD_NO_DEVPOOL=1: plain hipMalloc. */
int64_t sluamd_device_pool_trim(int dev);
/* "domain:bus:device.function" of HIP device `dev` (hipDeviceGetPCIBusId): lets a harness find the device's sysfs node, e.g. to sample its clock beside a
 * measurement (bench.py `device_clock`).  buf of at least 16 bytes. */
int sluamd_device_pci_bus_id(int dev, char *buf, int len);

/* ------------------------------------------------------------------------------------------------
 * Host-side producers of the L/U store (rows of SURVEY.md section 8(f) that the hot path needs when the
 * reference's pre-processing is not linked): symmetric-pattern symbolic factorisation + distribution
 * for a 1 x 1 x npdep grid.  Mirrors symbfact_dist / pddistribute3d (symbfact.c, pddistribute3d.c:1357)
 * in effect (same store formats), not in algorithm.
 * ------------------------------------------------------------------------------------------------ */
typedef struct sluamd_symb_s *sluamd_symb_t;

/* A: n x n CSR (rowptr/colind/nzval) of the ORIGINAL matrix; perm_c[old] = new is applied symmetrically
 * (A1 = Pc A Pc^T, i.e. RowPerm = NOROWPERM, ColPerm = MY_PERMC; pdgssvx3d.c:749-791); the etree
 * postorder is composed into perm_c_out like sp_colorder does (sp_colorder.c).  relax / maxsup play the
 * roles of sp_ienv_dist(2) / sp_ienv_dist(3) (sp_ienv.c:95-110). */
int sluamd_dsymbfact(sluamd_symb_t *s, int64_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                     const sluamd_int_t *perm_c, int32_t relax, int32_t maxsup, sluamd_int_t *perm_c_out);
/* The same producer for UNSYMMETRIC patterns with the reference's own supernode rules (symbfact.c:83-200: relax_snode :221-265, the
 * T2 / subset / maxsup boundary test of column_dfs :598-672): the exact structure of L and U of Pc A Pc^T under elimination without
 * pivoting, U as per-column skyline segments -- the store the reference's symbfact + pddistribute3d build for the same perm_c, relax
 * (sp_ienv_dist(2)) and maxsup (sp_ienv_dist(3)): xsup, every L block's row set and every Ufstnz entry agree (tests/test_symbolic_parity.py).
 * The result feeds the same consumers as sluamd_dsymbfact (sluamd_symb_view, sluamd_ddistribute_host, sluamd_dCreateLUHandleFromSymb[Grid],
 * sluamd_symb_partition, sluamd_symb_grid_footprint). */
int sluamd_dsymbfact_unsym(sluamd_symb_t *s, int64_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                           const sluamd_int_t *perm_c, int32_t relax, int32_t maxsup, sluamd_int_t *perm_c_out);
/* Fill-reducing ordering for matrices without geometry (the role of get_perm_c / METIS in the reference,
 * SRC/prec-independent/get_perm_c.c): nested dissection of the pattern of A + A^T by BFS level structures; perm_c[old] = new,
 * to be passed to sluamd_dsymbfact.  leaf = component size below which no further separator is sought (<= 0: 64). */
int sluamd_order_nd(int64_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind, int32_t leaf, sluamd_int_t *perm_c);
/* The synthetic operator of SURVEY.md 8(d) / BASELINE.json configs[1-2] for harnesses: 7-point Poisson on an nx x ny x nz grid, natural index (i ny + j) nz + k,
 * diagonal 6, off-diagonals -1, Dirichlet truncation, CSR with ascending column indices.  rowptr: nx ny nz + 1 entries, colind / nzval:
 * 7 n - 2 (nx ny + ny nz + nx nz) entries; returns that count (or a negative error code). */
int64_t sluamd_poisson3d(int32_t nx, int32_t ny, int32_t nz, sluamd_int_t *rowptr, sluamd_int_t *colind, double *nzval);
int sluamd_symb_info(sluamd_symb_t s, int32_t *nsupers, int64_t *nnzL, int64_t *nnzU, int64_t *lidx_len,
                     int64_t *uidx_len, double *flops);
/* borrow the store in the reference's formats (valid until sluamd_symb_free) */
int sluamd_symb_view(sluamd_symb_t s, sluamd_dLUview_t *view);
/* scatter A's values (original CSR + the same perm_c_out) into the zero-initialised store held by `s`
 * (host) -- what pddistribute3d does for a 1x1x1 grid */
int sluamd_ddistribute_host(sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                            const double *nzval, const sluamd_int_t *perm_c_final);
/* device-resident variant: create the LU handle straight from the symbolic structure; values are
 * zero-filled in HBM and A's entries scattered by a kernel (no host copy of the factors exists) */
int sluamd_dCreateLUHandleFromSymb(sluamd_handle_t *h, sluamd_symb_t s, const sluamd_int_t *rowptr,
                                   const sluamd_int_t *colind, const double *nzval,
                                   const sluamd_int_t *perm_c_final, const sluamd_options_t *opt);
/* copy the store out as flat arrays + offsets (any pointer may be NULL) */
int sluamd_symb_export(sluamd_symb_t s, sluamd_int_t *xsup, int64_t *lidx_off, sluamd_int_t *lidx, int64_t *lval_off,
                       double *lval, int64_t *uidx_off, sluamd_int_t *uidx, int64_t *uval_off, double *uval);
int sluamd_zCreateLUHandleFromSymb(sluamd_handle_t *h, sluamd_symb_t s, const sluamd_int_t *rowptr,
                                   const sluamd_int_t *colind, const sluamd_doublecomplex *nzval,
                                   const sluamd_int_t *perm_c_final, const sluamd_options_t *opt);
void sluamd_symb_free(sluamd_symb_t s);
/* elimination-forest partition for npdep Z layers (getForests' job, supernodalForest.c; tree ids in heap order like
 * getGridTrees, supernodal_etree.c:840-851): sn_tree[k] = forest of supernode k (see sluamd_dCreateLUHandleFromSymbGrid) */
int sluamd_symb_partition(sluamd_symb_t s, int32_t npdep, int32_t *sn_tree);
/* Capacity planning: stored factor values per world rank of an nprow x npcol x npdep grid (own slots, including the ancestor
 * panels every layer below a forest replicates -- dinit3DLUstructForest, pd3dcomm.c:334-800), the part of them that is such a
 * replica, and the index entries, from the symbolic structure alone.  Arrays of nprow*npcol*npdep entries; `replicated` /
 * `index_entries` may be NULL; sn_tree as for sluamd_dCreateLUHandleFromSymbGrid (NULL when npdep == 1). */
int sluamd_symb_grid_footprint(sluamd_symb_t s, int32_t nprow, int32_t npcol, int32_t npdep, const int32_t *sn_tree,
                               int64_t *values, int64_t *replicated, int64_t *index_entries);
/* re-run the device-side distribution (zero-fill + scatter of A) on such a handle: refactor loops */
int sluamd_dResetValues(sluamd_handle_t h);

/* ---- auxiliary (timing / tests) ---- */
int sluamd_device_synchronize(void);
int sluamd_set_profile(sluamd_handle_t h, int on);         /* per-kernel-family HIP-event timing in stats */
int sluamd_factor_info(sluamd_handle_t h, int *info, int *tiny_pivots);
/* Linv / Uinv of the diagonal block of supernode k as the handle keeps them for the panel solves and the triangular sweeps -- what pdCompute_Diag_Inv
 * (SRC/double/pdgstrs.c:842-959: dtrtri on the factored block) leaves in Llu->Linv_bc_ptr / Uinv_bc_ptr: column-major nsupc x nsupc each, unit-lower and
 * upper inverse with explicit zeros in the other triangle.  Host buffers; the rank must own the diagonal block; double handles after a factorisation. */
int sluamd_dGetDiagInv(sluamd_handle_t h, int32_t k, double *Linv, double *Uinv);
int sluamd_mfma_selftest(const double *A16x4, const double *B4x16, double *D16x16);

/* ---- iterative refinement: pdgsrfs3d (SRC/double/pdgsrfs.c:345-510) with its SpMV pdgsmv (SRC/double/pdgsmv.c) on the
 * device, SURVEY 8(f)-2.  The ORIGINAL matrix (CSR, 0-based) and perm_c are attached once; the factors in the handle are
 * those of Pc A Pc^T (Equil=NO, NOROWPERM -- the boundary's convention).  B, X: original ordering, column-major; X holds
 * the initial solution and is refined in place; berr[nrhs] = componentwise backward errors; *steps = refinement steps of
 * the last right-hand side (stat->RefineSteps).  Stopping rule as the reference: berr > eps, berr*2 <= previous, < 20. */
int sluamd_dAttachMatrix(sluamd_handle_t h, sluamd_int_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                         const double *nzval, const sluamd_int_t *perm_c);
int sluamd_pdgsrfs3d(sluamd_handle_t h, const double *B, int64_t ldb, double *X, int64_t ldx, int32_t nrhs, double *berr,
                     int32_t *steps);
int sluamd_pdgsrfs3d_dev(sluamd_handle_t h, const double *d_B, int64_t ldb, double *d_X, int64_t ldx, int32_t nrhs,
                         double *berr, int32_t *steps);

/* ------------------------------------------------------------------------------------------------
 * Process grids: nprow x npcol x npdep ranks, one rank per GPU (gridinfo3d_t, superlu_defs.h:385-420).
 *
 * The reference exchanges panels with MPI inside pdgstrf3d / pdgstrs3d:
 *   - inside a 2-D layer, per supernode k: the factored diagonal block goes down process column k % Pc and along process
 *     row k % Pr (dDiagFactIBCast, dtrfCommWrapper.c:32-118), the L panel along the process rows and the U panel down the
 *     process columns (dIBcastRecvLPanel / dIBcastRecvUPanel, dtrfCommWrapper.c:377-548, dcommunication_aux.c:32-430);
 *   - between layers, after every level of the elimination forest: the pairwise sum of the replicated ancestor panels
 *     (dreduceAllAncestors3d, pd3dcomm.c:1046-1081) and MPI_Allreduce(MIN) of info (pdgstrf3d.c:388-392);
 *   - in the solve: x_k down the process column, lsum along the process row, and the Z exchanges of the ancestor part of
 *     x (pdgstrs3d.c:1405-1535, :7180-7181).
 * Here all of it runs inside the library, in C, over a communicator object: RCCL called directly (ncclSend / ncclRecv
 * groups on the library's HIP streams, so an exchange overlaps the Schur tiles of the previous level), or a transport
 * supplied by the application (MPI in the reference-side binding of INTEGRATION.md), or an in-process world (tests).
 * World rank of grid position (row, col, z) = (z * nprow + row) * npcol + col  (layer-major, superlu_grid3d.c).
 * ------------------------------------------------------------------------------------------------ */
typedef struct sluamd_comm_s *sluamd_comm_t;

#define SLUAMD_UNIQUE_ID_BYTES 128
/* RCCL: rank 0 calls sluamd_comm_rccl_unique_id and ships the 128 bytes to every rank by any means (MPI_Bcast, a file,
 * a torch.distributed store); every rank then calls sluamd_comm_create_rccl (collective: ncclCommInitRank). */
int sluamd_comm_rccl_unique_id(void *id128);
int sluamd_comm_create_rccl(sluamd_comm_t *comm, const void *id128, int nprow, int npcol, int npdep, int myrow, int mycol,
                            int myz, int device);

/* Application-supplied transport on HOST buffers (the library stages device ranges through pinned memory).  isend / irecv
 * start one message to / from world rank `peer` (messages between one pair of ranks match in the order they were
 * started); waitall completes every message started since the last waitall; allreduce_min_i32 runs over the whole grid.
 * All return 0 on success. */
typedef struct sluamd_comm_callbacks {
    void *ctx;
    int (*isend)(void *ctx, const void *buf, int64_t bytes, int peer);
    int (*irecv)(void *ctx, void *buf, int64_t bytes, int peer);
    int (*waitall)(void *ctx);
    int (*allreduce_min_i32)(void *ctx, int32_t *v);
} sluamd_comm_callbacks_t;
int sluamd_comm_create_callbacks(sluamd_comm_t *comm, const sluamd_comm_callbacks_t *cb, int nprow, int npcol, int npdep,
                                 int myrow, int mycol, int myz);

/* In-process world: comms[nprow*npcol*npdep] (indexed by world rank), one per thread; any number of ranks per device. */
int sluamd_comm_create_local(sluamd_comm_t *comms, int nprow, int npcol, int npdep);
/* Transport self-test (collective): ring exchange of `bytes` bytes as one stream-ordered group between device fills, the
 * host-buffer group and the min-all-reduce -- every operation the drivers use of a transport; 0 when all data arrived intact.
 * On a one-rank communicator every message goes to the rank itself. */
int sluamd_comm_selftest(sluamd_comm_t comm, int64_t bytes);
int sluamd_comm_rank(sluamd_comm_t comm);
int sluamd_comm_size(sluamd_comm_t comm);
void sluamd_comm_destroy(sluamd_comm_t comm);

/* sluamd_dCreateLUHandle for one rank of a process grid (collective over `comm`): `lu` is that rank's dLocalLU_t view
 * (pointer arrays of ceil(nsupers/npcol) block columns and ceil(nsupers/nprow) block rows, local index lk = k / npcol,
 * lb = k / nprow; superlu_defs.h:270-279), `forests` its dtrf3Dpartition_t (required when npdep > 1).  The index arrays
 * of the panels a rank will receive are exchanged once here (the reference ships them with every panel message,
 * dIBcast_LPanel, dcommunication_aux.c:32-60).  The handle then works with the single-rank entry points:
 * sluamd_pdgstrf3d / sluamd_pdgstrs3d[_dev] / sluamd_dCopyLU2Host / sluamd_dSetValues become collective calls.
 * `comm` must outlive the handle. */
int sluamd_dCreateLUHandleGrid(sluamd_handle_t *h, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests,
                               const sluamd_options_t *opt, sluamd_comm_t comm);
/* complex16 on any nprow x npcol x npdep grid (pzgstrf3d.c: XY panel exchange -- ztrfCommWrapper.c, zcommunication_aux.c --, the 333-392 ancestor
 * reduction; the distributed pzgstrs3d): complex16 values travel as pairs of doubles through the double path's exchange plans; collective like
 * sluamd_dCreateLUHandleGrid */
int sluamd_zCreateLUHandleGrid(sluamd_handle_t *h, const sluamd_zLUview_t *lu, const sluamd_forest_view_t *forests,
                               const sluamd_options_t *opt, sluamd_comm_t comm);
/* Same from the library's own symbolic factorisation (every rank holds the complete structure `s`; no structure exchange):
 * the store of this rank's grid position is built and A's values are distributed on the device (pddistribute3d +
 * dinit3DLUstructForest, pddistribute3d.c:1357, pd3dcomm.c:334-800).  sn_tree (sluamd_symb_partition) may be NULL when
 * npdep == 1. */
int sluamd_dCreateLUHandleFromSymbGrid(sluamd_handle_t *h, sluamd_symb_t s, const sluamd_int_t *rowptr,
                                       const sluamd_int_t *colind, const double *nzval, const sluamd_int_t *perm_c_final,
                                       const sluamd_options_t *opt, const int32_t *sn_tree, sluamd_comm_t comm);
/* complex16 twin */
int sluamd_zCreateLUHandleFromSymbGrid(sluamd_handle_t *h, sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                                       const sluamd_doublecomplex *nzval, const sluamd_int_t *perm_c_final,
                                       const sluamd_options_t *opt, const int32_t *sn_tree, sluamd_comm_t comm);
/* Grid solve, replicated form (sluamd_pdgstrs3d[_dev] on a grid handle): every rank passes the COMPLETE permuted right-hand side
 * x = Pc*Pr*b (n x nrhs) and every rank receives the complete solution (direct all-gather of the owners' rows).
 *
 * Grid solve, distributed form = the boundary of the reference's pdgstrs3d[_newsolve] (pdgstrs3d.c:6604, :6935) including
 * pdReDistribute3d_B_to_X (:6265) and pdReDistribute3d_X_to_B (:6404): B is distributed by block rows over the processes of
 * layer 0 -- this rank holds rows [fst_row, fst_row + m_loc) of the ORIGINAL right-hand side, column-major with leading dimension
 * ldb (NRformat_loc); ranks of the other layers pass m_loc = 0 -- and is overwritten by rows of the solved vector y (L U y = x).
 * perm_in[i]  = row of the factored system that row i of B goes to: x[perm_in[i]] = B[i], i.e. ScalePermstruct->perm_c[perm_r[i]];
 * perm_out[i] = row of y returned in row i: B[i] = y[perm_out[i]].  The reference returns the rows of the PERMUTED solution
 *               (perm_out = identity; pdgssvx3d applies Pc^T itself); perm_out = perm_c gives x in the original order.
 * Both are replicated on every rank; NULL = identity.  Every row travels once to the rank that consumes it and every row of the
 * solution once back; nothing is replicated.  Collective on grid handles; works on single-rank handles too (m_loc = n). */
int sluamd_pdgstrs3d_dist(sluamd_handle_t h, double *B, int64_t ldb, int32_t nrhs, int64_t m_loc, int64_t fst_row,
                          const sluamd_int_t *perm_in, const sluamd_int_t *perm_out);
/* complex16 twin: the boundary of pzgstrs3d[_newsolve] (SRC/complex16/pzgstrs3d.c) */
int sluamd_pzgstrs3d_dist(sluamd_handle_t h, sluamd_doublecomplex *B, int64_t ldb, int32_t nrhs, int64_t m_loc, int64_t fst_row,
                          const sluamd_int_t *perm_in, const sluamd_int_t *perm_out);

#ifdef __cplusplus
}
#endif
#endif /* SUPERLU_DIST_AMD_H */
