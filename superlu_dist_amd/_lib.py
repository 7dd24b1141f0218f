"""ctypes binding of the C ABI in include/superlu_dist_amd.h (libsluamd.so, built in-tree by
superlu_dist_amd/csrc/Makefile).  No fallback: a missing library or a missing GPU raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SLUAMD_LIB") or os.path.join(_HERE, "libsluamd.so")   # SLUAMD_LIB: another build of the same library (A/B timing on one box)

int_t = C.c_int32
P_int = C.POINTER(C.c_int32)
P_dbl = C.POINTER(C.c_double)


class LUView(C.Structure):
    _fields_ = [("n", C.c_int64), ("nsupers", C.c_int32), ("xsup", P_int),
                ("nprow", C.c_int32), ("npcol", C.c_int32), ("npdep", C.c_int32),
                ("myrow", C.c_int32), ("mycol", C.c_int32), ("myzlayer", C.c_int32),
                ("Lrowind_bc_ptr", C.POINTER(P_int)), ("Lnzval_bc_ptr", C.POINTER(P_dbl)),
                ("Ufstnz_br_ptr", C.POINTER(P_int)), ("Unzval_br_ptr", C.POINTER(P_dbl))]


class ForestView(C.Structure):
    _fields_ = [("maxLvl", C.c_int32), ("myTreeIdxs", P_int), ("myZeroTrIdxs", P_int),
                ("numForests", C.c_int32), ("nNodes", P_int), ("nodeList", C.POINTER(P_int))]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("replace_tiny_pivot", C.c_int32), ("deterministic", C.c_int32),
                ("verbose", C.c_int32), ("info_rule", C.c_int32), ("reserved_i", C.c_int32), ("reserved", C.c_double * 3)]


class Stats(C.Structure):
    _fields_ = [("flops_schur_padded", C.c_double), ("flops_schur_exact", C.c_double), ("flops_panel", C.c_double),
                ("t_factor_ms", C.c_double), ("t_schur_ms", C.c_double), ("t_panel_ms", C.c_double),
                ("t_solve_ms", C.c_double), ("t_h2d_ms", C.c_double), ("t_d2h_ms", C.c_double),
                ("nnz_L", C.c_int64), ("nnz_U", C.c_int64), ("bytes_device", C.c_int64),
                ("num_levels", C.c_int32), ("num_launches", C.c_int32), ("tiny_pivots", C.c_int32),
                ("reserved_i", C.c_int32), ("schur_launches", C.c_int64), ("schur_tiles", C.c_int64),
                ("schur_bytes_alg", C.c_double), ("chain_units", C.c_int64), ("chain_levels", C.c_int32),
                ("solve_launches", C.c_int32), ("t_exchange_ms", C.c_double), ("t_reduce_ms", C.c_double),
                ("t_schur_big_ms", C.c_double), ("flops_schur_exact_big", C.c_double), ("schur_bytes_alg_big", C.c_double)]


EXPORTS = [
    "sluamd_default_options", "sluamd_dCreateLUHandle", "sluamd_dSetValues", "sluamd_pdgstrf3d",
    "sluamd_dCopyLU2Host", "sluamd_pdgstrs3d", "sluamd_pdgstrs3d_dev", "sluamd_pdgstrs3d_dist", "sluamd_pzgstrs3d_dist", "sluamd_dDestroyLUHandle",
    "sluamd_get_stats", "sluamd_setup_times", "sluamd_poisson3d", "sluamd_dGetDiagInv", "sluamd_last_error", "sluamd_device_count", "sluamd_device_pci_bus_id", "sluamd_device_pool_trim", "sluamd_plan_table", "sluamd_dsymbfact", "sluamd_dsymbfact_unsym", "sluamd_order_nd", "sluamd_symb_info",
    "sluamd_symb_view", "sluamd_symb_grid_footprint", "sluamd_ddistribute_host", "sluamd_dCreateLUHandleFromSymb", "sluamd_symb_free",
    "sluamd_zCreateLUHandle", "sluamd_zSetValues", "sluamd_pzgstrf3d", "sluamd_zCopyLU2Host", "sluamd_pzgstrs3d",
    "sluamd_dAttachMatrix", "sluamd_pdgsrfs3d", "sluamd_pdgsrfs3d_dev",
    "sluamd_comm_rccl_unique_id", "sluamd_comm_create_rccl", "sluamd_comm_create_callbacks", "sluamd_comm_create_local",
    "sluamd_comm_selftest", "sluamd_comm_rank", "sluamd_comm_size", "sluamd_comm_destroy", "sluamd_dCreateLUHandleGrid",
    "sluamd_dCreateLUHandleFromSymbGrid", "sluamd_zCreateLUHandleGrid", "sluamd_zCreateLUHandleFromSymbGrid",
]

# sluamd_comm_callbacks_t
ISEND_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
IRECV_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
WAITALL_T = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLMIN_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32))


class CommCallbacks(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("isend", ISEND_T), ("irecv", IRECV_T), ("waitall", WAITALL_T),
                ("allreduce_min_i32", ALLMIN_T)]


_lib = None


def bind(L):
    """Declare the C ABI's signatures on an opened library object."""
    L.sluamd_last_error.restype = C.c_char_p
    L.sluamd_dDestroyLUHandle.restype = None
    L.sluamd_symb_free.restype = None
    L.sluamd_default_options.restype = None
    L.sluamd_comm_destroy.restype = None
    L.sluamd_dsymbfact.argtypes = [C.POINTER(C.c_void_p), C.c_int64, P_int, P_int, P_int, C.c_int32, C.c_int32, P_int]
    L.sluamd_dsymbfact_unsym.argtypes = L.sluamd_dsymbfact.argtypes
    L.sluamd_order_nd.argtypes = [C.c_int64, P_int, P_int, C.c_int32, P_int]
    L.sluamd_symb_info.argtypes = [C.c_void_p, P_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64), P_dbl]
    L.sluamd_symb_view.argtypes = [C.c_void_p, C.POINTER(LUView)]
    L.sluamd_ddistribute_host.argtypes = [C.c_void_p, P_int, P_int, P_dbl, P_int]
    L.sluamd_symb_free.argtypes = [C.c_void_p]
    P64 = C.POINTER(C.c_int64)
    L.sluamd_symb_export.argtypes = [C.c_void_p, P_int, P64, P_int, P64, P_dbl, P64, P_int, P64, P_dbl]
    L.sluamd_dCreateLUHandle.argtypes = [C.POINTER(C.c_void_p), C.POINTER(LUView), C.POINTER(ForestView), C.POINTER(Options)]
    L.sluamd_dCreateLUHandleGrid.argtypes = [C.POINTER(C.c_void_p), C.POINTER(LUView), C.POINTER(ForestView), C.POINTER(Options), C.c_void_p]
    L.sluamd_dCreateLUHandleFromSymb.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, P_int, P_int, P_dbl, P_int, C.POINTER(Options)]
    L.sluamd_dCreateLUHandleFromSymbGrid.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, P_int, P_int, P_dbl, P_int, C.POINTER(Options),
                                                     P_int, C.c_void_p]
    L.sluamd_zCreateLUHandleFromSymbGrid.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, P_int, P_int, C.c_void_p, P_int, C.POINTER(Options),
                                                     P_int, C.c_void_p]
    L.sluamd_dSetValues.argtypes = [C.c_void_p, C.POINTER(LUView)]
    L.sluamd_pdgstrf3d.argtypes = [C.c_void_p, C.c_double, P_int]
    L.sluamd_factor_info.argtypes = [C.c_void_p, P_int, P_int]
    L.sluamd_dCopyLU2Host.argtypes = [C.c_void_p, C.POINTER(LUView)]
    L.sluamd_pdgstrs3d.argtypes = [C.c_void_p, P_dbl, C.c_int64, C.c_int32]
    L.sluamd_pdgstrs3d_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    L.sluamd_pdgstrs3d_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, P_int, P_int]
    L.sluamd_pzgstrs3d_dist.argtypes = L.sluamd_pdgstrs3d_dist.argtypes
    L.sluamd_dDestroyLUHandle.argtypes = [C.c_void_p]
    L.sluamd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.sluamd_mfma_selftest.argtypes = [P_dbl, P_dbl, P_dbl]
    L.sluamd_dResetValues.argtypes = [C.c_void_p]
    L.sluamd_zCreateLUHandle.argtypes = L.sluamd_dCreateLUHandle.argtypes     # zLUview is layout-identical to dLUview
    L.sluamd_zCreateLUHandleGrid.argtypes = L.sluamd_dCreateLUHandleGrid.argtypes
    L.sluamd_zSetValues.argtypes = [C.c_void_p, C.POINTER(LUView)]
    L.sluamd_zCopyLU2Host.argtypes = [C.c_void_p, C.POINTER(LUView)]
    L.sluamd_pzgstrf3d.argtypes = [C.c_void_p, C.c_double, P_int]
    L.sluamd_pzgstrs3d.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    L.sluamd_zCreateLUHandleFromSymb.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, P_int, P_int, C.c_void_p, P_int, C.POINTER(Options)]
    L.sluamd_symb_partition.argtypes = [C.c_void_p, C.c_int32, P_int]
    L.sluamd_symb_grid_footprint.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, P_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sluamd_set_profile.argtypes = [C.c_void_p, C.c_int]
    L.sluamd_setup_times.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.sluamd_poisson3d.argtypes = [C.c_int32, C.c_int32, C.c_int32, P_int, P_int, P_dbl]
    L.sluamd_poisson3d.restype = C.c_int64
    L.sluamd_dGetDiagInv.argtypes = [C.c_void_p, C.c_int32, P_dbl, P_dbl]
    L.sluamd_dAttachMatrix.argtypes = [C.c_void_p, C.c_int32, P_int, P_int, P_dbl, P_int]
    L.sluamd_pdgsrfs3d.argtypes = [C.c_void_p, P_dbl, C.c_int64, P_dbl, C.c_int64, C.c_int32, P_dbl, P_int]
    L.sluamd_pdgsrfs3d_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, P_dbl, P_int]
    L.sluamd_comm_rccl_unique_id.argtypes = [C.c_void_p]
    L.sluamd_comm_create_rccl.argtypes = [C.POINTER(C.c_void_p), C.c_void_p] + [C.c_int] * 7
    L.sluamd_comm_create_callbacks.argtypes = [C.POINTER(C.c_void_p), C.POINTER(CommCallbacks)] + [C.c_int] * 6
    L.sluamd_comm_create_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    L.sluamd_comm_selftest.argtypes = [C.c_void_p, C.c_int64]
    L.sluamd_comm_rank.argtypes = [C.c_void_p]
    L.sluamd_comm_size.argtypes = [C.c_void_p]
    L.sluamd_comm_destroy.argtypes = [C.c_void_p]
    return L


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError(f"{_SO} not built: run `make -C superlu_dist_amd/csrc` (or __graft_entry__.build()); "
                           "there is no CPU fallback for the hot path")
    _lib = bind(C.CDLL(_SO))
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {load().sluamd_last_error().decode()}")
