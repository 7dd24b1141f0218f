"""ctypes access to oracle/libslu_oracle.so (CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes, os, subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libslu_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
        _lib = ctypes.CDLL(_SO)
        _lib.slu_oracle_dfactor.restype = ctypes.c_int
        _lib.slu_oracle_zfactor.restype = ctypes.c_int
        _lib.slu_oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


class LUStore:
    """Flat copy of a 1x1x1 reference-format L/U store (see oracle/slu_oracle.c header)."""

    def __init__(self, n, xsup, Lrowind_off, Lrowind, Lnzval_off, Lnzval, Ufstnz_off, Ufstnz, Unzval_off, Unzval):
        self.n = int(n)
        self.xsup = np.ascontiguousarray(xsup, dtype=np.int32)
        self.nsupers = len(self.xsup) - 1
        self.Lrowind_off = np.ascontiguousarray(Lrowind_off, dtype=np.int64)
        self.Lrowind = np.ascontiguousarray(Lrowind, dtype=np.int32)
        self.Lnzval_off = np.ascontiguousarray(Lnzval_off, dtype=np.int64)
        self.dtype = np.complex128 if np.iscomplexobj(Lnzval) or np.iscomplexobj(Unzval) else np.float64
        self.Lnzval = np.array(Lnzval, dtype=self.dtype)
        self.Ufstnz_off = np.ascontiguousarray(Ufstnz_off, dtype=np.int64)
        self.Ufstnz = np.ascontiguousarray(Ufstnz, dtype=np.int32)
        self.Unzval_off = np.ascontiguousarray(Unzval_off, dtype=np.int64)
        self.Unzval = np.array(Unzval, dtype=self.dtype)

    @classmethod
    def from_golden(cls, g, rank=0, which="pre"):
        r = f"r{rank}__"
        return cls(int(g[r + "n"][0]), g[r + "xsup"], g[r + "Lrowind_off"], g[r + "Lrowind"], g[r + "Lnzval_off"],
                   g[r + f"Lnzval_{which}"], g[r + "Ufstnz_off"], g[r + "Ufstnz"], g[r + "Unzval_off"],
                   g[r + f"Unzval_{which}"])

    def copy(self):
        return LUStore(self.n, self.xsup, self.Lrowind_off, self.Lrowind, self.Lnzval_off, self.Lnzval.copy(),
                       self.Ufstnz_off, self.Ufstnz, self.Unzval_off, self.Unzval.copy())

    def _args(self):
        return (ctypes.c_int(self.n), ctypes.c_int(self.nsupers), _p(self.xsup, ctypes.c_int),
                _p(self.Lrowind_off, ctypes.c_int64), _p(self.Lrowind, ctypes.c_int),
                _p(self.Lnzval_off, ctypes.c_int64), self.Lnzval.ctypes.data_as(ctypes.c_void_p),
                _p(self.Ufstnz_off, ctypes.c_int64), _p(self.Ufstnz, ctypes.c_int),
                _p(self.Unzval_off, ctypes.c_int64), self.Unzval.ctypes.data_as(ctypes.c_void_p))

    @property
    def z(self):
        return self.dtype == np.complex128


def dfactor(store, order=None, replace_tiny=False, thresh=0.0):
    """In-place factorisation of `store`; returns (info, tiny, flops[schur_padded, panel])."""
    if order is None:
        order = np.arange(store.nsupers, dtype=np.int32)
    order = np.ascontiguousarray(order, dtype=np.int32)
    info = ctypes.c_int(0)
    flops = np.zeros(2)
    fn = lib().slu_oracle_zfactor if store.z else lib().slu_oracle_dfactor
    tiny = fn(*store._args(), _p(order, ctypes.c_int), ctypes.c_int(len(order)),
                                    ctypes.c_int(int(replace_tiny)), ctypes.c_double(thresh), ctypes.byref(info),
                                    _p(flops, ctypes.c_double))
    return info.value, tiny, flops


def dsolve(store, x):
    """Solve L U x = b on the permuted system; x (n x nrhs, Fortran order) overwritten and returned."""
    x = np.asfortranarray(np.array(x, dtype=store.dtype))
    if x.ndim == 1:
        x = np.asfortranarray(x[:, None])
    fn = lib().slu_oracle_zsolve if store.z else lib().slu_oracle_dsolve
    fn(*store._args(), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.shape[0]), ctypes.c_int(x.shape[1]))
    return x


def dsolve_level(store, x, nodes, direction):
    """In-place forward (+1) / backward (-1) solve restricted to the ascending supernode list `nodes`."""
    assert x.flags.f_contiguous and x.dtype == store.dtype
    nodes = np.ascontiguousarray(nodes, dtype=np.int32)
    L = lib()
    fn = ((L.slu_oracle_zsolve_fwd if direction > 0 else L.slu_oracle_zsolve_bwd) if store.z else
          (L.slu_oracle_dsolve_fwd if direction > 0 else L.slu_oracle_dsolve_bwd))
    fn(*store._args(), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.shape[0]), ctypes.c_int(x.shape[1]),
       _p(nodes, ctypes.c_int), ctypes.c_int(len(nodes)))
    return x


def dgsrfs(store, rowptr, colind, nzval, perm_c, B, X):
    """Iterative refinement (pdgsrfs3d restatement) of X (n x nrhs, Fortran order, updated in place and returned) for the
    ORIGINAL CSR matrix A, with `store` = factors of Pc A Pc^T.  Returns (X, berr[nrhs], steps of the last rhs)."""
    B = np.asfortranarray(np.array(B, dtype=store.dtype)); X = np.asfortranarray(np.array(X, dtype=store.dtype))
    if B.ndim == 1:
        B = np.asfortranarray(B[:, None]); X = np.asfortranarray(X[:, None])
    nrhs = B.shape[1]
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32); colind = np.ascontiguousarray(colind, dtype=np.int32)
    nzval = np.ascontiguousarray(nzval, dtype=store.dtype); perm_c = np.ascontiguousarray(perm_c, dtype=np.int32)
    berr = np.zeros(nrhs)
    fn = lib().slu_oracle_zgsrfs if store.z else lib().slu_oracle_dgsrfs
    fn.restype = ctypes.c_int
    steps = fn(*store._args(), _p(rowptr, ctypes.c_int), _p(colind, ctypes.c_int), nzval.ctypes.data_as(ctypes.c_void_p),
               _p(perm_c, ctypes.c_int), B.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(B.shape[0]),
               X.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(X.shape[0]), ctypes.c_int(nrhs), _p(berr, ctypes.c_double))
    return X, berr, steps


def num_threads():
    return lib().slu_oracle_num_threads()
