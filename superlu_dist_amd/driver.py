"""Host-side mirror of the reference's expert-driver interface for the hot path.

Names follow the reference (SRC/double/pdgssvx3d.c:519 `pdgssvx3d`, SRC/double/pdgstrf3d.c:121 `pdgstrf3d`,
SRC/double/pdgstrs3d.c:6604 `pdgstrs3d`): Python here is only the test/bench harness above the C ABI
(include/superlu_dist_amd.h) -- all numerics run in libsluamd.so on the GPU; nothing here computes on the CPU
except permutation bookkeeping and the residual check.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import LUView, ForestView, Options, Stats, P_int, P_dbl


def _pi(a):
    return a.ctypes.data_as(P_int)


def _pd(a):
    return a.ctypes.data_as(P_dbl)


class FlatStore:
    """One rank's L/U store as flat arrays + offsets over its LOCAL block columns / rows (what tests/golden holds),
    exposed to the C ABI through the reference's pointer-array view (Lrowind_bc_ptr[lk], Lnzval_bc_ptr[lk], ...)."""

    def __init__(self, n, xsup, Lrowind_off, Lrowind, Lnzval_off, Lnzval, Ufstnz_off, Ufstnz, Unzval_off, Unzval,
                 grid=(1, 1, 1), coords=(0, 0, 0)):
        self.n = int(n)
        self.xsup = np.ascontiguousarray(xsup, dtype=np.int32)
        self.nsupers = len(self.xsup) - 1
        self.Lrowind_off = np.asarray(Lrowind_off, dtype=np.int64)
        self.Lrowind = np.ascontiguousarray(Lrowind, dtype=np.int32)
        self.Lnzval_off = np.asarray(Lnzval_off, dtype=np.int64)
        self.z = bool(np.iscomplexobj(Lnzval) or np.iscomplexobj(Unzval))   # complex16 store (pzgstrf3d path)
        vt = np.complex128 if self.z else np.float64
        self.Lnzval = np.array(Lnzval, dtype=vt)
        self.Ufstnz_off = np.asarray(Ufstnz_off, dtype=np.int64)
        self.Ufstnz = np.ascontiguousarray(Ufstnz, dtype=np.int32)
        self.Unzval_off = np.asarray(Unzval_off, dtype=np.int64)
        self.Unzval = np.array(Unzval, dtype=vt)
        self.grid, self.coords = grid, coords
        self._build_view()

    @classmethod
    def from_golden(cls, g, rank=0, which="pre"):
        r = f"r{rank}__"
        grid = (int(g[r + "Pr"][0]), int(g[r + "Pc"][0]), int(g[r + "Pz"][0]))
        coords = (int(g[r + "myrow"][0]), int(g[r + "mycol"][0]), int(g[r + "myz"][0]))
        return cls(int(g[r + "n"][0]), g[r + "xsup"], g[r + "Lrowind_off"], g[r + "Lrowind"], g[r + "Lnzval_off"],
                   g[r + f"Lnzval_{which}"], g[r + "Ufstnz_off"], g[r + "Ufstnz"], g[r + "Unzval_off"],
                   g[r + f"Unzval_{which}"], grid, coords)

    def _build_view(self):
        ns = self.nsupers

        def ptrs(base, off, ctype, elem):
            nloc = len(off) - 1          # ceil(nsupers / npcol) block columns or ceil(nsupers / nprow) block rows
            arr = (C.POINTER(ctype) * max(nloc, 1))()
            addr = base.ctypes.data
            for k in range(nloc):
                if off[k + 1] > off[k]:
                    arr[k] = C.cast(addr + int(off[k]) * elem, C.POINTER(ctype))
            return arr
        self._lp = ptrs(self.Lrowind, self.Lrowind_off, C.c_int32, 4)
        self._lv = ptrs(self.Lnzval, self.Lnzval_off, C.c_double, self.Lnzval.itemsize)
        self._up = ptrs(self.Ufstnz, self.Ufstnz_off, C.c_int32, 4)
        self._uv = ptrs(self.Unzval, self.Unzval_off, C.c_double, self.Unzval.itemsize)
        v = LUView()
        v.n, v.nsupers, v.xsup = self.n, ns, _pi(self.xsup)
        v.nprow, v.npcol, v.npdep = self.grid
        v.myrow, v.mycol, v.myzlayer = self.coords
        v.Lrowind_bc_ptr = C.cast(self._lp, C.POINTER(P_int))
        v.Lnzval_bc_ptr = C.cast(self._lv, C.POINTER(P_dbl))
        v.Ufstnz_br_ptr = C.cast(self._up, C.POINTER(P_int))
        v.Unzval_br_ptr = C.cast(self._uv, C.POINTER(P_dbl))
        self.view = v


def order_nd(n, rowptr, colind, leaf=64):
    """perm_c[old] = new: nested dissection of the pattern of A + A^T without geometry (sluamd_order_nd)"""
    rp = np.ascontiguousarray(rowptr, dtype=np.int32); ci = np.ascontiguousarray(colind, dtype=np.int32)
    perm = np.zeros(n, dtype=np.int32)
    _lib.check(_lib.load().sluamd_order_nd(int(n), _pi(rp), _pi(ci), int(leaf), _pi(perm)), "sluamd_order_nd")
    return perm


class Symbolic:
    """sluamd_dsymbfact result (our symbfact_dist + pddistribute3d stand-in for a 1x1 layer)."""

    def __init__(self, n, rowptr, colind, perm_c=None, relax=32, maxsup=256, unsym=False):
        """unsym=True: sluamd_dsymbfact_unsym -- the exact unsymmetric structure with the reference's supernode rules (symbfact.c)"""
        L = _lib.load()
        self.n = int(n)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.colind = np.ascontiguousarray(colind, dtype=np.int32)
        self.perm_c = np.empty(self.n, dtype=np.int32)
        pin = None if perm_c is None else np.ascontiguousarray(perm_c, dtype=np.int32)
        self._h = C.c_void_p()
        fn = L.sluamd_dsymbfact_unsym if unsym else L.sluamd_dsymbfact
        _lib.check(fn(C.byref(self._h), self.n, _pi(self.rowptr), _pi(self.colind),
                      None if pin is None else _pi(pin), relax, maxsup, _pi(self.perm_c)), "sluamd_dsymbfact")
        ns = C.c_int32(); nl = C.c_int64(); nu = C.c_int64(); li = C.c_int64(); ui = C.c_int64(); fl = C.c_double()
        L.sluamd_symb_info(self._h, C.byref(ns), C.byref(nl), C.byref(nu), C.byref(li), C.byref(ui), C.byref(fl))
        self.nsupers, self.nnzL, self.nnzU, self.flops = ns.value, nl.value, nu.value, fl.value
        self.lidx_len, self.uidx_len = li.value, ui.value

    def distribute_host(self, nzval):
        L = _lib.load()
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        _lib.check(L.sluamd_ddistribute_host(self._h, _pi(self.rowptr), _pi(self.colind), _pd(nz), _pi(self.perm_c)),
                   "sluamd_ddistribute_host")

    def xsup(self):
        xs = np.empty(self.nsupers + 1, dtype=np.int32)
        _lib.check(_lib.load().sluamd_symb_export(self._h, _pi(xs), None, None, None, None, None, None, None, None), "sluamd_symb_export")
        return xs

    def grid_footprint(self, Pr, Pc, Pz, sn_tree=None):
        """stored factor values / replicated values / index entries per world rank of a Pr x Pc x Pz grid (sluamd_symb_grid_footprint)"""
        P = Pr * Pc * Pz
        vals, rep, idx = (np.zeros(P, dtype=np.int64) for _ in range(3))
        t = None if sn_tree is None else np.ascontiguousarray(sn_tree, dtype=np.int32)
        p64 = C.POINTER(C.c_int64)
        _lib.check(_lib.load().sluamd_symb_grid_footprint(self._h, Pr, Pc, Pz, None if t is None else _pi(t), vals.ctypes.data_as(p64),
                                                          rep.ctypes.data_as(p64), idx.ctypes.data_as(p64)), "sluamd_symb_grid_footprint")
        return vals, rep, idx

    def partition(self, npdep):
        """Tree id (heap order) of every supernode for a 1 x 1 x npdep grid."""
        t = np.zeros(self.nsupers, dtype=np.int32)
        _lib.check(_lib.load().sluamd_symb_partition(self._h, npdep, _pi(t)), "sluamd_symb_partition")
        return t

    def flat_store(self, values=True):
        """Copy the host store out as a FlatStore (tests / CPU-baseline harness)."""
        L = _lib.load()
        ns = self.nsupers
        xsup = np.empty(ns + 1, dtype=np.int32)
        lo = np.empty(ns + 1, dtype=np.int64); lvo = np.empty(ns + 1, dtype=np.int64)
        uo = np.empty(ns + 1, dtype=np.int64); uvo = np.empty(ns + 1, dtype=np.int64)
        li = np.empty(self.lidx_len, dtype=np.int32); ui = np.empty(self.uidx_len, dtype=np.int32)
        lv = np.zeros(self.nnzL if values else 0); uv = np.zeros(self.nnzU if values else 0)
        P64 = C.POINTER(C.c_int64)
        p64 = lambda a: a.ctypes.data_as(P64)
        _lib.check(L.sluamd_symb_export(self._h, _pi(xsup), p64(lo), _pi(li), p64(lvo), _pd(lv) if values else None,
                                        p64(uo), _pi(ui), p64(uvo), _pd(uv) if values else None), "sluamd_symb_export")
        if not values:
            lv = np.zeros(self.nnzL); uv = np.zeros(self.nnzU)
        return FlatStore(self.n, xsup, lo, li, lvo, lv, uo, ui, uvo, uv)

    def free(self):
        if self._h:
            _lib.load().sluamd_symb_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


PLAN_COLS = 20


def plan_table(h):
    L = _lib.load()
    rows = C.c_int64(0)
    _lib.check(L.sluamd_plan_table(h, None, 0, C.byref(rows)), "sluamd_plan_table")
    out = np.zeros((max(rows.value, 1), PLAN_COLS))
    _lib.check(L.sluamd_plan_table(h, out.ctypes.data_as(C.POINTER(C.c_double)), rows.value, C.byref(rows)), "sluamd_plan_table")
    return out[:rows.value]


class LUHandle:
    """Device-resident L/U (sluamd_handle_t): dCreateLUgpuHandle / pdgstrf3d_LUv1 / dCopyLUGPU2Host /
    dDestroyLUgpuHandle replacement (SRC/CplusplusFactor/LUgpuCHandle_interface_impl.cu:11-73)."""

    def __init__(self, h, store=None):
        self._h = h
        self.store = store
        self.z = False

    @staticmethod
    def _opts(replace_tiny=False, deterministic=False, device=-1, info_rule=0):
        o = Options()
        _lib.load().sluamd_default_options(C.byref(o))
        o.device = device; o.replace_tiny_pivot = int(replace_tiny); o.deterministic = int(deterministic)
        o.info_rule = int(info_rule)        # 1 = SLUAMD_INFO_REFERENCE: the zero-pivot rule of the reference's code (what the binding selects)
        return o

    @classmethod
    def from_store(cls, store, forests=None, **kw):
        L = _lib.load()
        h = C.c_void_p()
        o = cls._opts(**kw)
        fv = None
        keep = None
        if forests is not None:
            fv, keep = _forest_view(forests)
        create = L.sluamd_zCreateLUHandle if store.z else L.sluamd_dCreateLUHandle
        _lib.check(create(C.byref(h), C.byref(store.view), None if fv is None else C.byref(fv), C.byref(o)),
                   "sluamd_zCreateLUHandle" if store.z else "sluamd_dCreateLUHandle")
        obj = cls(h, store)
        obj.z = store.z
        obj._keep = keep
        return obj

    @classmethod
    def from_symbolic(cls, symb, nzval, **kw):
        L = _lib.load()
        h = C.c_void_p()
        o = cls._opts(**kw)
        if np.iscomplexobj(nzval):
            nz = np.ascontiguousarray(nzval, dtype=np.complex128)
            _lib.check(L.sluamd_zCreateLUHandleFromSymb(C.byref(h), symb._h, _pi(symb.rowptr), _pi(symb.colind),
                                                        nz.ctypes.data_as(C.c_void_p), _pi(symb.perm_c), C.byref(o)),
                       "sluamd_zCreateLUHandleFromSymb")
            obj = cls(h, None); obj.z = True
            return obj
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        _lib.check(L.sluamd_dCreateLUHandleFromSymb(C.byref(h), symb._h, _pi(symb.rowptr), _pi(symb.colind), _pd(nz),
                                                    _pi(symb.perm_c), C.byref(o)), "sluamd_dCreateLUHandleFromSymb")
        return cls(h, None)

    def set_values(self, store):
        L = _lib.load()
        _lib.check((L.sluamd_zSetValues if self.z else L.sluamd_dSetValues)(self._h, C.byref(store.view)), "sluamd_[dz]SetValues")

    def pdgstrf3d(self, thresh=0.0):
        """pdgstrf3d, or pzgstrf3d on a complex16 handle."""
        L = _lib.load()
        info = C.c_int32(0)
        _lib.check((L.sluamd_pzgstrf3d if self.z else L.sluamd_pdgstrf3d)(self._h, float(thresh), C.byref(info)), "sluamd_p[dz]gstrf3d")
        return info.value

    pzgstrf3d = pdgstrf3d

    def copy_to_host(self, store=None):
        store = store or self.store
        L = _lib.load()
        _lib.check((L.sluamd_zCopyLU2Host if self.z else L.sluamd_dCopyLU2Host)(self._h, C.byref(store.view)), "sluamd_[dz]CopyLU2Host")
        return store

    def pdgstrs3d(self, x):
        """pdgstrs3d, or pzgstrs3d on a complex16 handle."""
        x = np.asfortranarray(np.array(x, dtype=np.complex128 if self.z else np.float64))
        if x.ndim == 1:
            x = np.asfortranarray(x[:, None])
        L = _lib.load()
        if self.z:
            _lib.check(L.sluamd_pzgstrs3d(self._h, x.ctypes.data_as(C.c_void_p), x.shape[0], x.shape[1]), "sluamd_pzgstrs3d")
        else:
            _lib.check(L.sluamd_pdgstrs3d(self._h, _pd(x), x.shape[0], x.shape[1]), "sluamd_pdgstrs3d")
        return x

    pzgstrs3d = pdgstrs3d

    def pdgstrs3d_dist(self, B, perm=None, fst_row=0):
        """sluamd_pdgstrs3d_dist on a single-rank handle: B in the ORIGINAL row order (all n rows), perm[i] = row of the factored system."""
        B = np.asfortranarray(np.array(B, dtype=np.float64))
        if B.ndim == 1:
            B = np.asfortranarray(B[:, None])
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        _lib.check(_lib.load().sluamd_pdgstrs3d_dist(self._h, B.ctypes.data_as(C.c_void_p), max(B.shape[0], 1), B.shape[1], B.shape[0], int(fst_row),
                                                     None if pm is None else _pi(pm), None if pm is None else _pi(pm)), "sluamd_pdgstrs3d_dist")
        return B

    def pdgstrs3d_dev(self, ptr, ldx, nrhs):
        _lib.check(_lib.load().sluamd_pdgstrs3d_dev(self._h, C.c_void_p(ptr), ldx, nrhs), "sluamd_pdgstrs3d_dev")

    def reset_values(self):
        _lib.check(_lib.load().sluamd_dResetValues(self._h), "sluamd_dResetValues")

    def attach_matrix(self, n, rowptr, colind, nzval, perm_c):
        """Device copy of the ORIGINAL matrix (CSR) + perm_c for pdgsrfs3d."""
        rp = np.ascontiguousarray(rowptr, dtype=np.int32); ci = np.ascontiguousarray(colind, dtype=np.int32)
        v = np.ascontiguousarray(nzval, dtype=np.float64); pc = np.ascontiguousarray(perm_c, dtype=np.int32)
        _lib.check(_lib.load().sluamd_dAttachMatrix(self._h, int(n), _pi(rp), _pi(ci), _pd(v), _pi(pc)), "sluamd_dAttachMatrix")

    def pdgsrfs3d(self, b, x):
        """Iterative refinement of x (original ordering) for the attached matrix; returns (x, berr[nrhs], steps)."""
        b = np.asfortranarray(np.array(b, dtype=np.float64)); x = np.asfortranarray(np.array(x, dtype=np.float64))
        if b.ndim == 1:
            b = np.asfortranarray(b[:, None]); x = np.asfortranarray(x[:, None])
        berr = np.zeros(b.shape[1]); steps = C.c_int32(0)
        _lib.check(_lib.load().sluamd_pdgsrfs3d(self._h, _pd(b), b.shape[0], _pd(x), x.shape[0], b.shape[1], _pd(berr),
                                                C.byref(steps)), "sluamd_pdgsrfs3d")
        return x, berr, steps.value

    def set_profile(self, on=True):
        _lib.load().sluamd_set_profile(self._h, int(on))

    def stats(self):
        s = Stats()
        _lib.load().sluamd_get_stats(self._h, C.byref(s))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def plan_table(self):
        """this rank's plan, one row per (Z level, DAG level): columns as documented at sluamd_plan_table (include/superlu_dist_amd.h)"""
        return plan_table(self._h)

    def diag_inv(self, k, ns):
        """(Linv, Uinv) of the diagonal block of supernode k: ns x ns, column-major (sluamd_dGetDiagInv)"""
        li = np.zeros((ns, ns), order="F"); ui = np.zeros((ns, ns), order="F")
        _lib.check(_lib.load().sluamd_dGetDiagInv(self._h, int(k), _pd(li), _pd(ui)), "sluamd_dGetDiagInv")
        return li, ui

    def setup_times(self):
        """{phase: seconds} of this handle's creation (sluamd_setup_times)"""
        buf = C.create_string_buffer(4096)
        if _lib.load().sluamd_setup_times(self._h, buf, 4096):
            return {}
        out = {}
        for tok in buf.value.decode().split(";"):
            if "=" in tok:
                k, v = tok.split("=")
                out[k] = out.get(k, 0.0) + float(v)
        return out

    def destroy(self):
        if self._h:
            _lib.load().sluamd_dDestroyLUHandle(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _forest_view(forests):
    """forests = dict(maxLvl, myTreeIdxs, myZeroTrIdxs, nodeLists=[array or None per forest])"""
    fv = ForestView()
    mt = np.ascontiguousarray(forests["myTreeIdxs"], dtype=np.int32)
    mz = np.ascontiguousarray(forests["myZeroTrIdxs"], dtype=np.int32)
    lists = [np.ascontiguousarray(a if a is not None else [], dtype=np.int32) for a in forests["nodeLists"]]
    nn = np.array([len(a) for a in lists], dtype=np.int32)
    arr = (P_int * len(lists))()
    for i, a in enumerate(lists):
        arr[i] = _pi(a) if len(a) else None
    fv.maxLvl = int(forests["maxLvl"]); fv.myTreeIdxs = _pi(mt); fv.myZeroTrIdxs = _pi(mz)
    fv.numForests = len(lists); fv.nNodes = _pi(nn); fv.nodeList = C.cast(arr, C.POINTER(P_int))
    return fv, (mt, mz, lists, nn, arr)


def pivot_thresh(n, rowptr, colind, nzval):
    """thresh = smach_dist("Epsilon") * anorm of pdgstrf3d (pdgstrf3d.c:132-133): single-precision epsilon in LAPACK's
    slamch('E') sense (FLT_EPSILON * 0.5 = 2^-24, smach_dist.c:64) times the 1-norm of A (max column sum,
    dcomputeA_Norm(notran), pdgssvx3d.c)."""
    if len(nzval) == 0:
        return 0.0
    colsum = np.bincount(np.asarray(colind), weights=np.abs(nzval), minlength=int(n))
    return 0.5 * float(np.finfo(np.float32).eps) * float(colsum.max())


def pdgssvx3d(n, rowptr, colind, nzval, b, perm_c=None, relax=32, maxsup=256, replace_tiny=False, anorm=None,
              keep=False, refine=False):
    """Solve A x = b through the GPU hot path: symbolic (host) -> device-resident distribute -> pdgstrf3d ->
    pdgstrs3d, with Equil = NO, RowPerm = NOROWPERM, ColPerm = MY_PERMC/NATURAL, IterRefine = NOREFINE
    (the timing configuration of BASELINE.md section 4); refine=True adds IterRefine = SLU_DOUBLE (pdgsrfs3d on the device,
    double precision only) and puts `berr` / `refine_steps` into the stats.  Returns (x, info, stats[, handle, symb])."""
    symb = Symbolic(n, rowptr, colind, perm_c, relax, maxsup)
    h = LUHandle.from_symbolic(symb, nzval, replace_tiny=replace_tiny)
    thresh = pivot_thresh(n, rowptr, colind, nzval) if anorm is None else 0.5 * float(np.finfo(np.float32).eps) * anorm
    info = h.pdgstrf3d(thresh)
    b = np.asfortranarray(np.array(b, dtype=np.complex128 if np.iscomplexobj(nzval) else np.float64))
    if b.ndim == 1:
        b = np.asfortranarray(b[:, None])
    xp = np.zeros_like(b, order="F")
    xp[symb.perm_c, :] = b                                     # Pc*b
    y = h.pdgstrs3d(xp)
    x = np.asfortranarray(y[symb.perm_c, :])                   # Pc^T y
    st = h.stats()
    if refine:
        h.attach_matrix(n, rowptr, colind, nzval, symb.perm_c)
        x, berr, steps = h.pdgsrfs3d(b, x)
        st["berr"] = berr; st["refine_steps"] = steps
    if keep:
        return x, info, st, h, symb
    h.destroy(); symb.free()
    return x, info, st


pzgssvx3d = pdgssvx3d   # complex16 input (nzval complex) takes the pzgstrf3d / pzgstrs3d path of the same driver
