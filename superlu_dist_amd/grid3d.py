"""Process-grid harness above the C ABI (tests / bench only): creates the communicator object the library's C driver
needs and wraps the collective entry points.  The whole 3D algorithm -- pdgstrf3d's Z-level loop and ancestor reduction
(SRC/double/pdgstrf3d.c:333-392, pd3dcomm.c:1046-1081), the XY panel exchange (dtrfCommWrapper.c:32-118, :377-548), the
distributed solves (pdgstrs3d.c) -- runs inside libsluamd.so; nothing here touches the numerics.

Transports:
    local_comms(Pr, Pc, Pz)      in-process world (one thread per rank; any number of ranks per GPU)
    TorchComm(dist, ...)         host-buffer callbacks over torch.distributed (gloo): the CPU tests and multi-process
                                 single-GPU tests; the reference-side binding does the same with MPI
    rccl_comm(dist, ...)         RCCL called directly by the library (ncclSend / ncclRecv on its HIP streams); torch is only
                                 the out-of-band channel that ships the ncclUniqueId, like MPI_Bcast in an MPI application
World rank of grid position (row, col, z) = (z * Pr + row) * Pc + col.
"""
import ctypes as C
import numpy as np

from . import _lib


def grid_coords(rank, Pr, Pc, Pz):
    z, rem = divmod(rank, Pr * Pc)
    r, c = divmod(rem, Pc)
    return r, c, z


def default_grid(world):
    """Pr x Pc x Pz for `world` ranks: Z first (forests need no communication), then a near-square layer; 8 -> 2 x 2 x 2."""
    shapes = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2), 16: (2, 2, 4)}
    if world in shapes:
        return shapes[world]
    raise ValueError("world size must be 1, 2, 4, 8 or 16 (or pass an explicit grid)")


def local_comms(Pr, Pc, Pz):
    L = _lib.load()
    P = Pr * Pc * Pz
    arr = (C.c_void_p * P)()
    _lib.check(L.sluamd_comm_create_local(arr, Pr, Pc, Pz), "sluamd_comm_create_local")
    return [C.c_void_p(arr[i]) for i in range(P)]


class TorchComm:
    """sluamd_comm_callbacks_t over torch.distributed point-to-point (gloo, CPU tensors viewing the library's buffers)."""

    def __init__(self, dist, Pr, Pc, Pz, rank=None):
        import torch
        self.dist, self.torch = dist, torch
        self.rank = dist.get_rank() if rank is None else rank
        self.pending = []
        r, c, z = grid_coords(self.rank, Pr, Pc, Pz)

        def view(ptr, nbytes):
            buf = (C.c_char * nbytes).from_address(ptr)
            return torch.frombuffer(buf, dtype=torch.uint8)

        def isend(ctx, buf, nbytes, peer):
            try:
                self.pending.append(dist.isend(view(buf, nbytes), peer))
                return 0
            except Exception:
                return 1

        def irecv(ctx, buf, nbytes, peer):
            try:
                self.pending.append(dist.irecv(view(buf, nbytes), peer))
                return 0
            except Exception:
                return 1

        def waitall(ctx):
            try:
                for w in self.pending:
                    w.wait()
                self.pending = []
                return 0
            except Exception:
                return 1

        def allmin(ctx, pv):
            try:
                t = torch.tensor([pv[0]], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                pv[0] = int(t.item())
                return 0
            except Exception:
                return 1

        self._cbs = (_lib.ISEND_T(isend), _lib.IRECV_T(irecv), _lib.WAITALL_T(waitall), _lib.ALLMIN_T(allmin))
        cb = _lib.CommCallbacks(None, *self._cbs)
        self.handle = C.c_void_p()
        _lib.check(_lib.load().sluamd_comm_create_callbacks(C.byref(self.handle), C.byref(cb), Pr, Pc, Pz, r, c, z),
                   "sluamd_comm_create_callbacks")

    def destroy(self):
        if self.handle:
            _lib.load().sluamd_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def rccl_comm(dist, Pr, Pc, Pz, device):
    """RCCL communicator created by the library itself (ncclCommInitRank); dist only broadcasts the 128-byte unique id."""
    import torch
    L = _lib.load()
    rank = dist.get_rank()
    idbuf = (C.c_char * 128)()
    obj = [None]
    if rank == 0:
        try:
            _lib.check(L.sluamd_comm_rccl_unique_id(idbuf), "sluamd_comm_rccl_unique_id")
            obj = [bytes(idbuf)]
        except Exception as e:     # noqa: BLE001 -- the other ranks wait in the broadcast: tell them, then everybody raises
            obj = ["ERROR: " + str(e)]
    dist.broadcast_object_list(obj, src=0)
    if not isinstance(obj[0], (bytes, bytearray)):
        raise RuntimeError("sluamd_comm_rccl_unique_id failed on rank 0: " + str(obj[0]))
    idbuf = (C.c_char * 128).from_buffer_copy(obj[0])
    r, c, z = grid_coords(rank, Pr, Pc, Pz)
    h = C.c_void_p()
    _lib.check(L.sluamd_comm_create_rccl(C.byref(h), idbuf, Pr, Pc, Pz, r, c, z, device), "sluamd_comm_create_rccl")
    return h


class GridHandle:
    """One rank's device-resident store on a process grid (sluamd_dCreateLUHandleGrid / ...FromSymbGrid)."""

    def __init__(self, h, comm, n, z=False):
        self._h, self.comm, self.n, self.z = h, comm, n, z

    def plan_table(self):
        from .driver import plan_table
        return plan_table(self._h)

    @classmethod
    def from_store(cls, store, forests, comm, **opts):
        from .driver import LUHandle, _forest_view
        L = _lib.load()
        o = LUHandle._opts(**opts)
        fv, keep = (None, None) if forests is None else _forest_view(forests)
        h = C.c_void_p()
        create = L.sluamd_zCreateLUHandleGrid if store.z else L.sluamd_dCreateLUHandleGrid   
        _lib.check(create(C.byref(h), C.byref(store.view), None if fv is None else C.byref(fv), C.byref(o), comm), "sluamd_[dz]CreateLUHandleGrid")
        obj = cls(h, comm, store.n, store.z)
        obj._keep = (keep, store)
        return obj

    @classmethod
    def from_symbolic(cls, symb, nzval, comm, sn_tree=None, **opts):
        from .driver import LUHandle, _pi, _pd
        L = _lib.load()
        o = LUHandle._opts(**opts)
        t = None if sn_tree is None else np.ascontiguousarray(sn_tree, dtype=np.int32)
        h = C.c_void_p()
        if np.iscomplexobj(nzval):
            nz = np.ascontiguousarray(nzval, dtype=np.complex128)
            _lib.check(L.sluamd_zCreateLUHandleFromSymbGrid(C.byref(h), symb._h, _pi(symb.rowptr), _pi(symb.colind), nz.ctypes.data_as(C.c_void_p),
                                                            _pi(symb.perm_c), C.byref(o), None if t is None else t.ctypes.data_as(_lib.P_int), comm),
                       "sluamd_zCreateLUHandleFromSymbGrid")
            return cls(h, comm, symb.n, True)
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        _lib.check(L.sluamd_dCreateLUHandleFromSymbGrid(C.byref(h), symb._h, _pi(symb.rowptr), _pi(symb.colind), _pd(nz), _pi(symb.perm_c),
                                                        C.byref(o), None if t is None else t.ctypes.data_as(_lib.P_int), comm),
                   "sluamd_dCreateLUHandleFromSymbGrid")
        return cls(h, comm, symb.n)

    def pdgstrf3d(self, thresh=0.0):
        info = C.c_int32(0)
        L = _lib.load()
        _lib.check((L.sluamd_pzgstrf3d if self.z else L.sluamd_pdgstrf3d)(self._h, float(thresh), C.byref(info)), "sluamd_p[dz]gstrf3d")
        return info.value

    def pdgstrs3d(self, xp):
        """xp: complete permuted right-hand side (n x nrhs), replicated; returns the complete solution."""
        x = np.asfortranarray(np.array(xp, dtype=np.complex128 if self.z else np.float64))
        if x.ndim == 1:
            x = np.asfortranarray(x[:, None])
        L = _lib.load()
        if self.z:
            _lib.check(L.sluamd_pzgstrs3d(self._h, x.ctypes.data_as(C.c_void_p), x.shape[0], x.shape[1]), "sluamd_pzgstrs3d")
        else:
            _lib.check(L.sluamd_pdgstrs3d(self._h, x.ctypes.data_as(_lib.P_dbl), x.shape[0], x.shape[1]), "sluamd_pdgstrs3d")
        return x

    def pdgstrs3d_dist(self, B_loc, fst_row, perm=None, perm_out="same"):
        """Distributed boundary (sluamd_pdgstrs3d_dist): B_loc = this rank's rows [fst_row, fst_row + m_loc) of the ORIGINAL right-hand
        side (m_loc = 0 on layers z > 0), perm[i] = row of the factored system of original row i; returns rows perm_out[i] of the solved
        vector (default: the same permutation = x in the original order; None = the reference's convention, rows of the permuted solution)."""
        B = np.asfortranarray(np.array(B_loc, dtype=np.complex128 if self.z else np.float64))
        if B.ndim == 1:
            B = np.asfortranarray(B[:, None])
        m_loc, nrhs = B.shape
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        po = pm if isinstance(perm_out, str) else (None if perm_out is None else np.ascontiguousarray(perm_out, dtype=np.int32))
        fn = _lib.load().sluamd_pzgstrs3d_dist if self.z else _lib.load().sluamd_pdgstrs3d_dist
        _lib.check(fn(self._h, B.ctypes.data_as(C.c_void_p), max(m_loc, 1), nrhs, m_loc, int(fst_row),
                      None if pm is None else pm.ctypes.data_as(_lib.P_int),
                      None if po is None else po.ctypes.data_as(_lib.P_int)), "sluamd_p[dz]gstrs3d_dist")
        return B

    def copy_to_host(self, store):
        L = _lib.load()
        _lib.check((L.sluamd_zCopyLU2Host if self.z else L.sluamd_dCopyLU2Host)(self._h, C.byref(store.view)), "sluamd_[dz]CopyLU2Host")
        return store

    def reset_values(self):
        _lib.check(_lib.load().sluamd_dResetValues(self._h), "sluamd_dResetValues")

    def set_profile(self, on=True):
        """per-phase HIP-event timing (serial schedule): Schur / panel kernels, XY exchange phases, Z ancestor reduction"""
        _lib.load().sluamd_set_profile(self._h, int(on))

    def stats(self):
        s = _lib.Stats()
        _lib.load().sluamd_get_stats(self._h, C.byref(s))
        return {f[0]: getattr(s, f[0]) for f in _lib.Stats._fields_}

    def setup_times(self):
        """{phase: seconds} of this rank's handle creation (sluamd_setup_times)"""
        from .driver import LUHandle
        return LUHandle.setup_times(self)

    def destroy(self):
        if self._h:
            _lib.load().sluamd_dDestroyLUHandle(self._h)
            self._h = C.c_void_p()


def run_ranks(P, fn):
    """Run fn(rank) for rank in range(P) on P threads (ctypes calls release the GIL); re-raises the first failure."""
    import threading
    out, err = [None] * P, [None] * P

    def work(i):
        try:
            out[i] = fn(i)
        except BaseException as e:   # noqa: BLE001 -- reported to the caller below
            err[i] = e
    th = [threading.Thread(target=work, args=(i,)) for i in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out
