"""Z-sharded (1 x 1 x Pz) orchestration of the hot path: one rank = one GPU = one Z layer.

Mirrors the Z dimension of the reference's 3D algorithm:
  * pdgstrf3d's level loop (SRC/double/pdgstrf3d.c:333-385): factor my forest of level ilvl, then the pairwise ancestor
    reduction dreduceAllAncestors3d (SRC/double/pd3dcomm.c:1046-1081: sender = myGrid + 2^ilvl, receiver = myGrid when
    myGrid % 2^(ilvl+1) == 0) -- here ONE send/recv per contiguous value-arena slice of the shared ancestor forests
    followed by an add, over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests);
  * pdgstrs3d's forward / backward sweeps over the Z levels (pdgsTrForwardSolve3d :7312, pdgsTrBackSolve3d :7564) with
    dfsolveReduceLsum3d (:1646) / dp2pSolvedX3d (:1596) style exchanges of the ancestor parts of x.

The numeric work is done by a *backend* object (duck-typed):
    factor_level(ilvl, thresh) ; value_slices(alvl_from[, alvl_to]) -> [1-D tensors viewing the resident factors] ;
    solve_level(ilvl, direction, x) ; info() -> (info, tiny) ; n ; tree_rows(ilvl) -> [(row0, row1), ...]
`GpuLayer` below is the product backend (libsluamd.so).  tests/ supplies a CPU-oracle backend for the gloo tests.
"""
import ctypes as C
import numpy as np

from . import _lib


def max_level(npdep):
    lvl = 1
    while (1 << (lvl - 1)) < npdep:
        lvl += 1
    if (1 << (lvl - 1)) != npdep:
        raise ValueError("npdep must be a power of two (reference: superlu_gridinit3d)")
    return lvl


def path_trees(npdep, z):
    """Tree ids (heap order, root 0) handled by layer z at levels 0..maxLvl-1 (getGridTrees, supernodal_etree.c:840-851)."""
    t = [npdep - 1 + z]
    for _ in range(1, max_level(npdep)):
        t.append((t[-1] - 1) // 2)
    return t


def runs(sorted_nodes):
    """Contiguous runs [(k0, k1_exclusive), ...] of an ascending node list."""
    out = []
    for k in sorted_nodes:
        k = int(k)
        if out and out[-1][1] == k:
            out[-1][1] = k + 1
        else:
            out.append([k, k + 1])
    return [(a, b) for a, b in out]


class DistComm:
    """Point-to-point + all-reduce over torch.distributed (nccl on GPUs, gloo on CPU).

    npdep > 1 also creates the sub-communicators of the cooperative factorisation: for every Z level ilvl >= 1 the
    groups of 2^ilvl consecutive layers that share one ancestor forest (dist.new_group is collective: every rank
    creates every group, in the same order).  host_staging=True bounces device tensors through host memory (gloo
    without GPU support: the single-GPU multi-process tests)."""

    def __init__(self, dist, ranks=None, npdep=1, host_staging=False):
        self.dist = dist
        self.ranks = ranks            # layer z -> global rank (identity by default)
        self.host_staging = host_staging
        self.groups = {}
        if npdep > 1:
            for ilvl in range(1, max_level(npdep)):
                G = 1 << ilvl
                for z0 in range(0, npdep, G):
                    if G == npdep and ranks is None:
                        self.groups[(ilvl, z0)] = None                       # the world group
                    else:
                        self.groups[(ilvl, z0)] = dist.new_group([self._r(z) for z in range(z0, z0 + G)])

    def group(self, ilvl, z0):
        return ("g", self.groups[(ilvl, z0)])

    def _r(self, z):
        return z if self.ranks is None else self.ranks[z]

    def send(self, t, dst):
        t = t.contiguous()
        self.dist.send(t.cpu() if self.host_staging else t, self._r(dst))

    def recv(self, t, src):
        if self.host_staging:
            tmp = t.cpu()
            self.dist.recv(tmp, self._r(src))
            t.copy_(tmp)
        else:
            self.dist.recv(t, self._r(src))

    def allreduce_sum(self, t, group=None):
        pg = group[1] if group is not None else None
        if self.host_staging:
            tmp = t.cpu()
            self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM, group=pg)
            t.copy_(tmp)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=pg)

    def reduce_sum(self, t, dst, group=None):
        pg = group[1] if group is not None else None
        if self.host_staging:
            tmp = t.cpu()
            self.dist.reduce(tmp, self._r(dst), op=self.dist.ReduceOp.SUM, group=pg)
            t.copy_(tmp)
        else:
            self.dist.reduce(t, self._r(dst), op=self.dist.ReduceOp.SUM, group=pg)

    def broadcast(self, t, src, group=None):
        pg = group[1] if group is not None else None
        if self.host_staging:
            tmp = t.cpu()
            self.dist.broadcast(tmp, self._r(src), group=pg)
            t.copy_(tmp)
        else:
            self.dist.broadcast(t, self._r(src), group=pg)

    def allreduce_min_int(self, v, device):
        import torch
        t = torch.tensor([v], dtype=torch.int64, device="cpu" if self.host_staging else device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())


def _sync(backend):
    """torch ops (adds, copies, NCCL hand-offs) run on torch's current stream, the library on its own HIP streams:
    drain torch's stream before the next library call touches the same memory."""
    f = getattr(backend, "sync", None)
    if f is not None:
        f()


def pdgstrf3d(backend, comm, z, npdep, thresh):
    """Numeric factorisation over the Z levels; returns info (min over layers, 0 = none)."""
    import torch
    maxlvl = max_level(npdep)
    for ilvl in range(maxlvl):
        step = 1 << ilvl
        if z % step:
            break                                     # this layer is done (myZeroTrIdxs, supernodal_etree.c:853-869)
        backend.factor_level(ilvl, thresh)
        if ilvl < maxlvl - 1:
            slices = backend.value_slices(ilvl + 1)   # factors of ALL ancestor forests above this level
            if z % (2 * step) == 0:
                for sl in slices:
                    tmp = torch.empty_like(sl)
                    comm.recv(tmp, z + step)
                    sl += tmp                         # dzRecvLPanel/dzRecvUPanel: daxpy into the resident panel
            else:
                for sl in slices:
                    comm.send(sl, z - step)
            _sync(backend)
    info, _ = backend.info()
    big = backend.n + 1
    g = comm.allreduce_min_int(info if info else big, backend.device)
    return 0 if g == big else g


def pdgstrf3d_coop(backend, comm, z, npdep, thresh):
    """Numeric factorisation over the Z levels with COOPERATIVE ancestor forests: the 2^ilvl layers that share the
    forest of level ilvl factor it together instead of leaving all but one idle (what pdgstrf3d.c:333-385 does on a
    1 x 1 x Pz grid).  Storage of the forest is replicated in the group; block column jb is kept current by group
    member jb % G.  Per Z level:
      1. sum all-reduce over the group of THIS level's forest only: every member's copy holds exactly the partial Schur
         updates that member applied so far (A's entries sit on the group's first layer), so the sum over the group is
         the assembled forest -- what dreduceAllAncestors3d achieves pairwise -- and the copies of the HIGHER forests
         are not exchanged at all until their own level, where the (larger) group contains every contributor;
      2. per DAG level of the forest: owners factor diagonal blocks + L panels -> panel exchange (a broadcast from the
         owner when the level holds one supernode, else one sum all-reduce of the packed panels; the reference's
         dIBcastRecvLPanel) -> everyone: U-panel TRSM + Schur update of the destinations it owns;
      3. the U blocks are completed on the group's first layer: mask the non-owned ones, sum-reduce to that layer.
    Returns info (min over layers, 0 = none).  The solve (pdgstrs3d below) is unchanged: the first layer of each group
    holds the complete factors of its forest."""
    maxlvl = max_level(npdep)
    backend.factor_level(0, thresh)
    for ilvl in range(1, maxlvl):
        G = 1 << ilvl
        z0 = z - z % G
        g = z - z0
        grp = comm.group(ilvl, z0)
        for sl in backend.value_slices(ilvl, ilvl + 1):
            comm.allreduce_sum(sl, grp)
        nlev, max_stage = backend.coop_info(ilvl)
        stage = backend.stage_buffer(max_stage)
        for l in range(nlev):
            nodes = backend.coop_level_nodes(ilvl, l)
            if len(nodes) == 1:      # the usual case near the top of a separator: broadcast straight out of the arena
                backend.coop_panel(ilvl, l, G, g, thresh, None)
                for t in backend.panel_tensors(int(nodes[0])):
                    comm.broadcast(t, z0 + int(nodes[0]) % G, grp)
                backend.coop_update(ilvl, l, G, g, None)
            else:                    # many (small) panels with different owners: one packed sum all-reduce
                sz = backend.coop_panel(ilvl, l, G, g, thresh, stage)
                comm.allreduce_sum(stage[:sz], grp)
                backend.coop_update(ilvl, l, G, g, stage)
        backend.coop_mask_u(ilvl, G, g)
        for sl in backend.u_slices(ilvl):
            comm.reduce_sum(sl, z0, grp)              # only the group's first layer solves with this forest
    _sync(backend)
    info, _ = backend.info()
    big = backend.n + 1
    gmin = comm.allreduce_min_int(info if info else big, backend.device)
    return 0 if gmin == big else gmin


def init_rhs(backend, z, npdep, xp):
    """xp: (nrhs, n) tensor holding Pc*b everywhere; keep only the rows this layer owns (leaf forest + the ancestor
    forests it factors); the other layers' contributions arrive through the forward reduction."""
    import torch
    keep = torch.zeros_like(xp)
    for ilvl in range(max_level(npdep)):
        if z % (1 << ilvl) == 0:
            for a, b in backend.tree_rows(ilvl):
                keep[:, a:b] = xp[:, a:b]
    _sync(backend)
    return keep


def pdgstrs3d(backend, comm, z, npdep, x):
    """In-place solve; x is a contiguous (nrhs, n) tensor (== column-major n x nrhs) prepared by init_rhs.
    On return every layer holds the full solution."""
    import torch
    maxlvl = max_level(npdep)
    # forward sweep, leaves to root
    for ilvl in range(maxlvl):
        step = 1 << ilvl
        if z % step:
            break
        backend.solve_level(ilvl, +1, x)
        if ilvl < maxlvl - 1:
            rows = [r for al in range(ilvl + 1, maxlvl) for r in backend.tree_rows(al)]
            if z % (2 * step) == 0:
                for a, b in rows:
                    tmp = torch.empty((x.shape[0], b - a), dtype=x.dtype, device=x.device)
                    comm.recv(tmp, z + step)
                    x[:, a:b] += tmp
            else:
                for a, b in rows:
                    comm.send(x[:, a:b], z - step)
            _sync(backend)
    # backward sweep, root to leaves
    for ilvl in reversed(range(maxlvl)):
        step = 1 << ilvl
        if z % step:
            continue
        if ilvl < maxlvl - 1:
            rows = [r for al in range(ilvl + 1, maxlvl) for r in backend.tree_rows(al)]
            if z % (2 * step) == 0:
                if z + step < npdep:
                    for a, b in rows:
                        comm.send(x[:, a:b], z + step)
            else:
                for a, b in rows:
                    tmp = torch.empty((x.shape[0], b - a), dtype=x.dtype, device=x.device)
                    comm.recv(tmp, z - step)
                    x[:, a:b] = tmp
            _sync(backend)
        backend.solve_level(ilvl, -1, x)
    # assemble: every row is final on the layer that owns its forest
    out = torch.zeros_like(x)
    for ilvl in range(maxlvl):
        if z % (1 << ilvl) == 0:
            for a, b in backend.tree_rows(ilvl):
                out[:, a:b] = x[:, a:b]
    comm.allreduce_sum(out)
    x.copy_(out)
    _sync(backend)
    return x


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class GpuLayer:
    """Product backend: one Z layer's factors resident in HBM (sluamd_dCreateLUHandleFromSymb3D)."""

    def __init__(self, symb, nzval, npdep, z, device=-1, sn_tree=None):
        import torch
        from .driver import LUHandle, _pi, _pd
        L = _lib.load()
        self.L, self.symb, self.npdep, self.z, self.n = L, symb, npdep, z, symb.n
        self.device = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        ns = symb.nsupers
        if sn_tree is None:
            sn_tree = np.zeros(ns, dtype=np.int32)
            _lib.check(L.sluamd_symb_partition(symb._h, npdep, sn_tree.ctypes.data_as(_lib.P_int)), "sluamd_symb_partition")
        self.sn_tree = np.ascontiguousarray(sn_tree, dtype=np.int32)
        self.trees = path_trees(npdep, z)
        o = LUHandle._opts(device=device)
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        self._h = C.c_void_p()
        _lib.check(L.sluamd_dCreateLUHandleFromSymb3D(C.byref(self._h), symb._h, _pi(symb.rowptr), _pi(symb.colind), _pd(nz),
                                                      _pi(symb.perm_c), C.byref(o), npdep, z,
                                                      self.sn_tree.ctypes.data_as(_lib.P_int)), "sluamd_dCreateLUHandleFromSymb3D")
        self.handle = LUHandle(self._h, None)
        lo = np.zeros(ns + 1, dtype=np.int64); uo = np.zeros(ns + 1, dtype=np.int64)
        P64 = C.POINTER(C.c_int64)
        L.sluamd_local_offsets(self._h, lo.ctypes.data_as(P64), uo.ctypes.data_as(P64))
        dval = C.c_void_p(); nl = C.c_int64(); nu = C.c_int64()
        L.sluamd_arena(self._h, C.byref(dval), C.byref(nl), C.byref(nu))
        self.lval_off, self.uval_off, self.nnzL, self.nnzU = lo, uo, nl.value, nu.value
        tot = nl.value + nu.value
        self.arena = torch.as_tensor(_DevArray(dval.value, max(tot, 1)), device=self.device)[:tot]
        self.xsup = symb.xsup()
        self._nodes = [np.nonzero(self.sn_tree == t)[0] for t in self.trees]

    def tree_nodes(self, ilvl):
        return self._nodes[ilvl]

    def tree_rows(self, ilvl):
        return [(int(self.xsup[a]), int(self.xsup[b])) for a, b in runs(self._nodes[ilvl])]

    def value_slices(self, alvl_from, alvl_to=None):
        segs = []
        for al in range(alvl_from, len(self.trees) if alvl_to is None else alvl_to):
            for a, b in runs(self._nodes[al]):
                segs.append((int(self.lval_off[a]), int(self.lval_off[b])))
                segs.append((self.nnzL + int(self.uval_off[a]), self.nnzL + int(self.uval_off[b])))
        segs = sorted(s for s in segs if s[1] > s[0])
        merged = []
        for a, b in segs:                              # ancestors are adjacent in the local arena: few, large messages
            if merged and merged[-1][1] == a:
                merged[-1][1] = b
            else:
                merged.append([a, b])
        return [self.arena[a:b] for a, b in merged]

    def factor_level(self, ilvl, thresh):
        _lib.check(self.L.sluamd_pdgstrf3d_level(self._h, ilvl, float(thresh)), "sluamd_pdgstrf3d_level")

    # ---- cooperative mode (grid3d.pdgstrf3d_coop): everything queues on torch's current stream ----
    def coop_info(self, ilvl):
        import torch
        self.L.sluamd_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        cache = self.__dict__.setdefault("_coop_cache", {})
        if ilvl in cache:
            self._stage_sz = cache[ilvl][2]
            return cache[ilvl][0], cache[ilvl][1]
        nl = C.c_int32(); mx = C.c_int64()
        _lib.check(self.L.sluamd_coop_info(self._h, ilvl, C.byref(nl), C.byref(mx)), "sluamd_coop_info")
        self._stage_sz = []
        for l in range(nl.value):
            nn = C.c_int32(); sz = C.c_int64()
            self.L.sluamd_coop_level_size(self._h, ilvl, l, C.byref(nn), C.byref(sz))
            self._stage_sz.append(sz.value)
        cache[ilvl] = (nl.value, mx.value, self._stage_sz)
        return nl.value, mx.value

    def coop_level_nodes(self, ilvl, l):
        key = (ilvl, l)
        c = self.__dict__.setdefault("_lvl_nodes", {})
        if key not in c:
            nn = C.c_int32(); sz = C.c_int64()
            self.L.sluamd_coop_level_size(self._h, ilvl, l, C.byref(nn), C.byref(sz))
            buf = np.zeros(nn.value, dtype=np.int32)
            _lib.check(self.L.sluamd_coop_level_nodes(self._h, ilvl, l, buf.ctypes.data_as(_lib.P_int)), "sluamd_coop_level_nodes")
            c[key] = buf
        return c[key]

    def panel_tensors(self, k):
        import torch
        c = self.__dict__.setdefault("_panel_t", {})
        if k not in c:
            pl = C.c_void_p(); nl = C.c_int64(); pd = C.c_void_p(); nd = C.c_int64()
            _lib.check(self.L.sluamd_coop_panel_ptrs(self._h, k, C.byref(pl), C.byref(nl), C.byref(pd), C.byref(nd)), "sluamd_coop_panel_ptrs")
            c[k] = [torch.as_tensor(_DevArray(pl.value, nl.value), device=self.device),
                    torch.as_tensor(_DevArray(pd.value, nd.value), device=self.device)]
        return c[k]

    def stage_buffer(self, ndoubles):
        import torch
        if getattr(self, "_stage", None) is None or self._stage.numel() < ndoubles:
            self._stage = torch.empty(max(int(ndoubles), 1), dtype=torch.float64, device=self.device)
        return self._stage

    def coop_panel(self, ilvl, l, G, g, thresh, stage):
        sp = C.c_void_p(stage.data_ptr()) if stage is not None else None
        _lib.check(self.L.sluamd_coop_panel(self._h, ilvl, l, G, g, float(thresh), sp), "sluamd_coop_panel")
        return self._stage_sz[l]

    def coop_update(self, ilvl, l, G, g, stage):
        sp = C.c_void_p(stage.data_ptr()) if stage is not None else None
        _lib.check(self.L.sluamd_coop_update(self._h, ilvl, l, G, g, sp), "sluamd_coop_update")

    def coop_mask_u(self, ilvl, G, g):
        _lib.check(self.L.sluamd_coop_mask_u(self._h, ilvl, G, g), "sluamd_coop_mask_u")

    def u_slices(self, ilvl):
        out = []
        for a, b in runs(self._nodes[ilvl]):
            lo, hi = self.nnzL + int(self.uval_off[a]), self.nnzL + int(self.uval_off[b])
            if hi > lo:
                out.append(self.arena[lo:hi])
        return out

    def solve_level(self, ilvl, direction, x):
        assert x.is_contiguous() and x.dtype.itemsize == 8
        _lib.check(self.L.sluamd_pdgstrs3d_level(self._h, ilvl, direction, C.c_void_p(x.data_ptr()), x.shape[1], x.shape[0]),
                   "sluamd_pdgstrs3d_level")

    def sync(self):
        import torch
        torch.cuda.synchronize(self.device)

    def info(self):
        i = C.c_int32(); t = C.c_int32()
        self.L.sluamd_factor_info(self._h, C.byref(i), C.byref(t))
        return i.value, t.value

    def reset_values(self):
        self.handle.reset_values()

    def stats(self):
        return self.handle.stats()

    def destroy(self):
        self.handle.destroy()
