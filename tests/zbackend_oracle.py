"""CPU backend for superlu_dist_amd.grid3d built on the oracle (TEST INFRASTRUCTURE): lets the Z-sharded orchestration
run over gloo on a CPU-only box.  Every rank holds the full structure; values of supernodes it does not own start at 0."""
import numpy as np
import torch
import oracle as orc
from superlu_dist_amd import grid3d


class OracleLayer:
    def __init__(self, symb, nzval, npdep, z):
        self.n, self.npdep, self.z = symb.n, npdep, z
        self.device = torch.device("cpu")
        symb.distribute_host(nzval)
        fs = symb.flat_store()
        self.sn_tree = symb.partition(npdep)
        self.trees = grid3d.path_trees(npdep, z)
        self._nodes = [np.nonzero(self.sn_tree == t)[0].astype(np.int32) for t in self.trees]
        owned = np.zeros(fs.nsupers, dtype=bool)
        for l, nodes in enumerate(self._nodes):
            if z % (1 << l) == 0:
                owned[nodes] = True
        for k in np.nonzero(~owned)[0]:            # dinit3DLUstructForest: zero the replicas I do not own
            fs.Lnzval[fs.Lnzval_off[k]:fs.Lnzval_off[k + 1]] = 0.0
            fs.Unzval[fs.Unzval_off[k]:fs.Unzval_off[k + 1]] = 0.0
        self.store = orc.LUStore(fs.n, fs.xsup, fs.Lrowind_off, fs.Lrowind, fs.Lnzval_off, fs.Lnzval, fs.Ufstnz_off, fs.Ufstnz,
                                 fs.Unzval_off, fs.Unzval)
        self.xsup = fs.xsup
        self._tl = torch.from_numpy(self.store.Lnzval)
        self._tu = torch.from_numpy(self.store.Unzval)
        self._info = 0

    def tree_rows(self, ilvl):
        return [(int(self.xsup[a]), int(self.xsup[b])) for a, b in grid3d.runs(self._nodes[ilvl])]

    def value_slices(self, alvl_from, alvl_to=None):
        out = []
        for al in range(alvl_from, len(self.trees) if alvl_to is None else alvl_to):
            for a, b in grid3d.runs(self._nodes[al]):
                out.append(self._tl[int(self.store.Lnzval_off[a]):int(self.store.Lnzval_off[b])])
                out.append(self._tu[int(self.store.Unzval_off[a]):int(self.store.Unzval_off[b])])
        return [s for s in out if s.numel()]

    def factor_level(self, ilvl, thresh):
        info, tiny, _ = orc.dfactor(self.store, self._nodes[ilvl], False, thresh)
        if info and not self._info:
            self._info = info

    # ---- cooperative mode: one supernode per "DAG level" (sequential order is a valid schedule) ----
    def coop_info(self, ilvl):
        nodes = self._nodes[ilvl]
        mx = max([int(self.store.Lnzval_off[k + 1] - self.store.Lnzval_off[k]) for k in nodes], default=0)
        return len(nodes), mx

    def coop_level_nodes(self, ilvl, l):
        return self._nodes[ilvl][l:l + 1]

    def panel_tensors(self, k):
        return [self._tl[int(self.store.Lnzval_off[k]):int(self.store.Lnzval_off[k + 1])]]

    def stage_buffer(self, ndoubles):
        return torch.zeros(max(int(ndoubles), 1), dtype=torch.float64)

    def coop_panel(self, ilvl, l, G, g, thresh, stage):
        k = int(self._nodes[ilvl][l])
        info, _ = orc.dfactor_coop(self.store, [k], 1, G, g, False, thresh)
        if info and not self._info:
            self._info = info
        a, b = int(self.store.Lnzval_off[k]), int(self.store.Lnzval_off[k + 1])
        if stage is not None:
            if k % G == g:
                stage[:b - a] = self._tl[a:b]
            else:
                stage[:b - a] = 0.0
        return b - a

    def coop_update(self, ilvl, l, G, g, stage):
        k = int(self._nodes[ilvl][l])
        a, b = int(self.store.Lnzval_off[k]), int(self.store.Lnzval_off[k + 1])
        if stage is not None and k % G != g:
            self._tl[a:b] = stage[:b - a]
        orc.dfactor_coop(self.store, [k], 2, G, g)

    def _u_block_ranges(self, k):
        st = self.store
        ui = st.Ufstnz[st.Ufstnz_off[k]:st.Ufstnz_off[k + 1]]
        out = []
        if len(ui) == 0:
            return out
        klst = int(self.xsup[k + 1]); p = 3; r = int(st.Unzval_off[k])
        for _ in range(int(ui[0])):
            jb = int(ui[p]); nsj = int(self.xsup[jb + 1] - self.xsup[jb])
            nnz = int((klst - ui[p + 2:p + 2 + nsj]).sum())
            out.append((jb, r, r + nnz)); r += nnz; p += 2 + nsj
        return out

    def coop_mask_u(self, ilvl, G, g):
        for k in self._nodes[ilvl]:
            for jb, a, b in self._u_block_ranges(int(k)):
                if jb % G != g:
                    self._tu[a:b] = 0.0

    def u_slices(self, ilvl):
        out = [self._tu[int(self.store.Unzval_off[a]):int(self.store.Unzval_off[b])] for a, b in grid3d.runs(self._nodes[ilvl])]
        return [s for s in out if s.numel()]

    def solve_level(self, ilvl, direction, x):
        xf = x.numpy().T                            # (n, nrhs) Fortran-order view of the (nrhs, n) tensor
        assert xf.flags.f_contiguous
        orc.dsolve_level(self.store, xf, self._nodes[ilvl], direction)

    def info(self):
        return self._info, 0
