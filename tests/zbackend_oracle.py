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

    def value_slices(self, alvl_from):
        out = []
        for al in range(alvl_from, len(self.trees)):
            for a, b in grid3d.runs(self._nodes[al]):
                out.append(self._tl[int(self.store.Lnzval_off[a]):int(self.store.Lnzval_off[b])])
                out.append(self._tu[int(self.store.Unzval_off[a]):int(self.store.Unzval_off[b])])
        return [s for s in out if s.numel()]

    def factor_level(self, ilvl, thresh):
        info, tiny, _ = orc.dfactor(self.store, self._nodes[ilvl], False, thresh)
        if info and not self._info:
            self._info = info

    def solve_level(self, ilvl, direction, x):
        xf = x.numpy().T                            # (n, nrhs) Fortran-order view of the (nrhs, n) tensor
        assert xf.flags.f_contiguous
        orc.dsolve_level(self.store, xf, self._nodes[ilvl], direction)

    def info(self):
        return self._info, 0
