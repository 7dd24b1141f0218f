"""Synthetic inputs for the pdgstrf3d / pdgstrs3d hot path.

* ``poisson3d(N)``      7-point Poisson on an N^3 grid, natural index ((i*N+j)*N+k), diagonal 6,
                        off-diagonals -1, Dirichlet truncation.
* ``nd_perm_grid3d``    geometric nested-dissection column permutation for that grid; stands in for
                        METIS (absent in this image) and is passed to both the reference
                        (``ColPerm = MY_PERMC``, SRC/double/pdgssvx3d.c:749-791) and to our driver so both
                        factor the same permuted matrix.
* ``xtrue_rhs``         xtrue_i = +-1 alternating (reference: SRC/double/dutil_dist.c:598 dGenXtrue_dist),
                        b = A xtrue (dFillRHS_dist :619).
* ``random_unsym``      small diagonally dominant matrix with an UNSYMMETRIC pattern (ragged U skyline).
"""
import numpy as np


def poisson3d(N, nx=None, ny=None, nz=None):
    nx = nx or N; ny = ny or N; nz = nz or N
    n = nx * ny * nz
    if n >= 200000:      # bench sizes: the library's own generator (sluamd_poisson3d, the same operator entry for entry -- tests/test_host_symbolic.py compares the two)
        try:
            import ctypes as C
            from . import _lib
            L = _lib.load()
            nnz = 7 * n - 2 * (nx * ny + ny * nz + nx * nz)
            rp = np.empty(n + 1, dtype=np.int32); ci = np.empty(nnz, dtype=np.int32); v = np.empty(nnz)
            P_int, P_dbl = C.POINTER(C.c_int32), C.POINTER(C.c_double)
            got = L.sluamd_poisson3d(nx, ny, nz, rp.ctypes.data_as(P_int), ci.ctypes.data_as(P_int), v.ctypes.data_as(P_dbl))
            if got == nnz:
                return n, rp, ci, v
        except Exception:
            pass             # (library not built yet: the numpy construction below)
    idx = np.arange(n, dtype=np.int64)
    i = idx // (ny * nz); j = (idx // nz) % ny; k = idx % nz
    rows = [idx]; cols = [idx]; vals = [np.full(n, 6.0)]
    for cond, off in ((i > 0, -ny * nz), (i < nx - 1, ny * nz), (j > 0, -nz), (j < ny - 1, nz),
                      (k > 0, -1), (k < nz - 1, 1)):
        r = idx[cond]
        rows.append(r); cols.append(r + off); vals.append(np.full(r.size, -1.0))
    rows = np.concatenate(rows); cols = np.concatenate(cols); vals = np.concatenate(vals)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return n, rowptr.astype(np.int32), cols.astype(np.int32), vals


def nd_perm_grid3d(nx, ny, nz, leaf=64, sep_leaf=32):
    """Return perm_c with perm_c[old] = new (SuperLU convention) for the natural grid index.

    Recursive coordinate bisection (longest dimension, middle plane, separator last).  The separator plane itself is
    ordered by the SAME bisection rule restricted to its own two dimensions (and its separator lines likewise), so
    that the interface of every descendant sub-box with an ancestor separator is a contiguous index range: the L/U
    blocks coupling a supernode to an ancestor separator then come in full supernode-sized pieces instead of
    line-by-line fragments."""
    out = []
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    gid = (ii * ny + jj) * nz + kk

    def rec(x0, x1, y0, y1, z0, z1, cap):
        dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
        if dx <= 0 or dy <= 0 or dz <= 0:
            return
        if dx * dy * dz <= cap or max(dx, dy, dz) < 3:
            out.append(gid[x0:x1, y0:y1, z0:z1].ravel())
            return
        if dx >= dy and dx >= dz:
            m = x0 + dx // 2
            rec(x0, m, y0, y1, z0, z1, cap); rec(m + 1, x1, y0, y1, z0, z1, cap)
            rec(m, m + 1, y0, y1, z0, z1, sep_leaf)
        elif dy >= dz:
            m = y0 + dy // 2
            rec(x0, x1, y0, m, z0, z1, cap); rec(x0, x1, m + 1, y1, z0, z1, cap)
            rec(x0, x1, m, m + 1, z0, z1, sep_leaf)
        else:
            m = z0 + dz // 2
            rec(x0, x1, y0, y1, z0, m, cap); rec(x0, x1, y0, y1, m + 1, z1, cap)
            rec(x0, x1, y0, y1, m, m + 1, sep_leaf)

    rec(0, nx, 0, ny, 0, nz, leaf)
    order = np.concatenate(out)           # order[new] = old
    perm = np.empty(nx * ny * nz, dtype=np.int32)
    perm[order] = np.arange(order.size, dtype=np.int32)
    return perm


def xtrue_rhs(n, rowptr, colind, vals, nrhs=1):
    """xtrue = +1/-1 alternating per row (reference dGenXtrue_dist); b = A*xtrue."""
    x = np.where((np.arange(n) % 2) == 1, 1.0, -1.0)   # i odd ? 1 : -1  (dutil_dist.c:598)
    xt = np.tile(x[:, None], (1, nrhs)).copy(order="F")
    b = csr_matvec(n, rowptr, colind, vals, xt)
    return xt, b


def csr_matvec(n, rowptr, colind, vals, x):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    try:        # compiled CSR product when scipy is there (bench / tests: a 7 M-entry matrix costs 0.1 s through np.add.at)
        import scipy.sparse as _sp
        return np.asfortranarray(_sp.csr_matrix((vals, colind, rowptr), shape=(n, n)) @ x)
    except ImportError:
        pass
    prod = vals[:, None] * x[colind, :]
    out = np.zeros((n, x.shape[1]), dtype=prod.dtype)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    np.add.at(out, rows, prod)
    return np.asfortranarray(out)


def random_unsym(n, density=0.02, seed=0, diag_scale=None):
    rng = np.random.default_rng(seed)
    nnz_off = int(density * n * n)
    r = rng.integers(0, n, nnz_off); c = rng.integers(0, n, nnz_off)
    keep = r != c
    r, c = r[keep], c[keep]
    key = np.unique(r.astype(np.int64) * n + c)
    r = (key // n).astype(np.int64); c = (key % n).astype(np.int64)
    v = rng.uniform(-1.0, 1.0, r.size)
    rowsum = np.zeros(n); np.add.at(rowsum, r, np.abs(v))
    colsum = np.zeros(n); np.add.at(colsum, c, np.abs(v))
    d = np.maximum(rowsum, colsum) + 1.0
    if diag_scale is not None:          # NOT diagonally dominant: unpivoted LU loses digits -> exercises iterative refinement
        d = diag_scale * d
    rows = np.concatenate([r, np.arange(n)]); cols = np.concatenate([c, np.arange(n)])
    vals = np.concatenate([v, d])
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = np.zeros(n + 1, dtype=np.int64); np.add.at(rowptr, rows + 1, 1)
    return n, np.cumsum(rowptr).astype(np.int32), cols.astype(np.int32), vals


def stencil3d_unsym(N, drop=0.3, seed=0, reach=2):
    """Irregular unsymmetric-PATTERN matrix on an N^3 grid (stand-in for unstructured FE matrices such as SuiteSparse
    audikw_1, which is not available offline): every grid point couples to a random subset of the points within `reach`
    grid steps along each axis (entries dropped independently with probability `drop`, so A(i,j) != 0 does not imply
    A(j,i) != 0), uniform(-1, 1) values, diagonally dominant."""
    rng = np.random.default_rng(seed)
    n = N * N * N
    idx = np.arange(n)
    ii, jj, kk = idx // (N * N), (idx // N) % N, idx % N
    rows, cols = [], []
    for di in range(-reach, reach + 1):
        for dj in range(-reach, reach + 1):
            for dk in range(-reach, reach + 1):
                if (di, dj, dk) == (0, 0, 0) or abs(di) + abs(dj) + abs(dk) > reach:
                    continue
                ok = (ii + di >= 0) & (ii + di < N) & (jj + dj >= 0) & (jj + dj < N) & (kk + dk >= 0) & (kk + dk < N)
                ok &= rng.random(n) >= drop
                rows.append(idx[ok]); cols.append(((ii + di) * N + (jj + dj)) * N + (kk + dk))
                cols[-1] = cols[-1][ok]
    r = np.concatenate(rows); c = np.concatenate(cols)
    v = rng.uniform(-1.0, 1.0, r.size)
    rowsum = np.zeros(n); np.add.at(rowsum, r, np.abs(v))
    colsum = np.zeros(n); np.add.at(colsum, c, np.abs(v))
    d = np.maximum(rowsum, colsum) + 1.0
    rows = np.concatenate([r, idx]); cols = np.concatenate([c, idx]); vals = np.concatenate([v, d])
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = np.zeros(n + 1, dtype=np.int64); np.add.at(rowptr, rows + 1, 1)
    return n, np.cumsum(rowptr).astype(np.int32), cols.astype(np.int32), vals


def elasticity3d_like(N, dof=3, drop=0.1, seed=0, shuffle=True):
    """Stand-in for BASELINE.json configs[3] (SuiteSparse audikw_1: symmetric positive definite, 3-D structural mesh with 3 unknowns
    per node, n = 943 695, ~82 entries per row -- not available offline): N^3 nodes x `dof` unknowns, every node coupled to its
    27-point neighbourhood through dense dof x dof blocks (81 entries per row inside the mesh), a fraction `drop` of the node pairs
    removed at random (irregular row lengths and separators), symmetric values, strictly diagonally dominant (SPD: unpivoted LU is
    stable); the nodes are renumbered at random so that nothing but the graph tells an ordering where the separators are.
    N = 68, dof = 3 gives n = 943 296."""
    rng = np.random.default_rng(seed)
    idx = np.arange(N ** 3).reshape(N, N, N)
    pairs_a, pairs_b = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if (dx, dy, dz) <= (0, 0, 0):
                    continue                         # each undirected node pair once
                sa = idx[max(0, -dx):N - max(0, dx), max(0, -dy):N - max(0, dy), max(0, -dz):N - max(0, dz)]
                sb = idx[max(0, dx):N + min(0, dx), max(0, dy):N + min(0, dy), max(0, dz):N + min(0, dz)]
                pairs_a.append(sa.ravel()); pairs_b.append(sb.ravel())
    a = np.concatenate(pairs_a); b = np.concatenate(pairs_b)
    keep = rng.random(a.size) >= drop
    a, b = a[keep], b[keep]
    if shuffle:
        ren = rng.permutation(N ** 3)
        a, b = ren[a], ren[b]
    nn = N ** 3
    # block entries: pair (a, b) -> dof x dof block W and its transpose at (b, a)
    W = -rng.random((a.size, dof, dof))
    ii = (a[:, None, None] * dof + np.arange(dof)[None, :, None]) + np.zeros((1, 1, dof), dtype=np.int64)
    jj = (b[:, None, None] * dof + np.arange(dof)[None, None, :]) + np.zeros((1, dof, 1), dtype=np.int64)
    rows = np.concatenate([ii.ravel(), jj.ravel()]); cols = np.concatenate([jj.ravel(), ii.ravel()]); vals = np.concatenate([W.ravel(), W.ravel()])
    # diagonal blocks: symmetric coupling of the node's own unknowns, diagonal = 1 + sum of the row's magnitudes
    D = -rng.random((nn, dof, dof)); D = 0.5 * (D + D.transpose(0, 2, 1))
    di = (np.arange(nn)[:, None, None] * dof + np.arange(dof)[None, :, None]) + np.zeros((1, 1, dof), dtype=np.int64)
    dj = (np.arange(nn)[:, None, None] * dof + np.arange(dof)[None, None, :]) + np.zeros((1, dof, 1), dtype=np.int64)
    off = di.ravel() != dj.ravel()
    rows = np.concatenate([rows, di.ravel()[off]]); cols = np.concatenate([cols, dj.ravel()[off]]); vals = np.concatenate([vals, D.ravel()[off]])
    import scipy.sparse as sp
    n = nn * dof
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    rowsum = np.asarray(abs(A).sum(axis=1)).ravel()
    A = (A + sp.diags(1.0 + rowsum)).tocsr()
    A.sort_indices()
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def write_triplet_dat(path, n, rowptr, colind, vals):
    """'.dat' triplet file the reference reads (SRC/double/dreadtriple.c:43-92, complex: SRC/complex16/zreadtriple.c):
    header 'm n nnz', 1-based, 'row col value' or 'row col re im'."""
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    cplx = np.iscomplexobj(vals)
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(vals)}\n")
        for r, c, v in zip(rows, colind, vals):
            if cplx:
                f.write(f"{r + 1} {c + 1} {float(v.real)!r} {float(v.imag)!r}\n")
            else:
                f.write(f"{r + 1} {c + 1} {float(v)!r}\n")


def complex_shift(vals, rowptr, colind, seed=0):
    """Complex test values on an existing pattern: v*(1 + 0.3i*u) off the diagonal, diagonal += (1 + 0.5i)
    (keeps diagonal dominance; the cg20-style 'complex grid operator' of BASELINE.json config 5)."""
    rng = np.random.default_rng(seed)
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    out = vals.astype(np.complex128) * (1.0 + 0.3j * rng.uniform(-1, 1, len(vals)))
    d = rows == colind
    out[d] = vals[d] * (1.0 + 0.0j) + (1.0 + 0.5j)
    return out


def read_matrix_market(path):
    """MatrixMarket coordinate file -> CSR (n, rowptr, colind, values) with symmetric / skew-symmetric storage expanded to the full
    pattern and duplicates summed: what dreadMM_dist does for the reference (SRC/double/dreadMM.c).  real / integer / pattern / complex."""
    with open(path) as f:
        banner = f.readline().split()
        if len(banner) != 5 or banner[0] != "%%MatrixMarket" or banner[2].lower() != "coordinate":
            raise ValueError("not a MatrixMarket coordinate file: " + path)
        field, sym = banner[3].lower(), banner[4].lower()
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        m, n, nz = (int(t) for t in line.split())
        if m != n:
            raise ValueError("matrix is not square")
        data = np.loadtxt(f, ndmin=2) if nz else np.zeros((0, 3))
    r = data[:, 0].astype(np.int64) - 1; c = data[:, 1].astype(np.int64) - 1
    if field == "pattern":
        val = np.ones(len(r))
    elif field == "complex":
        val = data[:, 2] + 1j * data[:, 3]
    else:
        val = data[:, 2].astype(np.float64)
    if sym in ("symmetric", "skew-symmetric", "hermitian"):
        off = r != c
        r2, c2 = c[off], r[off]
        v2 = -val[off] if sym == "skew-symmetric" else (np.conj(val[off]) if sym == "hermitian" else val[off])
        r, c, val = np.concatenate([r, r2]), np.concatenate([c, c2]), np.concatenate([val, v2])
    elif sym != "general":
        raise ValueError("unsupported MatrixMarket symmetry " + sym)
    import scipy.sparse as sp
    A = sp.coo_matrix((val, (r, c)), shape=(n, n)).tocsr()      # sums duplicates
    A.sort_indices()
    v = A.data.copy()
    if field == "pattern":      # a value that makes the unpivoted factorisation well defined: unit off-diagonals, dominant diagonal
        deg = np.diff(A.indptr)
        rows = np.repeat(np.arange(n), deg)
        v[A.indices == rows] = (deg + 1.0)[rows[A.indices == rows]]
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), v


def write_matrix_market(path, n, rowptr, colind, values, symmetric=False):
    """CSR -> MatrixMarket coordinate real (symmetric=True stores the lower triangle only, like the SuiteSparse files)."""
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    keep = rows >= colind if symmetric else np.ones(len(colind), dtype=bool)
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real " + ("symmetric" if symmetric else "general") + "\n% written by superlu_dist_amd.matgen\n")
        f.write(f"{n} {n} {int(keep.sum())}\n")
        np.savetxt(f, np.column_stack([rows[keep] + 1, colind[keep] + 1, values[keep]]), fmt="%d %d %.17g")
