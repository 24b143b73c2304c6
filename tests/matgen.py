"""Synthetic inputs (test + bench infrastructure, numpy only).

Generators emit CSR in (row, col) order, which Csr::read expects from its
caller (core/matrix/csr.cpp:453-470).
"""
import numpy as np


def poisson_2d_5pt(nx, ny=None):
    """5-point stencil (-1 N, -1 W, 4 C, -1 E, -1 S) on an nx x ny grid,
    row = i*ny + j, ascending columns.  SURVEY.md 8(d) config P2 at 1000x1000."""
    ny = nx if ny is None else ny
    n = nx * ny
    i, j = np.divmod(np.arange(n, dtype=np.int64), ny)
    cols = np.stack([np.arange(n) - ny, np.arange(n) - 1, np.arange(n),
                     np.arange(n) + 1, np.arange(n) + ny], axis=1)
    valid = np.stack([i > 0, j > 0, np.ones(n, bool), j < ny - 1, i < nx - 1], axis=1)
    vals = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), (n, 5))
    row_ptrs = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(valid.sum(axis=1), out=row_ptrs[1:])
    return (n, row_ptrs, cols[valid].astype(np.int32),
            np.ascontiguousarray(vals[valid], dtype=np.float64))


def poisson_3d_7pt(nx, ny=None, nz=None):
    """7-point stencil on nx x ny x nz, row = (i*ny + j)*nz + k (config P3)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64)
    k = idx % nz
    j = (idx // nz) % ny
    i = idx // (nz * ny)
    offs = [-ny * nz, -nz, -1, 0, 1, nz, ny * nz]
    valid = np.stack([i > 0, j > 0, k > 0, np.ones(n, bool), k < nz - 1,
                      j < ny - 1, i < nx - 1], axis=1)
    cols = np.stack([idx + o for o in offs], axis=1)
    vals = np.broadcast_to(np.array([-1.0, -1, -1, 6.0, -1, -1, -1]), (n, 7))
    row_ptrs = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(valid.sum(axis=1), out=row_ptrs[1:])
    return (n, row_ptrs, cols[valid].astype(np.int32),
            np.ascontiguousarray(vals[valid], dtype=np.float64))


def random_csr(nrows, ncols, min_nnz, max_nnz, seed, sort=True, dist="uniform"):
    """Random CSR like gko::test::generate_random_matrix
    (core/test/utils/matrix_generator.hpp): per-row nnz count from a
    distribution, distinct columns, normal(0,1)-like values."""
    rng = np.random.default_rng(seed)
    if dist == "uniform":
        counts = rng.integers(min_nnz, max_nnz + 1, size=nrows)
    else:  # heavy tail: most rows short, a few very long
        counts = np.minimum(max_nnz, min_nnz + rng.geometric(0.15, size=nrows) - 1)
        long_rows = rng.choice(nrows, size=max(1, nrows // 100), replace=False)
        counts[long_rows] = max_nnz
    counts = np.minimum(counts, ncols)
    row_ptrs = np.zeros(nrows + 1, dtype=np.int32)
    np.cumsum(counts, out=row_ptrs[1:])
    nnz = int(row_ptrs[-1])
    cols = np.empty(nnz, dtype=np.int32)
    for r in range(nrows):
        c = rng.choice(ncols, size=counts[r], replace=False)
        if sort:
            c.sort()
        cols[row_ptrs[r]:row_ptrs[r + 1]] = c
    vals = rng.standard_normal(nnz)
    return row_ptrs, cols, vals


def stencil_3d_27pt(nx):
    """27-point stencil on an nx^3 grid (FEM-like: up to 27 entries per row,
    ascending columns): diagonal 26, off-diagonals -1."""
    idx = np.arange(nx ** 3, dtype=np.int64)
    i, j, k = idx // (nx * nx), (idx // nx) % nx, idx % nx
    cols, valid = [], []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for dk in (-1, 0, 1):
                ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < nx) & (k + dk >= 0) & (k + dk < nx)
                cols.append(idx + (di * nx + dj) * nx + dk)
                valid.append(ok)
    cols, valid = np.stack(cols, axis=1), np.stack(valid, axis=1)
    vals = np.where(np.arange(27) == 13, 26.0, -1.0)[None, :].repeat(len(idx), axis=0)
    row_ptrs = np.zeros(len(idx) + 1, dtype=np.int32)
    np.cumsum(valid.sum(axis=1), out=row_ptrs[1:])
    return len(idx), row_ptrs, cols[valid].astype(np.int32), np.ascontiguousarray(vals[valid])


def random_rows_csr(nrows, ncols, counts, seed, local=None):
    """Large random CSR without a Python loop over rows: row r gets counts[r]
    columns (sorted; duplicates are merged by nudging, so rows stay valid), either
    uniform over all columns or within +-local of the diagonal."""
    rng = np.random.default_rng(seed)
    counts = np.minimum(np.asarray(counts, dtype=np.int64), ncols)
    row_ptrs = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptrs[1:])
    nnz = int(row_ptrs[-1])
    rows = np.repeat(np.arange(nrows, dtype=np.int64), counts)
    if local is None:
        cols = rng.integers(0, ncols, size=nnz)
    else:
        cols = np.clip(rows * ncols // nrows + rng.integers(-local, local + 1, size=nnz), 0, ncols - 1)
    order = np.lexsort((cols, rows))
    cols = cols[order]
    # duplicates inside a row: keep the first, drop the rest
    keep = np.ones(nnz, dtype=bool)
    keep[1:] = (cols[1:] != cols[:-1]) | (rows[1:] != rows[:-1])
    rows, cols = rows[keep], cols[keep]
    rp = np.zeros(nrows + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=nrows), out=rp[1:])
    return rp, cols.astype(np.int32), rng.standard_normal(len(cols))


def read_mtx(path):
    """Minimal MatrixMarket reader (coordinate real/integer general|symmetric,
    array real general).  Returns ('coo', nrows, ncols, rows, cols, vals)
    sorted row-major like gko::read (include/ginkgo/core/base/mtx_io.hpp:89-90),
    or ('array', nrows, ncols, values[nrows, ncols])."""
    with open(path) as f:
        header = f.readline().lower().split()
        assert header[0] == "%%matrixmarket"
        layout, field, sym = header[2], header[3], header[4]
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        dims = [int(t) for t in line.split()]
        body = f.read().split()
    if layout == "array":
        nrows, ncols = dims
        vals = np.array(body, dtype=np.float64).reshape(ncols, nrows).T.copy()
        return ("array", nrows, ncols, vals)
    nrows, ncols, nnz = dims
    per = 2 if field == "pattern" else 3
    data = np.array(body, dtype=np.float64).reshape(nnz, per)
    rows = data[:, 0].astype(np.int64) - 1
    cols = data[:, 1].astype(np.int64) - 1
    vals = data[:, 2] if per == 3 else np.ones(nnz)
    if sym == "symmetric":
        off = rows != cols
        rows, cols, vals = (np.concatenate([rows, cols[off]]),
                            np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, vals[off]]))
    order = np.lexsort((cols, rows))
    return ("coo", nrows, ncols, rows[order].astype(np.int32),
            cols[order].astype(np.int32), vals[order].copy())


def coo_to_csr(nrows, rows, cols, vals):
    row_ptrs = np.zeros(nrows + 1, dtype=np.int32)
    np.add.at(row_ptrs, rows.astype(np.int64) + 1, 1)
    np.cumsum(row_ptrs, out=row_ptrs)
    return row_ptrs, cols.astype(np.int32), vals.astype(np.float64)


def dense_to_csr(a):
    a = np.asarray(a, dtype=np.float64)
    rows, cols = np.nonzero(a)
    return coo_to_csr(a.shape[0], rows.astype(np.int32), cols.astype(np.int32), a[rows, cols])


def rel_err(a, b):
    """GKO_ASSERT_MTX_NEAR metric: sqrt(sum|a-b|^2 / max(sum|a|^2, sum|b|^2))
    (core/test/utils/assertions.hpp:205-233)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.sum(a * a)), float(np.sum(b * b)))
    num = float(np.sum((a - b) ** 2))
    if den == 0.0:
        return 0.0 if num == 0.0 else float("inf")
    return (num / den) ** 0.5


# ---- offline stand-ins of the SuiteSparse matrices named by BASELINE.json ----
# (thermal2 / atmosmodd are not in the container)

def permute_symmetric(n, rp, ci, v, seed=42):
    """P A P^T with a fixed random permutation, rows and columns sorted."""
    rng = np.random.default_rng(seed)
    inv = np.argsort(rng.permutation(n))
    rows = np.repeat(np.arange(n), np.diff(rp))
    pr, pc = inv[rows], inv[ci]
    order = np.lexsort((pc, pr))
    return coo_to_csr(n, pr[order].astype(np.int32), pc[order].astype(np.int32), v[order])


def t2_like_permuted(g=1108, seed=42):
    """"T2-like" of SURVEY 8(d): 2-D 5-pt Poisson g x g (n = 1 227 664 at 1108)
    under a fixed random symmetric permutation that destroys the banded locality."""
    n, rp, ci, v = poisson_2d_5pt(g)
    rp2, ci2, v2 = permute_symmetric(n, rp, ci, v, seed)
    return n, rp2, ci2, v2


def diffusion_2d_patch_ordered(g=1104, patch=(4, 8), seed=7, contrast=3.0, coarse=16):
    """Second thermal2 stand-in: -div(k grad u) on a g x g cell grid, k piecewise
    constant on coarse x coarse cell blocks with log10 k uniform in
    +-contrast/2, harmonic-mean edge coefficients, Dirichlet boundary; cells
    numbered patch by patch (patch[0] x patch[1] cells = 32 consecutive rows),
    the locality a FEM ordering has and a lexicographic grid lacks: one block of
    block-Jacobi(32) is one patch.  SPD, <= 5 entries per row, columns sorted."""
    rng = np.random.default_rng(seed)
    pi, pj = patch
    assert g % pi == 0 and g % pj == 0
    nb = (g + coarse - 1) // coarse
    kc = 10.0 ** (rng.uniform(-contrast / 2, contrast / 2, size=(nb, nb)))
    k = np.repeat(np.repeat(kc, coarse, axis=0), coarse, axis=1)[:g, :g]
    i, j = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    # cell (i, j) -> index: patches row-major, cells row-major inside a patch
    idx = ((i // pi) * (g // pj) + (j // pj)) * (pi * pj) + (i % pi) * pj + (j % pj)
    hm = lambda a, b: 2.0 * a * b / (a + b)
    n = g * g
    rows, cols, vals = [], [], []
    diag = np.zeros((g, g))
    for di, dj in ((-1, 0), (0, -1), (0, 1), (1, 0)):
        ii, jj = i + di, j + dj
        inside = (ii >= 0) & (ii < g) & (jj >= 0) & (jj < g)
        w = np.where(inside, hm(k, k[np.clip(ii, 0, g - 1), np.clip(jj, 0, g - 1)]), k)  # boundary: own k
        diag += w
        rows.append(idx[inside])
        cols.append(idx[ii[inside], jj[inside]])
        vals.append(-w[inside])
    rows.append(idx.ravel())
    cols.append(idx.ravel())
    vals.append(diag.ravel())
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((cols, rows))
    rp, ci, v = coo_to_csr(n, rows[order].astype(np.int32), cols[order].astype(np.int32), vals[order])
    return n, rp, ci, v


def at_like(g=108, upwind=0.5):
    """"AT-like" of SURVEY 8(d): 3-D 7-pt convection-diffusion g^3 (n = 1 259 712
    at 108), nonsymmetric (upwind term on the k-1 neighbour)."""
    n, rp, ci, v = poisson_3d_7pt(g)
    v = v.copy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows - 1] -= upwind
    v[ci == rows] += upwind
    return n, rp, ci, v
