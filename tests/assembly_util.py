"""Seeded triplet sets shaped like the fixture of the reference's own test
(test/base/device_matrix_data_kernels.cpp:63-108): 100 x 200, 1000 random
entries in [1, 2) + 1000 explicit zeros, unique locations, shuffled; a copy with
1000 duplicated locations carrying new random values."""
import numpy as np


def fixture(seed=82754, nrows=100, ncols=200, n_rand=1000, n_zero=1000, n_dup=1000):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, nrows, n_rand + n_zero).astype(np.int32)
    cols = rng.integers(0, ncols, n_rand + n_zero).astype(np.int32)
    vals = np.concatenate([rng.uniform(1.0, 2.0, n_rand), np.zeros(n_zero)])
    # row-major order, then keep the first entry of every location
    order = np.lexsort((np.arange(len(rows)), cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    first = np.ones(len(rows), bool)
    first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    sorted_t = (rows[first].copy(), cols[first].copy(), vals[first].copy())
    perm = rng.permutation(len(sorted_t[0]))
    host = tuple(a[perm].copy() for a in sorted_t)
    keep = host[2] != 0.0
    nonzero = tuple(a[keep].copy() for a in host)
    loc = rng.integers(0, len(host[0]), n_dup)
    dup = (np.concatenate([host[0], host[0][loc]]), np.concatenate([host[1], host[1][loc]]),
           np.concatenate([host[2], rng.uniform(1.0, 2.0, n_dup)]))
    return {"host": host, "sorted": sorted_t, "nonzero": nonzero, "duplicate": dup}


def sum_duplicates_numpy(rows, cols, vals):
    """stable sort + left-to-right sums starting from 0.0 (independent of the oracle)"""
    order = np.lexsort((np.arange(len(rows)), cols, rows))
    r, c, v = rows[order], cols[order], vals[order]
    out_r, out_c, out_v = [], [], []
    for i in range(len(r)):
        if i == 0 or r[i] != r[i - 1] or c[i] != c[i - 1]:
            out_r.append(r[i]); out_c.append(c[i]); out_v.append(0.0)
        out_v[-1] = out_v[-1] + v[i]
    return np.array(out_r, np.int32), np.array(out_c, np.int32), np.array(out_v)
