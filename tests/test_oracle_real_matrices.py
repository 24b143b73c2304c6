"""CPU: the oracle on the real matrices the reference ships as test data
(tests/golden/ani4.mtx, 1138_bus.mtx = copies of matrices/test/*.mtx), against
scipy as an independent witness: SpMV, the exact ILU(0) the ParILU fixed point
converges to, the triangular solves on its factors."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

import ilu_util
import matgen

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    kind, n, m, rows, cols, vals = matgen.read_mtx(os.path.join(G, name))
    rp, ci, v = matgen.coo_to_csr(n, rows, cols, vals)
    return n, rp, ci, v


@pytest.mark.parametrize("name,nnz", [("ani4.mtx", 20971), ("1138_bus.mtx", 4054)])
def test_spmv_matches_scipy(oracle, name, nnz):
    n, rp, ci, v = load(name)
    assert rp[-1] == nnz   # symmetric storage of 1138_bus expanded: 2 * 2596 - 1138
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    x = np.sin(0.1 * np.arange(n)).reshape(n, 1)
    y = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, x, 1, y, 1)
    assert matgen.rel_err(y, A @ x) <= 1e-15


def test_par_ilu_fixed_point_is_ilu0_and_trs_solves_it(oracle):
    n, rp, ci, v = load("ani4.mtx")
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    L = sp.csr_matrix((f["L"][2], f["L"][1], f["L"][0]), shape=(n, n))
    U = sp.csr_matrix((f["U"][2], f["U"][1], f["U"][0]), shape=(n, n))
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    # ILU(0): (L U)_ij == A_ij on the sparsity pattern of A
    LU = (L @ U).tocsr()
    rows = np.repeat(np.arange(n), np.diff(rp))
    assert np.allclose(np.asarray(LU[rows, ci]).ravel(), v, rtol=1e-10, atol=1e-12)
    assert np.allclose(L.diagonal(), 1.0)
    b = np.cos(0.05 * np.arange(n)).reshape(n, 1)
    y = np.zeros((n, 1))
    oracle.ref_lower_trs_solve(n, 1, f["L"][0], f["L"][1], f["L"][2], 0, b, 1, y, 1)
    assert matgen.rel_err(y, spl.spsolve_triangular(L, b, lower=True)) <= 1e-12
    z = np.zeros((n, 1))
    oracle.ref_upper_trs_solve(n, 1, f["U"][0], f["U"][1], f["U"][2], 0, y, 1, z, 1)
    assert matgen.rel_err(z, spl.spsolve_triangular(U, y, lower=False)) <= 1e-10
