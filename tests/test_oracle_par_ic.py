"""Pins the oracle's ParIC kernels against
reference/test/factorization/par_ic_kernels.cpp (tests/golden/trs_ilu.json)."""
import json
import os

import numpy as np

import ilu_util
import matgen
from krylov_util import dense_to_csr

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trs_ilu.json")))["par_ic"]
TOL = G["tol"]


def sparse(A):
    """gko::initialize<Csr>(dense list): only the nonzeros are stored"""
    A = np.asarray(A, np.float64)
    n = A.shape[0]
    rp, ci, v = [0], [], []
    for r in range(n):
        for c in range(n):
            if A[r, c] != 0.0:
                ci.append(c); v.append(A[r, c])
        rp.append(len(ci))
    return n, np.array(rp, np.int32), np.array(ci, np.int32), np.array(v)


def test_kernel_init(oracle):
    n, rp, ci, v = sparse(G["mtx_l_system"])
    oracle.ref_par_ic_init_factor(n, rp, ci, v)
    assert matgen.rel_err(ilu_util.csr_to_dense(n, n, rp, ci, v), G["mtx_l_init_expect"]) <= TOL


def test_kernel_compute(oracle):
    n, rp, ci, v = sparse(G["mtx_l_system"])
    oracle.ref_par_ic_compute_factor(n, v.copy(), rp, ci, v)
    assert matgen.rel_err(ilu_util.csr_to_dense(n, n, rp, ci, v), G["mtx_l_it_expect"]) <= TOL


def test_generate(oracle):
    for A, L in ((G["identity"], G["identity"]), (G["banded"], G["banded_l_expect"]),
                 (G["mtx_system"], G["mtx_l_it_expect"])):
        n, rp, ci, v = sparse(A)
        f = oracle_ic = ilu_util.oracle_par_ic(oracle, n, rp, ci, v)
        assert matgen.rel_err(ilu_util.csr_to_dense(n, n, *f["L"]), L) <= TOL
        assert matgen.rel_err(ilu_util.csr_to_dense(n, n, *f["Lt"]), np.array(L).T) <= TOL


def test_ic0_of_poisson_reproduces_the_pattern_entries(oracle):
    # exact IC(0): (L L^T)(i,j) = A(i,j) on the sparsity pattern of the lower triangle
    n, rp, ci, v = matgen.poisson_2d_5pt(9)
    f = ilu_util.oracle_par_ic(oracle, n, rp, ci, v)
    L = ilu_util.csr_to_dense(n, n, *f["L"])
    A = ilu_util.csr_to_dense(n, n, rp, ci, v)
    P = (np.tril(A) != 0)
    assert np.allclose((L @ L.T)[P], A[P], rtol=0, atol=1e-13)
