"""Pins the oracle's triangular solves and ParILU chain against the
reference's known answers (tests/golden/trs_ilu.json)."""
import json
import os

import numpy as np
import pytest

import ilu_util
import matgen

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trs_ilu.json")))


@pytest.mark.parametrize("which", ["lower", "upper"])
def test_trs_known_answers(oracle, which):
    g = G[which]
    fn = oracle.ref_lower_trs_solve if which == "lower" else oracle.ref_upper_trs_solve
    for case in g["cases"]:
        rp, ci, v = matgen.dense_to_csr(g[case["matrix"]])
        b = np.array(case["b"], np.float64)
        x = np.zeros_like(b)
        fn(len(b), b.shape[1], rp, ci, v, int(case["unit"]), b, b.shape[1], x, b.shape[1])
        assert matgen.rel_err(x, case["expect"]) <= case["tol"], case["name"]


@pytest.mark.parametrize("case", G["add_diagonal"], ids=lambda c: c["name"])
def test_add_diagonal_elements(oracle, case):
    rp = np.array(case["row_ptrs"], np.int32)
    ci = np.array(case["col_idxs"] or [0], np.int32)
    v = np.array(case["vals"] or [0.0])
    nc, nv = np.zeros(20, np.int32), np.zeros(20)
    nnz = oracle.ref_add_diagonal_elements(case["nrows"], case["ncols"], rp, ci, v, nc, nv)
    assert list(rp) == case["expect_row_ptrs"]
    assert list(nc[:nnz]) == case["expect_col_idxs"] and list(nv[:nnz]) == case["expect_vals"]


@pytest.mark.parametrize("case", G["par_ilu"]["cases"], ids=lambda c: c["name"])
def test_par_ilu_known_factors(oracle, case):
    a = np.array(case["A"], np.float64)
    n = a.shape[0]
    rp, ci, v = matgen.dense_to_csr(a)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v, iterations=0)
    L = ilu_util.csr_to_dense(n, n, *f["L"])
    U = ilu_util.csr_to_dense(n, n, *f["U"])
    assert matgen.rel_err(L, case["L"]) <= case["tol"]
    assert matgen.rel_err(U, case["U"]) <= case["tol"]


def test_transpose_roundtrip(oracle):
    rp, ci, v = matgen.random_csr(37, 23, 0, 9, seed=2)
    trp, tc, tv = np.zeros(24, np.int32), np.zeros(len(ci), np.int32), np.zeros(len(ci))
    oracle.ref_csr_transpose(37, 23, rp, ci, v, trp, tc, tv)
    a = ilu_util.csr_to_dense(37, 23, rp, ci, v)
    assert np.array_equal(ilu_util.csr_to_dense(23, 37, trp, tc, tv), a.T)
    for r in range(23):
        assert np.all(np.diff(tc[trp[r]:trp[r + 1]]) > 0)


def test_par_ilu_kernel_known_answers(oracle):
    """reference/test/factorization/par_ilu_kernels.cpp:306-446: initialize_row_ptrs_l_u, initialize_l_u and ONE
    sweep of compute_l_u_factors on mtx_small (sequential sweep = exact factors, as the reference test expects)"""
    k = G["par_ilu"]["kernels"]
    a = np.array(k["A"], np.float64)
    n = 3
    rp, ci, v = matgen.dense_to_csr(a)
    lrp, urp = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    oracle.ref_initialize_row_ptrs_l_u(n, rp, ci, lrp, urp)
    assert list(lrp) == k["l_row_ptrs"] and list(urp) == k["u_row_ptrs"]
    lc, lv, uc, uv = np.zeros(6, np.int32), np.zeros(6), np.zeros(6, np.int32), np.zeros(6)
    oracle.ref_initialize_l_u(n, rp, ci, v, lrp, lc, lv, urp, uc, uv)
    assert np.array_equal(ilu_util.csr_to_dense(n, n, lrp, lc, lv), np.array(k["L_init"], np.float64))
    assert np.array_equal(ilu_util.csr_to_dense(n, n, urp, uc, uv), np.array(k["U_init"], np.float64))
    utrp, utc, utv = np.zeros(n + 1, np.int32), np.zeros(6, np.int32), np.zeros(6)
    oracle.ref_csr_transpose(n, n, urp, uc, uv, utrp, utc, utv)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    oracle.ref_par_ilu_compute_l_u_factors(1, len(ci), rows, ci, v, lrp, lc, lv, utrp, utc, utv)
    assert matgen.rel_err(ilu_util.csr_to_dense(n, n, lrp, lc, lv), k["L_after_one_sweep"]) <= k["tol"]
    assert matgen.rel_err(ilu_util.csr_to_dense(n, n, utrp, utc, utv).T, k["U_after_one_sweep"]) <= k["tol"]
    # the zero matrix: add_diagonal_elements first (par_ilu_kernels.cpp:330-354), then identity patterns
    z = k["zero_matrix"]
    zrp, nc, nv = np.zeros(n + 1, np.int32), np.zeros(n, np.int32), np.ones(n)
    oracle.ref_add_diagonal_elements(n, n, zrp, np.zeros(1, np.int32), np.zeros(1), nc, nv)
    assert list(zrp) == [0, 1, 2, 3] and list(nc) == [0, 1, 2] and not nv.any()
    oracle.ref_initialize_row_ptrs_l_u(n, zrp, nc, lrp, urp)
    assert list(lrp) == z["l_row_ptrs"] and list(urp) == z["u_row_ptrs"]
    # KernelInitializeLUZeroMatrix (:391-406): the EMPTY matrix itself (no stored diagonal -> "set it to 1 by
    # default", factorization_kernels.cpp:218-219) into identity-shaped factors
    lc, lv, uc, uv = np.zeros(3, np.int32), np.zeros(3), np.zeros(3, np.int32), np.zeros(3)
    oracle.ref_initialize_l_u(n, np.zeros(n + 1, np.int32), np.zeros(1, np.int32), np.zeros(1), lrp, lc, lv, urp, uc, uv)
    assert np.array_equal(ilu_util.csr_to_dense(n, n, lrp, lc, lv), np.eye(n))
    assert np.array_equal(ilu_util.csr_to_dense(n, n, urp, uc, uv), np.eye(n))
