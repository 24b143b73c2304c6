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
