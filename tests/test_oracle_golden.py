"""Pins the CPU oracle (oracle/*.c) against the reference's own known-answer
tests, transcribed as data into tests/golden/ (each file names its source)."""
import json
import os

import numpy as np
import pytest

import matgen

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = np.finfo(np.float64).eps


def load(name):
    return json.load(open(os.path.join(G, name)))


def csr_of(m):
    return (np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32),
            np.array(m["vals"], np.float64))


@pytest.mark.parametrize("case", load("csr_spmv.json")["cases"], ids=lambda c: c["name"])
def test_csr_spmv_known_answers(oracle, case):
    g = load("csr_spmv.json")
    m = g["matrices"][case["matrix"]]
    rp, ci, v = csr_of(m)
    b = np.array(case["b"], np.float64)
    nrhs = b.shape[1]
    if "alpha" in case:
        c = np.array(case["c"], np.float64)
        oracle.ref_csr_advanced_spmv(m["nrows"], nrhs, case["alpha"], rp, ci, v, b, nrhs,
                                     case["beta"], c, nrhs)
        c2 = np.array(case["c"], np.float64)
        oracle.omp_csr_advanced_spmv(m["nrows"], nrhs, case["alpha"], rp, ci, v, b, nrhs,
                                     case["beta"], c2, nrhs)
    else:
        c = np.full((m["nrows"], nrhs), np.nan)
        oracle.ref_csr_spmv(m["nrows"], nrhs, rp, ci, v, b, nrhs, c, nrhs)
        c2 = np.full((m["nrows"], nrhs), np.nan)
        oracle.omp_csr_spmv(m["nrows"], nrhs, rp, ci, v, b, nrhs, c2, nrhs)
    # the reference asserts EXPECT_EQ: exact
    assert np.array_equal(c, np.array(case["expect"]))
    assert np.array_equal(c2, np.array(case["expect"]))


def _strided(mat, stride, pad=-1.0):
    mat = np.array(mat, np.float64)
    stride = stride or mat.shape[1]
    buf = np.full((mat.shape[0], stride), pad)
    buf[:, :mat.shape[1]] = mat
    return buf


@pytest.mark.parametrize("case", load("dense_blas1.json")["cases"], ids=lambda c: c["name"])
def test_dense_known_answers(oracle, case):
    op = case["op"]
    expect = np.array(case["expect"], np.float64)
    stride = case.get("stride")
    if op in ("scale", "inv_scale"):
        x = _strided(case["x"], stride)
        nr, nc = np.array(case["x"]).shape
        alpha = np.array(case["alpha"], np.float64)
        getattr(oracle, "ref_dense_" + op)(nr, nc, alpha, len(alpha), x, x.shape[1])
        out = x
    elif op in ("add_scaled", "sub_scaled"):
        y = _strided(case["y"], stride)
        x = _strided(case["x"], stride)
        nr, nc = np.array(case["x"]).shape
        alpha = np.array(case["alpha"], np.float64)
        getattr(oracle, "ref_dense_" + op)(nr, nc, alpha, len(alpha), x, x.shape[1], y, y.shape[1])
        out = y
    elif op == "fill":
        x = _strided(case["x"], stride)
        nr, nc = np.array(case["x"]).shape
        oracle.ref_dense_fill(nr, nc, x, x.shape[1], case["value"])
        out = x
    elif op == "sqrt":
        x = _strided(case["x"], stride)
        nr, nc = np.array(case["x"]).shape
        oracle.ref_dense_compute_sqrt(nr, nc, x, x.shape[1])
        out = x
    else:
        x = _strided(case["x"], stride)
        nr, nc = np.array(case["x"]).shape
        res = np.full((1, nc), np.nan)
        if op == "dot":
            y = _strided(case["y"], stride)
            oracle.ref_dense_compute_dot(nr, nc, x, x.shape[1], y, y.shape[1], res)
        elif op == "norm2":
            oracle.ref_dense_compute_norm2(nr, nc, x, x.shape[1], res)
        elif op == "squared_norm2":
            oracle.ref_dense_compute_squared_norm2(nr, nc, x, x.shape[1], res)
        elif op == "norm1":
            oracle.ref_dense_compute_norm1(nr, nc, x, x.shape[1], res)
        assert np.array_equal(res, expect)
        return
    nc = expect.shape[1]
    assert np.array_equal(out[:, :nc], expect)
    if out.shape[1] > nc:  # padding untouched (ASSERT_EQ(get_values()[3], in_stride))
        assert np.all(out[:, nc:] == -1.0)


@pytest.mark.parametrize("case", load("cg.json")["kernel_cases"], ids=lambda c: c["name"])
def test_cg_kernel_known_answers(oracle, case):
    A = lambda k: np.array(case[k], np.float64)
    stop = np.array(case.get("stop", [0, 0]), np.uint8)
    if case["op"] == "step_1":
        p, z = A("p"), A("z")
        oracle.ref_cg_step_1(2, 2, p, 2, z, 2, A("rho"), A("prev_rho"), stop)
        assert np.array_equal(p, A("expect_p"))
    elif case["op"] == "step_2":
        x, r = A("x"), A("r")
        oracle.ref_cg_step_2(2, 2, x, 2, r, 2, A("p"), 2, A("q"), 2, A("beta"), A("rho"), stop)
        assert np.array_equal(x, A("expect_x"))
        assert np.array_equal(r, A("expect_r"))
    else:
        b = _strided(case["b"], case["b_stride"])
        r = np.zeros((2, 2)); z = np.ones((2, 2)); p = np.ones((2, 2)); q = np.ones((2, 2))
        prev_rho = np.zeros(2); rho = np.ones(2)
        stop = np.array([1, 1], np.uint8)
        oracle.ref_cg_initialize(2, 2, b, b.shape[1], r, 2, z, 2, p, 2, q, 2, prev_rho, rho, stop)
        assert np.array_equal(r, A("expect_r"))
        assert not z.any() and not p.any() and not q.any()
        assert np.array_equal(rho, A("expect_rho")) and np.array_equal(prev_rho, A("expect_prev_rho"))
        assert not stop.any()


@pytest.mark.parametrize("case", load("cg.json")["solve_cases"], ids=lambda c: c["name"])
def test_cg_solve_known_answers(oracle, case):
    rp, ci, v = matgen.dense_to_csr(case["A"])
    b = np.array(case["b"], np.float64)
    x = np.array(case["x0"], np.float64)
    iters = oracle.ref_cg_solve(len(b), rp, ci, v, b, x, case["max_iters"], case["reduction"], 0, None, 0)
    assert iters < case["max_iters"]
    assert matgen.rel_err(x, case["expect_x"]) <= case["tol"]


def test_simple_solver_example_matches_documented_output(oracle):
    """examples/simple-solver: CG on the 19x19 Trefethen matrix reproduces
    doc/results.dox (6 significant digits as printed)."""
    g = load("cg.json")["simple_solver"]
    kind, n, _, rows, cols, vals = matgen.read_mtx(os.path.join(G, "simple_solver_A.mtx"))
    rp, ci, v = matgen.coo_to_csr(n, rows, cols, vals)
    b = matgen.read_mtx(os.path.join(G, "simple_solver_b.mtx"))[3][:, 0].copy()
    x = matgen.read_mtx(os.path.join(G, "simple_solver_x0.mtx"))[3][:, 0].copy()
    iters = oracle.ref_cg_solve(n, rp, ci, v, b, x, g["max_iters"], g["reduction"], 0, None, 0)
    assert iters <= g["max_iters"]
    printed = np.array([float(f"{t:.6g}") for t in x])
    assert np.array_equal(printed, np.array(g["expect_x"]))
    res = b.copy()
    oracle.ref_csr_advanced_spmv(n, 1, -1.0, rp, ci, v, x, 1, 1.0, res, 1)
    nrm = np.zeros(1)
    oracle.ref_dense_compute_norm2(n, 1, res, 1, nrm)
    assert float(f"{nrm[0]:.6g}") == g["expect_residual_norm"]


def test_stop_kernels(oracle):
    # reference/test/stop/residual_norm_kernels.cpp semantics: converge sets
    # bit7 + id (+ finalized bit6), all_converged / one_changed flags
    st = np.zeros(3, np.uint8)
    flags = np.zeros(2, np.uint8)
    oracle.ref_residual_norm(3, np.array([0.5, 2.0, 0.01]), np.array([1.0, 1.0, 1.0]), 1.0, 2, 1, st, flags)
    assert list(st) == [0x80 | 0x40 | 2, 0, 0x80 | 0x40 | 2] and list(flags) == [0, 1]
    oracle.ref_set_all_statuses(3, 5, 0, st)
    assert list(st) == [0xC2, 5, 0xC2]
    oracle.ref_implicit_residual_norm(3, np.array([0.25, 0.25, 0.25]), np.ones(3), 0.6, 3, 0, st, flags)
    assert list(st) == [0xC2, 5, 0xC2] and list(flags) == [1, 1]


def _dense_to_csr(dense):
    a = np.array(dense, np.float64)
    rp = np.zeros(a.shape[0] + 1, np.int32)
    np.cumsum((a != 0).sum(1), out=rp[1:])
    ci = np.concatenate([np.nonzero(r)[0] for r in a]).astype(np.int32)
    return a.shape, rp, ci, a[a != 0]


def test_csr_utility_known_answers(oracle):
    """csr::transpose, is_sorted_by_column_index, sort_by_column_index, extract_diagonal of the oracle against the
    reference's own small cases (reference/test/matrix/csr_kernels.cpp:1044-1076, 1299-1344)"""
    u = load("formats.json")["csr_utilities"]
    for t in u["transposes"]:
        (nr, nc), rp, ci, v = _dense_to_csr(t["dense"])
        trp, tc, tv = np.zeros(nc + 1, np.int32), np.zeros(len(ci), np.int32), np.zeros(len(ci))
        oracle.ref_csr_transpose(nr, nc, rp, ci, v, trp, tc, tv)
        (_, _), erp, ec, ev = _dense_to_csr(t["expect"])
        assert np.array_equal(trp, erp) and np.array_equal(tc, ec) and np.array_equal(tv, ev), t["name"]
    s, un = u["mtx3_sorted"], u["mtx3_unsorted"]
    rp = np.array(s["row_ptrs"], np.int32)
    assert oracle.ref_csr_is_sorted_by_column_index(3, rp, np.array(s["col_idxs"], np.int32)) == 1
    assert oracle.ref_csr_is_sorted_by_column_index(3, rp, np.array(un["col_idxs"], np.int32)) == 0
    for m in (s, un):   # SortSortedMatrix / SortUnsortedMatrix
        c, v = np.array(m["col_idxs"], np.int32), np.array(m["vals"], np.float64)
        oracle.ref_csr_sort_by_column_index(3, rp, c, v)
        assert list(c) == s["col_idxs"] and list(v) == s["vals"]
    diag = np.full(3, -1.0)
    oracle.ref_csr_extract_diagonal(3, rp, np.array(un["col_idxs"], np.int32), np.array(un["vals"], np.float64), diag)
    assert list(diag) == u["mtx3_diagonal"]
