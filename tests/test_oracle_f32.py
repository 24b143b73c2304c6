"""Pins the single-precision oracle (oracle/f32.c) against the reference's known-answer vectors: the reference's tests
are TYPED_TESTs over float and double with the same numbers (reference/test/matrix/csr_kernels.cpp:358-452,
dense_kernels.cpp:318-706, solver/cg_kernels.cpp:153-266, :447-477), so the double fixtures of tests/golden/ serve the
float instantiation too, with r<float> = 10 * eps(float) where the reference asserts a tolerance
(core/test/utils.hpp:212-220)."""
import json
import os

import numpy as np
import pytest

import matgen
from test_oracle_golden import _strided, load

F = np.float32
R_FLOAT = 10 * float(np.finfo(np.float32).eps)


@pytest.mark.parametrize("case", load("csr_spmv.json")["cases"], ids=lambda c: c["name"])
def test_csr_spmv_known_answers_f32(oracle, case):
    m = load("csr_spmv.json")["matrices"][case["matrix"]]
    rp, ci, v = np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32), np.array(m["vals"], F)
    b = np.array(case["b"], F)
    nrhs = b.shape[1]
    if "alpha" in case:
        c = np.array(case["c"], F)
        oracle.ref_csr_advanced_spmv_f32(m["nrows"], nrhs, case["alpha"], rp, ci, v, b, nrhs, case["beta"], c, nrhs)
    else:
        c = np.full((m["nrows"], nrhs), np.nan, F)
        oracle.ref_csr_spmv_f32(m["nrows"], nrhs, rp, ci, v, b, nrhs, c, nrhs)
    assert np.array_equal(c, np.array(case["expect"], F))     # EXPECT_EQ in the reference: exact


@pytest.mark.parametrize("case", [c for c in load("dense_blas1.json")["cases"] if c["op"] in
                                  ("scale", "inv_scale", "add_scaled", "sub_scaled", "fill", "dot", "norm2")], ids=lambda c: c["name"])
def test_dense_known_answers_f32(oracle, case):
    op = case["op"]
    expect = np.array(case["expect"], F)
    stride = case.get("stride")
    S = lambda k: _strided(case[k], stride).astype(F)
    nr, nc = np.array(case["x"]).shape
    if op in ("scale", "inv_scale"):
        x, alpha = S("x"), np.array(case["alpha"], F)
        getattr(oracle, f"ref_dense_{op}_f32")(nr, nc, alpha, len(alpha), x, x.shape[1])
        out = x
    elif op in ("add_scaled", "sub_scaled"):
        x, y, alpha = S("x"), S("y"), np.array(case["alpha"], F)
        getattr(oracle, f"ref_dense_{op}_f32")(nr, nc, alpha, len(alpha), x, x.shape[1], y, y.shape[1])
        out = y
    elif op == "fill":
        x = S("x")
        oracle.ref_dense_fill_f32(nr, nc, x, x.shape[1], case["value"])
        out = x
    else:
        x = S("x")
        res = np.full((1, nc), np.nan, F)
        if op == "dot":
            y = S("y")
            oracle.ref_dense_compute_dot_f32(nr, nc, x, x.shape[1], y, y.shape[1], res)
        else:
            oracle.ref_dense_compute_norm2_f32(nr, nc, x, x.shape[1], res)
        assert matgen.rel_err(res.astype(np.float64), expect.astype(np.float64)) <= R_FLOAT
        return
    assert matgen.rel_err(out[:, :expect.shape[1]].astype(np.float64), expect.astype(np.float64)) <= R_FLOAT
    if out.shape[1] > expect.shape[1]:
        assert np.all(out[:, expect.shape[1]:] == -1.0)


@pytest.mark.parametrize("case", load("cg.json")["kernel_cases"], ids=lambda c: c["name"])
def test_cg_kernel_known_answers_f32(oracle, case):
    A = lambda k: np.array(case[k], F)
    stop = np.array(case.get("stop", [0, 0]), np.uint8)
    if case["op"] == "step_1":
        p, z = A("p"), A("z")
        oracle.ref_cg_step_1_f32(2, 2, p, 2, z, 2, A("rho"), A("prev_rho"), stop)
        assert matgen.rel_err(p.astype(np.float64), A("expect_p").astype(np.float64)) <= R_FLOAT
    elif case["op"] == "step_2":
        x, r = A("x"), A("r")
        oracle.ref_cg_step_2_f32(2, 2, x, 2, r, 2, A("p"), 2, A("q"), 2, A("beta"), A("rho"), stop)
        assert matgen.rel_err(x.astype(np.float64), A("expect_x").astype(np.float64)) <= R_FLOAT
        assert matgen.rel_err(r.astype(np.float64), A("expect_r").astype(np.float64)) <= R_FLOAT
    else:
        b = _strided(case["b"], case["b_stride"]).astype(F)
        r = np.zeros((2, 2), F); z = np.ones((2, 2), F); p = np.ones((2, 2), F); q = np.ones((2, 2), F)
        prev_rho = np.zeros(2, F); rho = np.ones(2, F)
        stop = np.array([1, 1], np.uint8)
        oracle.ref_cg_initialize_f32(2, 2, b, b.shape[1], r, 2, z, 2, p, 2, q, 2, prev_rho, rho, stop)
        assert np.array_equal(r, A("expect_r")) and not z.any() and not p.any() and not q.any()
        assert np.array_equal(rho, A("expect_rho")) and np.array_equal(prev_rho, A("expect_prev_rho")) and not stop.any()


def test_cg_solves_stencil_system_f32(oracle):
    """Cg<float> SolvesStencilSystem (reference/test/solver/cg_kernels.cpp:255-266): tolerance r<float>"""
    case = load("cg.json")["solve_cases"][0]
    assert case["name"] == "SolvesStencilSystem"
    rp, ci, v = matgen.dense_to_csr(case["A"])
    b, x = np.array(case["b"], F), np.array(case["x0"], F)
    iters = oracle.ref_cg_solve_f32(len(b), rp, ci, v.astype(F), b, x, case["max_iters"], R_FLOAT, 0)
    assert iters < case["max_iters"]
    assert matgen.rel_err(x.astype(np.float64), case["expect_x"]) <= R_FLOAT
