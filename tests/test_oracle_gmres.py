"""Pins the oracle's GMRES kernels and driver against
reference/test/solver/gmres_kernels.cpp (tests/golden/gmres.json)."""
import json
import os

import numpy as np
import pytest

import matgen

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmres.json")))
R = 2.220446049250313e-15


def arr(a):
    return np.array([[float(v) for v in row] for row in a], np.float64)


@pytest.mark.parametrize("case", G["hessenberg_qr"], ids=lambda c: c["name"])
def test_hessenberg_qr(oracle, case):
    cos, sin, rnc, hess = arr(case["cos"]), arr(case["sin"]), arr(case["rnc"]), arr(case["hess"])
    fin = np.array(case["final_iter_nums"], np.uint64)
    rn = np.full(2, np.nan)
    oracle.ref_gmres_hessenberg_qr(2, sin, cos, rn, rnc, hess, 2, case["iter"], fin, np.zeros(2, np.uint8))
    assert list(fin) == case["expect_final_iter_nums"]
    for got, key in ((cos, "expect_cos"), (sin, "expect_sin"), (hess, "expect_hess"), (rnc, "expect_rnc")):
        assert matgen.rel_err(got, arr(case[key])) <= R, key
    assert matgen.rel_err(rn, case["expect_residual_norm"]) <= R


def test_solve_krylov_multi_axpy_restart(oracle):
    g = G["solve_krylov"]
    y = np.full((2, 2), np.nan)
    oracle.ref_gmres_solve_krylov(2, arr(g["rnc"]), arr(g["hess"]), 4, y, np.array(g["final_iter_nums"], np.uint64),
                                  np.zeros(2, np.uint8))
    assert matgen.rel_err(y, g["expect_y"]) <= R
    g = G["multi_axpy"]
    x = np.full((3, 2), np.nan)
    st = np.array(g["stop_in"], np.uint8)
    oracle.ref_gmres_multi_axpy(3, 2, arr(g["krylov"]), 2, arr(g["y"]), x, 2, np.array(g["final_iter_nums"], np.uint64), st)
    assert matgen.rel_err(x, g["expect_x"]) <= R and list(st) == g["stop_out"]
    b = arr(G["restart"]["b"])
    nrm = np.sqrt((b * b).sum(axis=0))
    rnc = np.full((3, 2), np.nan)
    kb = np.full((9, 2), 9999.0)
    fin = np.full(2, 999, np.uint64)
    oracle.ref_gmres_restart(3, 2, b, 2, nrm, rnc, kb, 2, fin)
    assert list(fin) == [0, 0] and np.array_equal(rnc[0], nrm)
    assert matgen.rel_err(kb[:3], b / nrm) <= R and np.all(kb[3:] == 9999.0)


@pytest.mark.parametrize("case", G["solves"], ids=lambda c: c["name"])
def test_solves(oracle, case):
    rp, ci, v = matgen.dense_to_csr(case["A"])
    b = np.array(case["b"], np.float64)
    x = np.zeros_like(b)
    fr = np.zeros(1)
    it = oracle.ref_gmres_solve(len(b), rp, ci, v, None, None, b, x, case["krylov_dim"], case["max_iters"],
                                case["reduction"], 0, fr)
    assert it <= case["max_iters"]
    assert matgen.rel_err(x, case["expect_x"]) <= case["tol"]
