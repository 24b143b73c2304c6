"""Pins the oracle's BiCGSTAB / FCG / CGS kernels and drivers against the
known answers of reference/test/solver/{bicgstab,fcg,cgs}_kernels.cpp
(tests/golden/krylov.json)."""
import json
import os

import numpy as np
import pytest

import matgen
from krylov_util import KERNEL_ARGS, dense_to_csr

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "krylov.json")))


@pytest.mark.parametrize("case", G["kernels"], ids=lambda c: c["solver"] + "_" + c["name"])
def test_kernel_known_answers(oracle, case):
    vecs, scalars = KERNEL_ARGS[(case["solver"], case["op"])]
    data = {k: np.array(case[k], np.float64) for k in vecs + scalars}
    stop = np.array(case["stop"], np.uint8)
    args = [2, 2]
    for k in vecs:
        args += [data[k], 2]
    args += [data[k] for k in scalars] + [stop]
    getattr(oracle, f"ref_{case['solver']}_{case['op']}")(*args)
    for k, e in case["expect"].items():
        got = stop if k == "stop" else data[k]
        assert np.array_equal(got, np.array(e, got.dtype)), (case["name"], k)


def test_initialize_kernels(oracle):
    # Kernel Initialize tests: r = b (and t / r_tld), every other vector 0,
    # scalars 1 except rho = 0 for fcg / cgs; statuses reset
    b = np.full((2, 2), 2.0)
    mk = lambda: np.full((2, 2), 7.0)
    sc = lambda: np.full(2, 5.0)
    st = np.full(2, 1, np.uint8)
    r, rr, y, s, t, z, v, p = (mk() for _ in range(8))
    prev_rho, rho, alpha, beta, gamma, omega = (sc() for _ in range(6))
    oracle.ref_bicgstab_initialize(2, 2, b, 2, r, 2, rr, 2, y, 2, s, 2, t, 2, z, 2, v, 2, p, 2, prev_rho, rho, alpha,
                                   beta, gamma, omega, st)
    assert np.array_equal(r, b) and not any(a.any() for a in (rr, y, s, t, z, v, p))
    assert all(np.array_equal(a, np.ones(2)) for a in (prev_rho, rho, alpha, beta, gamma, omega)) and not st.any()
    st[:] = 1
    r, z, p, q, t = (mk() for _ in range(5))
    prev_rho, rho, rho_t = (sc() for _ in range(3))
    oracle.ref_fcg_initialize(2, 2, b, 2, r, 2, z, 2, p, 2, q, 2, t, 2, prev_rho, rho, rho_t, st)
    assert np.array_equal(r, b) and np.array_equal(t, b) and not any(a.any() for a in (z, p, q))
    assert np.array_equal(prev_rho, np.ones(2)) and np.array_equal(rho_t, np.ones(2)) and not rho.any() and not st.any()
    st[:] = 1
    r, r_tld, p, q, u, u_hat, v_hat, t = (mk() for _ in range(8))
    alpha, beta, gamma, prev_rho, rho = (sc() for _ in range(5))
    oracle.ref_cgs_initialize(2, 2, b, 2, r, 2, r_tld, 2, p, 2, q, 2, u, 2, u_hat, 2, v_hat, 2, t, 2, alpha, beta, gamma,
                              prev_rho, rho, st)
    assert np.array_equal(r, b) and np.array_equal(r_tld, b) and not any(a.any() for a in (p, q, u, u_hat, v_hat, t))
    assert all(np.array_equal(a, np.ones(2)) for a in (alpha, beta, gamma, prev_rho)) and not rho.any() and not st.any()


@pytest.mark.parametrize("case", G["solves"], ids=lambda c: c["solver"] + "_" + c["name"])
def test_solve_known_answers(oracle, case):
    n, rp, ci, v = dense_to_csr(case["A"])
    x = np.zeros(n)
    if case["solver"] == "ir":
        it = oracle.ref_ir_solve(n, rp, ci, v, case["relaxation_factor"], np.array(case["b"]), x, case["max_iters"],
                                 case["reduction"], 0)
    else:
        it = getattr(oracle, f"ref_{case['solver']}_solve")(n, rp, ci, v, np.array(case["b"]), x, case["max_iters"],
                                                           case["reduction"], 0)
    assert it <= case["max_iters"]
    assert matgen.rel_err(x, case["expect_x"]) <= case["tol"], (it, x)


@pytest.mark.parametrize("solver", ["bicgstab", "fcg", "cgs", "bicg"])
def test_solves_poisson(oracle, solver):
    n, rp, ci, v = matgen.poisson_2d_5pt(24)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros(n)
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b.reshape(n, 1), 1)
    x = np.zeros(n)
    it = getattr(oracle, f"ref_{solver}_solve")(n, rp, ci, v, b, x, 1000, 1e-12, 0)
    assert 0 < it < 200 and matgen.rel_err(x, xs) < 1e-9
