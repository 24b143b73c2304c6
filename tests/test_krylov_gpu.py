"""GPU parity (through the C ABI) of BiCGSTAB / FCG / CGS: the step kernels
bit-exact against the oracle (known answers of the reference's tests, then
random data with stopped columns and zero denominators), the drivers against
the reference's solve answers and the oracle's iteration counts."""
import json
import os

import numpy as np
import pytest
import torch

import matgen
from gkomi import solvers
from gpu_util import dev, host, stream_ptr
from krylov_util import KERNEL_ARGS, dense_to_csr

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "krylov.json")))


def run_kernel(gk, oracle, solver, op, n, nrhs, data, stop, strides=None):
    """runs ref_<solver>_<op> on copies and gkomi_<solver>_<op>_f64 on the device; returns both result dicts"""
    vecs, scalars = KERNEL_ARGS[(solver, op)]
    strides = strides or {k: nrhs for k in vecs}
    e = {k: v.copy() for k, v in data.items()}
    es = stop.copy()
    args = [n, nrhs]
    for k in vecs:
        args += [e[k], strides[k]]
    args += [e[k] for k in scalars] + [es]
    getattr(oracle, f"ref_{solver}_{op}")(*args)
    d = {k: dev(v) for k, v in data.items()}
    ds = dev(stop)
    args = [stream_ptr(), n, nrhs]
    for k in vecs:
        args += [d[k], strides[k]]
    args += [d[k] for k in scalars] + [ds]
    getattr(gk, f"{solver}_{op}_f64")(*args)
    got = {k: host(v) for k, v in d.items()}
    got["stop"], e["stop"] = host(ds), es
    return got, e


@pytest.mark.parametrize("case", G["kernels"], ids=lambda c: c["solver"] + "_" + c["name"])
def test_kernel_known_answers(gk, oracle, case):
    vecs, scalars = KERNEL_ARGS[(case["solver"], case["op"])]
    data = {k: np.array(case[k], np.float64) for k in vecs + scalars}
    got, _ = run_kernel(gk, oracle, case["solver"], case["op"], 2, 2, data, np.array(case["stop"], np.uint8))
    for k, exp in case["expect"].items():
        assert np.array_equal(got[k], np.array(exp, got[k].dtype)), (case["name"], k)


@pytest.mark.parametrize("key", sorted(KERNEL_ARGS), ids=lambda k: "_".join(k))
@pytest.mark.parametrize("n,nrhs,pad", [(1, 1, 0), (1000, 1, 0), (777, 3, 2), (100003, 2, 0)])
def test_kernels_bitexact_random(gk, oracle, key, n, nrhs, pad):
    solver, op = key
    vecs, scalars = KERNEL_ARGS[key]
    rng = np.random.default_rng(hash(key) % 1000 + n)
    stride = nrhs + pad
    data = {k: rng.standard_normal((n, stride)) for k in vecs}
    for k in scalars:
        data[k] = rng.standard_normal(nrhs)
        if nrhs > 1:
            data[k][rng.integers(0, nrhs)] = 0.0      # a zero denominator somewhere
    stop = np.zeros(nrhs, np.uint8)
    if nrhs > 2:
        stop[1] = 1                                    # a stopped column
    if op == "finalize":
        stop = np.array(([1, 65, 0] * nrhs)[:nrhs], np.uint8)
    got, exp = run_kernel(gk, oracle, solver, op, n, nrhs, data, stop, {k: stride for k in vecs})
    for k in exp:
        assert got[k].tobytes() == exp[k].tobytes(), (key, k)


def test_initialize_kernels(gk):
    n, nrhs = 1000, 3
    b = dev(np.random.default_rng(0).standard_normal((n, nrhs)))
    mk = lambda: torch.full((n, nrhs), 7.0, dtype=torch.float64, device="cuda:0")
    sc = lambda: torch.full((nrhs,), 5.0, dtype=torch.float64, device="cuda:0")
    st = torch.ones(nrhs, dtype=torch.uint8, device="cuda:0")
    r, rr, y, s, t, z, v, p = (mk() for _ in range(8))
    sca = [sc() for _ in range(6)]
    gk.bicgstab_initialize_f64(stream_ptr(), n, nrhs, b, nrhs, r, nrhs, rr, nrhs, y, nrhs, s, nrhs, t, nrhs, z, nrhs, v,
                               nrhs, p, nrhs, *sca, st)
    assert torch.equal(r, b) and not any(bool(a.any()) for a in (rr, y, s, t, z, v, p))
    assert all(bool((a == 1).all()) for a in sca) and not bool(st.any())
    st.fill_(1)
    r, z, p, q, t = (mk() for _ in range(5))
    prev_rho, rho, rho_t = sc(), sc(), sc()
    gk.fcg_initialize_f64(stream_ptr(), n, nrhs, b, nrhs, r, nrhs, z, nrhs, p, nrhs, q, nrhs, t, nrhs, prev_rho, rho,
                          rho_t, st)
    assert torch.equal(r, b) and torch.equal(t, b) and not any(bool(a.any()) for a in (z, p, q))
    assert bool((prev_rho == 1).all()) and bool((rho_t == 1).all()) and not bool(rho.any()) and not bool(st.any())
    st.fill_(1)
    r, r_tld, p, q, u, u_hat, v_hat, t = (mk() for _ in range(8))
    alpha, beta, gamma, prev_rho, rho = (sc() for _ in range(5))
    gk.cgs_initialize_f64(stream_ptr(), n, nrhs, b, nrhs, r, nrhs, r_tld, nrhs, p, nrhs, q, nrhs, u, nrhs, u_hat, nrhs,
                          v_hat, nrhs, t, nrhs, alpha, beta, gamma, prev_rho, rho, st)
    assert torch.equal(r, b) and torch.equal(r_tld, b) and not any(bool(a.any()) for a in (p, q, u, u_hat, v_hat, t))
    assert all(bool((a == 1).all()) for a in (alpha, beta, gamma, prev_rho)) and not bool(rho.any())


@pytest.mark.parametrize("case", G["solves"], ids=lambda c: c["solver"] + "_" + c["name"])
def test_solve_known_answers(gk, oracle, case):
    n, rp, ci, v = dense_to_csr(case["A"])
    if case["solver"] == "ir":
        res = solvers.ir_solve(gk, n, dev(rp), dev(ci), dev(v), dev(np.array(case["b"])),
                               relaxation_factor=case["relaxation_factor"], max_iters=case["max_iters"],
                               reduction=case["reduction"])
        assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= 4 * case["tol"], res
        xe = np.zeros(n)
        ite = oracle.ref_ir_solve(n, rp, ci, v, case["relaxation_factor"], np.array(case["b"]), xe, case["max_iters"],
                                  case["reduction"], 0)
        assert abs(res["iterations"] - ite) <= 1
        return
    solve = (lambda *a, **k: solvers.bicg_solve(gk, *a[1:], **k)) if case["solver"] == "bicg" else \
        (lambda *a, **k: solvers.krylov_solve(gk, *a, **k))
    res = solve(case["solver"], n, dev(rp), dev(ci), dev(v), dev(np.array(case["b"])),
                max_iters=case["max_iters"], reduction=case["reduction"])
    # the reference's tolerance holds for its sequential dots; the device sums in
    # a two-stage order (deterministic, but different): allow 4x
    assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= 4 * case["tol"], res
    xe = np.zeros(n)
    ite = getattr(oracle, f"ref_{case['solver']}_solve")(n, rp, ci, v, np.array(case["b"]), xe, case["max_iters"],
                                                        case["reduction"], 0)
    # near machine precision the last iterations are rounding noise (the
    # "DivergenceCheck" systems are badly conditioned on purpose)
    assert abs(res["iterations"] - ite) <= max(2, ite // 4)


@pytest.mark.parametrize("solver", ["bicgstab", "fcg", "cgs"])
@pytest.mark.parametrize("problem", ["poisson", "convection"])
def test_solves_like_the_oracle(gk, oracle, solver, problem):
    if problem == "poisson":
        n, rp, ci, v = matgen.poisson_2d_5pt(40)
    else:
        if solver == "fcg":
            pytest.skip("FCG needs a symmetric matrix")
        n, rp, ci, v = matgen.poisson_3d_7pt(12)
        v = v.copy()
        rows = np.repeat(np.arange(n), np.diff(rp))
        v[ci == rows - 1] -= 0.5
        v[ci == rows] += 0.5
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    xe = np.zeros(n)
    ite = getattr(oracle, f"ref_{solver}_solve")(n, rp, ci, v, b[:, 0].copy(), xe, 2000, 1e-10, 0)
    res = solvers.krylov_solve(gk, solver, n, dev(rp), dev(ci), dev(v), dev(b[:, 0].copy()), max_iters=2000,
                               reduction=1e-10)
    assert res["converged"] and res["rel_residual"] <= 1e-10
    # the criterion lives on the device: how often the host looks changes nothing
    for every in (1, 3, 50):
        again = solvers.krylov_solve(gk, solver, n, dev(rp), dev(ci), dev(v), dev(b[:, 0].copy()), max_iters=2000,
                                     reduction=1e-10, check_every=every)
        assert again["iterations"] == res["iterations"] and again["converged"]
        assert host(again["x"]).tobytes() == host(res["x"]).tobytes()
        assert again["residual_norm"][0] == res["residual_norm"][0]
    # reductions are summed in a different order: the iterates drift by rounding
    assert abs(res["iterations"] - ite) <= max(2, ite // 10), (res["iterations"], ite)
    assert matgen.rel_err(host(res["x"]), xs) < 1e-7
    # the reported residual is the true norm of the final residual vector
    r = b[:, 0] - (np.add.reduceat(v * host(res["x"])[ci], rp[:-1]))
    assert abs(np.linalg.norm(r) - res["residual_norm"][0]) <= 1e-6 * np.linalg.norm(b) + 1e-3 * res["residual_norm"][0]


def test_multiple_rhs_preconditioner_and_iteration_limit(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(32)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    rng = np.random.default_rng(4)
    xs = rng.standard_normal((n, 3))
    b = np.zeros((n, 3))
    oracle.ref_csr_spmv(n, 3, rp, ci, v, xs, 3, b, 3)
    pc = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=4, nrhs=3)
    for solver in ("bicgstab", "fcg", "cgs"):
        plain = solvers.krylov_solve(gk, solver, n, rpd, cid, vd, dev(b), max_iters=2000, reduction=1e-11)
        assert plain["converged"] and matgen.rel_err(host(plain["x"]), xs) < 1e-8
        pre = solvers.krylov_solve(gk, solver, n, rpd, cid, vd, dev(b), max_iters=2000, reduction=1e-11, precond=pc)
        assert pre["converged"] and matgen.rel_err(host(pre["x"]), xs) < 1e-8
        capped = solvers.krylov_solve(gk, solver, n, rpd, cid, vd, dev(b), max_iters=3, reduction=1e-11)
        assert capped["iterations"] == 3 and not capped["converged"]


def test_bicg_like_the_oracle_and_with_jacobi(gk, oracle):
    n, rp, ci, v = matgen.poisson_3d_7pt(12)
    v = v.copy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows - 1] -= 0.5
    v[ci == rows] += 0.5
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    xe = np.zeros(n)
    ite = oracle.ref_bicg_solve(n, rp, ci, v, b[:, 0].copy(), xe, 2000, 1e-10, 0)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    res = solvers.bicg_solve(gk, n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10)
    assert res["converged"] and abs(res["iterations"] - ite) <= max(2, ite // 10)
    assert matgen.rel_err(host(res["x"]), xs) < 1e-7
    again = solvers.bicg_solve(gk, n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10, check_every=1)
    assert again["iterations"] == res["iterations"] and host(again["x"]).tobytes() == host(res["x"]).tobytes()
    pc = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=1)     # scalar Jacobi = its own transpose
    pre = solvers.bicg_solve(gk, n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10, precond=pc,
                             precond_t=pc)
    assert pre["converged"] and matgen.rel_err(host(pre["x"]), xs) < 1e-7


def test_ir_with_inner_preconditioner_and_richardson(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(24)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.cos(0.2 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    # Richardson with relaxation 1/4 (< 2 / lambda_max = 1/4 for the 5-point stencil) vs the oracle
    xe = np.zeros(n)
    ite = oracle.ref_ir_solve(n, rp, ci, v, 0.24, b[:, 0].copy(), xe, 300, 1e-3, 0)
    res = solvers.ir_solve(gk, n, rpd, cid, vd, dev(b[:, 0].copy()), relaxation_factor=0.24, max_iters=300, reduction=1e-3)
    assert res["iterations"] == ite and matgen.rel_err(host(res["x"]), xe) < 1e-12
    # block-Jacobi as inner "solver": converges in fewer sweeps than scalar Richardson
    pc = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=8)
    inner = solvers.ir_solve(gk, n, rpd, cid, vd, dev(b[:, 0].copy()), relaxation_factor=0.9, inner=pc, max_iters=300,
                             reduction=1e-3)
    assert inner["converged"] and inner["iterations"] < res["iterations"] or not res["converged"]
    r = b[:, 0] - np.add.reduceat(v * host(inner["x"])[ci], rp[:-1])
    assert np.linalg.norm(r) <= 1.01e-3 * np.linalg.norm(b)


@pytest.mark.parametrize("solver", ["cg", "gmres", "bicgstab", "fcg", "cgs"])
def test_solvers_on_every_matrix_format(gk, oracle, solver):
    """config 4 of BASELINE.json solves on ELL / SELL-P: the *_solve_op_f64 drivers
    take the system matrix in any format.  ELL and SELL-P SpMV are bit-identical to
    CSR's, so the iterates are too; COO / Hybrid sum with atomics (rounding)."""
    from gkomi import formats
    n, rp, ci, v = matgen.poisson_2d_5pt(40)
    if solver in ("gmres", "bicgstab", "cgs"):
        v = v.copy()
        rows = np.repeat(np.arange(n), np.diff(rp))
        v[ci == rows - 1] -= 0.3
        v[ci == rows] += 0.3
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    xs = np.sin(0.3 * np.arange(n))
    b = A.apply(dev(xs.reshape(n, 1)), torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")).reshape(n)
    pc = solvers.jacobi_generate(gk, n, A.row_ptrs, A.col_idxs, A.vals, max_block_size=4)
    kw = dict(max_iters=2000, reduction=1e-10, precond=pc)
    base = solvers.solve_op(gk, solver, A, b, **kw)
    assert base["converged"] and matgen.rel_err(host(base["x"]), xs) < 1e-7
    # the CSR entry point and the operator entry point run the same loop
    if solver == "cg":
        direct = solvers.cg_solve(gk, n, A.row_ptrs, A.col_idxs, A.vals, b, mode=0, **kw)
    elif solver == "gmres":
        direct = solvers.gmres_solve(gk, n, A.row_ptrs, A.col_idxs, A.vals, b, **kw)
    else:
        direct = solvers.krylov_solve(gk, solver, n, A.row_ptrs, A.col_idxs, A.vals, b, **kw)
    assert direct["iterations"] == base["iterations"] and host(direct["x"]).tobytes() == host(base["x"]).tobytes()
    for fmt in ("ell", "sellp", "coo", "hybrid"):
        M = A.to(fmt, kind=0, num_columns=3) if fmt == "hybrid" else A.to(fmt)  # column_limit(3): some COO part
        res = solvers.solve_op(gk, solver, M, b, **kw)
        assert res["converged"] and matgen.rel_err(host(res["x"]), xs) < 1e-7, fmt
        if fmt in ("ell", "sellp"):
            assert res["iterations"] == base["iterations"] and host(res["x"]).tobytes() == host(base["x"]).tobytes()
        else:
            assert abs(res["iterations"] - base["iterations"]) <= max(2, base["iterations"] // 10)


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "fcg", "cgs"])
@pytest.mark.parametrize("precond", [False, True])
def test_fused_drivers_on_every_matrix_format(gk, solver, precond):
    """The fused single-rhs drivers with the system matrix behind a callback: ELL
    and SELL-P carry the dot-product epilogue in their SpMV (same row -> thread map
    and block sums as the CSR kernel), so the iterates are bit-identical to the
    fused CSR entry point's; CSR behind its callback is the CSR entry point; COO /
    Hybrid take apply + a partials kernel (different partial sums: rounding).
    A Csr that carries its srow ("csr_split": what the C++ mirror's Csr and formats.Csr pass by
    default) runs the nonzero-split kernel with the dot epilogue: one partial per tile instead of
    one per 256 rows, so the iterates agree to rounding, not bit for bit."""
    from gkomi import formats
    n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)   # n = 1517: not a multiple of the block
    if solver in ("bicgstab", "cgs"):
        v = v.copy()
        rows = np.repeat(np.arange(n), np.diff(rp))
        v[ci == rows - 1] -= 0.3
        v[ci == rows] += 0.3
    A = formats.Csr.from_host(gk, n, n, rp, ci, v, split=False)
    S = formats.Csr.from_host(gk, n, n, rp, ci, v)
    assert S.srow() is not None and A.srow() is None
    xs = np.sin(0.3 * np.arange(n))
    b = A.apply(dev(xs.reshape(n, 1)), torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")).reshape(n)
    assert host(S.apply(dev(xs.reshape(n, 1)), torch.zeros((n, 1), dtype=torch.float64, device="cuda:0"))).tobytes() == \
        host(b).tobytes()
    pc = solvers.jacobi_generate(gk, n, A.row_ptrs, A.col_idxs, A.vals, max_block_size=4) if precond else None
    kw = dict(max_iters=2000, reduction=1e-10, precond=pc, check_every=4)
    if solver == "cg":
        base = solvers.cg_solve(gk, n, A.row_ptrs, A.col_idxs, A.vals, b, mode=1, **kw)
    else:
        base = solvers.krylov_solve(gk, solver, n, A.row_ptrs, A.col_idxs, A.vals, b, fused=True, **kw)
    assert base["converged"] and matgen.rel_err(host(base["x"]), xs) < 1e-7
    for fmt in ("csr", "csr_split", "ell", "sellp", "coo", "hybrid"):
        M = A if fmt == "csr" else S if fmt == "csr_split" else (
            A.to(fmt, kind=0, num_columns=3) if fmt == "hybrid" else A.to(fmt))
        res = solvers.solve_op(gk, solver, M, b, fused=True, **kw)
        assert res["converged"] and matgen.rel_err(host(res["x"]), xs) < 1e-7, fmt
        if fmt in ("csr", "ell", "sellp"):
            assert res["iterations"] == base["iterations"], fmt
            assert host(res["x"]).tobytes() == host(base["x"]).tobytes(), fmt
        else:
            assert abs(res["iterations"] - base["iterations"]) <= max(2, base["iterations"] // 10), fmt
            if fmt == "csr_split":
                assert matgen.rel_err(host(res["x"]), host(base["x"])) < 1e-8


# ---- fused single-rhs BiCGSTAB (6 launches per iteration) ---------------------------
def _convection(n3=12):
    n, rp, ci, v = matgen.poisson_3d_7pt(n3)
    v = v.copy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows - 1] -= 0.5
    v[ci == rows] += 0.5
    return n, rp, ci, v


@pytest.mark.parametrize("case", [c for c in G["solves"] if c["solver"] == "bicgstab"], ids=lambda c: c["name"])
def test_fused_bicgstab_known_answers(gk, oracle, case):
    n, rp, ci, v = dense_to_csr(case["A"])
    res = solvers.krylov_solve(gk, "bicgstab", n, dev(rp), dev(ci), dev(v), dev(np.array(case["b"])),
                               max_iters=case["max_iters"], reduction=case["reduction"], fused=True)
    assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= 4 * case["tol"], res
    xe = np.zeros(n)
    ite = oracle.ref_bicgstab_solve(n, rp, ci, v, np.array(case["b"]), xe, case["max_iters"], case["reduction"], 0)
    if "DivergenceCheck" in case["name"]:
        # badly conditioned on purpose: near machine precision the iteration
        # count is rounding noise; the answer above is what the reference checks
        assert res["iterations"] <= 2 * ite + 2
    else:
        assert abs(res["iterations"] - ite) <= max(2, ite // 4)


@pytest.mark.parametrize("problem", ["poisson", "convection", "odd_size"])
def test_fused_bicgstab_like_the_oracle_and_the_reference_sequence(gk, oracle, problem):
    if problem == "poisson":
        n, rp, ci, v = matgen.poisson_2d_5pt(40)
    elif problem == "convection":
        n, rp, ci, v = _convection()
    else:
        n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)   # n = 1517: odd, not a multiple of any block
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    xe = np.zeros(n)
    ite = oracle.ref_bicgstab_solve(n, rp, ci, v, b[:, 0].copy(), xe, 2000, 1e-10, 0)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    res = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10,
                               fused=True)
    assert res["converged"] and res["rel_residual"] <= 1e-10
    # BiCGSTAB's residual is erratic: a different (fixed) summation order of the
    # dots moves the iteration at which 1e-10 is crossed by a few
    assert abs(res["iterations"] - ite) <= max(3, ite // 4), (res["iterations"], ite)
    assert matgen.rel_err(host(res["x"]), xs) < 1e-7
    unfused = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10)
    assert abs(res["iterations"] - unfused["iterations"]) <= max(3, ite // 4)
    assert matgen.rel_err(host(res["x"]), host(unfused["x"])) < 1e-8
    for every in (1, 3, 50):
        again = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000,
                                     reduction=1e-10, check_every=every, fused=True)
        assert again["iterations"] == res["iterations"] and again["converged"]
        assert host(again["x"]).tobytes() == host(res["x"]).tobytes()
        assert again["residual_norm"][0] == res["residual_norm"][0]
    # the reported norm belongs to the residual of the returned x
    r = b[:, 0] - np.add.reduceat(v * host(res["x"])[ci], rp[:-1])
    assert abs(np.linalg.norm(r) - res["residual_norm"][0]) <= 1e-6 * np.linalg.norm(b) + 1e-3 * res["residual_norm"][0]


def test_fused_bicgstab_stops_in_the_half_step_and_at_the_iteration_limit(gk, oracle):
    # a multiple of the identity converges in the first half step: s = r - alpha v = 0,
    # x += alpha y is the finalize branch (core/solver/bicgstab.cpp:196-203)
    n = 5000
    rp = np.arange(n + 1, dtype=np.int32)
    ci = np.arange(n, dtype=np.int32)
    v = np.full(n, 4.0)
    b = np.cos(0.01 * np.arange(n))
    for fused in (False, True):
        res = solvers.krylov_solve(gk, "bicgstab", n, dev(rp), dev(ci), dev(v), dev(b), max_iters=50, reduction=1e-12,
                                   fused=fused)
        assert res["converged"] and res["iterations"] == 0, res
        assert matgen.rel_err(host(res["x"]), b / 4.0) < 1e-15
    # iteration limit: x after exactly 3 iterations equals the unfused x to rounding
    n, rp, ci, v = _convection(10)
    b = np.sin(0.1 * np.arange(n))
    a = solvers.krylov_solve(gk, "bicgstab", n, dev(rp), dev(ci), dev(v), dev(b), max_iters=3, reduction=1e-14, fused=True)
    u = solvers.krylov_solve(gk, "bicgstab", n, dev(rp), dev(ci), dev(v), dev(b), max_iters=3, reduction=1e-14)
    assert a["iterations"] == 3 and not a["converged"] and u["iterations"] == 3
    assert matgen.rel_err(host(a["x"]), host(u["x"])) < 1e-12
    # the reported norm is that of the residual of the returned x, also at the iteration limit
    r3 = b - np.add.reduceat(v * host(a["x"])[ci], rp[:-1])
    assert abs(np.linalg.norm(r3) - a["residual_norm"][0]) <= 1e-9 * np.linalg.norm(b)
    assert abs(u["residual_norm"][0] - a["residual_norm"][0]) <= 1e-9 * np.linalg.norm(b)
    # zero iterations allowed: x stays the initial guess
    z = solvers.krylov_solve(gk, "bicgstab", n, dev(rp), dev(ci), dev(v), dev(b), max_iters=0, reduction=1e-14, fused=True)
    assert z["iterations"] == 0 and not z["converged"] and not host(z["x"]).any()


def test_fused_bicgstab_with_preconditioners_and_formats(gk, oracle):
    from gkomi import formats
    n, rp, ci, v = _convection()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    bd = dev(b[:, 0].copy())
    plain = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, fused=True)
    jac = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=8)
    ilu = solvers.par_ilu_generate(gk, n, rpd, cid, vd, iterations=5)
    for pc in (jac, ilu):
        pre = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=pc,
                                   fused=True)
        ref = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=pc)
        assert pre["converged"] and matgen.rel_err(host(pre["x"]), xs) < 1e-7
        assert abs(pre["iterations"] - ref["iterations"]) <= max(2, ref["iterations"] // 5)
        assert pre["iterations"] <= plain["iterations"]
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    for fmt in ("ell", "sellp", "coo", "hybrid"):
        res = solvers.solve_op(gk, "bicgstab", A.to(fmt), bd, max_iters=2000, reduction=1e-10, fused=True)
        assert res["converged"] and matgen.rel_err(host(res["x"]), xs) < 1e-7, fmt
        assert abs(res["iterations"] - plain["iterations"]) <= max(2, plain["iterations"] // 5), fmt


# ---- fused single-rhs FCG (3 launches per iteration) ----------------------------------
@pytest.mark.parametrize("case", [c for c in G["solves"] if c["solver"] == "fcg"], ids=lambda c: c["name"])
def test_fused_fcg_known_answers(gk, oracle, case):
    n, rp, ci, v = dense_to_csr(case["A"])
    res = solvers.krylov_solve(gk, "fcg", n, dev(rp), dev(ci), dev(v), dev(np.array(case["b"])),
                               max_iters=case["max_iters"], reduction=case["reduction"], fused=True)
    assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= 4 * case["tol"], res
    xe = np.zeros(n)
    ite = oracle.ref_fcg_solve(n, rp, ci, v, np.array(case["b"]), xe, case["max_iters"], case["reduction"], 0)
    assert res["iterations"] <= 2 * ite + 2 if "DivergenceCheck" in case["name"] else abs(res["iterations"] - ite) <= max(2, ite // 4)


@pytest.mark.parametrize("problem", ["poisson", "odd_size", "nonzero_guess"])
def test_fused_fcg_like_the_oracle_and_the_reference_sequence(gk, oracle, problem):
    n, rp, ci, v = matgen.poisson_2d_5pt(40) if problem != "odd_size" else matgen.poisson_2d_5pt(37, 41)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    x0 = np.cos(0.1 * np.arange(n)) if problem == "nonzero_guess" else np.zeros(n)
    xe = x0.copy()
    ite = oracle.ref_fcg_solve(n, rp, ci, v, b[:, 0].copy(), xe, 2000, 1e-10, 0)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    res = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, dev(b[:, 0].copy()), x=dev(x0.copy()), max_iters=2000,
                               reduction=1e-10, fused=True)
    assert res["converged"] and res["rel_residual"] <= 1e-10
    assert abs(res["iterations"] - ite) <= max(2, ite // 10), (res["iterations"], ite)
    assert matgen.rel_err(host(res["x"]), xs) < 1e-7
    unfused = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, dev(b[:, 0].copy()), x=dev(x0.copy()), max_iters=2000,
                                   reduction=1e-10)
    assert abs(res["iterations"] - unfused["iterations"]) <= max(2, ite // 10)
    for every in (1, 3, 50):
        again = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, dev(b[:, 0].copy()), x=dev(x0.copy()), max_iters=2000,
                                     reduction=1e-10, check_every=every, fused=True)
        assert again["iterations"] == res["iterations"] and again["converged"]
        assert host(again["x"]).tobytes() == host(res["x"]).tobytes()
        assert again["residual_norm"][0] == res["residual_norm"][0]
    r = b[:, 0] - np.add.reduceat(v * host(res["x"])[ci], rp[:-1])
    assert abs(np.linalg.norm(r) - res["residual_norm"][0]) <= 1e-6 * np.linalg.norm(b) + 1e-3 * res["residual_norm"][0]


def test_fused_fcg_with_preconditioner_formats_and_iteration_limit(gk, oracle):
    from gkomi import formats
    n, rp, ci, v = matgen.poisson_2d_5pt(48)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    bd = dev(b[:, 0].copy())
    plain = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, fused=True)
    jac = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=8)
    pre = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=jac, fused=True)
    ref = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=jac)
    assert pre["converged"] and matgen.rel_err(host(pre["x"]), xs) < 1e-7
    assert abs(pre["iterations"] - ref["iterations"]) <= max(2, ref["iterations"] // 10)
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    for fmt in ("ell", "sellp", "coo", "hybrid"):
        res = solvers.solve_op(gk, "fcg", A.to(fmt), bd, max_iters=2000, reduction=1e-10, fused=True)
        assert res["converged"] and matgen.rel_err(host(res["x"]), xs) < 1e-7, fmt
        assert abs(res["iterations"] - plain["iterations"]) <= max(2, plain["iterations"] // 10), fmt
    a = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bd, max_iters=3, reduction=1e-14, fused=True)
    u = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bd, max_iters=3, reduction=1e-14)
    assert a["iterations"] == 3 and not a["converged"] and matgen.rel_err(host(a["x"]), host(u["x"])) < 1e-12
    r3 = b[:, 0] - np.add.reduceat(v * host(a["x"])[ci], rp[:-1])
    assert abs(np.linalg.norm(r3) - a["residual_norm"][0]) <= 1e-9 * np.linalg.norm(b)
    assert abs(u["residual_norm"][0] - a["residual_norm"][0]) <= 1e-9 * np.linalg.norm(b)
    z = solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bd, max_iters=0, reduction=1e-14, fused=True)
    assert z["iterations"] == 0 and not z["converged"] and not host(z["x"]).any()


# ---- fused single-rhs CGS (5 launches per iteration) ----------------------------------
@pytest.mark.parametrize("case", [c for c in G["solves"] if c["solver"] == "cgs"], ids=lambda c: c["name"])
def test_fused_cgs_known_answers(gk, oracle, case):
    n, rp, ci, v = dense_to_csr(case["A"])
    res = solvers.krylov_solve(gk, "cgs", n, dev(rp), dev(ci), dev(v), dev(np.array(case["b"])),
                               max_iters=case["max_iters"], reduction=case["reduction"], fused=True)
    assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= 4 * case["tol"], res
    xe = np.zeros(n)
    ite = oracle.ref_cgs_solve(n, rp, ci, v, np.array(case["b"]), xe, case["max_iters"], case["reduction"], 0)
    assert res["iterations"] <= 2 * ite + 2 if "DivergenceCheck" in case["name"] else abs(res["iterations"] - ite) <= max(2, ite // 4)


@pytest.mark.parametrize("problem", ["poisson", "convection", "odd_size"])
def test_fused_cgs_like_the_oracle_and_the_reference_sequence(gk, oracle, problem):
    if problem == "poisson":
        n, rp, ci, v = matgen.poisson_2d_5pt(40)
    elif problem == "convection":
        n, rp, ci, v = _convection()
    else:
        n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    xe = np.zeros(n)
    ite = oracle.ref_cgs_solve(n, rp, ci, v, b[:, 0].copy(), xe, 2000, 1e-10, 0)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    res = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10, fused=True)
    assert res["converged"] and res["rel_residual"] <= 1e-10
    assert abs(res["iterations"] - ite) <= max(3, ite // 4), (res["iterations"], ite)
    assert matgen.rel_err(host(res["x"]), xs) < 1e-6
    unfused = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10)
    assert abs(res["iterations"] - unfused["iterations"]) <= max(3, ite // 4)
    for every in (1, 3, 50):
        again = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, dev(b[:, 0].copy()), max_iters=2000, reduction=1e-10,
                                     check_every=every, fused=True)
        assert again["iterations"] == res["iterations"] and again["converged"]
        assert host(again["x"]).tobytes() == host(res["x"]).tobytes()
        assert again["residual_norm"][0] == res["residual_norm"][0]


def test_fused_cgs_with_preconditioner_formats_and_iteration_limit(gk, oracle):
    from gkomi import formats
    n, rp, ci, v = _convection()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    bd = dev(b[:, 0].copy())
    plain = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, fused=True)
    for pc in (solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=8),
               solvers.par_ilu_generate(gk, n, rpd, cid, vd, iterations=5)):
        pre = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=pc, fused=True)
        ref = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=pc)
        assert pre["converged"] and matgen.rel_err(host(pre["x"]), xs) < 1e-6
        assert abs(pre["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 4)
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    for fmt in ("ell", "sellp", "coo", "hybrid"):
        res = solvers.solve_op(gk, "cgs", A.to(fmt), bd, max_iters=2000, reduction=1e-10, fused=True)
        assert res["converged"] and matgen.rel_err(host(res["x"]), xs) < 1e-6, fmt
        assert abs(res["iterations"] - plain["iterations"]) <= max(3, plain["iterations"] // 4), fmt
    a = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bd, max_iters=3, reduction=1e-14, fused=True)
    u = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bd, max_iters=3, reduction=1e-14)
    assert a["iterations"] == 3 and not a["converged"] and matgen.rel_err(host(a["x"]), host(u["x"])) < 1e-12
    r3 = b[:, 0] - np.add.reduceat(v * host(a["x"])[ci], rp[:-1])
    assert abs(np.linalg.norm(r3) - a["residual_norm"][0]) <= 1e-9 * np.linalg.norm(b)
    assert abs(u["residual_norm"][0] - a["residual_norm"][0]) <= 1e-9 * np.linalg.norm(b)
    z = solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bd, max_iters=0, reduction=1e-14, fused=True)
    assert z["iterations"] == 0 and not z["converged"] and not host(z["x"]).any()


def test_bicg_with_block_jacobi_and_its_transpose(gk, oracle):
    """Bicg applies M^-1 to r and M^-T to r2 (core/solver/bicg.cpp:169-171); for
    block-Jacobi the transposed preconditioner is Jacobi::transpose =
    gkomi_jacobi_transpose_f64_i32.  With it BiCG converges in fewer iterations
    than without a preconditioner and to the same solution."""
    n, rp, ci, v = _convection()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.sin(0.3 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    bd = dev(b[:, 0].copy())
    plain = solvers.bicg_solve(gk, n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10)
    jac = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=12)   # one grid line per block: nonsymmetric blocks
    jac_t = solvers.jacobi_transpose(gk, jac)
    assert not torch.equal(jac.blocks, jac_t.blocks)
    pre = solvers.bicg_solve(gk, n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=jac, precond_t=jac_t)
    assert pre["converged"] and pre["iterations"] < plain["iterations"]
    assert matgen.rel_err(host(pre["x"]), xs) < 1e-7
    # adaptive storage transposes too
    ad = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=12, storage_optimization=solvers.AUTODETECT)
    ad_t = solvers.jacobi_transpose(gk, ad)
    pa = solvers.bicg_solve(gk, n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, precond=ad, precond_t=ad_t)
    assert pa["converged"] and matgen.rel_err(host(pa["x"]), xs) < 1e-7
    assert abs(pa["iterations"] - pre["iterations"]) <= max(3, pre["iterations"] // 4)


@pytest.mark.parametrize("solver", ["bicgstab", "fcg", "cgs"])
@pytest.mark.parametrize("fused", [False, True], ids=["sequence", "fused"])
def test_empty_and_one_row_systems(gk, solver, fused):
    """Degenerate sizes must not trip the launch geometry: n = 0 (norms are 0, the
    criterion 0 < 0 never fires: the iteration limit stops it) and n = 1."""
    rp = dev(np.zeros(1, np.int32))
    empty_i = torch.zeros(0, dtype=torch.int32, device="cuda:0")
    empty_d = torch.zeros(0, dtype=torch.float64, device="cuda:0")
    res = solvers.krylov_solve(gk, solver, 0, rp, empty_i, empty_d, torch.zeros((0, 1), dtype=torch.float64, device="cuda:0"),
                               max_iters=2, reduction=1e-10, fused=fused)
    assert res["iterations"] == 2 and not res["converged"]
    one = solvers.krylov_solve(gk, solver, 1, dev(np.array([0, 1], np.int32)), dev(np.array([0], np.int32)), dev(np.array([4.0])),
                               dev(np.array([2.0])), max_iters=5, reduction=1e-12, fused=fused)
    assert one["converged"] and abs(float(host(one["x"])[0]) - 0.5) < 1e-15


def test_cg_and_gmres_on_empty_and_one_row_systems(gk):
    rp = dev(np.zeros(1, np.int32))
    empty_i = torch.zeros(0, dtype=torch.int32, device="cuda:0")
    empty_d = torch.zeros(0, dtype=torch.float64, device="cuda:0")
    b0 = torch.zeros((0, 1), dtype=torch.float64, device="cuda:0")
    for mode in (0, 1):
        res = solvers.cg_solve(gk, 0, rp, empty_i, empty_d, b0, max_iters=2, reduction=1e-10, mode=mode)
        assert res["iterations"] == 2 and not res["converged"]
        one = solvers.cg_solve(gk, 1, dev(np.array([0, 1], np.int32)), dev(np.array([0], np.int32)), dev(np.array([4.0])),
                               dev(np.array([[2.0]])), max_iters=5, reduction=1e-12, mode=mode)
        assert one["converged"] and abs(float(host(one["x"]).reshape(-1)[0]) - 0.5) < 1e-15
    res = solvers.gmres_solve(gk, 0, rp, empty_i, empty_d, b0, krylov_dim=3, max_iters=2, reduction=1e-10)
    assert res["iterations"] == 2 and not res["converged"]
    one = solvers.gmres_solve(gk, 1, dev(np.array([0, 1], np.int32)), dev(np.array([0], np.int32)), dev(np.array([4.0])),
                              dev(np.array([[2.0]])), krylov_dim=3, max_iters=5, reduction=1e-12)
    assert one["converged"] and abs(float(host(one["x"]).reshape(-1)[0]) - 0.5) < 1e-14
