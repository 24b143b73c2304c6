"""The hot path on the two real matrices the reference ships as test data
(matrices/test/ani4.mtx: 3081 x 3081 anisotropic FEM, 20971 entries, the matrix
of the reference's ParILU / ParIC kernel tests, test/factorization/
par_ilu_kernels.cpp:76; matrices/test/1138_bus.mtx: SuiteSparse HB/1138_bus,
symmetric positive definite, 4054 entries).  The files under tests/golden/ are
copies of those DATA files.  Every format's apply, the ParILU chain, the
triangular solves on its (irregular) factors and preconditioned solves, against
the oracle."""
import os

import numpy as np
import pytest
import torch

import ilu_util
import matgen
from gpu_util import dev, host

import gkomi.solvers as solvers
from gkomi import formats

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    kind, n, m, rows, cols, vals = matgen.read_mtx(os.path.join(G, name))
    assert kind == "coo" and n == m
    rp, ci, v = matgen.coo_to_csr(n, rows, cols, vals)
    return n, rp, ci, v


@pytest.mark.parametrize("name", ["ani4.mtx", "1138_bus.mtx"])
@pytest.mark.parametrize("nrhs", [1, 3])
def test_every_format_applies_like_the_oracle(gk, oracle, name, nrhs):
    n, rp, ci, v = load(name)
    rng = np.random.default_rng(5)
    b = rng.standard_normal((n, nrhs))
    expect = np.zeros((n, nrhs))
    oracle.ref_csr_spmv(n, nrhs, rp, ci, v, b, nrhs, expect, nrhs)
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    bd = dev(b)
    for fmt in formats.FORMATS:
        got = host(A.to(fmt).apply(bd, torch.full((n, nrhs), float("nan"), dtype=torch.float64, device="cuda:0")))
        # r<double>: the automatic CSR strategy may pick a tree-ordered kernel for
        # long rows, COO / Hybrid add row segments with atomics
        assert matgen.rel_err(got, expect) <= 1e-14, fmt
    # the file read on the device (assembly path) gives the same matrix
    B = formats.read_matrix(gk, os.path.join(G, name))
    assert B.nnz == len(v) and np.array_equal(host(B.row_ptrs), rp) and np.array_equal(host(B.col_idxs), ci)
    assert np.array_equal(host(B.vals), v)


def test_par_ilu_on_ani4_like_the_reference_test(gk, oracle):
    """test/factorization/par_ilu_kernels.cpp: the device chain against the
    reference chain on ani4 -- index arrays bit-exact, converged factors equal
    to the sequential sweep, the default sweep count within the reference's
    5e-2."""
    n, rp, ci, v = load("ani4.mtx")
    e = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    # par_ilu_kernels.cpp:302-307: 200 sweeps reproduce the sequential factors to r<double>
    f = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=200)
    for key in ("A", "L", "U"):
        assert np.array_equal(host(f[key][0]), e[key][0]), key
        assert np.array_equal(host(f[key][1]), e[key][1]), key
    assert matgen.rel_err(host(f["L"][2]), e["L"][2]) <= 1e-13
    assert matgen.rel_err(host(f["U"][2]), e["U"][2]) <= 1e-13
    f0 = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=0)
    assert matgen.rel_err(host(f0["L"][2]), e["L"][2]) <= 5e-2
    assert matgen.rel_err(host(f0["U"][2]), e["U"][2]) <= 5e-2


def test_triangular_solves_on_irregular_factors_bitexact(gk, oracle):
    """The ILU(0) factors of ani4 have no grid structure: dependencies at
    irregular distances, inside and across the 512-row chunks of the solve."""
    n, rp, ci, v = load("ani4.mtx")
    e = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    rng = np.random.default_rng(2)
    b = rng.standard_normal((n, 2))
    nb = gk.trs_workspace_bytes()
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    for which, (frp, fci, fv), unit in (("lower", e["L"], 1), ("lower", e["L"], 0), ("upper", e["U"], 0)):
        expect = np.zeros((n, 2))
        getattr(oracle, f"ref_{which}_trs_solve")(n, 2, frp, fci, fv, unit, b, 2, expect, 2)
        x = torch.zeros((n, 2), dtype=torch.float64, device="cuda:0")
        getattr(gk, f"{which}_trs_solve_f64_i32")(s, n, 2, dev(frp), dev(fci), dev(fv), unit, dev(b), 2, x, 2, ws, nb)
        assert np.array_equal(host(x), expect), (which, unit)


def test_preconditioned_solves_on_the_real_matrices(gk, oracle):
    # ani4: nonsymmetric storage (general), GMRES / BiCGSTAB with ParILU
    n, rp, ci, v = load("ani4.mtx")
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.cos(0.01 * np.arange(n))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    bd = dev(b[:, 0].copy())
    ilu = solvers.par_ilu_generate(gk, n, rpd, cid, vd, iterations=5)
    plain = solvers.gmres_solve(gk, n, rpd, cid, vd, bd, krylov_dim=30, max_iters=5000, reduction=1e-10)
    pre = solvers.gmres_solve(gk, n, rpd, cid, vd, bd, krylov_dim=30, max_iters=5000, reduction=1e-10, precond=ilu)
    assert pre["converged"] and pre["iterations"] < plain["iterations"]
    assert matgen.rel_err(host(pre["x"]).reshape(-1), xs) < 1e-6
    bi = solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bd, max_iters=5000, reduction=1e-10, precond=ilu, fused=True)
    assert bi["converged"] and matgen.rel_err(host(bi["x"]), xs) < 1e-6
    r = b[:, 0] - np.add.reduceat(v * host(bi["x"])[ci], rp[:-1])
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b)
    # 1138_bus: SPD and badly conditioned (kappa ~ 1e7): CG with block-Jacobi and IC
    n, rp, ci, v = load("1138_bus.mtx")
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    xs = np.ones(n)
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xs.reshape(n, 1), 1, b, 1)
    bd = dev(b)
    none = solvers.cg_solve(gk, n, rpd, cid, vd, bd, max_iters=20000, reduction=1e-10)
    xe = np.zeros(n)
    ite = oracle.ref_cg_solve(n, rp, ci, v, b[:, 0].copy(), xe, 20000, 1e-10, 0, None, 0)
    # thousands of iterations on kappa ~ 1e7: the counts agree to a few percent
    assert none["converged"] and abs(none["iterations"] - ite) <= max(5, ite // 10), (none["iterations"], ite)
    jac = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=8)
    pj = solvers.cg_solve(gk, n, rpd, cid, vd, bd, max_iters=20000, reduction=1e-10, precond=jac)
    ic = solvers.par_ic_generate(gk, n, rpd, cid, vd, iterations=5)
    pi = solvers.cg_solve(gk, n, rpd, cid, vd, bd, max_iters=20000, reduction=1e-10, precond=ic)
    assert pj["converged"] and pi["converged"]
    assert pi["iterations"] < pj["iterations"] < none["iterations"]
    for res in (none, pj, pi):
        assert matgen.rel_err(host(res["x"]).reshape(-1), xs) < 1e-4   # kappa * 1e-10
