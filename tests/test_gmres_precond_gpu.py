"""GPU parity tests (C ABI): GMRES kernels and driver against the oracle and
the reference's known answers; CG and GMRES with the native block-Jacobi and
ILU preconditioner callbacks (configs 3 and 4 in miniature).  Solver
tolerance: north-star's 1e-6 relative, iteration counts within +-1 (+-2 with a
preconditioner) of the oracle's."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

import gkomi.solvers as solvers
import ilu_util
import matgen
from gpu_util import dev, host, stream_ptr
from test_oracle_gmres import arr

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gmres.json")))
R = 2.220446049250313e-15
U64 = torch.int64


@pytest.mark.parametrize("case", G["hessenberg_qr"], ids=lambda c: c["name"])
def test_hessenberg_qr_known_answers(gk, case):
    cos, sin, rnc, hess = (dev(arr(case[k])) for k in ("cos", "sin", "rnc", "hess"))
    fin = dev(np.array(case["final_iter_nums"], np.int64))
    rn = dev(np.full(2, np.nan))
    gk.gmres_hessenberg_qr_f64(stream_ptr(), 2, sin, cos, rn, rnc, hess, 2, case["iter"], fin,
                               dev(np.zeros(2, np.uint8)))
    assert list(host(fin)) == case["expect_final_iter_nums"]
    for got, key in ((cos, "expect_cos"), (sin, "expect_sin"), (hess, "expect_hess"), (rnc, "expect_rnc")):
        assert matgen.rel_err(host(got), arr(case[key])) <= R, key
    assert matgen.rel_err(host(rn), case["expect_residual_norm"]) <= R


def test_small_kernels_known_answers_and_bitexact(gk, oracle):
    s = stream_ptr()
    g = G["solve_krylov"]
    y = dev(np.full((2, 2), np.nan))
    gk.gmres_solve_krylov_f64(s, 2, dev(arr(g["rnc"])), dev(arr(g["hess"])), 4, y,
                              dev(np.array(g["final_iter_nums"], np.int64)), dev(np.zeros(2, np.uint8)))
    assert matgen.rel_err(host(y), g["expect_y"]) <= R
    g = G["multi_axpy"]
    x = dev(np.full((3, 2), np.nan))
    st = dev(np.array(g["stop_in"], np.uint8))
    gk.gmres_multi_axpy_f64(s, 3, 2, dev(arr(g["krylov"])), 2, dev(arr(g["y"])), x, 2,
                            dev(np.array(g["final_iter_nums"], np.int64)), st)
    assert matgen.rel_err(host(x), g["expect_x"]) <= R and list(host(st)) == g["stop_out"]
    # random sizes, bit-exact against the oracle (same loops per entry)
    rng = np.random.default_rng(0)
    n, k, d = 1000, 3, 7
    kb = rng.standard_normal(((d + 1) * n, k))
    yv = rng.standard_normal((d, k))
    fin = np.array([7, 3, 5], np.uint64)
    st0 = np.array([0, 2, 0x42], np.uint8)
    e = np.full((n, k), -5.0)
    ste = st0.copy()
    oracle.ref_gmres_multi_axpy(n, k, kb, k, yv, e, k, fin, ste)
    x = dev(np.full((n, k), -5.0))
    std = dev(st0)
    gk.gmres_multi_axpy_f64(s, n, k, dev(kb), k, dev(yv), x, k, dev(fin.astype(np.int64)), std)
    assert np.array_equal(host(x), e) and np.array_equal(host(std), ste)
    b = rng.standard_normal((n, k))
    nrm = np.sqrt((b * b).sum(axis=0))
    rnc = dev(np.full((d + 1, k), np.nan))
    kbd = dev(np.full(((d + 1) * n, k), 9999.0))
    find = dev(np.full(k, 999, np.int64))
    gk.gmres_restart_f64(s, n, k, dev(b), k, dev(nrm), rnc, kbd, k, find)
    ernc, ekb, efin = np.full((d + 1, k), np.nan), np.full(((d + 1) * n, k), 9999.0), np.full(k, 999, np.uint64)
    oracle.ref_gmres_restart(n, k, b, k, nrm, ernc, ekb, k, efin)
    assert np.array_equal(host(kbd), ekb) and np.array_equal(host(rnc)[0], ernc[0]) and not host(find).any()
    res = dev(np.zeros((n, k)))
    gs, gc = dev(np.ones((d, k))), dev(np.ones((d, k)))
    st = dev(np.full(k, 9, np.uint8))
    gk.gmres_initialize_f64(s, n, k, d, dev(b), k, res, k, gs, gc, st)
    assert np.array_equal(host(res), b) and not host(gs).any() and not host(gc).any() and not host(st).any()


@pytest.mark.parametrize("case", G["solves"], ids=lambda c: c["name"])
def test_known_answer_solves(gk, case):
    rp, ci, v = matgen.dense_to_csr(case["A"])
    b = np.array(case["b"], np.float64)
    res = solvers.gmres_solve(gk, len(b), dev(rp), dev(ci), dev(v), dev(b), krylov_dim=case["krylov_dim"],
                              max_iters=case["max_iters"], reduction=case["reduction"])
    assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= max(case["tol"], 1e-11)


def convection_diffusion_3d(g, upwind=0.5):
    """7-pt stencil with an upwind convection term: nonsymmetric (config 4's AT-like stand-in)."""
    n, rp, ci, v = matgen.poisson_3d_7pt(g)
    v = v.copy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows - 1] -= upwind
    v[ci == rows] += upwind
    return n, rp, ci, v


@pytest.mark.parametrize("krylov_dim", [5, 30])
def test_gmres_matches_oracle_nonsymmetric(gk, oracle, krylov_dim):
    n, rp, ci, v = convection_diffusion_3d(12)
    b = np.cos(0.3 * np.arange(n))
    xe = np.zeros(n)
    fr = np.zeros(1)
    it = oracle.ref_gmres_solve(n, rp, ci, v, None, None, b, xe, krylov_dim, 2000, 1e-10, 0, fr)
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=krylov_dim, max_iters=2000,
                              reduction=1e-10)
    assert res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-6
    # true residual
    r = b.copy().reshape(n, 1)
    oracle.ref_csr_advanced_spmv(n, 1, -1.0, rp, ci, v, host(res["x"]).reshape(n, 1), 1, 1.0, r, 1)
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("g3,krylov_dim", [(27, 30), (45, 10), (64, 30)], ids=["19683_odd", "91125_odd", "262144"])
def test_gmres_single_launch_arnoldi_matches_oracle(gk, oracle, g3, krylov_dim):
    """n >= 64 * #CU: the modified Gram-Schmidt sweep, the Givens update and the criterion of an
    iteration are ONE launch (gmres_arnoldi_persistent_kernel): same iteration count as the
    oracle's loop (+-1: the sums are grouped per workgroup), same solution, restarts included,
    odd n included."""
    n, rp, ci, v = convection_diffusion_3d(g3)
    b = np.cos(0.3 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_gmres_solve(n, rp, ci, v, None, None, b, xe, krylov_dim, 3000, 1e-10, 0, np.zeros(1))
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=krylov_dim, max_iters=3000,
                              reduction=1e-10)
    assert res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-6
    r = b.copy().reshape(n, 1)
    oracle.ref_csr_advanced_spmv(n, 1, -1.0, rp, ci, v, host(res["x"]).reshape(n, 1), 1, 1.0, r, 1)
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b)
    again = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=krylov_dim, max_iters=3000,
                                reduction=1e-10)
    assert again["iterations"] == res["iterations"] and host(again["x"]).tobytes() == host(res["x"]).tobytes()
    # iteration limit in the middle of a restart cycle
    cut = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=krylov_dim, max_iters=krylov_dim + 3,
                              reduction=1e-30)
    assert cut["iterations"] == krylov_dim + 3 and not cut["converged"]


def test_gmres_odd_number_of_rows(gk, oracle):
    # n odd: every second Krylov basis vector is only 8-byte aligned (the fused
    # Arnoldi kernels then run their 8-byte variant)
    n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)
    assert n % 2 == 1
    v = v.copy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows - 1] -= 0.4
    v[ci == rows] += 0.4
    b = np.cos(0.3 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_gmres_solve(n, rp, ci, v, None, None, b, xe, 20, 2000, 1e-10, 0, np.zeros(1))
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=20, max_iters=2000, reduction=1e-10)
    assert res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-6


def test_gmres_iteration_limit_and_multiple_rhs(gk, oracle):
    n, rp, ci, v = convection_diffusion_3d(8)
    b = np.ones(n)
    xe = np.zeros(n)
    it = oracle.ref_gmres_solve(n, rp, ci, v, None, None, b, xe, 4, 9, 1e-30, 0, np.zeros(1))
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=4, max_iters=9, reduction=1e-30)
    assert it == 9 and res["iterations"] == 9 and not res["converged"]
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-10
    # two right-hand sides at once equal two single solves (to rounding)
    b2 = np.stack([np.ones(n), np.cos(np.arange(n))], axis=1)
    r2 = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b2), krylov_dim=20, max_iters=500, reduction=1e-10)
    for j in range(2):
        r1 = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b2[:, j].copy()), krylov_dim=20, max_iters=500,
                                 reduction=1e-10)
        assert matgen.rel_err(host(r2["x"])[:, j], host(r1["x"])) <= 1e-6


def _oracle_ilu_callback(oracle, n, f):
    lrp, lc, lv = f["L"]
    urp, uc, uv = f["U"]
    FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))

    def cb(_, pin, pout):
        vin = np.ctypeslib.as_array(pin, shape=(n, 1)).copy()
        mid = np.zeros((n, 1))
        oracle.ref_lower_trs_solve(n, 1, lrp, lc, lv, 0, vin, 1, mid, 1)
        out = np.zeros((n, 1))
        oracle.ref_upper_trs_solve(n, 1, urp, uc, uv, 0, mid, 1, out, 1)
        np.ctypeslib.as_array(pout, shape=(n, 1))[:] = out
        return 0
    fn = FN(cb)
    return fn, ctypes.cast(fn, ctypes.c_void_p).value


def test_gmres_with_parilu_preconditioner(gk, oracle):
    """Config 4 in miniature: GMRES(30) + ParILU on a nonsymmetric 3-D stencil."""
    n, rp, ci, v = convection_diffusion_3d(14)
    b = np.cos(0.3 * np.arange(n))
    fe = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    keep, fnptr = _oracle_ilu_callback(oracle, n, fe)
    xe = np.zeros(n)
    it = oracle.ref_gmres_solve(n, rp, ci, v, fnptr, None, b, xe, 30, 500, 1e-10, 0, np.zeros(1))
    it_plain = oracle.ref_gmres_solve(n, rp, ci, v, None, None, b, np.zeros(n), 30, 500, 1e-10, 0, np.zeros(1))
    assert it < it_plain  # the preconditioner helps
    fg = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=40)
    pre = solvers.ilu_from_factors(gk, n, fg["L"], fg["U"])
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=30, max_iters=500, reduction=1e-10,
                              precond=pre)
    assert res["converged"] and abs(res["iterations"] - it) <= 2
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-6


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("max_bs", [1, 4, 32])
def test_cg_with_block_jacobi(gk, oracle, mode, max_bs):
    """Config 3 in miniature: CG + block-Jacobi on an SPD matrix whose banded
    locality a fixed symmetric permutation destroyed (T2-like stand-in)."""
    n, rp, ci, v = matgen.poisson_2d_5pt(48, 52)
    rng = np.random.default_rng(42)
    perm = rng.permutation(n)
    inv = np.argsort(perm)
    rows = np.repeat(np.arange(n), np.diff(rp))
    pr, pc = inv[rows], inv[ci]
    order = np.lexsort((pc, pr))
    rp2, ci2, v2 = matgen.coo_to_csr(n, pr[order].astype(np.int32), pc[order].astype(np.int32), v[order])
    # scale rows/cols so that Jacobi has something to do
    d = 1.0 + rng.random(n) * 9.0
    rows2 = np.repeat(np.arange(n), np.diff(rp2))
    v2 = v2 * np.sqrt(d[rows2] * d[ci2])
    b = np.ones(n)
    rpd, cid, vd = dev(rp2), dev(ci2), dev(v2)
    pre = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=max_bs)
    plain = solvers.cg_solve(gk, n, rpd, cid, vd, dev(b), max_iters=5000, reduction=1e-10, mode=mode)
    res = solvers.cg_solve(gk, n, rpd, cid, vd, dev(b), max_iters=5000, reduction=1e-10, mode=mode, precond=pre,
                           check_every=5)
    assert res["converged"] and plain["converged"]
    assert res["iterations"] < plain["iterations"]
    assert matgen.rel_err(host(res["x"]), host(plain["x"])) <= 1e-6
    r = b.copy().reshape(n, 1)
    oracle.ref_csr_advanced_spmv(n, 1, -1.0, rp2, ci2, v2, host(res["x"]).reshape(n, 1), 1, 1.0, r, 1)
    assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(b) * 10


def test_gmres_meeting_timeout_falls_back_before_touching_x(gk, oracle, monkeypatch):
    """when a meeting of the single-launch Arnoldi step times out, the launches after
    it return early and leave the Hessenberg column / Givens terms / stop record stale.  The driver
    now looks at the flag right after every poll -- before update_solution, before the stop and
    the max_iters exits -- and solves again, from the x of the last completed restart, with the
    launch-per-vector kernels.  GKOMI_MEET_MAX_POLLS=1 (test hook) makes every meeting time out."""
    n, rp, ci, v = convection_diffusion_3d(27)
    b = np.cos(0.3 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_gmres_solve(n, rp, ci, v, None, None, b, xe, 30, 3000, 1e-10, 0, np.zeros(1))
    healthy = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=30, max_iters=3000, reduction=1e-10)
    monkeypatch.setenv("GKOMI_MEET_MAX_POLLS", "1")
    before = gk.gmres_meeting_fallbacks()
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=30, max_iters=3000, reduction=1e-10)
    assert gk.gmres_meeting_fallbacks() == before + 1
    assert res["converged"] and abs(res["iterations"] - it) <= 1
    assert np.all(np.isfinite(host(res["x"]))) and matgen.rel_err(host(res["x"]), xe) <= 1e-6
    assert matgen.rel_err(host(res["x"]), host(healthy["x"])) <= 1e-9
    # the iteration limit inside the first restart cycle: no garbage in x, no false "converged"
    cut = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=30, max_iters=7, reduction=1e-30)
    assert gk.gmres_meeting_fallbacks() == before + 2
    assert cut["iterations"] == 7 and not cut["converged"] and np.all(np.isfinite(host(cut["x"])))
    monkeypatch.delenv("GKOMI_MEET_MAX_POLLS")
    cut_ok = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), krylov_dim=30, max_iters=7, reduction=1e-30)
    assert matgen.rel_err(host(cut["x"]), host(cut_ok["x"])) <= 1e-9


def test_gmres_back_substitution_from_lds_is_bit_identical(gk, oracle, monkeypatch):
    """One right-hand side: solve_krylov runs out of LDS (gmres_solve_krylov_single_kernel) -- the reference's loop
    (common/unified/solver/gmres_kernels.cpp solve_krylov), operand by operand in the same order: the solution of
    a solve with restarts has the same bits as with the one-thread-per-column kernel of the C ABI."""
    n, rp, ci, v = convection_diffusion_3d(20)
    b = np.cos(0.3 * np.arange(n))
    args = dict(krylov_dim=7, max_iters=2000, reduction=1e-10)
    res = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), **args)
    monkeypatch.setenv("GKOMI_GMRES_SOLVE_KRYLOV", "generic")
    ref = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), **args)
    assert res["converged"] and res["iterations"] == ref["iterations"] and res["iterations"] > 14
    assert host(res["x"]).tobytes() == host(ref["x"]).tobytes()


@pytest.mark.parametrize("g3,krylov_dim", [(27, 30), (64, 30), (80, 10), (100, 30), (117, 12)],
                         ids=["19683_odd_1pair", "262144_3pairs", "512000_3pairs", "1e6_5pairs", "1601613_odd_8pairs"])
def test_gmres_blocked_sweep_against_the_one_by_one_sweep(gk, oracle, g3, krylov_dim, monkeypatch):
    """The blocked modified Gram-Schmidt sweep (one meeting per 2-4 basis vectors, every register tiling of
    gmres_arnoldi_blocked_kernel, 16-byte and 8-byte loads) against the sweep that meets once per vector
    (GKOMI_GMRES_ARNOLDI=sweep; that one is pinned to the oracle by the tests above): the two differ in the rounding
    of the Hessenberg entries only (core/solver/gmres.cpp:300-319 is an identity in the sums either way) -- same
    iteration count +- 1, same solution to 1e-9, true residual at the goal."""
    n, rp, ci, v = convection_diffusion_3d(g3)
    b = np.cos(0.3 * np.arange(n))
    a = [dev(rp), dev(ci), dev(v)]
    args = dict(krylov_dim=krylov_dim, max_iters=3000, reduction=1e-10)
    res = solvers.gmres_solve(gk, n, *a, dev(b), **args)
    again = solvers.gmres_solve(gk, n, *a, dev(b), **args)
    monkeypatch.setenv("GKOMI_GMRES_ARNOLDI", "sweep")
    ref = solvers.gmres_solve(gk, n, *a, dev(b), **args)
    assert res["converged"] and ref["converged"] and abs(res["iterations"] - ref["iterations"]) <= 1
    assert res["iterations"] > krylov_dim                       # restarts included
    assert again["iterations"] == res["iterations"] and host(again["x"]).tobytes() == host(res["x"]).tobytes()
    assert matgen.rel_err(host(res["x"]), host(ref["x"])) <= 1e-9
    r = b.copy().reshape(n, 1)
    oracle.ref_csr_advanced_spmv(n, 1, -1.0, rp, ci, v, host(res["x"]).reshape(n, 1), 1, 1.0, r, 1)
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b)
