"""GPU parity tests of the <float, int32> instantiation (csrc/f32.hip, gkomi_*_f32) against the single-precision oracle
(oracle/f32.c): csr::spmv / advanced_spmv bit-exact for any row lengths (reference order, float intermediates, no
contraction), elementwise dense and CG kernels bit-exact, reductions within 4 eps32 sqrt(n) of the sequential sum, the
Cg driver on the reference's stencil system (known answer) and on a Poisson system against the oracle's Cg<float>."""
import numpy as np
import pytest
import torch

import matgen
from gpu_util import stream_ptr
from test_oracle_golden import load

pytestmark = pytest.mark.gpu
F = np.float32
EPS32 = float(np.finfo(np.float32).eps)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def spmv(gk, n, m, rp, ci, v, b, c=None, alpha=None, beta=None):
    nrhs = b.shape[1]
    out = torch.full((n, nrhs), float("nan"), dtype=torch.float32, device="cuda:0") if c is None else dev(c)
    al = dev(np.array([alpha], F)) if alpha is not None else None
    be = dev(np.array([beta], F)) if beta is not None else None
    gk.csr_spmv_f32_i32(stream_ptr(), n, m, nrhs, len(v), dev(rp), dev(ci), dev(v), dev(b), b.shape[1], out, nrhs, al, be)
    return host(out)


@pytest.mark.parametrize("case", load("csr_spmv.json")["cases"], ids=lambda c: c["name"])
def test_csr_known_answers(gk, case):
    m = load("csr_spmv.json")["matrices"][case["matrix"]]
    rp, ci, v = np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32), np.array(m["vals"], F)
    b = np.array(case["b"], F)
    if "alpha" in case:
        got = spmv(gk, m["nrows"], m["ncols"], rp, ci, v, b, np.array(case["c"], F), case["alpha"], case["beta"])
    else:
        got = spmv(gk, m["nrows"], m["ncols"], rp, ci, v, b)
    assert np.array_equal(got, np.array(case["expect"], F))


@pytest.mark.parametrize("shape", ["532x231", "poisson", "long_rows", "empty_rows", "one_row"])
@pytest.mark.parametrize("nrhs", [1, 3])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_csr_bitexact_vs_oracle(gk, oracle, shape, nrhs, advanced):
    rng = np.random.default_rng(5)
    if shape == "532x231":
        n, m = 532, 231
        rp, ci, v = matgen.random_csr(n, m, 1, 231, seed=42, sort=False)     # test/matrix/csr_kernels2.cpp:73-77
    elif shape == "poisson":
        n, rp, ci, v = matgen.poisson_2d_5pt(173, 181)
        m = n
    elif shape == "long_rows":
        n, m = 700, 9000
        counts = rng.integers(0, 30, size=n); counts[3] = 8000; counts[400] = 5000
        rp, ci, v = matgen.random_rows_csr(n, m, counts, 3)
    elif shape == "empty_rows":
        n, m = 5000, 300
        counts = rng.integers(0, 4, size=n); counts[100:3000] = 0; counts[-20:] = 0
        rp, ci, v = matgen.random_rows_csr(n, m, counts, 4)
    else:
        n, m = 1, 4000
        rp, ci, v = matgen.random_rows_csr(n, m, np.array([3500]), 6)
    rp, ci, v = rp.astype(np.int32), ci.astype(np.int32), v.astype(F)
    b = rng.standard_normal((m, nrhs)).astype(F)
    c0 = rng.standard_normal((n, nrhs)).astype(F)
    if advanced:
        expect = c0.copy()
        oracle.ref_csr_advanced_spmv_f32(n, nrhs, -0.75, rp, ci, v, b, nrhs, 1.5, expect, nrhs)
        got = spmv(gk, n, m, rp, ci, v, b, c0, -0.75, 1.5)
    else:
        expect = np.full((n, nrhs), np.nan, F)
        oracle.ref_csr_spmv_f32(n, nrhs, rp, ci, v, b, nrhs, expect, nrhs)
        got = spmv(gk, n, m, rp, ci, v, b)
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("ncols,stride", [(1, 1), (3, 3), (3, 5)])
def test_dense_kernels_vs_oracle(gk, oracle, ncols, stride):
    rng = np.random.default_rng(ncols + stride)
    n = 20011
    x = rng.standard_normal((n, stride)).astype(F)
    y0 = rng.standard_normal((n, stride)).astype(F)
    for alpha in (np.array([0.75], F), rng.standard_normal(ncols).astype(F)):
        for op in ("scale", "inv_scale", "add_scaled", "sub_scaled"):
            ye, yd = y0.copy(), dev(y0)
            if op in ("scale", "inv_scale"):
                getattr(oracle, f"ref_dense_{op}_f32")(n, ncols, alpha, len(alpha), ye, stride)
                getattr(gk, f"dense_{op}_f32")(stream_ptr(), n, ncols, dev(alpha), len(alpha), yd, stride)
            else:
                getattr(oracle, f"ref_dense_{op}_f32")(n, ncols, alpha, len(alpha), x, stride, ye, stride)
                getattr(gk, f"dense_{op}_f32")(stream_ptr(), n, ncols, dev(alpha), len(alpha), dev(x), stride, yd, stride)
            assert np.array_equal(host(yd), ye), op     # padding columns untouched, entries bit-exact
    yd = dev(y0)
    gk.dense_fill_f32(stream_ptr(), n, ncols, yd, stride, 2.5)
    ye = y0.copy(); ye[:, :ncols] = 2.5
    assert np.array_equal(host(yd), ye)
    yd = dev(y0)
    gk.dense_copy_f32(stream_ptr(), n, ncols, dev(x), stride, yd, stride)
    ye = y0.copy(); ye[:, :ncols] = x[:, :ncols]
    assert np.array_equal(host(yd), ye)
    # reductions: two-stage on the device, sequential in the oracle
    nb = gk.dense_reduction_workspace_bytes_f32(n, ncols)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    res = torch.zeros(ncols, dtype=torch.float32, device="cuda:0")
    e = np.zeros(ncols, F)
    gk.dense_compute_dot_f32(stream_ptr(), n, ncols, dev(x), stride, dev(y0), stride, res, ws, nb)
    oracle.ref_dense_compute_dot_f32(n, ncols, x, stride, y0, stride, e)
    scale = np.sum(np.abs(x[:, :ncols].astype(np.float64) * y0[:, :ncols]), axis=0)
    assert np.all(np.abs(host(res).astype(np.float64) - e) <= 4 * EPS32 * np.sqrt(n) * scale)
    gk.dense_compute_norm2_f32(stream_ptr(), n, ncols, dev(x), stride, res, ws, nb)
    oracle.ref_dense_compute_norm2_f32(n, ncols, x, stride, e)
    assert np.all(np.abs(host(res).astype(np.float64) - e) <= 4 * EPS32 * np.sqrt(n) * e)


@pytest.mark.parametrize("case", load("cg.json")["kernel_cases"], ids=lambda c: c["name"])
def test_cg_kernel_known_answers(gk, oracle, case):
    """reference/test/solver/cg_kernels.cpp:153-252 in float: the device kernels against the oracle's (bit-exact)"""
    A = lambda k: np.array(case[k], F)
    stop = np.array(case.get("stop", [0, 0]), np.uint8)
    if case["op"] == "step_1":
        p, z = A("p"), A("z")
        pd = dev(p)
        oracle.ref_cg_step_1_f32(2, 2, p, 2, z, 2, A("rho"), A("prev_rho"), stop)
        gk.cg_step_1_f32(stream_ptr(), 2, 2, pd, 2, dev(z), 2, dev(A("rho")), dev(A("prev_rho")), dev(stop))
        assert np.array_equal(host(pd), p)
    elif case["op"] == "step_2":
        x, r = A("x"), A("r")
        xd, rd = dev(x), dev(r)
        oracle.ref_cg_step_2_f32(2, 2, x, 2, r, 2, A("p"), 2, A("q"), 2, A("beta"), A("rho"), stop)
        gk.cg_step_2_f32(stream_ptr(), 2, 2, xd, 2, rd, 2, dev(A("p")), 2, dev(A("q")), 2, dev(A("beta")), dev(A("rho")), dev(stop))
        assert np.array_equal(host(xd), x) and np.array_equal(host(rd), r)
    else:
        from test_oracle_golden import _strided
        b = _strided(case["b"], case["b_stride"]).astype(F)
        t = lambda fill, shape=(2, 2): torch.full(shape, fill, dtype=torch.float32, device="cuda:0")
        r, z, p, q, prev_rho, rho = t(0.0), t(1.0), t(1.0), t(1.0), t(0.0, (2,)), t(1.0, (2,))
        stopd = dev(np.array([1, 1], np.uint8))
        gk.cg_initialize_f32(stream_ptr(), 2, 2, dev(b), b.shape[1], r, 2, z, 2, p, 2, q, 2, prev_rho, rho, stopd)
        assert np.array_equal(host(r), A("expect_r")) and not host(z).any() and not host(p).any() and not host(q).any()
        assert np.array_equal(host(rho), A("expect_rho")) and np.array_equal(host(prev_rho), A("expect_prev_rho"))
        assert not host(stopd).any()


def cg(gk, n, rp, ci, v, b, x0, max_iters, reduction):
    nb = gk.cg_workspace_bytes_f32(n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    x = dev(x0.astype(F))
    info = np.zeros(4)
    gk.cg_solve_f32(stream_ptr(), n, len(v), dev(rp.astype(np.int32)), dev(ci.astype(np.int32)), dev(v.astype(F)), dev(b.astype(F)), x,
                    max_iters, reduction, 0, ws, nb, info)
    return host(x), int(info[0]), bool(info[1]), info[2], info[3]


def test_cg_driver(gk, oracle):
    r_float = 10 * EPS32
    case = load("cg.json")["solve_cases"][0]      # SolvesStencilSystem, cg_kernels.cpp:255-266
    rp, ci, v = matgen.dense_to_csr(case["A"])
    x, iters, conv, _, _ = cg(gk, 3, rp, ci, v, np.array(case["b"]), np.array(case["x0"]), case["max_iters"], r_float)
    assert conv and iters < case["max_iters"] and matgen.rel_err(x.astype(np.float64), case["expect_x"]) <= r_float
    # a 2-D Poisson system against the oracle's Cg<float>: same recurrences, reductions in another order
    n, rp, ci, v = matgen.poisson_2d_5pt(40)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(n)
    x, iters, conv, res, base = cg(gk, n, rp, ci, v, b, np.zeros(n), 1000, 1e-5)
    xe = np.zeros(n, F)
    ite = oracle.ref_cg_solve_f32(n, rp.astype(np.int32), ci.astype(np.int32), v.astype(F), b.astype(F), xe, 1000, 1e-5, 0)
    assert conv and abs(iters - ite) <= max(2, ite // 20)
    assert res < 1e-5 * base
    assert matgen.rel_err(x.astype(np.float64), xe.astype(np.float64)) <= 1e-3     # both solved to 1e-5 of ||b||, kappa ~ 700
    # the iteration limit stops an unconverged solve and says so
    x, iters, conv, _, _ = cg(gk, n, rp, ci, v, b, np.zeros(n), 5, 1e-7)
    assert iters == 5 and not conv


@pytest.mark.parametrize("nnz_target", [1, 5, 8, 9, 10, 11, 2047, 2048, 2049, 2051, 4099])
def test_csr_quads_and_tails(gk, oracle, nnz_target):
    """nonzero counts around the 2048-nonzero tile of the float kernel and tiny ones, rows scattered over the workgroups
    (written for a variant with 16-B loads, which lost 12.2 -> 16.2 us and was dropped; the cases stay)"""
    rng = np.random.default_rng(nnz_target)
    ncols = 97
    nrows = max(1, nnz_target // 3 + 300)
    counts = np.zeros(nrows, np.int64)
    left = nnz_target
    order = rng.permutation(nrows)
    for r in order:
        k = min(left, int(rng.integers(0, 8)))
        counts[r] = k
        left -= k
        if left == 0:
            break
    counts[order[0]] += left
    rp, ci, v = matgen.random_rows_csr(nrows, ncols, np.minimum(counts, ncols), 7)
    rp, ci, v = rp.astype(np.int32), ci.astype(np.int32), v.astype(F)
    b = rng.standard_normal((ncols, 1)).astype(F)
    expect = np.full((nrows, 1), np.nan, F)
    oracle.ref_csr_spmv_f32(nrows, 1, rp, ci, v, b, 1, expect, 1)
    assert np.array_equal(spmv(gk, nrows, ncols, rp, ci, v, b), expect)
