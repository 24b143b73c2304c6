"""GPU parity tests of the dense BLAS-1, CG step and stopping kernels through
the C ABI: known answers of reference/test/matrix/dense_kernels.cpp and
reference/test/solver/cg_kernels.cpp, then random data against the oracle
(the reference's cross-executor tests, test/matrix/dense_kernels.cpp,
test/solver/cg_kernels.cpp)."""
import json
import os

import numpy as np
import pytest
import torch

import matgen
from gpu_util import dev, host, stream_ptr

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
R = 10 * np.finfo(np.float64).eps  # r<double>::value, core/test/utils.hpp:212-220


def load(name):
    return json.load(open(os.path.join(G, name)))


def _strided(mat, stride, pad=-1.0):
    mat = np.array(mat, np.float64)
    stride = stride or mat.shape[1]
    buf = np.full((mat.shape[0], stride), pad)
    buf[:, :mat.shape[1]] = mat
    return buf


def _ws(gk, nrows, ncols):
    nbytes = gk.dense_reduction_workspace_bytes(nrows, ncols)
    return torch.empty(max(nbytes, 8), dtype=torch.uint8, device="cuda:0"), nbytes


@pytest.mark.parametrize("case", load("dense_blas1.json")["cases"], ids=lambda c: c["name"])
def test_dense_known_answers(gk, case):
    op = case["op"]
    expect = np.array(case["expect"], np.float64)
    stride = case.get("stride")
    s = stream_ptr()
    nr, nc = np.array(case["x"]).shape
    xd = dev(_strided(case["x"], stride))
    st = xd.shape[1]
    if op in ("scale", "inv_scale"):
        al = dev(np.array(case["alpha"], np.float64))
        getattr(gk, f"dense_{op}_f64")(s, nr, nc, al, al.numel(), xd, st)
        out = host(xd)
    elif op in ("add_scaled", "sub_scaled"):
        yd = dev(_strided(case["y"], stride))
        al = dev(np.array(case["alpha"], np.float64))
        getattr(gk, f"dense_{op}_f64")(s, nr, nc, al, al.numel(), xd, st, yd, st)
        out = host(yd)
    elif op == "fill":
        gk.dense_fill_f64(s, nr, nc, xd, st, case["value"])
        out = host(xd)
    elif op == "sqrt":
        gk.dense_compute_sqrt_f64(s, nr, nc, xd, st)
        out = host(xd)
    else:
        res = torch.full((1, nc), float("nan"), dtype=torch.float64, device="cuda:0")
        ws, nb = _ws(gk, nr, nc)
        if op == "dot":
            yd = dev(_strided(case["y"], stride))
            gk.dense_compute_dot_f64(s, nr, nc, xd, st, yd, st, res, ws, nb)
        elif op == "norm2":
            gk.dense_compute_norm2_f64(s, nr, nc, xd, st, res, ws, nb)
        elif op == "squared_norm2":
            gk.dense_compute_squared_norm2_f64(s, nr, nc, xd, st, res, ws, nb)
        elif op == "norm1":
            gk.dense_compute_norm1_f64(s, nr, nc, xd, st, res, ws, nb)
        assert np.array_equal(host(res), expect)
        return
    nc_e = expect.shape[1]
    assert np.array_equal(out[:, :nc_e], expect)
    if out.shape[1] > nc_e:
        assert np.all(out[:, nc_e:] == -1.0)


@pytest.mark.parametrize("shape", [(1, 1), (2, 1), (1023, 1), (100001, 1), (1 << 20, 1), (777, 3), (4096, 16)])
@pytest.mark.parametrize("padded", [False, True])
def test_dense_elementwise_bitexact_vs_oracle(gk, oracle, shape, padded):
    """Elementwise steps are one rounded product + one rounded sum per entry:
    bit-exact against the oracle (compiled -ffp-contract=off like the kernels)."""
    nr, nc = shape
    stride = nc + (2 if padded else 0)
    rng = np.random.default_rng(nr * 31 + nc)
    x = rng.standard_normal((nr, stride))
    y = rng.standard_normal((nr, stride))
    s = stream_ptr()
    for acols in (1, nc):
        alpha = rng.standard_normal(acols) + 2.5
        al = dev(alpha)
        for op in ("scale", "inv_scale"):
            e = x.copy()
            getattr(oracle, "ref_dense_" + op)(nr, nc, alpha, acols, e, stride)
            xd = dev(x)
            getattr(gk, f"dense_{op}_f64")(s, nr, nc, al, acols, xd, stride)
            assert np.array_equal(host(xd), e), op
        for op in ("add_scaled", "sub_scaled"):
            e = y.copy()
            getattr(oracle, "ref_dense_" + op)(nr, nc, alpha, acols, x, stride, e, stride)
            yd = dev(y)
            getattr(gk, f"dense_{op}_f64")(s, nr, nc, al, acols, dev(x), stride, yd, stride)
            assert np.array_equal(host(yd), e), op
    xd = dev(x)
    out = torch.full((nr, stride), -3.0, dtype=torch.float64, device="cuda:0")
    gk.dense_copy_f64(s, nr, nc, xd, stride, out, stride)
    assert np.array_equal(host(out)[:, :nc], x[:, :nc]) and np.all(host(out)[:, nc:] == -3.0)
    gk.dense_fill_f64(s, nr, nc, out, stride, 0.25)
    assert np.all(host(out)[:, :nc] == 0.25) and np.all(host(out)[:, nc:] == -3.0)


@pytest.mark.parametrize("shape", [(1, 1), (63, 1), (4097, 1), (1_000_000, 1), (1 << 22, 1), (50000, 3), (1000, 17)])
def test_dense_reductions_vs_oracle(gk, oracle, shape):
    nr, nc = shape
    rng = np.random.default_rng(nr + nc)
    x = rng.standard_normal((nr, nc))
    y = rng.standard_normal((nr, nc))
    xd, yd = dev(x), dev(y)
    s = stream_ptr()
    ws, nb = _ws(gk, nr, nc)
    res = torch.empty((1, nc), dtype=torch.float64, device="cuda:0")
    e = np.zeros((1, nc))
    # tolerance: sequential vs tree sums differ by O(eps*sqrt(n)) relative to sum|x y|
    def close(got, exp, scale):
        return np.all(np.abs(got - exp) <= 4 * np.finfo(np.float64).eps * max(1.0, np.sqrt(nr)) * np.maximum(scale, 1e-300))
    gk.dense_compute_dot_f64(s, nr, nc, xd, nc, yd, nc, res, ws, nb)
    oracle.ref_dense_compute_dot(nr, nc, x, nc, y, nc, e)
    assert close(host(res), e, np.sum(np.abs(x * y), axis=0))
    first = host(res).copy()
    gk.dense_compute_dot_f64(s, nr, nc, xd, nc, yd, nc, res, ws, nb)
    assert np.array_equal(first, host(res))  # no atomics: reproducible
    gk.dense_compute_norm2_f64(s, nr, nc, xd, nc, res, ws, nb)
    oracle.ref_dense_compute_norm2(nr, nc, x, nc, e)
    tol = 4 * np.finfo(np.float64).eps * max(1.0, np.sqrt(nr))  # sequential-vs-tree sum
    assert np.all(np.abs(host(res) - e) <= tol * e)
    gk.dense_compute_squared_norm2_f64(s, nr, nc, xd, nc, res, ws, nb)
    oracle.ref_dense_compute_squared_norm2(nr, nc, x, nc, e)
    assert np.all(np.abs(host(res) - e) <= tol * e)
    gk.dense_compute_norm1_f64(s, nr, nc, xd, nc, res, ws, nb)
    oracle.ref_dense_compute_norm1(nr, nc, x, nc, e)
    assert np.all(np.abs(host(res) - e) <= tol * e)


def test_reduction_workspace_too_small_is_an_error(gk):
    import gkomi
    x = torch.ones(1000, dtype=torch.float64, device="cuda:0")
    res = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    with pytest.raises(gkomi.GkomiError) as e:
        gk.dense_compute_norm2_f64(stream_ptr(), 1000, 1, x, 1, res, x, 0)
    assert e.value.code == -4


def test_row_gather(gk, oracle):
    rng = np.random.default_rng(2)
    src = rng.standard_normal((500, 3))
    rows = rng.integers(0, 500, size=123).astype(np.int32)
    out = torch.zeros((123, 3), dtype=torch.float64, device="cuda:0")
    gk.dense_row_gather_f64_i32(stream_ptr(), 123, 3, dev(rows), dev(src), 3, out, 3)
    assert np.array_equal(host(out), src[rows])


@pytest.mark.parametrize("case", load("cg.json")["kernel_cases"], ids=lambda c: c["name"])
def test_cg_kernel_known_answers(gk, case):
    A = lambda k: dev(np.array(case[k], np.float64))
    s = stream_ptr()
    stop = dev(np.array(case.get("stop", [0, 0]), np.uint8))
    if case["op"] == "step_1":
        p = A("p")
        gk.cg_step_1_f64(s, 2, 2, p, 2, A("z"), 2, A("rho"), A("prev_rho"), stop)
        assert np.array_equal(host(p), np.array(case["expect_p"]))
    elif case["op"] == "step_2":
        x, r = A("x"), A("r")
        gk.cg_step_2_f64(s, 2, 2, x, 2, r, 2, A("p"), 2, A("q"), 2, A("beta"), A("rho"), stop)
        assert np.array_equal(host(x), np.array(case["expect_x"]))
        assert np.array_equal(host(r), np.array(case["expect_r"]))
    else:
        b = dev(_strided(case["b"], case["b_stride"]))
        mk = lambda v: torch.full((2, 2), v, dtype=torch.float64, device="cuda:0")
        r, z, p, q = mk(0.0), mk(1.0), mk(1.0), mk(1.0)
        prev_rho, rho = dev(np.zeros(2)), dev(np.ones(2))
        stop = dev(np.array([1, 1], np.uint8))
        gk.cg_initialize_f64(s, 2, 2, b, 3, r, 2, z, 2, p, 2, q, 2, prev_rho, rho, stop)
        assert np.array_equal(host(r), np.array(case["expect_r"]))
        assert not host(z).any() and not host(p).any() and not host(q).any()
        assert np.array_equal(host(rho), [0.0, 0.0]) and np.array_equal(host(prev_rho), [1.0, 1.0])
        assert not host(stop).any()


@pytest.mark.parametrize("shape", [(1, 1), (777, 1), (1_000_001, 1), (513, 4)])
def test_cg_steps_bitexact_vs_oracle(gk, oracle, shape):
    n, k = shape
    rng = np.random.default_rng(n)
    x, r, p, q, z = (rng.standard_normal((n, k)) for _ in range(5))
    rho = rng.standard_normal(k) + 3
    prev_rho = rng.standard_normal(k) + 3
    beta = rng.standard_normal(k) + 3
    stop = np.zeros(k, np.uint8)
    if k > 1:
        stop[1] = 1
        prev_rho[2] = 0.0
        beta[2] = 0.0
    s = stream_ptr()
    pe = p.copy()
    oracle.ref_cg_step_1(n, k, pe, k, z, k, rho, prev_rho, stop)
    pd = dev(p)
    gk.cg_step_1_f64(s, n, k, pd, k, dev(z), k, dev(rho), dev(prev_rho), dev(stop))
    assert np.array_equal(host(pd), pe)
    xe, re_ = x.copy(), r.copy()
    oracle.ref_cg_step_2(n, k, xe, k, re_, k, p, k, q, k, beta, rho, stop)
    xd, rd = dev(x), dev(r)
    gk.cg_step_2_f64(s, n, k, xd, k, rd, k, dev(p), k, dev(q), k, dev(beta), dev(rho), dev(stop))
    assert np.array_equal(host(xd), xe) and np.array_equal(host(rd), re_)


def test_stop_kernels_match_oracle(gk, oracle):
    rng = np.random.default_rng(11)
    for nrhs in (1, 3, 300):
        tau = rng.random(nrhs)
        orig = np.ones(nrhs)
        st0 = (rng.integers(0, 2, nrhs) * 3).astype(np.uint8)
        for fin in (0, 1):
            st = st0.copy(); fl = np.zeros(2, np.uint8)
            oracle.ref_residual_norm(nrhs, tau, orig, 0.5, 2, fin, st, fl)
            std = dev(st0); fld = dev(np.full(2, 9, np.uint8))
            hflags = np.full(2, 7, np.uint8)
            gk.residual_norm_f64(stream_ptr(), nrhs, dev(tau), dev(orig), 0.5, 2, fin, std, fld, hflags)
            assert np.array_equal(host(std), st) and np.array_equal(host(fld), fl)
            assert np.array_equal(hflags, fl)  # blocking host copy like the reference's
            st = st0.copy()
            oracle.ref_implicit_residual_norm(nrhs, tau, orig, 0.7, 4, fin, st, fl)
            std = dev(st0)
            gk.implicit_residual_norm_f64(stream_ptr(), nrhs, dev(tau), dev(orig), 0.7, 4, fin, std, fld, None)
            assert np.array_equal(host(std), st) and np.array_equal(host(fld), fl)
            st = st0.copy()
            oracle.ref_set_all_statuses(nrhs, 5, fin, st)
            std = dev(st0)
            gk.set_all_statuses(stream_ptr(), nrhs, 5, fin, std)
            assert np.array_equal(host(std), st)
