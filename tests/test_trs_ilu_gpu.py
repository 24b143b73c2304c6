"""GPU parity tests (C ABI) of the sparse triangular solves (= ILU apply),
the ParILU chain and csr::transpose against the oracle.  Triangular solves
are bit-exact (per row the subtractions run in storage order).  ParILU sweeps
are asynchronous: parity by tolerance after enough sweeps, like
test/factorization/par_ilu_kernels.cpp:277-309."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

import ilu_util
import matgen
from gpu_util import dev, host, stream_ptr

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trs_ilu.json")))


def trs(gk, which, n, rp, ci, v, unit, b):
    fn = gk.lower_trs_solve_f64_i32 if which == "lower" else gk.upper_trs_solve_f64_i32
    nrhs = b.shape[1]
    x = torch.full((n, nrhs), 777.0, dtype=torch.float64, device="cuda:0")
    nb = gk.trs_workspace_bytes()
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
    fn(stream_ptr(), n, nrhs, dev(rp), dev(ci), dev(v), int(unit), dev(b), nrhs, x, nrhs, ws, nb)
    flag = ctypes.c_int(-1)
    gk.trs_check_overrun(stream_ptr(), ws, ctypes.addressof(flag))
    assert flag.value == 0, "triangular solve hit its spin bound"
    return host(x)


@pytest.mark.parametrize("which", ["lower", "upper"])
def test_trs_known_answers(gk, which):
    g = G[which]
    for case in g["cases"]:
        rp, ci, v = matgen.dense_to_csr(g[case["matrix"]])
        b = np.array(case["b"], np.float64)
        x = trs(gk, which, len(b), rp, ci, v, case["unit"], b)
        assert matgen.rel_err(x, case["expect"]) <= case["tol"], case["name"]


def random_triangular(n, lower, seed, max_off=6, band=None):
    """Well-conditioned random triangular CSR with the other triangle partly
    filled (must be ignored) and entries in a random order inside each row."""
    rng = np.random.default_rng(seed)
    rp, ci, v = [0], [], []
    for r in range(n):
        lo, hi = (0, r) if lower else (r + 1, n)
        if band is not None:
            lo, hi = (max(0, r - band), r) if lower else (r + 1, min(n, r + 1 + band))
        cnt = min(hi - lo, int(rng.integers(0, max_off + 1)))
        deps = list(rng.choice(np.arange(lo, hi), size=cnt, replace=False)) if cnt else []
        olo, ohi = (r + 1, n) if lower else (0, r)
        other = list(rng.choice(np.arange(olo, ohi), size=min(2, ohi - olo), replace=False)) if ohi > olo else []
        cols = deps + other + [r]
        rng.shuffle(cols)
        for c in cols:
            ci.append(int(c))
            v.append(float(rng.standard_normal() * 0.3 + (4.0 if c == r else 0.0)))
        rp.append(len(ci))
    return np.array(rp, np.int32), np.array(ci, np.int32), np.array(v)


@pytest.mark.parametrize("which", ["lower", "upper"])
@pytest.mark.parametrize("n,band", [(1, None), (63, None), (257, None), (5000, 40), (100_000, 700)])
@pytest.mark.parametrize("unit", [False, True])
def test_trs_bitexact_vs_oracle(gk, oracle, which, n, band, unit):
    rp, ci, v = random_triangular(n, which == "lower", seed=n + unit, band=band)
    rng = np.random.default_rng(5)
    for nrhs in (1, 2):
        b = rng.standard_normal((n, nrhs))
        e = np.zeros_like(b)
        (oracle.ref_lower_trs_solve if which == "lower" else oracle.ref_upper_trs_solve)(
            n, nrhs, rp, ci, v, int(unit), b, nrhs, e, nrhs)
        x = trs(gk, which, n, rp, ci, v, unit, b)
        assert np.array_equal(x, e)


def test_trs_long_dependency_chain(gk, oracle):
    # bidiagonal: every row waits for the previous one (worst case for a
    # sync-free solve: 20000 sequential hand-offs, many inside one wave)
    n = 20000
    rp = np.arange(0, 2 * n + 1, 2, dtype=np.int32) - 1
    rp[0] = 0
    ci = np.empty(2 * n - 1, np.int32)
    v = np.empty(2 * n - 1)
    ci[0], v[0] = 0, 2.0
    ci[1::2], v[1::2] = np.arange(0, n - 1), -1.0
    ci[2::2], v[2::2] = np.arange(1, n), 2.0
    b = np.ones((n, 1))
    e = np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, rp, ci, v, 0, b, 1, e, 1)
    assert np.array_equal(trs(gk, "lower", n, rp, ci, v, False, b), e)


def test_trs_nan_results_do_not_hang(gk, oracle):
    # 0/0 on the diagonal gives NaN; later rows must still complete
    rp, ci, v = matgen.dense_to_csr([[1e-300, 0, 0], [1.0, 2.0, 0], [1.0, 1.0, 3.0]])
    v = v.copy()
    v[0] = 0.0
    b = np.zeros((3, 1))
    x = trs(gk, "lower", 3, rp, ci, v, False, b)
    assert np.isnan(x).all()


def test_ilu0_apply_on_poisson_matches_oracle(gk, oracle):
    """Ilu::apply = L^-1 then U^-1 (ilu.hpp:265-286) with the exact ILU(0)
    factors of a 3-D 7-pt stencil (config 4's structure: ~3*grid levels)."""
    n, rp, ci, v = matgen.poisson_3d_7pt(24)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    lrp, lc, lv = f["L"]
    urp, uc, uv = f["U"]
    b = np.sin(0.1 * np.arange(n)).reshape(n, 1)
    y = np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, lrp, lc, lv, 0, b, 1, y, 1)
    e = np.zeros_like(b)
    oracle.ref_upper_trs_solve(n, 1, urp, uc, uv, 0, y, 1, e, 1)
    yg = trs(gk, "lower", n, lrp, lc, lv, False, b)
    assert np.array_equal(yg, y)
    assert np.array_equal(trs(gk, "upper", n, urp, uc, uv, False, yg), e)


@pytest.mark.parametrize("case", G["add_diagonal"], ids=lambda c: c["name"])
def test_add_diagonal_known_answers(gk, case):
    n, m = case["nrows"], case["ncols"]
    rpd = dev(np.array(case["row_ptrs"], np.int32))
    cid = dev(np.array(case["col_idxs"] or [0], np.int32))
    vd = dev(np.array(case["vals"] or [0.0]))
    nb = gk.factorization_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    missing = ctypes.c_int64(-1)
    gk.factorization_count_missing_diagonal_i32(stream_ptr(), n, m, rpd, cid, ws, nb, ctypes.addressof(missing))
    total = len(case["expect_vals"])
    assert missing.value == total - len(case["vals"])
    nc = torch.zeros(total, dtype=torch.int32, device="cuda:0")
    nv = torch.zeros(total, dtype=torch.float64, device="cuda:0")
    gk.factorization_add_diagonal_elements_f64_i32(stream_ptr(), n, m, rpd, cid, vd, nc, nv, ws)
    assert list(host(rpd)) == case["expect_row_ptrs"]
    assert list(host(nc)) == case["expect_col_idxs"] and list(host(nv)) == case["expect_vals"]


def test_par_ilu_kernel_known_answers(gk):
    """reference/test/factorization/par_ilu_kernels.cpp:306-446 through the C ABI: initialize_row_ptrs_l_u,
    initialize_l_u, compute_l_u_factors on mtx_small (sweeps until the exact factors: the device sweep is
    asynchronous, the reference's sequential one is exact after one), and the zero-matrix variants"""
    k = G["par_ilu"]["kernels"]
    n = 3
    rp, ci, v = matgen.dense_to_csr(np.array(k["A"], np.float64))
    s = stream_ptr()
    sb = gk.prefix_sum_workspace_bytes(n + 1)
    sws = torch.empty(max(sb, 8), dtype=torch.uint8, device="cuda:0")
    lrp = torch.zeros(n + 1, dtype=torch.int32, device="cuda:0")
    urp = torch.zeros(n + 1, dtype=torch.int32, device="cuda:0")
    gk.factorization_initialize_row_ptrs_l_u_i32(s, n, dev(rp), dev(ci), lrp, urp, sws, sb)
    assert list(host(lrp)) == k["l_row_ptrs"] and list(host(urp)) == k["u_row_ptrs"]
    lc, uc = (torch.zeros(6, dtype=torch.int32, device="cuda:0") for _ in range(2))
    lv, uv = (torch.zeros(6, dtype=torch.float64, device="cuda:0") for _ in range(2))
    gk.factorization_initialize_l_u_f64_i32(s, n, dev(rp), dev(ci), dev(v), lrp, lc, lv, urp, uc, uv)
    assert np.array_equal(ilu_util.csr_to_dense(n, n, host(lrp), host(lc), host(lv)), np.array(k["L_init"], np.float64))
    assert np.array_equal(ilu_util.csr_to_dense(n, n, host(urp), host(uc), host(uv)), np.array(k["U_init"], np.float64))
    f = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=5)
    assert matgen.rel_err(ilu_util.csr_to_dense(n, n, *(host(t) for t in f["L"])), k["L_after_one_sweep"]) <= k["tol"]
    assert matgen.rel_err(ilu_util.csr_to_dense(n, n, *(host(t) for t in f["U"])), k["U_after_one_sweep"]) <= k["tol"]
    z = k["zero_matrix"]
    zrp = torch.zeros(n + 1, dtype=torch.int32, device="cuda:0")
    one = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    lrp.copy_(dev(np.array(z["l_row_ptrs"], np.int32)))
    urp.copy_(dev(np.array(z["u_row_ptrs"], np.int32)))
    gk.factorization_initialize_l_u_f64_i32(s, n, zrp, one, torch.zeros(1, dtype=torch.float64, device="cuda:0"), lrp, lc, lv, urp, uc, uv)
    assert np.array_equal(ilu_util.csr_to_dense(n, n, host(lrp), host(lc)[:3], host(lv)[:3]), np.eye(n))
    assert np.array_equal(ilu_util.csr_to_dense(n, n, host(urp), host(uc)[:3], host(uv)[:3]), np.eye(n))


@pytest.mark.parametrize("case", G["par_ilu"]["cases"], ids=lambda c: c["name"])
def test_par_ilu_known_factors(gk, case):
    a = np.array(case["A"], np.float64)
    n = a.shape[0]
    rp, ci, v = matgen.dense_to_csr(a)
    f = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=30)
    L = ilu_util.csr_to_dense(n, n, *[host(t) for t in f["L"]])
    U = ilu_util.csr_to_dense(n, n, *[host(t) for t in f["U"]])
    assert matgen.rel_err(L, case["L"]) <= max(case["tol"], 1e-14)
    assert matgen.rel_err(U, case["U"]) <= max(case["tol"], 1e-14)


@pytest.mark.parametrize("name", ["poisson2d", "poisson3d", "random_dd"])
def test_par_ilu_chain_vs_oracle(gk, oracle, name):
    if name == "poisson2d":
        n, rp, ci, v = matgen.poisson_2d_5pt(40, 37)
    elif name == "poisson3d":
        n, rp, ci, v = matgen.poisson_3d_7pt(12, 11, 13)
    else:
        n = 400
        rp, ci, v = matgen.random_csr(n, n, 2, 9, seed=12)
        # diagonally dominant, some rows without a stored diagonal
        a = ilu_util.csr_to_dense(n, n, rp, ci, v) * 0.1
        a[np.arange(n), np.arange(n)] = 3.0
        a[7, 7] = 0.0
        a[123, 123] = 0.0
        rp, ci, v = matgen.dense_to_csr(a)
    e = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    f = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=60)
    # integer parts of the chain: bit-exact
    for key in ("A", "L", "U"):
        assert np.array_equal(host(f[key][0]), e[key][0]), key
        assert np.array_equal(host(f[key][1]), e[key][1]), key
    assert np.array_equal(host(f["A"][2]), e["A"][2])
    # factors: converged fixed point == the sequential sweep
    assert matgen.rel_err(host(f["L"][2]), e["L"][2]) <= 1e-12
    assert matgen.rel_err(host(f["U"][2]), e["U"][2]) <= 1e-12
    # the reference's own bar for the default sweep count: 5e-2
    f10 = ilu_util.gpu_par_ilu(gk, torch, n, dev(rp), dev(ci), dev(v), iterations=0)
    assert matgen.rel_err(host(f10["L"][2]), e["L"][2]) <= 5e-2
    assert matgen.rel_err(host(f10["U"][2]), e["U"][2]) <= 5e-2


def test_transpose_bitexact_vs_oracle(gk, oracle):
    for (nr, nc, seed) in ((37, 23, 2), (1000, 1000, 3), (5, 2000, 4)):
        rp, ci, v = matgen.random_csr(nr, nc, 0, 9, seed=seed)
        nnz = len(ci)
        etrp, etc_, etv = np.zeros(nc + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        oracle.ref_csr_transpose(nr, nc, rp, ci, v, etrp, etc_, etv)
        nb = gk.csr_transpose_workspace_bytes(nc)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        trp = torch.zeros(nc + 1, dtype=torch.int32, device="cuda:0")
        tc = torch.zeros(nnz, dtype=torch.int32, device="cuda:0")
        tv = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
        gk.csr_transpose_f64_i32(stream_ptr(), nr, nc, nnz, dev(rp), dev(ci), dev(v), trp, tc, tv, ws, nb)
        assert np.array_equal(host(trp), etrp) and np.array_equal(host(tc), etc_) and np.array_equal(host(tv), etv)


def test_sort_by_column_index_and_is_sorted(gk, oracle):
    """csr::sort_by_column_index / is_sorted_by_column_index (csr_kernels.cpp:969-1009)"""
    import ctypes
    n = 2000
    rp, ci, v = matgen.random_csr(n, 1500, 0, 40, seed=9, sort=False)
    ws = torch.zeros(8, dtype=torch.uint8, device="cuda:0")
    flag = ctypes.c_int(-1)
    gk.csr_is_sorted_by_column_index_i32(stream_ptr(), n, dev(rp), dev(ci), ws, 8, ctypes.addressof(flag))
    assert flag.value == 0 == oracle.ref_csr_is_sorted_by_column_index(n, rp, ci)
    ec, ev = ci.copy(), v.copy()
    oracle.ref_csr_sort_by_column_index(n, rp, ec, ev)
    cd, vd = dev(ci), dev(v)
    gk.csr_sort_by_column_index_f64_i32(stream_ptr(), n, dev(rp), cd, vd)
    assert np.array_equal(host(cd), ec) and host(vd).tobytes() == ev.tobytes()
    gk.csr_is_sorted_by_column_index_i32(stream_ptr(), n, dev(rp), cd, ws, 8, ctypes.addressof(flag))
    assert flag.value == 1 == oracle.ref_csr_is_sorted_by_column_index(n, rp, ec)
    gk.csr_is_sorted_by_column_index_i32(stream_ptr(), 0, dev(rp[:1]), cd, ws, 8, ctypes.addressof(flag))
    assert flag.value == 1


def test_csr_utilities_known_answers(gk):
    """the reference's own small cases (reference/test/matrix/csr_kernels.cpp:1044-1076, 1299-1344; fixture
    tests/golden/formats.json csr_utilities) through the C ABI: transpose, is_sorted, sort, extract_diagonal"""
    import ctypes, json, os
    from test_oracle_golden import _dense_to_csr
    u = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats.json")))["csr_utilities"]
    for t in u["transposes"]:
        (nr, nc), rp, ci, v = _dense_to_csr(t["dense"])
        nnz = len(ci)
        nb = gk.csr_transpose_workspace_bytes(nc)
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda:0")
        trp = torch.zeros(nc + 1, dtype=torch.int32, device="cuda:0")
        tc = torch.zeros(nnz, dtype=torch.int32, device="cuda:0")
        tv = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
        gk.csr_transpose_f64_i32(stream_ptr(), nr, nc, nnz, dev(rp), dev(ci), dev(v), trp, tc, tv, ws, nb)
        (_, _), erp, ec, ev = _dense_to_csr(t["expect"])
        assert np.array_equal(host(trp), erp) and np.array_equal(host(tc), ec) and np.array_equal(host(tv), ev), t["name"]
    s, un = u["mtx3_sorted"], u["mtx3_unsorted"]
    rp = dev(np.array(s["row_ptrs"], np.int32))
    ws = torch.zeros(8, dtype=torch.uint8, device="cuda:0")
    flag = ctypes.c_int(-1)
    for m, expect in ((s, 1), (un, 0)):
        gk.csr_is_sorted_by_column_index_i32(stream_ptr(), 3, rp, dev(np.array(m["col_idxs"], np.int32)), ws, 8, ctypes.addressof(flag))
        assert flag.value == expect
        c, v = dev(np.array(m["col_idxs"], np.int32)), dev(np.array(m["vals"], np.float64))
        gk.csr_sort_by_column_index_f64_i32(stream_ptr(), 3, rp, c, v)
        assert list(host(c)) == s["col_idxs"] and list(host(v)) == s["vals"]
    diag = torch.full((3,), -1.0, dtype=torch.float64, device="cuda:0")
    gk.csr_extract_diagonal_f64_i32(stream_ptr(), 3, rp, dev(np.array(un["col_idxs"], np.int32)), dev(np.array(un["vals"], np.float64)), diag)
    assert list(host(diag)) == u["mtx3_diagonal"]


# ---- analysed solves: LowerTrs / UpperTrs::generate + apply (csrc/trs_levels.hip) ----------

def trs_plan(gk, which, n, rp, ci, v, unit, b, check=None):
    import gkomi.solvers as solvers
    plan = solvers.TrsPlan(gk, n, dev(rp), dev(ci), dev(v), which == "lower")
    nrhs = b.shape[1]
    x = torch.full((n, nrhs), 777.0, dtype=torch.float64, device="cuda:0")
    plan.solve(dev(b), x, unit)
    assert not plan.overrun(), "analysed triangular solve hit its spin bound"
    if check is not None:
        check(plan)
    return host(x)


def reference_levels(n, rp, ci, lower):
    lvl = np.zeros(n, np.int64)
    order = range(n) if lower else range(n - 1, -1, -1)
    for r in order:
        cols = ci[rp[r]:rp[r + 1]]
        deps = cols[cols < r] if lower else cols[cols > r]
        if len(deps):
            lvl[r] = lvl[deps].max() + 1
    return lvl


@pytest.mark.parametrize("which", ["lower", "upper"])
def test_trs_plan_known_answers(gk, which):
    g = G[which]
    for case in g["cases"]:
        rp, ci, v = matgen.dense_to_csr(g[case["matrix"]])
        b = np.array(case["b"], np.float64)
        x = trs_plan(gk, which, len(b), rp, ci, v, case["unit"], b)
        assert matgen.rel_err(x, case["expect"]) <= case["tol"], case["name"]


@pytest.mark.parametrize("which", ["lower", "upper"])
@pytest.mark.parametrize("n,band", [(1, None), (63, None), (64, None), (257, None), (5000, 40), (100_000, 700)])
@pytest.mark.parametrize("unit", [False, True])
def test_trs_plan_bitexact_vs_oracle(gk, oracle, which, n, band, unit):
    rp, ci, v = random_triangular(n, which == "lower", seed=n + unit, band=band)
    rng = np.random.default_rng(5)

    def check(plan):
        if n <= 5000:   # the analysis itself: level count and the level-sorted permutation
            lvl = reference_levels(n, rp, ci, which == "lower")
            assert plan.nlevels == int(lvl.max()) + 1 and plan.nslices == (n + 63) // 64
            perm = host(plan.plan[256:256 + 4 * plan.nslices * 64].view(torch.int32))
            assert np.array_equal(perm[:n], np.argsort(lvl, kind="stable"))
            assert np.all(perm[n:] == -1)

    for nrhs in (1, 2):
        b = rng.standard_normal((n, nrhs))
        e = np.zeros_like(b)
        (oracle.ref_lower_trs_solve if which == "lower" else oracle.ref_upper_trs_solve)(
            n, nrhs, rp, ci, v, int(unit), b, nrhs, e, nrhs)
        x = trs_plan(gk, which, n, rp, ci, v, unit, b, check)
        assert np.array_equal(x, e)


def test_trs_plan_rows_longer_than_the_register_window(gk, oracle):
    # up to 40 dependencies per row: several windows of 8, padding inside a slice
    n = 3000
    rp, ci, v = random_triangular(n, True, seed=9, max_off=40, band=300)
    b = np.random.default_rng(2).standard_normal((n, 1))
    e = np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, rp, ci, v, 0, b, 1, e, 1)
    assert np.array_equal(trs_plan(gk, "lower", n, rp, ci, v, False, b), e)


def test_trs_plan_long_dependency_chain(gk, oracle):
    # every row its own level: slices straddle 64 levels, lanes publish one by one
    n = 3000
    rp = np.arange(0, 2 * n + 1, 2, dtype=np.int32) - 1
    rp[0] = 0
    ci = np.empty(2 * n - 1, np.int32)
    v = np.empty(2 * n - 1)
    ci[0], v[0] = 0, 2.0
    ci[1::2], v[1::2] = np.arange(0, n - 1), -1.0
    ci[2::2], v[2::2] = np.arange(1, n), 2.0
    b = np.ones((n, 1))
    e = np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, rp, ci, v, 0, b, 1, e, 1)
    assert np.array_equal(trs_plan(gk, "lower", n, rp, ci, v, False, b,
                                   lambda plan: (plan.nlevels == n) or pytest.fail("levels")), e)


def test_trs_plan_nan_results_do_not_hang(gk, oracle):
    rp, ci, v = matgen.dense_to_csr([[1e-300, 0, 0], [1.0, 2.0, 0], [1.0, 1.0, 3.0]])
    v = v.copy()
    v[0] = 0.0
    x = trs_plan(gk, "lower", 3, rp, ci, v, False, np.zeros((3, 1)))
    assert np.isnan(x).all()


def test_trs_plan_ilu0_apply_and_refresh(gk, oracle):
    """Ilu::apply with analysed factors on config 4's structure; new values through the
    numeric phase alone (the symbolic analysis depends on the sparsity pattern only)."""
    import gkomi.solvers as solvers
    n, rp, ci, v = matgen.poisson_3d_7pt(24)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    lrp, lc, lv = f["L"]
    urp, uc, uv = f["U"]
    b = np.sin(0.1 * np.arange(n)).reshape(n, 1)
    y, e = np.zeros_like(b), np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, lrp, lc, lv, 0, b, 1, y, 1)
    oracle.ref_upper_trs_solve(n, 1, urp, uc, uv, 0, y, 1, e, 1)
    pl = solvers.TrsPlan(gk, n, dev(lrp), dev(lc), dev(lv), True)
    pu = solvers.TrsPlan(gk, n, dev(urp), dev(uc), dev(uv), False)
    assert pl.nlevels == pu.nlevels == 3 * 23 + 1
    yd = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    xd = torch.zeros_like(yd)
    pl.solve(dev(b), yd)
    pu.solve(yd, xd)
    assert np.array_equal(host(yd), y) and np.array_equal(host(xd), e)
    lv2 = lv * 1.25
    y2 = np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, lrp, lc, lv2, 0, b, 1, y2, 1)
    pl.refresh(dev(lv2))
    pl.solve(dev(b), yd)
    assert np.array_equal(host(yd), y2) and not pl.overrun() and not pu.overrun()


@pytest.mark.parametrize("analyse", [False, "force", "bricks"])
def test_trs_overrun_is_sticky_and_surfaces_as_an_error(gk, oracle, monkeypatch, analyse):
    """a triangular solve that gives up used to erase its own flag at the next
    solve and no driver looked at it.  Now the flag is sticky and the solver drivers return
    GKOMI_ETRS_OVERRUN for an Ilu preconditioner whose solves gave up.
    The contract is the reference's `nan_produced` guard, cuda/solver/common_trs_kernels.cuh:444-449.
    Independent of the library's default brick shape: "force" rules the brick plan
    out (level plan), "bricks" asks for bricks of 256 rows and checks there is a hand-off to give up on."""
    import gkomi
    import gkomi.solvers as solvers
    # (more than 4096 rows: a smaller factor is solved by one workgroup behind barriers and has nothing to give up on)
    n, rp, ci, v = matgen.poisson_2d_5pt(72)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    L = tuple(dev(a) for a in f["L"])
    U = tuple(dev(a) for a in f["U"])

    def make():
        if analyse == "bricks":
            p = solvers.ilu_from_factors(gk, n, L, U, analyse=True, bricks="force", brick_rows=256)
            assert p.l_bricks is not None and p.u_bricks is not None
            assert p.l_bricks.nbricks > 1 and p.u_bricks.nbricks > 1   # there is a brick-to-brick hand-off to poll
            assert p.l_bricks.coarse_levels > 1
        else:
            p = solvers.ilu_from_factors(gk, n, L, U, analyse=analyse, bricks=False)
            assert p.l_bricks is None and p.u_bricks is None
            assert (p.l_plan is not None) == (analyse == "force")
        return p

    pre = make()
    b = dev(np.ones(n))
    ok = solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), b, krylov_dim=30, max_iters=200, reduction=1e-10, precond=pre)
    assert ok["converged"]
    monkeypatch.setenv("GKOMI_TRS_MAX_ROUNDS", "1")      # test hook: give up after one look at a dependency
    monkeypatch.setenv("GKOMI_TRS_MAX_POLLS", "1")       # ... the same for the brick plan
    with pytest.raises(gkomi._lib.GkomiError) as e:
        solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), b, krylov_dim=30, max_iters=20, reduction=1e-10, precond=pre)
    assert e.value.code == -6                            # GKOMI_ETRS_OVERRUN
    monkeypatch.delenv("GKOMI_TRS_MAX_ROUNDS")
    monkeypatch.delenv("GKOMI_TRS_MAX_POLLS")
    # sticky: a later, healthy solve with the same preconditioner still reports it
    with pytest.raises(gkomi._lib.GkomiError):
        solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), b, max_iters=5, reduction=1e-10, precond=pre)
    # a fresh preconditioner (fresh workspace / plans) is healthy again
    pre2 = make()
    assert solvers.gmres_solve(gk, n, dev(rp), dev(ci), dev(v), b, krylov_dim=30, max_iters=200, reduction=1e-10,
                               precond=pre2)["converged"]


def test_trs_single_brick_has_nothing_to_wait_for(gk, oracle, monkeypatch):
    """The complement of the overrun test: a factor that fits ONE brick has no inflow, so even with
    the polls cut to one the solve finishes, bit-exact, and raises no flag."""
    import gkomi.solvers as solvers
    n, rp, ci, v = matgen.poisson_2d_5pt(24)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    lrp, lc, lv = f["L"]
    bk = solvers.TrsBricks(gk, n, dev(lrp), dev(lc), dev(lv), True, brick_rows=1024)
    assert bk.nbricks == 1
    b = np.sin(0.1 * np.arange(n)).reshape(n, 1)
    y = np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, lrp, lc, lv, 0, b, 1, y, 1)
    monkeypatch.setenv("GKOMI_TRS_MAX_POLLS", "1")
    yd = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    bk.solve(dev(b), yd)
    assert np.array_equal(host(yd), y) and not bk.overrun()


@pytest.mark.parametrize("which", ["lower", "upper"])
@pytest.mark.parametrize("n", [2, 511, 512, 513, 1024, 1025, 2049, 4096, 4097])
def test_trs_plan_small_factors_one_workgroup(gk, oracle, which, n):
    """Factors of up to 4096 rows with at most 8 dependencies per row are solved by ONE workgroup (x in LDS, the
    rows' dependencies in registers, a barrier per level: trs_small_solve_kernel); 1 / 2 / 4 / 8 positions per thread
    at the sizes around 512, 1024, 2048; 4097 rows take the level-scheduled waves.  Bit-exact either way, two
    right-hand sides, unit and stored diagonal; gkomi_trs_use_plan sends every one of them to the plan."""
    rp, ci, v = random_triangular(n, which == "lower", seed=n, max_off=8, band=min(n, 300))
    rng = np.random.default_rng(n)
    import gkomi.solvers as solvers
    plan = solvers.TrsPlan(gk, n, dev(rp), dev(ci), dev(v), which == "lower")
    assert plan.max_deps <= 8 and bool(gk.trs_use_plan(n, plan.nlevels, plan.max_deps)) == (n <= 4096 or n >= 64 * plan.nlevels)
    for unit in (False, True):
        b = rng.standard_normal((n, 2))
        e = np.zeros_like(b)
        (oracle.ref_lower_trs_solve if which == "lower" else oracle.ref_upper_trs_solve)(n, 2, rp, ci, v, int(unit), b, 2, e, 2)
        x = torch.full((n, 2), 777.0, dtype=torch.float64, device="cuda:0")
        plan.solve(dev(b), x, unit)
        assert np.array_equal(host(x), e), unit
    assert not plan.overrun()


def test_trs_small_factor_of_the_reference_test_matrix_ani4(gk, oracle):
    """ILU(0) factors of matrices/test/ani4.mtx (3081 rows, irregular FEM pattern, 183 levels of 17 rows): Ilu::apply
    through the plan the generate step picks -- pieces of the level order as bricks (thin factor), or the single-workgroup
    solve where the cost model prefers it -- bit-exact against the oracle's two solves; and the single-workgroup solve
    itself on the same factors."""
    import gkomi.solvers as solvers
    kind, n, nc, rows, cols, vals = matgen.read_mtx(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ani4.mtx"))
    rp, ci, v = matgen.coo_to_csr(n, rows, cols, vals)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    pre = solvers.ilu_from_factors(gk, n, tuple(dev(a) for a in f["L"]), tuple(dev(a) for a in f["U"]))
    assert (pre.l_plan is not None or pre.l_bricks is not None) and (pre.u_plan is not None or pre.u_bricks is not None)
    b = np.sin(0.1 * np.arange(n)).reshape(n, 1) + 2.0
    y, e = np.zeros_like(b), np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, *f["L"], 0, b, 1, y, 1)
    oracle.ref_upper_trs_solve(n, 1, *f["U"], 0, y, 1, e, 1)
    z = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    pre.apply(dev(b), z)
    assert np.array_equal(host(z), e)
    pre2 = solvers.ilu_from_factors(gk, n, tuple(dev(a) for a in f["L"]), tuple(dev(a) for a in f["U"]), bricks=False)
    assert pre2.l_plan is not None and pre2.u_plan is not None       # <= 4096 rows: the single-workgroup solve of the level plan
    z.zero_()
    pre2.apply(dev(b), z)
    assert np.array_equal(host(z), e)
