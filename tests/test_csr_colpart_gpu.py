"""GPU parity tests of the column-partitioned CSR copy (gkomi_csr_colpart_*, csrc/csr_colpart.hip; strategy "csrp" of
the Python mirror): csr::spmv / advanced_spmv (reference/matrix/csr_kernels.cpp:75-128) with every (row, column
block) group added left to right and a row's groups in block order -- tolerance parity like the load-balanced
strategy (the bar: 4 eps sqrt(longest row) relative to sum |a_ij b_j| per entry), bit-exact where a row lies in one
block."""
import ctypes

import numpy as np
import pytest
import torch

import matgen
from gkomi import formats
from gpu_util import dev, host, stream_ptr
from test_csr_spmv_gpu import _oracle_apply

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def bound(rp, ci, v, b, alpha=1.0):
    """sum_j |alpha a_ij b_j| per row: the scale of the rounding error of any summation order"""
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    return np.bincount(rows, weights=np.abs(alpha * v * b[ci, 0]), minlength=n).reshape(n, 1)


def powerlaw(n, ncols, seed):
    rng = np.random.default_rng(seed)
    counts = np.minimum(n // 4, np.maximum(1, (4 * rng.pareto(1.3, size=n)).astype(np.int64)))
    counts[17] = 60000
    counts[n // 2: n // 2 + 3000] = 0      # a run of empty rows: empty virtual rows in every block
    return matgen.random_rows_csr(n, ncols, counts, seed)


@pytest.mark.parametrize("nb", [2, 4, 8])
@pytest.mark.parametrize("kind", ["uniform", "powerlaw"])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_partitioned_apply_matches_oracle(gk, oracle, kind, nb, advanced):
    n = ncols = 120000
    if kind == "uniform":
        rng = np.random.default_rng(3)
        rp, ci, v = matgen.random_rows_csr(n, ncols, rng.integers(4, 20, size=n), 4)
    else:
        rp, ci, v = powerlaw(n, ncols, 5)
    rng = np.random.default_rng(9)
    b = rng.standard_normal((ncols, 1))
    c0 = rng.standard_normal((n, 1))
    M = formats.Csr.from_host(gk, n, ncols, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    assert M.colpart(nb) is not None
    info = (ctypes.c_int64 * 8)()
    gk.csr_colpart_info(M._colpart[0], ctypes.addressof(info))
    assert info[0] == nb and info[1] == nb * n
    if advanced:
        expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, -0.5, 2.0)
        got = host(M.apply(dev(b), dev(c0), -0.5, 2.0))
        scale = bound(rp, ci, v, b, 0.5) + np.abs(2.0 * c0)
    else:
        expect = _oracle_apply(oracle, n, rp, ci, v, b)
        got = host(M.apply(dev(b), torch.full((n, 1), np.nan, dtype=torch.float64, device="cuda:0")))
        scale = bound(rp, ci, v, b)
    longest = int(np.diff(rp).max())
    assert np.all(np.abs(got - expect) <= 4 * EPS * np.sqrt(longest) * scale + 1e-300)
    # rows whose columns all lie in one block: one group, the reference's own order and bits (simple apply)
    if not advanced:
        width = -(-ncols // nb)
        rows = np.repeat(np.arange(n), np.diff(rp))
        lo = np.full(n, nb, np.int64); hi = np.full(n, -1, np.int64)
        np.minimum.at(lo, rows, ci // width); np.maximum.at(hi, rows, ci // width)
        one_block = (lo == hi) & (np.diff(rp) <= 64)   # (longer rows may be cut by a tile of the load-balanced kernel)
        if kind == "uniform":
            assert np.array_equal(got[one_block], expect[one_block])
        empty = np.diff(rp) == 0
        assert np.array_equal(got[empty], np.zeros((int(empty.sum()), 1)))


def test_policy_and_refresh(gk, oracle):
    """blocks_for: only shapes where b overflows an L2 but not 16 MB, with enough nonzeros per row; refresh re-gathers
    changed values; a strided b / c; the mirror falls back to the automatic kernels where the copy does not pay"""
    assert gk.csr_colpart_blocks_for(1000000, 1000000, 16000000) == 4
    assert gk.csr_colpart_blocks_for(1000000, 300000, 16000000) == 0       # b within an L2
    assert gk.csr_colpart_blocks_for(1000000, 1500000, 16000000) == 8
    assert gk.csr_colpart_blocks_for(4000000, 4000000, 64000000) == 8      # 32 MB of b: 8 slices of 4 MB
    assert gk.csr_colpart_blocks_for(9000000, 9000000, 144000000) == 0     # 72 MB of b: slices beyond 6 MiB
    assert gk.csr_colpart_blocks_for(3000000, 3000000, 24000000) == 4      # 8 per row: no more than 4 blocks
    assert gk.csr_colpart_blocks_for(1000000, 1000000, 5000000) == 4       # 5 per row: groups of 1.25 (the timing decides)
    assert gk.csr_colpart_blocks_for(1000000, 1000000, 4000000) == 2       # 4 per row
    assert gk.csr_colpart_blocks_for(1000000, 1000000, 3000000) == 0       # 3 per row
    n = ncols = 600000
    rng = np.random.default_rng(1)
    rp, ci, v = matgen.random_rows_csr(n, ncols, rng.integers(6, 12, size=n), 2)
    M = formats.Csr.from_host(gk, n, ncols, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    assert gk.csr_colpart_blocks_for(n, ncols, M.nnz) == 4
    if M.colpart() is None:      # the TIMED analysis declined (a noisy box): the rest of the test needs a copy
        M._colpart = None
        assert M.colpart(4) is not None
    info = (ctypes.c_int64 * 8)()
    gk.csr_colpart_info(M._colpart[0], ctypes.addressof(info))
    assert info[0] in (2, 4) and info[1] == info[0] * n     # the analysis timed both and kept one
    b2 = rng.standard_normal((ncols, 3))
    bd, cd = dev(b2), torch.zeros((n, 2), dtype=torch.float64, device="cuda:0")
    M.apply(bd[:, 1:2], cd[:, 0:1])            # strides 3 and 2
    expect = _oracle_apply(oracle, n, rp, ci, v, np.ascontiguousarray(b2[:, 1:2]))
    assert matgen.rel_err(host(cd[:, 0:1]), expect) <= 1e-14 and not host(cd[:, 1]).any()
    v2 = v * rng.uniform(0.5, 2.0, size=len(v))
    M.vals.copy_(dev(v2))
    M.values_changed()
    got = host(M.apply(bd[:, 1:2], cd[:, 0:1]))
    assert matgen.rel_err(got, _oracle_apply(oracle, n, rp, ci, v2, np.ascontiguousarray(b2[:, 1:2]))) <= 1e-14
    # a banded matrix of a fitting shape: the gather statistic says "not scattered", no copy, the automatic kernel's bits
    rp, ci, v = matgen.random_rows_csr(n, ncols, rng.integers(6, 12, size=n), 8, local=500)
    Bd = formats.Csr.from_host(gk, n, ncols, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    assert gk.csr_colpart_blocks_for(n, ncols, Bd.nnz) == 4 and Bd.colpart() is None
    bb = rng.standard_normal((ncols, 1))
    assert np.array_equal(host(Bd.apply(dev(bb), torch.zeros((n, 1), dtype=torch.float64, device="cuda:0"))),
                          _oracle_apply(oracle, n, rp, ci, v, bb))
    # small matrix: the strategy keeps the automatic kernel (bit-exact)
    rp, ci, v = matgen.random_csr(500, 400, 0, 9, seed=3)
    S = formats.Csr.from_host(gk, 500, 400, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    bb = rng.standard_normal((400, 1))
    assert S.colpart() is None
    assert np.array_equal(host(S.apply(dev(bb), torch.zeros((500, 1), dtype=torch.float64, device="cuda:0"))),
                          _oracle_apply(oracle, 500, rp, ci, v, bb))


def test_create_rejects_bad_arguments(gk):
    from gkomi._lib import GkomiError
    rp, ci, v = matgen.random_csr(100, 100, 1, 5, seed=1)
    plan = torch.empty(int(gk.csr_colpart_plan_bytes(100, len(v), 4)), dtype=torch.uint8, device="cuda:0")
    h = ctypes.c_void_p(0)
    for nb in (1, 3, 16):
        with pytest.raises(GkomiError):
            gk.csr_colpart_create_f64_i32(stream_ptr(), 100, 100, len(v), dev(rp), dev(ci), dev(v), nb, plan, plan.numel(), ctypes.addressof(h))
    with pytest.raises(GkomiError):   # plan too small
        gk.csr_colpart_create_f64_i32(stream_ptr(), 100, 100, len(v), dev(rp), dev(ci), dev(v), 4, plan, 64, ctypes.addressof(h))
    assert h.value is None


@pytest.mark.parametrize("solver", ["gmres", "bicgstab"])
def test_solver_drivers_take_the_copy_as_system_matrix(gk, solver):
    """gkomi_csr_colpart_matrix_apply_cb behind the *_solve_op_f64 drivers (a solve holds its matrix const: the copy
    cannot go stale there): same convergence as with the plain CSR matrix, same solution to the solve's tolerance"""
    from gkomi import solvers
    n = 500000
    rng = np.random.default_rng(12)
    rp, ci, v = matgen.random_rows_csr(n, n, np.full(n, 8), 13)
    v = 0.1 * rng.standard_normal(len(v))
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[rows == ci] = 0.0
    # a strongly diagonally dominant nonsymmetric system: add 2 to the diagonal through one more CSR entry per row
    rp2 = rp + np.arange(n + 1, dtype=rp.dtype)
    ci2 = np.insert(ci, rp[1:], np.arange(n)).astype(np.int32)   # (appended at the row's end: unsorted rows are fine)
    v2 = np.insert(v, rp[1:], 2.0)
    b = dev(rng.standard_normal((n, 1)))
    out = {}
    for name in ("csr", "csrp"):
        M = formats.Csr.from_host(gk, n, n, rp2, ci2, v2, strategy=formats.Csr.CSR_STRATEGIES[name])
        if name == "csrp":
            assert M.colpart(2) is not None    # forced: with 4 MB of b the timed analysis may decline the copy
        out[name] = solvers.solve_op(gk, solver, M, b, max_iters=200, reduction=1e-10, krylov_dim=30)
        assert out[name]["converged"]
    assert abs(out["csr"]["iterations"] - out["csrp"]["iterations"]) <= 1
    assert matgen.rel_err(host(out["csrp"]["x"]), host(out["csr"]["x"])) <= 1e-8


def test_analysis_declines_where_the_copy_does_not_pay(gk, oracle):
    """A scattered pattern with few nonzeros per row (the T2-like randomly permuted 2-D matrix, ~5 per row on 1.2 M
    columns: blocks_for says 0; forced through a shape that passes blocks_for, the timed analysis decides): whatever it
    decides, the strategy's apply is the matrix's product"""
    n, rp, ci, v = matgen.t2_like_permuted(1108)
    M = formats.Csr.from_host(gk, n, n, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    rng = np.random.default_rng(4)
    b = rng.standard_normal((n, 1))
    got = host(M.apply(dev(b), torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")))
    expect = _oracle_apply(oracle, n, rp, ci, v, b)
    if M.colpart() is None:
        assert np.array_equal(got, expect)       # the automatic kernel: the reference's bits
    else:
        assert matgen.rel_err(got, expect) <= 1e-14
    # 6-7 per row, uniformly random on 600 k columns: passes the shape test; the analysis times it
    rp, ci, v = matgen.random_rows_csr(600000, 600000, rng.integers(6, 8, size=600000), 21)
    S = formats.Csr.from_host(gk, 600000, 600000, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    b = rng.standard_normal((600000, 1))
    got = host(S.apply(dev(b), torch.zeros((600000, 1), dtype=torch.float64, device="cuda:0")))
    assert matgen.rel_err(got, _oracle_apply(oracle, 600000, rp, ci, v, b)) <= 1e-14
    print("copy built for 6-7 per row:", S.colpart() is not None)
