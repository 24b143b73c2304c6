"""GPU parity tests of the CSR SpMV path (through the C ABI) against the
oracle.  Mirrors reference/test/matrix/csr_kernels.cpp:358-452 (known
answers) and test/matrix/csr_kernels2.cpp:228-455 (every strategy x
{simple, advanced} x {1, 3 rhs} x {sorted, unsorted}) on the same 532x231
random shape (csr_kernels2.cpp:73-77)."""
import json
import os

import numpy as np
import pytest
import torch

import matgen
from gpu_util import DevCsr, csr_apply, dev, host, sync

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAM, VECTOR = 1, 2
STRATEGIES = {"auto": 0, "stream": STREAM, "stream_v1": STREAM | (1 << 8),
              "stream_v2": STREAM | (2 << 8), "stream_v3": STREAM | (3 << 8),
              "stream_v4": STREAM | (4 << 8), "stream_noswz": STREAM | (1 << 16),
              "stream_v5": STREAM | (5 << 8), "stream_v8": STREAM | (8 << 8), "stream_v9": STREAM | (9 << 8),
              "stream_v13": STREAM | (13 << 8), "stream_v11_noswz": STREAM | (11 << 8) | (1 << 16),
              "stream_spmm": STREAM | (20 << 8), "stream_v15_pad": STREAM | (15 << 8),
              "stream_v16_nt_pad": STREAM | (16 << 8) | (1 << 16), "stream_v14_nt": STREAM | (14 << 8) | (1 << 16),
              "vector": VECTOR, "vector64": VECTOR | (64 << 8), "vector2": VECTOR | (2 << 8), "balanced": 3}
BITEXACT = {k for k in STRATEGIES if k.startswith("stream") or k == "auto"}


@pytest.mark.parametrize("strategy", sorted(STRATEGIES))
def test_known_answers(gk, oracle, strategy):
    g = json.load(open(os.path.join(G, "csr_spmv.json")))
    for case in g["cases"]:
        m = g["matrices"][case["matrix"]]
        A = DevCsr(m["nrows"], m["ncols"], m["row_ptrs"], m["col_idxs"], m["vals"])
        b = dev(np.array(case["b"], np.float64))
        if "alpha" in case:
            c = dev(np.array(case["c"], np.float64))
            csr_apply(gk, A, b, c, case["alpha"], case["beta"], STRATEGIES[strategy])
        else:
            c = csr_apply(gk, A, b, strategy=STRATEGIES[strategy])
        # EXPECT_EQ in the reference: exact (small integers/halves: any order is exact)
        assert np.array_equal(host(c), np.array(case["expect"])), (strategy, case["name"])


def _oracle_apply(oracle, n, rp, ci, v, b, c=None, alpha=None, beta=None):
    nrhs = b.shape[1]
    if alpha is None:
        out = np.full((n, nrhs), np.nan)
        oracle.ref_csr_spmv(n, nrhs, rp, ci, v, b, nrhs, out, nrhs)
    else:
        out = c.copy()
        oracle.ref_csr_advanced_spmv(n, nrhs, alpha, rp, ci, v, b, nrhs, beta, out, nrhs)
    return out


@pytest.mark.parametrize("sort", [True, False], ids=["sorted", "unsorted"])
@pytest.mark.parametrize("nrhs", [1, 3, 7])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("strategy", sorted(STRATEGIES))
def test_random_532x231(gk, oracle, strategy, advanced, nrhs, sort):
    rp, ci, v = matgen.random_csr(532, 231, 1, 231, seed=42, sort=sort)
    rng = np.random.default_rng(15)
    b = rng.standard_normal((231, nrhs))
    c0 = rng.standard_normal((532, nrhs))
    A = DevCsr(532, 231, rp, ci, v)
    if advanced:
        expect = _oracle_apply(oracle, 532, rp, ci, v, b, c0, 2.0, -1.0)
        got = host(csr_apply(gk, A, dev(b), dev(c0), 2.0, -1.0, STRATEGIES[strategy]))
    else:
        expect = _oracle_apply(oracle, 532, rp, ci, v, b)
        got = host(csr_apply(gk, A, dev(b), strategy=STRATEGIES[strategy]))
    if strategy in BITEXACT and not (strategy == "auto"):
        assert np.array_equal(got, expect)  # same summation order as the reference
    else:
        # r<double> = 10 eps is the reference's bound for tree-ordered sums
        assert matgen.rel_err(got, expect) <= 1e-14


@pytest.mark.parametrize("strategy", ["stream", "stream_v1", "stream_v3", "stream_v5", "stream_v8", "stream_v11_noswz", "stream_v15_pad",
                                      "stream_v16_nt_pad", "vector", "balanced", "auto"])
def test_ragged_and_empty_rows(gk, oracle, strategy):
    # empty rows, rows longer than one LDS tile (8192), an empty last row
    rng = np.random.default_rng(7)
    counts = np.array([0, 0, 5, 9000, 1, 0, 20000, 3, 0], dtype=np.int64)
    ncols = 25000
    rp = np.zeros(len(counts) + 1, np.int32)
    np.cumsum(counts, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(ncols, size=k, replace=False)) for k in counts]).astype(np.int32)
    v = rng.standard_normal(int(rp[-1]))
    b = rng.standard_normal((ncols, 1))
    A = DevCsr(len(counts), ncols, rp, ci, v)
    expect = _oracle_apply(oracle, len(counts), rp, ci, v, b)
    got = host(csr_apply(gk, A, dev(b), strategy=STRATEGIES[strategy]))
    if strategy.startswith("stream"):
        assert np.array_equal(got, expect)
    else:
        assert matgen.rel_err(got, expect) <= 1e-14
    assert got[0, 0] == 0.0 and got[-1, 0] == 0.0


def test_simple_apply_never_reads_c(gk, oracle):
    # beta = 0 semantics: NaNs in c must not propagate (SURVEY 8b numerical contract)
    n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)
    b = np.random.default_rng(3).standard_normal((n, 1))
    A = DevCsr(n, n, rp, ci, v)
    for s in ("stream", "vector"):
        c = torch.full((n, 1), float("nan"), dtype=torch.float64, device="cuda:0")
        csr_apply(gk, A, dev(b), c, strategy=STRATEGIES[s])
        assert np.array_equal(host(c), _oracle_apply(oracle, n, rp, ci, v, b)) or s == "vector"
        assert not np.isnan(host(c)).any()


def test_strided_rhs_and_output(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(29, 31)
    rng = np.random.default_rng(5)
    bfull = rng.standard_normal((n, 5))
    cfull = np.full((n, 4), -7.0)
    A = DevCsr(n, n, rp, ci, v)
    bd, cd = dev(bfull), dev(cfull)
    # 2 rhs living in a 5-wide / 4-wide buffer
    gk.csr_spmv_f64_i32(torch.cuda.current_stream().cuda_stream, n, n, 2, A.nnz, A.row_ptrs, A.col_idxs,
                        A.vals, bd, 5, cd, 4, None, None, STREAM, 5)
    expect = np.full((n, 2), np.nan)
    oracle.ref_csr_spmv(n, 2, rp, ci, v, bfull, 5, expect, 2)
    got = host(cd)
    assert np.array_equal(got[:, :2], expect)
    assert np.all(got[:, 2:] == -7.0)


@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("nrhs,b_stride,c_stride", [(2, 2, 2), (4, 4, 4), (4, 5, 7), (5, 6, 6), (8, 8, 8), (11, 12, 11), (15, 16, 16), (17, 17, 18)])
def test_several_right_hand_sides_read_the_matrix_once(gk, oracle, nrhs, b_stride, c_stride, advanced):
    """The multi-rhs kernel (4 / 2 columns per pass + single-column remainder),
    aligned (16-B loads of b) and odd strides, rows longer than one LDS tile,
    empty rows: bit-exact against the oracle per (row, column)."""
    rng = np.random.default_rng(nrhs * 100 + b_stride)
    counts = np.concatenate([rng.integers(0, 9, size=700), [0, 4000, 0, 1, 1700], rng.integers(0, 40, size=300)])
    ncols = 5000
    rp = np.zeros(len(counts) + 1, np.int32)
    np.cumsum(counts, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(ncols, size=k, replace=False)) for k in counts]).astype(np.int32)
    v = rng.standard_normal(int(rp[-1]))
    n = len(counts)
    bfull = rng.standard_normal((ncols, b_stride))
    cfull = rng.standard_normal((n, c_stride))
    A = DevCsr(n, ncols, rp, ci, v)
    bd, cd = dev(bfull), dev(cfull)
    expect = cfull.copy()
    if advanced:
        al, be = dev(np.array([0.5])), dev(np.array([-2.0]))
        oracle.ref_csr_advanced_spmv(n, nrhs, 0.5, rp, ci, v, bfull, b_stride, -2.0, expect, c_stride)
    else:
        al = be = None
        oracle.ref_csr_spmv(n, nrhs, rp, ci, v, bfull, b_stride, expect, c_stride)
    for strategy in (0, STRATEGIES["stream_spmm"]):
        cd.copy_(dev(cfull))
        gk.csr_spmv_f64_i32(torch.cuda.current_stream().cuda_stream, n, ncols, nrhs, A.nnz, A.row_ptrs, A.col_idxs,
                            A.vals, bd, b_stride, cd, c_stride, al, be, strategy, 0 if strategy else 4000)
        if strategy == 0:
            # automatic with a 4000-long row picks the sub-wave kernel: tolerance
            assert matgen.rel_err(host(cd)[:, :nrhs], expect[:, :nrhs]) <= 1e-14
        else:
            assert np.array_equal(host(cd), expect)  # padding columns untouched too


def test_full_size_poisson_four_right_hand_sides(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    rng = np.random.default_rng(4)
    b = rng.standard_normal((n, 4))
    A = DevCsr(n, n, rp, ci, v)
    expect = _oracle_apply(oracle, n, rp, ci, v, b)
    got = host(csr_apply(gk, A, dev(b), strategy=0))
    assert np.array_equal(got, expect)
    # each column equals the single-column apply
    for j in range(4):
        col = host(csr_apply(gk, A, dev(np.ascontiguousarray(b[:, j:j + 1])), strategy=0))
        assert np.array_equal(col[:, 0], got[:, j])


def test_misaligned_arrays_fall_back(gk, oracle):
    # sub-views that break the 16-B/8-B alignment the stream kernel wants
    n, rp, ci, v = matgen.poisson_2d_5pt(23, 19)
    b = np.random.default_rng(9).standard_normal((n, 1))
    vals_buf = dev(np.concatenate([[0.0], v]))
    cols_buf = dev(np.concatenate([[0], ci]).astype(np.int32))
    rpd = dev(rp)
    c = torch.empty((n, 1), dtype=torch.float64, device="cuda:0")
    gk.csr_spmv_f64_i32(torch.cuda.current_stream().cuda_stream, n, n, 1, int(rp[-1]), rpd, cols_buf[1:],
                        vals_buf[1:], dev(b), 1, c, 1, None, None, 0, 5)
    assert matgen.rel_err(host(c), _oracle_apply(oracle, n, rp, ci, v, b)) <= 1e-14


def test_max_row_nnz(gk):
    rp, ci, v = matgen.random_csr(10000, 500, 0, 77, seed=1)
    out = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    gk.csr_max_row_nnz_i32(torch.cuda.current_stream().cuda_stream, 10000, dev(rp), out)
    assert int(out.item()) == int(np.max(np.diff(rp)))


def test_full_size_poisson_p2_bitexact_and_linear(gk, oracle):
    """BASELINE config[1]: 1M-row 5-pt Poisson.  Bit-exact against the oracle
    (it finishes in ~30 ms), plus size-independent properties: A*1 vanishes in
    the interior, linearity, run-to-run determinism."""
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    assert n == 1_000_000 and rp[-1] == 4_996_000
    x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
    A = DevCsr(n, n, rp, ci, v)
    xd = dev(x)
    expect = _oracle_apply(oracle, n, rp, ci, v, x)
    for s in ("stream", "stream_v1", "stream_v2", "stream_noswz", "stream_v5", "stream_v8", "stream_v9", "stream_v13",
              "stream_v14_nt", "stream_v15_pad", "stream_v16_nt_pad"):
        got = host(csr_apply(gk, A, xd, strategy=STRATEGIES[s]))
        assert np.array_equal(got, expect), s
    got2 = host(csr_apply(gk, A, xd, strategy=STRATEGIES["stream"]))
    assert np.array_equal(got2, expect)  # deterministic
    ones = torch.ones((n, 1), dtype=torch.float64, device="cuda:0")
    y1 = host(csr_apply(gk, A, ones, strategy=0)).reshape(1000, 1000)
    assert np.all(y1[1:-1, 1:-1] == 0.0) and y1[0, 0] == 2.0 and y1[0, 1] == 1.0
    # linearity: A(2x + 1) == 2 A x + A 1 up to rounding
    y = host(csr_apply(gk, A, 2 * xd + ones, strategy=0))
    assert matgen.rel_err(y, 2 * expect + y1.reshape(n, 1)) <= 1e-14
    got_v = host(csr_apply(gk, A, xd, strategy=STRATEGIES["vector"]))
    assert matgen.rel_err(got_v, expect) <= 1e-14


@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("nrhs", [1, 2])
def test_load_balanced_on_skewed_matrices(gk, oracle, advanced, nrhs):
    """A few rows hold most of the nonzeros (power-law like): the automatic
    strategy must pick the nonzero-split kernel and agree with the oracle."""
    rng = np.random.default_rng(21)
    n, ncols = 20000, 30000
    counts = rng.integers(0, 6, size=n)
    counts[rng.choice(n, size=12, replace=False)] = rng.integers(8000, 25000, size=12)
    counts[0] = 0
    counts[-1] = 17000
    rp = np.zeros(n + 1, np.int32)
    np.cumsum(counts, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(ncols, size=k, replace=False)) for k in counts]).astype(np.int32)
    v = rng.standard_normal(int(rp[-1]))
    b = rng.standard_normal((ncols, nrhs))
    c0 = rng.standard_normal((n, nrhs))
    A = DevCsr(n, ncols, rp, ci, v)
    assert A.max_row_nnz > 64 * (A.nnz // n + 1)
    for strat in ("balanced", "auto"):
        if advanced:
            expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, -0.5, 2.0)
            got = host(csr_apply(gk, A, dev(b), dev(c0), -0.5, 2.0, STRATEGIES[strat]))
        else:
            expect = _oracle_apply(oracle, n, rp, ci, v, b)
            got = host(csr_apply(gk, A, dev(b), strategy=STRATEGIES[strat]))
        assert matgen.rel_err(got, expect) <= 1e-14, strat
        assert got[0, 0] == (2.0 * c0[0, 0] if advanced else 0.0)  # empty row: just beta*c / 0
