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
from gpu_util import DevCsr, csr_apply, csr_apply_srow, dev, host, make_srow, sync

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
              "vector": VECTOR, "vector64": VECTOR | (64 << 8), "vector2": VECTOR | (2 << 8), "balanced": 3,
              "balanced_serial": 3 | (1 << 8)}
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
                                      "stream_v16_nt_pad", "vector", "balanced", "balanced_serial", "auto"])
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
    for strat in ("balanced", "balanced_serial", "auto"):
        if advanced:
            expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, -0.5, 2.0)
            got = host(csr_apply(gk, A, dev(b), dev(c0), -0.5, 2.0, STRATEGIES[strat]))
        else:
            expect = _oracle_apply(oracle, n, rp, ci, v, b)
            got = host(csr_apply(gk, A, dev(b), strategy=STRATEGIES[strat]))
        assert matgen.rel_err(got, expect) <= 1e-14, strat
        assert got[0, 0] == (2.0 * c0[0, 0] if advanced else 0.0)  # empty row: just beta*c / 0


@pytest.mark.parametrize("with_srow", [False, True], ids=["search", "srow"])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_load_balanced_with_long_runs_of_empty_rows(gk, oracle, advanced, with_srow):
    """Long rows AND a run of 300 000 empty rows inside one tile: that tile goes by its nonzeros (a row lookup per
    position) instead of walking the rows 256 at a time; also an empty run in front of the row that contains a
    tile's first nonzero, and trailing empty rows."""
    from gpu_util import csr_apply_srow, make_srow
    rng = np.random.default_rng(5)
    n, ncols = 420000, 50000
    counts = rng.integers(0, 5, size=n)
    counts[2000:302000] = 0
    counts[302000] = 9000            # a long row right behind the run
    counts[310000:311500] = 1        # > 4 x 256 rows starting in one tile, none empty
    counts[350000] = 4000
    counts[400000:] = 0
    rp, ci, v = matgen.random_rows_csr(n, ncols, counts, seed=6)
    b = rng.standard_normal((ncols, 1))
    c0 = rng.standard_normal((n, 1))
    A = DevCsr(n, ncols, rp, ci, v)
    srow, tile = make_srow(gk, A) if with_srow else (None, 0)

    def run(c=None):
        if with_srow:
            return csr_apply_srow(gk, A, dev(b), srow, tile, c, *((-0.5, 2.0) if advanced else (None, None)), 3)
        return csr_apply(gk, A, dev(b), c, *((-0.5, 2.0) if advanced else (None, None)), STRATEGIES["balanced"])
    expect = _oracle_apply(oracle, n, rp, ci, v, b, *((c0, -0.5, 2.0) if advanced else ()))
    got = host(run(dev(c0) if advanced else None))
    assert matgen.rel_err(got, expect) <= 1e-14
    assert np.array_equal(got[2000:302000], 2.0 * c0[2000:302000] if advanced else np.zeros((300000, 1)))
    out = run(dev(c0) if advanced else None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(out)
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) / 10 < 0.6, "ms per apply (0.02-0.05 measured; 1.2 ms with the row walk)"


@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_load_balanced_long_rows_are_reduced_by_whole_waves(gk, oracle, advanced):
    """Power-law shape with a 200 k-nonzero row (130 tiles of one row), rows around the
    cooperative threshold (127 / 128 / 129 / 130 nonzeros), rows that end exactly on a tile
    boundary, an odd nnz: the load-balanced kernel adds segments of more than 128 products
    with a whole wave (lane-strided partial sums + tree), so the bar is the tolerance of a
    tree-ordered sum, 4 eps sqrt(longest row), per entry relative to sum |a_ij b_j|; rows of
    at most 128 nonzeros that lie inside one 1536-nonzero tile keep the reference's bits;
    two runs give the same bits for rows that are not cut by a tile boundary."""
    rng = np.random.default_rng(77)
    n, ncols = 30000, 400000
    counts = np.maximum(1, (4 * rng.pareto(1.3, size=n)).astype(np.int64))
    counts = np.minimum(counts, 5000)
    counts[17] = 200_000
    counts[100:104] = [127, 128, 129, 130]
    counts[5000] = 1536 * 3           # whole tiles
    counts[20000] = 0
    counts[-1] = 2001
    rp, ci, v = matgen.random_rows_csr(n, ncols, counts, seed=78)
    if int(rp[-1]) % 2 == 0:       # (the generator merges duplicate columns) odd nnz: drop the last nonzero
        rp = rp.copy()
        rp[-1] -= 1
        ci, v = ci[:-1], v[:-1]
    counts = np.diff(rp)
    nnz = int(rp[-1])
    assert nnz % 2 == 1 and counts.max() >= 150_000
    b = rng.standard_normal((ncols, 1))
    c0 = rng.standard_normal((n, 1))
    A = DevCsr(n, ncols, rp, ci, v)
    if advanced:
        expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, -0.5, 2.0)
        run = lambda s: host(csr_apply(gk, A, dev(b), dev(c0), -0.5, 2.0, STRATEGIES[s]))
        scale = np.abs(2.0 * c0)
    else:
        expect = _oracle_apply(oracle, n, rp, ci, v, b)
        run = lambda s: host(csr_apply(gk, A, dev(b), strategy=STRATEGIES[s]))
        scale = np.zeros((n, 1))
    # magnitude of each row's sum: sum |a_ij| |b_j| (+ |beta c|)
    absrow = np.zeros(n)
    np.add.at(absrow, np.repeat(np.arange(n), counts), np.abs(v) * np.abs(b[ci, 0]) * (0.5 if advanced else 1.0))
    bound = 4 * np.finfo(np.float64).eps * np.sqrt(counts.max()) * (absrow.reshape(n, 1) + scale) + 1e-300
    for strat in ("balanced", "auto"):
        got = run(strat)
        assert np.all(np.abs(got - expect) <= bound), (strat, int(np.argmax(np.abs(got - expect) / bound)))
    got = run("balanced")
    first_tile, last_tile = rp[:-1] // 1536, (np.maximum(rp[1:], rp[:-1] + 1) - 1) // 1536
    inside = first_tile == last_tile
    if not advanced:
        short = inside & (counts <= 128)
        assert np.array_equal(got[short], expect[short])     # the reference's order
    again = run("balanced")
    assert np.array_equal(again[inside], got[inside])        # fixed order: same bits


COLBLOCK = 1 << 25


@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("nrhs", [1, 2])
@pytest.mark.parametrize("with_srow", [False, True], ids=["search", "srow"])
def test_column_windows_of_the_load_balanced_kernel(gk, oracle, advanced, nrhs, with_srow):
    """GKOMI_CSR_COLBLOCK: one pass per 4 MiB window of b (1.2 M columns: 3 windows; two columns of b:
    5), partial sums added up in c.  Power-law rows with uniformly random columns, an empty row, a row of
    one tile and a half; b holds Inf at the first entry of the second window in a column no row touches
    (a product outside the window is replaced by +0.0, never multiplied by zero).  Tolerance of a
    re-grouped sum; the automatic strategy with the flag takes the same path."""
    rng = np.random.default_rng(5)
    n, ncols = 20000, 1_200_000
    counts = np.maximum(1, (3 * rng.pareto(1.3, size=n)).astype(np.int64))
    counts = np.minimum(counts, 3000)
    counts[7] = 0
    counts[11] = 2304
    rp, ci, v = matgen.random_rows_csr(n, ncols, counts, seed=6)
    counts = np.diff(rp)
    b = rng.standard_normal((ncols, nrhs))
    passes = -(-8 * ncols * nrhs // (4096 << 10))
    width = -(-ncols // passes)
    untouched = width if width not in set(ci.tolist()) else None
    if untouched is not None:
        b[untouched, :] = np.inf
    c0 = rng.standard_normal((n, nrhs))
    A = DevCsr(n, ncols, rp, ci, v)
    srow, tile = make_srow(gk, A) if with_srow else (None, 0)
    absrow = np.zeros(n)
    finite_b = np.where(np.isfinite(b), b, 0.0)
    for j in range(nrhs):
        np.add.at(absrow, np.repeat(np.arange(n), counts), np.abs(v) * np.abs(finite_b[ci, j]))
    for strat in (3 | COLBLOCK, COLBLOCK):
        if advanced:
            expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, -0.5, 2.0)
            got = host(csr_apply_srow(gk, A, dev(b), srow, tile, dev(c0), -0.5, 2.0, strat))
        else:
            expect = _oracle_apply(oracle, n, rp, ci, v, b)
            got = host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=strat))
        assert np.all(np.isfinite(got))
        bound = 4 * np.finfo(np.float64).eps * np.sqrt(counts.max()) * (absrow.reshape(n, 1) + np.abs(2.0 * c0)) + 1e-300
        assert np.all(np.abs(got - expect) <= bound), strat
        assert np.all(got[7] == (2.0 * c0[7] if advanced else 0.0))


def test_gather_analysis_flags_spread_columns_only(gk):
    """gkomi_csr_analyse_gather_i32: uniformly random columns over 1 M columns -> COLBLOCK (a tile's gathers
    range over ~8 MB of b); a banded matrix with one far column per row (arrow) and a stencil -> no flag;
    random columns over 200 k columns (1.6 MB of b: fits L2) -> no flag."""
    import ctypes
    def analyse(ncols, ci):
        scratch = torch.zeros(2, dtype=torch.float64, device="cuda:0")
        flags, foot = ctypes.c_int(-1), ctypes.c_int64(-1)
        gk.csr_analyse_gather_i32(torch.cuda.current_stream().cuda_stream, ncols, len(ci), dev(ci.astype(np.int32)), scratch,
                                  ctypes.addressof(flags), ctypes.addressof(foot))
        return flags.value, foot.value
    rng = np.random.default_rng(1)
    f, foot = analyse(1_000_000, np.sort(rng.integers(0, 1_000_000, size=(100_000, 8)), axis=1).ravel())
    assert f == COLBLOCK and 6_000_000 < foot <= 8_000_000
    rows = np.repeat(np.arange(100_000), 8)
    band = np.clip(rows + rng.integers(-50, 50, size=len(rows)), 0, 999_999)
    band[::8] = 999_999    # the arrow's column
    f, foot = analyse(1_000_000, band)
    assert f == 0
    n, rp, ci, v = matgen.poisson_2d_5pt(600)
    assert analyse(n, ci)[0] == 0
    assert analyse(200_000, rng.integers(0, 200_000, size=500_000))[0] == 0
    assert analyse(10, np.zeros(0, np.int32)) == (0, 0)
