"""GPU parity tests of the <double, int64> instantiation (gkomi_csr_*_i64): csr::spmv / advanced_spmv, Csr::make_srow,
the index conversions and CG on an int64 matrix, against the oracle run on the same matrix with int32 indices
(reference/matrix/csr_kernels.cpp:75-128 is the same loop for both index types: the bits must agree).  The
reference instantiates every kernel of the path for {int32, int64} (include/ginkgo/core/base/types.hpp:544-560);
int64 is the one a 288 GB GPU needs: the last test applies a matrix of more than 2^31 nonzeros."""
import numpy as np
import pytest
import torch

import gkomi.formats as formats
import gkomi.solvers as solvers
import matgen
from gpu_util import dev, host, stream_ptr

pytestmark = pytest.mark.gpu


def oracle_apply(oracle, n, rp, ci, v, b, c=None, alpha=None, beta=None):
    nrhs = b.shape[1]
    if alpha is None:
        out = np.full((n, nrhs), np.nan)
        oracle.ref_csr_spmv(n, nrhs, rp, ci, v, b, nrhs, out, nrhs)
    else:
        out = c.copy()
        oracle.ref_csr_advanced_spmv(n, nrhs, alpha, rp, ci, v, b, nrhs, beta, out, nrhs)
    return out


def known_answer_matrix():
    # reference/test/matrix/csr_kernels.cpp:358-400: [1 3 2; 0 5 0] (2, 1, 4)^T = (13, 5)^T; alpha -1, beta 2, y (1, 2) -> (-11, -1)
    return 2, 3, np.array([0, 3, 4]), np.array([0, 1, 2, 1]), np.array([1.0, 3.0, 2.0, 5.0])


@pytest.mark.parametrize("split", [False, True], ids=["stream", "split"])
def test_known_answers_int64(gk, split):
    n, m, rp, ci, v = known_answer_matrix()
    A = formats.Csr64.from_host(gk, n, m, rp, ci, v, split=split)
    b = dev(np.array([[2.0], [1.0], [4.0]]))
    y = A.apply(b, torch.full((2, 1), float("nan"), dtype=torch.float64, device="cuda:0"))
    assert np.array_equal(host(y), [[13.0], [5.0]])
    y = A.apply(b, dev(np.array([[1.0], [2.0]])), alpha=-1.0, beta=2.0)
    assert np.array_equal(host(y), [[-11.0], [-1.0]])


@pytest.mark.parametrize("split", [False, True], ids=["stream", "split"])
@pytest.mark.parametrize("nrhs", [1, 3])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("sort", [True, False], ids=["sorted", "unsorted"])
def test_random_532x231_int64(gk, oracle, split, nrhs, advanced, sort):
    """test/matrix/csr_kernels2.cpp:228-455 on the int64 instantiation: rows of 1..231 nonzeros (longer than the split
    kernel's 64-nonzero look-ahead: finished from memory), unsorted columns, several right-hand sides"""
    rp, ci, v = matgen.random_csr(532, 231, 1, 231, seed=42, sort=sort)
    rng = np.random.default_rng(15)
    b = rng.standard_normal((231, nrhs))
    c0 = rng.standard_normal((532, nrhs))
    A = formats.Csr64.from_host(gk, 532, 231, rp, ci, v, split=split)
    if advanced:
        expect = oracle_apply(oracle, 532, rp, ci, v, b, c0, 2.0, -1.0)
        got = host(A.apply(dev(b), dev(c0), alpha=2.0, beta=-1.0))
    else:
        expect = oracle_apply(oracle, 532, rp, ci, v, b)
        got = host(A.apply(dev(b), torch.full((532, nrhs), float("nan"), dtype=torch.float64, device="cuda:0")))
    assert np.array_equal(got, expect)


def test_ragged_and_empty_rows_int64(gk, oracle):
    rng = np.random.default_rng(7)
    counts = np.array([0, 0, 5, 9000, 1, 0, 20000, 3, 0], dtype=np.int64)
    ncols = 25000
    rp = np.zeros(len(counts) + 1, np.int32)
    np.cumsum(counts, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(ncols, size=k, replace=False)) for k in counts]).astype(np.int32)
    v = rng.standard_normal(int(rp[-1]))
    b = rng.standard_normal((ncols, 1))
    expect = oracle_apply(oracle, len(counts), rp, ci, v, b)
    for split in (False, True):
        A = formats.Csr64.from_host(gk, len(counts), ncols, rp, ci, v, split=split)
        got = host(A.apply(dev(b), torch.full((len(counts), 1), float("nan"), dtype=torch.float64, device="cuda:0")))
        assert np.array_equal(got, expect), split
    assert A.max_row_nnz() == 20000


def test_make_srow_and_index_conversions_int64(gk):
    rp, ci, v = matgen.random_csr(3000, 500, 0, 40, seed=3)
    nnz = int(rp[-1])
    rp64 = dev(rp.astype(np.int64))
    for tile in (1536, 2048, 3072):
        ne = int(gk.csr_srow_entries(nnz, tile))
        srow = torch.full((ne,), -1, dtype=torch.int64, device="cuda:0")
        gk.csr_make_srow_i64(stream_ptr(), 3000, nnz, rp64, tile, srow, ne)
        expect = np.minimum(np.searchsorted(rp, np.arange(ne - 1) * tile, side="left"), 3000)
        assert np.array_equal(host(srow)[:-1], expect)
        assert host(srow)[-1] == np.max(np.diff(expect))   # the most rows that start in one tile
    # ptrs -> idxs -> ptrs (reference/components/format_conversion_kernels.cpp:50-95)
    idxs = torch.full((nnz,), -1, dtype=torch.int64, device="cuda:0")
    gk.convert_ptrs_to_idxs_i64(stream_ptr(), rp64, 3000, idxs)
    assert np.array_equal(host(idxs), np.repeat(np.arange(3000), np.diff(rp)))
    back = torch.full((3001,), -7, dtype=torch.int64, device="cuda:0")
    nb = gk.prefix_sum_workspace_bytes(3001)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda:0")
    gk.convert_idxs_to_ptrs_i64(stream_ptr(), idxs, nnz, 3000, back, ws, nb)
    assert np.array_equal(host(back), rp)
    sizes = torch.zeros(3000, dtype=torch.uint64, device="cuda:0")
    gk.convert_ptrs_to_sizes_i64(stream_ptr(), rp64, 3000, sizes)
    assert np.array_equal(host(sizes.view(torch.int64)), np.diff(rp))


@pytest.mark.parametrize("g", [1, 2, 7, 40])
def test_device_generated_poisson_matrix_equals_the_host_one(gk, g):
    n, rp, ci, v = matgen.poisson_3d_7pt(g)
    A = formats.Csr64.poisson_3d_7pt(gk, g)
    assert np.array_equal(host(A.row_ptrs), rp) and np.array_equal(host(A.col_idxs), ci) and np.array_equal(host(A.vals), v)
    rp32 = torch.empty(n + 1, dtype=torch.int32, device="cuda:0")
    ci32 = torch.empty(len(v), dtype=torch.int32, device="cuda:0")
    v32 = torch.empty(len(v), dtype=torch.float64, device="cuda:0")
    gk.diag_poisson3d_7pt_f64_i32(stream_ptr(), g, rp32, ci32, v32)
    assert np.array_equal(host(rp32), rp) and np.array_equal(host(ci32), ci) and np.array_equal(host(v32), v)


def test_full_size_poisson_p2_int64_bitexact(gk, oracle):
    """BASELINE config[1] through the int64 instantiation: bit-exact against the oracle, both kernels"""
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
    expect = oracle_apply(oracle, n, rp, ci, v, x)
    for split in (True, False):
        A = formats.Csr64.from_host(gk, n, n, rp, ci, v, split=split)
        got = host(A.apply(dev(x), torch.full((n, 1), float("nan"), dtype=torch.float64, device="cuda:0")))
        assert np.array_equal(got, expect), split


def test_p3_int64_equals_int32_bit_for_bit(gk):
    """BASELINE config 5's matrix (256^3 7-pt, 117 M nonzeros) in both index types: the same bits, and the
    A 1 = 0 interior property"""
    g = 256
    A64 = formats.Csr64.poisson_3d_7pt(gk, g)
    n = g ** 3
    A32 = formats.Csr(gk, n, n, A64.row_ptrs.to(torch.int32), A64.col_idxs.to(torch.int32), A64.vals)
    x = torch.sin(0.01 * torch.arange(n, dtype=torch.float64, device="cuda:0")).reshape(n, 1)
    y64 = A64.apply(x, torch.empty_like(x))
    y32 = A32.apply(x, torch.empty_like(x))
    assert torch.equal(y64, y32)
    ones = torch.ones_like(x)
    y1 = A64.apply(ones, torch.empty_like(x)).reshape(g, g, g)
    assert bool(torch.all(y1[1:-1, 1:-1, 1:-1] == 0.0)) and float(y1[0, 0, 0]) == 3.0 and float(y1[0, 0, 1]) == 2.0


def test_cg_on_an_int64_matrix(gk, oracle):
    """Cg::apply on Csr<double, int64> through the operator driver (gkomi_csr64_matrix_apply_cb): the oracle's
    iteration count and solution (core/solver/cg.cpp:107-193)"""
    n, rp, ci, v = matgen.poisson_2d_5pt(200)
    s = np.sin(np.arange(n, dtype=np.float64))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, (s / np.linalg.norm(s)).reshape(n, 1), 1, b, 1)
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b[:, 0].copy(), xe, 5000, 1e-10, 0, None, 0)
    A = formats.Csr64.from_host(gk, n, n, rp, ci, v)
    for fused in (False, True):
        res = solvers.solve_op(gk, "cg", A, dev(b), max_iters=5000, reduction=1e-10, fused=fused, check_every=8)
        assert res["converged"] and abs(res["iterations"] - it) <= 1
        assert matgen.rel_err(host(res["x"]).reshape(n), xe) <= 1e-7


def test_more_than_2_31_nonzeros(gk):
    """700^3 7-pt Poisson: 343 M rows, 2.398 G nonzeros (> 2^31), 38 GB of values + int64 columns, generated on the
    device; one apply, checked by size-independent properties: A 1 vanishes in the interior and counts the missing
    neighbours on the boundary; A x for x = k (the fastest grid coordinate) vanishes away from the k faces."""
    g = 700
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free device memory")
    A = formats.Csr64.poisson_3d_7pt(gk, g)
    assert A.nnz == 7 * g ** 3 - 6 * g * g and A.nnz > 2 ** 31
    assert int(A.row_ptrs[-1].item()) == A.nnz and int(A.col_idxs[-1].item()) == g ** 3 - 1
    n = g ** 3
    ones = torch.ones((n, 1), dtype=torch.float64, device="cuda:0")
    y = A.apply(ones, torch.full((n, 1), float("nan"), dtype=torch.float64, device="cuda:0")).reshape(g, g, g)
    assert bool(torch.all(y[1:-1, 1:-1, 1:-1] == 0.0))
    assert float(y[0, 0, 0]) == 3.0 and float(y[-1, -1, -1]) == 3.0 and float(y[0, 5, 5]) == 1.0 and float(y[g // 2, 0, g - 1]) == 2.0
    assert float(y.sum().item()) == 6.0 * g * g          # one unit per boundary face cell
    del ones
    xk = (torch.arange(n, dtype=torch.float64, device="cuda:0") % g).reshape(n, 1)
    y = A.apply(xk, y.reshape(n, 1)).reshape(g, g, g)
    # interior in k: 6k - (k-1) - (k+1) - 4k = 0 wherever the four (i, j) neighbours exist
    assert bool(torch.all(y[1:-1, 1:-1, 1:-1] == 0.0))
    assert float(y[5, 5, 0]) == -1.0 and float(y[5, 5, g - 1]) == 6.0 * (g - 1) - (g - 2) - 4.0 * (g - 1)


@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_long_runs_of_empty_rows_int64(gk, oracle, advanced):
    """the sparse-rows mode of the nonzero-split kernel (rows handed out by index when a tile owns more than 2048 of
    them) in the int64 instantiation: the mark sits in bit 63 of the srow entries; same bits as the oracle"""
    rng = np.random.default_rng(8)
    nrows, ncols = 300007, 7001
    counts = rng.integers(0, 6, size=nrows)
    counts[4000:200000] = 0
    counts[250000:251000] = 1
    counts[290000:] = 0
    rp, ci, v = matgen.random_rows_csr(nrows, ncols, counts, 9)
    b = rng.standard_normal((ncols, 1))
    c0 = rng.standard_normal((nrows, 1))
    # (explicit nonzero-split strategy: with fewer nonzeros than rows the automatic one goes by rows)
    A = formats.Csr64.from_host(gk, nrows, ncols, rp, ci, v, strategy=4, split=True)
    srow = host(A.srow())
    assert srow[-1] > 2048 and np.all(srow[:-1] < 0)       # marked
    if advanced:
        expect = oracle_apply(oracle, nrows, rp.astype(np.int32), ci.astype(np.int32), v, b, c0, 1.5, -0.25)
        got = host(A.apply(dev(b), dev(c0), alpha=1.5, beta=-0.25))
    else:
        expect = oracle_apply(oracle, nrows, rp.astype(np.int32), ci.astype(np.int32), v, b)
        got = host(A.apply(dev(b), torch.full((nrows, 1), float("nan"), dtype=torch.float64, device="cuda:0")))
    assert np.array_equal(got, expect)
