"""GPU parity tests of the COO tile kernels (csrc/coo_spmv.hip): the atomic-free
path for row-sorted matrices (gkomi_coo_spmv_sorted_f64_i32 /
gkomi_coo_spmv2_sorted_f64_i32) and the any-order multi-column pass behind
gkomi_coo_spmv2_f64_i32, against the oracle's COO loops
(reference/matrix/coo_kernels.cpp:63-131).

Tolerance: a row of at most 128 nonzeros that lie inside one 1536-nonzero tile is summed in the
reference's order -- bit-exact for c = A b; a longer row segment is reduced by a whole wave
(lane-strided partial sums + fixed tree, the role of the reference's segmented scan); the advanced / apply2 forms add the
finished row sum to beta*c (the reference adds product by product) and rows cut
by a tile boundary add per-tile partial sums: <= 1e-14 relative.  The sorted path
has no atomics: two runs give the same bits.  Edge cases as in
test/matrix/coo_kernels.cpp of the reference: empty matrix, empty rows (leading,
trailing, long runs), rows longer than many tiles, nnz = 1, strided b / c."""
import ctypes

import numpy as np
import pytest
import torch

import matgen
from gpu_util import dev, host

pytestmark = pytest.mark.gpu
TILE = 1536


def stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(gk, nnz, nrhs):
    nb = gk.coo_sorted_workspace_bytes(nnz, nrhs)
    return torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda:0"), nb


def rows_of(rp):
    return np.repeat(np.arange(len(rp) - 1, dtype=np.int32), np.diff(rp)).astype(np.int32)


def cases():
    out = {}
    n, rp, ci, v = matgen.poisson_2d_5pt(300, 300)
    out["poisson300"] = (n, n, rows_of(rp), ci, v)
    # empty rows: leading, trailing, runs shorter and longer than the in-thread limit (16)
    rng = np.random.default_rng(5)
    counts = rng.integers(0, 4, size=6000)
    counts[:40] = 0
    counts[-700:] = 0
    counts[1000:1010] = 0
    counts[2000:2300] = 0
    rp, ci, v = matgen.random_rows_csr(6000, 5000, counts, seed=6)
    out["empty_rows"] = (6000, 5000, rows_of(rp), ci, v)
    # more than 64 long gaps inside one tile (the workgroup's gap list overflows)
    counts = np.zeros(40000, dtype=np.int64)
    counts[::100] = 3
    rp, ci, v = matgen.random_rows_csr(40000, 777, counts, seed=7)
    out["many_long_gaps"] = (40000, 777, rows_of(rp), ci, v)
    # rows of 1..231 nonzeros
    rp, ci, v = matgen.random_csr(532, 231, 1, 231, seed=42)
    out["rows_1_231"] = (532, 231, rows_of(rp), ci, v)
    # rows that run through several tiles; one through more than 64 of them
    counts = np.full(300, 7, dtype=np.int64)
    counts[3] = 5000
    counts[100] = 130 * TILE + 17
    counts[101] = 2 * TILE
    counts[299] = 4000
    rp, ci, v = matgen.random_rows_csr(300, 400000, counts, seed=8)
    out["long_rows"] = (300, 400000, rows_of(rp), ci, v)
    # the whole matrix is one row / one entry / exactly one tile
    rng = np.random.default_rng(9)
    out["one_row"] = (3, 50000, np.full(4 * TILE, 1, np.int32), np.sort(rng.choice(50000, 4 * TILE, replace=False)).astype(np.int32),
                      rng.standard_normal(4 * TILE))
    out["one_entry"] = (5, 5, np.array([2], np.int32), np.array([4], np.int32), np.array([3.5]))
    rp, ci, v = matgen.random_rows_csr(512, 900, np.full(512, 3), seed=10)
    out["exactly_one_tile"] = (512, 900, rows_of(rp), ci, v)
    return out


CASES = cases()


def expected(oracle, mode, nrows, rows, ci, v, b, c0, alpha, beta):
    nnz, nrhs = len(v), b.shape[1]
    e = c0.copy()
    if mode == "spmv":
        oracle.ref_coo_spmv(nrows, nnz, nrhs, rows, ci, v, b, nrhs, e, nrhs)
    elif mode == "advanced":
        oracle.ref_coo_advanced_spmv(nrows, nnz, nrhs, alpha, rows, ci, v, b, nrhs, beta, e, nrhs)
    elif mode == "spmv2":
        oracle.ref_coo_spmv2(nnz, nrhs, rows, ci, v, b, nrhs, e, nrhs)
    else:
        oracle.ref_coo_advanced_spmv2(nnz, nrhs, alpha, rows, ci, v, b, nrhs, e, nrhs)
    return e


def analyse(gk, rows_d, nnz, ws, nb):
    flag, longest = ctypes.c_int(0), ctypes.c_int64(0)
    gk.coo_analyse_rows_i32(stream(), nnz, rows_d, ws, nb, ctypes.addressof(flag), ctypes.addressof(longest))
    return flag.value, longest.value


def violation(gk, ws):
    flag = ctypes.c_int(0)
    gk.coo_sorted_check(stream(), ws, ctypes.addressof(flag))
    return flag.value


def run_sorted(gk, mode, nrows, ncols, rows_d, ci_d, v_d, b_d, c_d, alpha, beta, ws, nb, hint=-1):
    nnz, nrhs = v_d.numel() if rows_d is not None else 0, b_d.shape[1]
    al = dev(np.array([alpha])) if mode in ("advanced", "advanced_spmv2") else None
    be = dev(np.array([beta])) if mode == "advanced" else None
    if mode in ("spmv", "advanced"):
        gk.coo_spmv_sorted_f64_i32(stream(), nrows, ncols, nrhs, nnz, rows_d, ci_d, v_d, b_d, b_d.stride(0), c_d,
                                   c_d.stride(0), al, be, hint, ws, nb)
    else:
        gk.coo_spmv2_sorted_f64_i32(stream(), nrows, ncols, nrhs, nnz, rows_d, ci_d, v_d, b_d, b_d.stride(0), c_d,
                                    c_d.stride(0), al, hint, ws, nb)
    return c_d


def uncut_rows(nrows, rows):
    """rows of at most 128 nonzeros, all in one tile (and rows without nonzeros), for each of
    the tile sizes in use: 1536 / 1024 / 512 nonzeros for 1 / 2 / 4 columns per pass"""
    nnz = len(rows)
    cut = np.bincount(rows, minlength=nrows) > 128
    for tile in (TILE, 1024, 512):   # (8 columns per pass: 512 too)
        for t in range(tile, nnz, tile):
            if rows[t - 1] == rows[t]:
                cut[rows[t]] = True
    return ~cut


def longest_row(rows):
    return int(np.max(np.diff(np.flatnonzero(np.concatenate(([True], rows[1:] != rows[:-1], [True]))))))


@pytest.mark.parametrize("variant", ["carries", "halo"])
@pytest.mark.parametrize("mode", ["spmv", "advanced", "spmv2", "advanced_spmv2"])
@pytest.mark.parametrize("nrhs", [1, 2, 3, 4, 7, 8, 13])
@pytest.mark.parametrize("case", sorted(CASES))
def test_sorted_against_the_oracle(gk, oracle, case, nrhs, mode, variant):
    nrows, ncols, rows, ci, v = CASES[case]
    nnz = len(v)
    longest = longest_row(rows)
    if variant == "halo" and longest > 64:
        pytest.skip("the one-launch variant takes rows of at most 64 nonzeros")
    hint = longest if variant == "halo" else -1
    rng = np.random.default_rng(nrhs)
    b = rng.standard_normal((ncols, nrhs))
    c0 = rng.standard_normal((nrows, nrhs))
    alpha, beta = -0.75, 1.5
    e = expected(oracle, mode, nrows, rows, ci, v, b, c0, alpha, beta)
    rows_d, ci_d, v_d, b_d = dev(rows), dev(ci), dev(v), dev(b)
    ws, nb = workspace(gk, nnz, nrhs)
    assert analyse(gk, rows_d, nnz, ws, nb) == (1, min(longest, 65))
    got = host(run_sorted(gk, mode, nrows, ncols, rows_d, ci_d, v_d, b_d, dev(c0), alpha, beta, ws, nb, hint))
    # rows of 2e5 nonzeros: partial sums per tile vs the reference's one running sum, eps * sqrt(length)
    assert matgen.rel_err(got, e) <= (1e-13 if case in ("long_rows", "one_row") else 1e-14)
    if mode == "spmv" and variant == "halo":
        assert np.array_equal(got, e)                   # every row in the reference's order
    elif mode == "spmv":
        keep = uncut_rows(nrows, rows)
        assert np.array_equal(got[keep], e[keep])      # the reference's order inside a row
    again = host(run_sorted(gk, mode, nrows, ncols, rows_d, ci_d, v_d, b_d, dev(c0), alpha, beta, ws, nb, hint))
    assert again.tobytes() == got.tobytes()            # no atomics: same bits every run
    assert violation(gk, ws) == 0


@pytest.mark.parametrize("hint", [-1, 3])
@pytest.mark.parametrize("mode", ["spmv", "advanced", "spmv2"])
def test_sorted_strided_vectors_leave_the_padding_alone(gk, oracle, mode, hint):
    nrows, ncols, rows, ci, v = CASES["empty_rows"]
    nnz, nrhs = len(v), 3
    rng = np.random.default_rng(3)
    bb = rng.standard_normal((ncols, 5))
    cc = rng.standard_normal((nrows, 6))
    b, c0 = np.ascontiguousarray(bb[:, :nrhs]), np.ascontiguousarray(cc[:, :nrhs])
    e = expected(oracle, mode, nrows, rows, ci, v, b, c0, -0.75, 1.5)
    ws, nb = workspace(gk, nnz, nrhs)
    c_d = dev(cc)
    run_sorted(gk, mode, nrows, ncols, dev(rows), dev(ci), dev(v), dev(bb)[:, :nrhs], c_d[:, :nrhs], -0.75, 1.5, ws, nb, hint)
    got = host(c_d)
    assert matgen.rel_err(got[:, :nrhs], e) <= 1e-14
    assert np.array_equal(got[:, nrhs:], cc[:, nrhs:])


@pytest.mark.parametrize("mode", ["spmv", "advanced", "spmv2"])
def test_sorted_empty_matrix(gk, mode):
    nrows, ncols, nrhs = 70, 30, 2
    rng = np.random.default_rng(1)
    b, c0 = rng.standard_normal((ncols, nrhs)), rng.standard_normal((nrows, nrhs))
    ws, nb = workspace(gk, 0, nrhs)
    empty_i, empty_v = torch.zeros(1, dtype=torch.int32, device="cuda:0"), torch.zeros(1, dtype=torch.float64, device="cuda:0")
    al = dev(np.array([-0.75])) if mode == "advanced" else None
    be = dev(np.array([1.5])) if mode == "advanced" else None
    c_d = dev(c0)
    if mode == "spmv2":
        gk.coo_spmv2_sorted_f64_i32(stream(), nrows, ncols, nrhs, 0, empty_i, empty_i, empty_v, dev(b), nrhs, c_d, nrhs, None, -1, ws, nb)
    else:
        gk.coo_spmv_sorted_f64_i32(stream(), nrows, ncols, nrhs, 0, empty_i, empty_i, empty_v, dev(b), nrhs, c_d, nrhs, al, be, -1, ws, nb)
    e = {"spmv": np.zeros_like(c0), "advanced": 1.5 * c0, "spmv2": c0}[mode]
    assert np.array_equal(host(c_d), e)


def test_row_analysis_and_workspace_checks(gk):
    rows = np.array([0, 0, 1, 3, 3, 2, 4], np.int32)
    ws, nb = workspace(gk, len(rows), 1)
    assert analyse(gk, dev(rows), len(rows), ws, nb)[0] == 0
    assert analyse(gk, dev(np.sort(rows)), len(rows), ws, nb) == (1, 2)
    big = np.arange(3_000_000, dtype=np.int32) // 3
    assert analyse(gk, dev(big), len(big), ws, nb) == (1, 3)
    big[2_345_678] = 5
    assert analyse(gk, dev(big), len(big), ws, nb)[0] == 0
    runs = np.repeat(np.arange(5, dtype=np.int32), [3, 64, 1, 65, 2])
    assert analyse(gk, dev(runs), len(runs), ws, nb) == (1, 65)       # capped: "more than 64"
    assert analyse(gk, dev(runs[:68]), 68, ws, nb) == (1, 64)
    # too small a workspace is refused, not overrun
    nrows, ncols, r, ci, v = CASES["poisson300"]
    c = torch.zeros((nrows, 1), dtype=torch.float64, device="cuda:0")
    with pytest.raises(Exception, match="workspace"):
        gk.coo_spmv_sorted_f64_i32(stream(), nrows, ncols, 1, len(v), dev(r), dev(ci), dev(v), c, 1, c, 1, None, None, -1, ws, 64)


def test_a_hint_that_is_too_small_is_reported(gk):
    """rows of 100 nonzeros through the one-launch kernel with hint 64: the result is
    wrong by contract, and gkomi_coo_sorted_check says so."""
    counts = np.full(200, 100, dtype=np.int64)
    rp, ci, v = matgen.random_rows_csr(200, 5000, counts, seed=3)
    rows = rows_of(rp)
    ws, nb = workspace(gk, len(v), 1)
    assert analyse(gk, dev(rows), len(v), ws, nb) == (1, 65)
    b = dev(np.ones((5000, 1)))
    c = torch.zeros((200, 1), dtype=torch.float64, device="cuda:0")
    gk.coo_spmv_sorted_f64_i32(stream(), 200, 5000, 1, len(v), dev(rows), dev(ci), dev(v), b, 1, c, 1, None, None, 64, ws, nb)
    assert violation(gk, ws) == 1


@pytest.mark.parametrize("nrhs", [2, 4, 5, 8, 11])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_any_order_several_columns(gk, oracle, nrhs, advanced):
    """gkomi_coo_spmv_f64_i32 with nrhs >= 2: the tile is read once per 4 / 2 columns
    (atomics, any order of the entries)."""
    nrows, ncols, rows, ci, v = CASES["poisson300"]
    nnz = len(v)
    perm = np.random.default_rng(2).permutation(nnz)
    rng = np.random.default_rng(nrhs)
    b, c0 = rng.standard_normal((ncols, nrhs)), rng.standard_normal((nrows, nrhs))
    e = expected(oracle, "advanced" if advanced else "spmv", nrows, rows, ci, v, b, c0, -0.75, 1.5)
    al = dev(np.array([-0.75])) if advanced else None
    be = dev(np.array([1.5])) if advanced else None
    for order in (np.arange(nnz), perm):
        c_d = dev(c0)
        gk.coo_spmv_f64_i32(stream(), nrows, ncols, nrhs, nnz, dev(rows[order]), dev(ci[order]), dev(v[order]), dev(b), nrhs,
                            c_d, nrhs, al, be)
        assert matgen.rel_err(host(c_d), e) <= 1e-14


def test_format_object_picks_the_sorted_path(gk, oracle):
    from gkomi import formats
    n, rp, ci, v = matgen.poisson_2d_5pt(64, 70)
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    M = A.to("coo")
    x = dev(np.sin(0.1 * np.arange(n)).reshape(n, 1))
    y_csr = A.apply(x, torch.zeros_like(x))
    y = M.apply(x, torch.full_like(x, 7.0))
    assert M._sorted is True and M.max_row_nnz == 5
    y_carry = M.apply(x, torch.full_like(x, 7.0), hint=-1)
    assert matgen.rel_err(host(y_carry), host(y_csr)) <= 1e-15
    assert np.array_equal(host(y), host(y_csr))        # one launch, every row in CSR's (= the reference's) order
    assert matgen.rel_err(host(y), host(y_csr)) <= 1e-15
    y2 = M.apply(x, torch.full_like(x, 7.0), sorted_rows=False)
    assert matgen.rel_err(host(y2), host(y_csr)) <= 1e-15
    acc = M.apply2(x, y_csr.clone())
    assert matgen.rel_err(host(acc), 2 * host(y_csr)) <= 1e-15
