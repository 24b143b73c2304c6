"""GPU parity tests of the nonzero-split CSR SpMV (GKOMI_CSR_SPLIT over srow,
gkomi_csr_spmv_srow_f64_i32): bit-exact against the oracle
(reference/matrix/csr_kernels.cpp:75-128) for every tile size and load order,
on the shapes of test/matrix/csr_kernels2.cpp:228-455 and on the edge cases of
the cut by nonzeros: empty rows at tile boundaries, trailing empty rows, rows
longer than the caller's hint, odd / tiny nnz, nnz an exact multiple of the tile."""
import numpy as np
import pytest
import torch

import matgen
from gpu_util import DevCsr, csr_apply, csr_apply_srow, dev, host, make_srow
from test_csr_spmv_gpu import _oracle_apply

pytestmark = pytest.mark.gpu
SPLIT = 4
TILES = [1024, 1536, 2048, 3072]
VARIANTS = {"plain": 0, "nt": 2, "noswz": 1 << 8, "nt_noswz": (1 << 8) | 2}


def strat(variant):
    v = VARIANTS[variant]
    return SPLIT | ((v & 0xff) << 8) | ((v >> 8) << 16)


def check_srow(srow, rp, nnz, tile):
    """srow[t] = first row with row_ptrs[row] >= t*tile (lower bound over rows 0..nrows)."""
    nrows = len(rp) - 1
    t = np.arange(nnz // tile + 2, dtype=np.int64) * tile
    expect = np.minimum(np.searchsorted(np.asarray(rp[:nrows], np.int64), t, side="left"), nrows)
    got = host(srow)
    # ... behind the tile starts, the most rows that start in one tile; beyond 2048 the kernel hands rows out by
    # index (sparse-rows mode), which every tile start says in its top bit
    assert got[-1] == np.max(np.diff(expect))
    marked = got[-1] > 2048
    assert np.array_equal(got[:-1] < 0, np.full(len(expect), marked))
    assert np.array_equal(got[:-1] & 0x7fffffff, expect.astype(np.int32))


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_poisson_bit_exact(gk, oracle, tile, variant, advanced):
    n, rp, ci, v = matgen.poisson_2d_5pt(173, 181)
    rng = np.random.default_rng(11)
    b = rng.standard_normal((n, 1))
    c0 = rng.standard_normal((n, 1))
    A = DevCsr(n, n, rp, ci, v)
    srow, _ = make_srow(gk, A, tile)
    check_srow(srow, rp, A.nnz, tile)
    if advanced:
        expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, -0.75, 1.5)
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, dev(c0), -0.75, 1.5, strat(variant)))
    else:
        expect = _oracle_apply(oracle, n, rp, ci, v, b)
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=strat(variant)))
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("sort", [True, False], ids=["sorted", "unsorted"])
@pytest.mark.parametrize("nrhs", [1, 3])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("hint", ["exact", "too_small", "unknown"])
def test_random_532x231(gk, oracle, hint, advanced, nrhs, sort):
    # rows of 1..231 entries: far beyond the over-read of the tile, finished from memory
    rp, ci, v = matgen.random_csr(532, 231, 1, 231, seed=42, sort=sort)
    rng = np.random.default_rng(15)
    b = rng.standard_normal((231, nrhs))
    c0 = rng.standard_normal((532, nrhs))
    A = DevCsr(532, 231, rp, ci, v)
    srow, tile = make_srow(gk, A)
    h = {"exact": A.max_row_nnz, "too_small": 3, "unknown": -1}[hint]
    if advanced:
        expect = _oracle_apply(oracle, 532, rp, ci, v, b, c0, 2.0, -1.0)
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, dev(c0), 2.0, -1.0, SPLIT, hint=h))
    else:
        expect = _oracle_apply(oracle, 532, rp, ci, v, b)
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=SPLIT, hint=h))
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_short_random_rows_with_empty_rows(gk, oracle, tile, seed):
    # 0..9 entries per row, ~10 % empty rows, runs of empty rows across tile boundaries
    rng = np.random.default_rng(seed)
    nrows, ncols = 20011, 5003
    counts = rng.integers(0, 10, size=nrows)
    counts[(rng.integers(0, nrows, size=40)[:, None] + np.arange(300)[None, :]) % nrows] = 0
    counts[-57:] = 0                       # trailing empty rows: owned by the last tile
    rp, ci, v = matgen.random_rows_csr(nrows, ncols, counts, seed)
    b = rng.standard_normal((ncols, 1))
    A = DevCsr(nrows, ncols, rp, ci, v)
    srow, _ = make_srow(gk, A, tile)
    check_srow(srow, rp, A.nnz, tile)
    expect = _oracle_apply(oracle, nrows, rp, ci, v, b)
    for variant in ("plain", "nt", "nt_noswz"):
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=strat(variant)))
        assert np.array_equal(got, expect), variant
    # automatic strategy with srow takes the same kernel
    assert np.array_equal(host(csr_apply_srow(gk, A, dev(b), srow, tile)), expect)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
def test_long_runs_of_empty_rows(gk, oracle, tile, advanced):
    """A tile that owns hundreds of thousands of empty rows (the non-local block of a distributed matrix, a
    selection matrix): the kernel hands rows out by index instead of walking them in one workgroup; same bits."""
    rng = np.random.default_rng(tile)
    nrows, ncols = 400003, 9001
    counts = rng.integers(0, 7, size=nrows)
    counts[5000:260000] = 0          # one tile owns 255 000 empty rows
    counts[300000:300900] = 1        # 900 one-entry rows: more than 2 x 256 rows starting in one tile
    counts[380000:] = 0              # trailing empty rows
    rp, ci, v = matgen.random_rows_csr(nrows, ncols, counts, 5)
    b = rng.standard_normal((ncols, 1))
    c0 = rng.standard_normal((nrows, 1))
    A = DevCsr(nrows, ncols, rp, ci, v)
    srow, _ = make_srow(gk, A, tile)
    check_srow(srow, rp, A.nnz, tile)
    assert host(srow)[-1] > 2048
    for variant in ("plain", "nt_noswz"):
        if advanced:
            expect = _oracle_apply(oracle, nrows, rp, ci, v, b, c0, 1.25, -0.5)
            got = host(csr_apply_srow(gk, A, dev(b), srow, tile, dev(c0), 1.25, -0.5, strat(variant)))
        else:
            expect = _oracle_apply(oracle, nrows, rp, ci, v, b)
            got = host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=strat(variant)))
        assert np.array_equal(got, expect), variant
    # and it is fast: the whole matrix is ~1 M nonzeros
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    db = dev(b)
    out = csr_apply_srow(gk, A, db, srow, tile, strategy=SPLIT)
    e0.record()
    for _ in range(10):
        csr_apply_srow(gk, A, db, srow, tile, out, strategy=SPLIT)
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) / 10 < 0.6, "ms per apply (0.01-0.05 measured; > 1 ms when one workgroup walks the rows)"


def test_non_local_block_of_a_distributed_matrix(gk, oracle):
    """x = 1 * A_nonlocal * halo + 1 * x as core/distributed/matrix.cpp:330-331 asks it of csr::advanced_spmv: 2 M rows,
    nonzeros only in the first and last 65 536 of them.  Fewer nonzeros than rows: the automatic strategy takes the
    row-cut kernel (its grid follows the rows); the explicit split strategy is correct too (sparse-rows mode)."""
    n, halo = 2097152, 131072
    counts = np.zeros(n, np.int64)
    counts[:65536] = 1
    counts[-65536:] = 1
    rp = np.zeros(n + 1, np.int32)
    np.cumsum(counts, out=rp[1:])
    ci = np.arange(halo, dtype=np.int32)
    rng = np.random.default_rng(2)
    v = rng.standard_normal(halo)
    b = rng.standard_normal((halo, 1))
    c0 = rng.standard_normal((n, 1))
    A = DevCsr(n, halo, rp, ci, v)
    srow, tile = make_srow(gk, A)
    expect = _oracle_apply(oracle, n, rp, ci, v, b, c0, 1.0, 1.0)
    for strategy in (0, SPLIT):
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, dev(c0), 1.0, 1.0, strategy))
        assert np.array_equal(got, expect), strategy
    out = dev(c0)
    db = dev(b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    csr_apply_srow(gk, A, db, srow, tile, out, 1.0, 1.0, 0)
    e0.record()
    for _ in range(10):
        csr_apply_srow(gk, A, db, srow, tile, out, 1.0, 1.0, 0)
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) / 10 < 0.5, "ms per apply of the automatic strategy (0.05 measured, most of it the test's own scalar uploads; 5 ms when one workgroup walks the rows)"


@pytest.mark.parametrize("nnz_target", [2, 3, 1535, 1536, 1537, 3072, 2 * 1536 + 1])
def test_tiny_odd_and_exact_multiple_nnz(gk, oracle, nnz_target):
    # one entry per row except a few longer ones, nnz hits the tile size exactly / +-1
    tile = 1536
    nrows = nnz_target - min(4, nnz_target - 1)
    counts = np.ones(nrows, np.int64)
    counts[nrows // 2] += nnz_target - nrows
    rp = np.zeros(nrows + 1, np.int32)
    np.cumsum(counts, out=rp[1:])
    assert rp[-1] == nnz_target
    rng = np.random.default_rng(nnz_target)
    ncols = 97
    ci = np.concatenate([np.sort(rng.choice(ncols, size=k, replace=False)) for k in counts]).astype(np.int32)
    v = rng.standard_normal(nnz_target)
    b = rng.standard_normal((ncols, 1))
    A = DevCsr(nrows, ncols, rp, ci, v)
    srow, _ = make_srow(gk, A, tile)
    check_srow(srow, rp, A.nnz, tile)
    expect = _oracle_apply(oracle, nrows, rp, ci, v, b)
    assert np.array_equal(host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=SPLIT)), expect)
    assert np.array_equal(host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=SPLIT | (2 << 8))), expect)


def test_ragged_rows_longer_than_a_tile(gk, oracle):
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
    for tile in TILES:
        srow, _ = make_srow(gk, A, tile)
        got = host(csr_apply_srow(gk, A, dev(b), srow, tile, strategy=SPLIT, hint=5))
        assert np.array_equal(got, expect), tile
    # automatic: long rows do not take the split kernel, result within r<double>
    srow, tile = make_srow(gk, A)
    got = host(csr_apply_srow(gk, A, dev(b), srow, tile))
    assert matgen.rel_err(got, expect) <= 1e-14


def test_split_needs_srow_and_null_srow_is_the_plain_entry(gk, oracle):
    import gkomi
    n, rp, ci, v = matgen.poisson_2d_5pt(40)
    b = np.random.default_rng(3).standard_normal((n, 1))
    A = DevCsr(n, n, rp, ci, v)
    with pytest.raises(gkomi._lib.GkomiError):
        csr_apply(gk, A, dev(b), strategy=SPLIT)
    expect = _oracle_apply(oracle, n, rp, ci, v, b)
    assert np.array_equal(host(csr_apply_srow(gk, A, dev(b), None, 0)), expect)


def test_simple_apply_never_reads_c_and_nan_safe(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)
    b = np.random.default_rng(3).standard_normal((n, 1))
    A = DevCsr(n, n, rp, ci, v)
    srow, tile = make_srow(gk, A)
    c = torch.full((n, 1), float("nan"), dtype=torch.float64, device="cuda:0")
    csr_apply_srow(gk, A, dev(b), srow, tile, c, strategy=SPLIT)
    assert np.array_equal(host(c), _oracle_apply(oracle, n, rp, ci, v, b))


def test_full_size_p2_bit_exact_and_properties(gk, oracle):
    """BASELINE config 2 at full size through the srow path."""
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
    A = DevCsr(n, n, rp, ci, v)
    srow, tile = make_srow(gk, A)
    check_srow(srow, rp, A.nnz, tile)
    expect = _oracle_apply(oracle, n, rp, ci, v, x)
    got = host(csr_apply_srow(gk, A, dev(x), srow, tile, hint=5))
    assert np.array_equal(got, expect)
    ones = host(csr_apply_srow(gk, A, dev(np.ones((n, 1))), srow, tile, hint=5))
    interior = np.ones((1000, 1000), bool)
    interior[[0, -1], :] = False
    interior[:, [0, -1]] = False
    assert np.all(ones[interior.ravel()] == 0.0)          # A 1 = 0 away from the boundary
    again = host(csr_apply_srow(gk, A, dev(x), srow, tile, hint=5))
    assert np.array_equal(got, again)                     # run-to-run deterministic


def test_residency_tracker_selects_the_nontemporal_streams_by_itself(gk, oracle):
    """An automatic apply (no GKOMI_CSR_STREAMING: what Csr::apply through the reference interface passes) reads the
    matrix with nontemporal loads when more than the 256 MiB Infinity Cache of OTHER CSR applies went by since its last
    apply -- the library's own byte count (csr_probably_evicted).  The same matrix again and again: never; four 85 MB
    matrices in rotation: every apply after the first round.  Results are bit-identical either way."""
    g = 1030
    n, rp, ci, v = matgen.poisson_2d_5pt(g)
    x = dev(np.sin(0.01 * np.arange(n)).reshape(n, 1))
    mats = []
    for k in range(4):
        A = DevCsr(n, n, rp, ci, v * (1.0 + k))
        mats.append((A, make_srow(gk, A, 1536)))
    expect = np.empty((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, host(x), 1, expect, 1)
    # (the tracker is keyed by the address of the values: torch hands out addresses that earlier tests' matrices had --
    # one apply each makes these four known, then the first one is the most recent again)
    for A, (srow, tile) in mats + mats[:1]:
        csr_apply_srow(gk, A, x, srow, tile, hint=5)
    before = gk.diag_csr_evicted_applies()
    A, (srow, tile) = mats[0]
    for _ in range(6):
        y = csr_apply_srow(gk, A, x, srow, tile, hint=5)
    assert gk.diag_csr_evicted_applies() == before and np.array_equal(host(y), expect)
    for rnd in range(3):
        for A, (srow, tile) in mats:
            y = csr_apply_srow(gk, A, x, srow, tile, hint=5)
    assert gk.diag_csr_evicted_applies() - before >= 8          # rounds 2 and 3: every apply
    y = csr_apply_srow(gk, mats[0][0], x, mats[0][1][0], mats[0][1][1], hint=5)
    assert np.array_equal(host(y), expect)
