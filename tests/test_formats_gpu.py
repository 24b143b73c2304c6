"""GPU parity tests (through the C ABI) of ELL / SELL-P / COO / Hybrid SpMV,
the CSR conversions and the index components against the oracle.
Bit-exact for ELL, SELL-P, all conversions and index kernels; COO/Hybrid-COO
within r<double> (one fp64 atomic per row segment), exact when every row lies
inside one tile and the output starts from zero."""
import json
import os

import numpy as np
import pytest
import torch

import formats_util as fu
import matgen
from gpu_util import dev, host, stream_ptr

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats.json")))
U8 = torch.uint8


def ws_for(gk, n):
    nb = gk.prefix_sum_workspace_bytes(n)
    return torch.empty(max(nb, 8), dtype=U8, device="cuda:0"), nb


def scal(x):
    return None if x is None else dev(np.array([x], np.float64))


def dev_to_ell(gk, nrows, rpd, cid, vd, stride=None):
    mx = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    gk.csr_max_row_nnz_i32(stream_ptr(), nrows, rpd, mx)
    k = int(mx.item())
    stride = nrows if stride is None else stride
    cols = torch.full((max(stride * k, 1),), -1, dtype=torch.int32, device="cuda:0")
    vals = torch.zeros(max(stride * k, 1), dtype=torch.float64, device="cuda:0")
    gk.csr_convert_to_ell_f64_i32(stream_ptr(), nrows, rpd, cid, vd, k, stride, cols, vals)
    return k, stride, cols, vals


def dev_to_sellp(gk, nrows, rpd, cid, vd, slice_size=64, stride_factor=1):
    nsl = (nrows + slice_size - 1) // slice_size
    sets = torch.zeros(nsl + 1, dtype=torch.int64, device="cuda:0")
    lens = torch.zeros(max(nsl, 1), dtype=torch.int64, device="cuda:0")
    ws, nb = ws_for(gk, nsl + 1)
    gk.sellp_compute_slice_sets_i32(stream_ptr(), rpd, nrows, slice_size, stride_factor, sets, lens, ws, nb)
    total = int(sets[nsl].item()) * slice_size  # exec->copy_val_to_host (csr.cpp:352)
    cols = torch.full((max(total, 1),), -1, dtype=torch.int32, device="cuda:0")
    vals = torch.zeros(max(total, 1), dtype=torch.float64, device="cuda:0")
    gk.csr_convert_to_sellp_f64_i32(stream_ptr(), nrows, rpd, cid, vd, slice_size, sets, lens, cols, vals)
    return sets, lens, cols, vals


def dev_to_hybrid(gk, nrows, ncols, rpd, cid, vd, kind=4, percent=0.8, ratio=1e-4, num_columns=0):
    import ctypes
    res = ctypes.c_int64(0)
    gk.hybrid_ell_width_i32(stream_ptr(), rpd, nrows, kind, percent, ratio, num_columns, ctypes.addressof(res))
    ell_lim = min(int(res.value), ncols)
    crp = torch.zeros(nrows + 1, dtype=torch.int64, device="cuda:0")
    ws, nb = ws_for(gk, nrows + 1)
    gk.hybrid_compute_coo_row_ptrs_i32(stream_ptr(), rpd, nrows, ell_lim, crp, ws, nb)
    coo_nnz = int(crp[nrows].item())
    ell_cols = torch.full((max(ell_lim * nrows, 1),), -1, dtype=torch.int32, device="cuda:0")
    ell_vals = torch.zeros(max(ell_lim * nrows, 1), dtype=torch.float64, device="cuda:0")
    cr = torch.zeros(max(coo_nnz, 1), dtype=torch.int32, device="cuda:0")
    cc = torch.zeros(max(coo_nnz, 1), dtype=torch.int32, device="cuda:0")
    cv = torch.zeros(max(coo_nnz, 1), dtype=torch.float64, device="cuda:0")
    gk.csr_convert_to_hybrid_f64_i32(stream_ptr(), nrows, rpd, cid, vd, crp, ell_lim, nrows, ell_cols, ell_vals, cr, cc, cv)
    return dict(ell_lim=ell_lim, ell_stride=nrows, ell_cols=ell_cols, ell_vals=ell_vals, coo_nnz=coo_nnz,
                coo_rows=cr, coo_cols=cc, coo_vals=cv, coo_row_ptrs=crp)


def csr(name):
    m = G["csr"][name]
    return m["nrows"], m["ncols"], np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32), np.array(m["vals"])


# ---- known answers of the reference's tests ---------------------------------

def test_conversions_known_answers(gk):
    n, nc, rp, ci, v = csr("mtx")
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    idxs = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    gk.convert_ptrs_to_idxs_i32(stream_ptr(), rpd, n, idxs)
    assert list(host(idxs)) == G["to_coo"]["row_idxs"]
    k, st, cols, vals = dev_to_ell(gk, n, rpd, cid, vd)
    assert (k, st) == (3, 2) and list(host(cols)) == G["to_ell"]["col_idxs"] and list(host(vals)) == G["to_ell"]["vals"]
    sets, lens, cols, vals = dev_to_sellp(gk, n, rpd, cid, vd)
    g = G["to_sellp"]
    assert list(host(sets)) == g["slice_sets"] and list(host(lens)) == g["slice_lengths"]
    for kk, e in g["checks"]["col_idxs"].items():
        assert host(cols)[int(kk)] == e
    for kk, e in g["checks"]["vals"].items():
        assert host(vals)[int(kk)] == e
    h = dev_to_hybrid(gk, n, nc, rpd, cid, vd, kind=4)
    g = G["to_hybrid_automatic"]
    assert h["ell_lim"] == 0 and h["coo_nnz"] == 4
    assert list(host(h["coo_rows"])) == g["coo_row_idxs"] and list(host(h["coo_vals"])) == g["coo_vals"]
    n, nc, rp, ci, v = csr("mtx2")
    h = dev_to_hybrid(gk, n, nc, dev(rp), dev(ci), dev(v), kind=0, num_columns=2)
    g = G["to_hybrid_column2"]
    assert h["ell_lim"] == 2 and list(host(h["ell_vals"])) == g["ell_vals"] and list(host(h["ell_cols"])) == g["ell_col_idxs"]
    assert list(host(h["coo_rows"])) == [0] and list(host(h["coo_cols"])) == [2] and list(host(h["coo_vals"])) == [2.0]


def test_prefix_sum_known_answer(gk):
    for dt, fn in ((torch.int32, gk.prefix_sum_i32), (torch.int64, gk.prefix_sum_i64)):
        v = torch.tensor(G["prefix_sum"]["vals"], dtype=dt, device="cuda:0")
        ws, nb = ws_for(gk, v.numel())
        fn(stream_ptr(), v, v.numel(), ws, nb)
        assert list(host(v)) == G["prefix_sum"]["expected"]


@pytest.mark.parametrize("case", G["applies"], ids=lambda c: c["name"])
def test_applies_known_answers(gk, oracle, case):
    n, nc, rp, ci, v = csr("mtx")
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    b = dev(np.array(case["b"], np.float64))
    nrhs = b.shape[1]
    al, be = scal(case.get("alpha")), scal(case.get("beta"))
    expect = np.array(case["expect"])

    def c0():
        return dev(np.array(case["c"], np.float64)) if al is not None else torch.full((n, nrhs), float("nan"), dtype=torch.float64, device="cuda:0")
    s = stream_ptr()
    for stride in (None, 16):
        k, st, cols, vals = dev_to_ell(gk, n, rpd, cid, vd, stride)
        c = c0()
        gk.ell_spmv_f64_i32(s, n, nc, nrhs, k, st, cols, vals, b, nrhs, c, nrhs, al, be)
        assert np.array_equal(host(c), expect), f"ell stride {st}"
    for ss, sf in ((64, 1), (2, 2)):
        sets, lens, cols, vals = dev_to_sellp(gk, n, rpd, cid, vd, ss, sf)
        c = c0()
        gk.sellp_spmv_f64_i32(s, n, nc, nrhs, ss, sets, lens, cols, vals, b, nrhs, c, nrhs, al, be)
        assert np.array_equal(host(c), expect), f"sellp {ss} {sf}"
    rows = dev(np.array(G["to_coo"]["row_idxs"], np.int32))
    c = c0()
    gk.coo_spmv_f64_i32(s, n, nc, nrhs, 4, rows, cid, vd, b, nrhs, c, nrhs, al, be)
    assert np.array_equal(host(c), expect), "coo"
    h = G["apply_layouts"]["hybrid_mtx3"]
    c = c0()
    gk.hybrid_spmv_f64_i32(s, n, nc, nrhs, 2, 2, dev(np.array(h["ell_col_idxs"], np.int32)), dev(np.array(h["ell_vals"])),
                           1, dev(np.array(h["coo_row_idxs"], np.int32)), dev(np.array(h["coo_col_idxs"], np.int32)),
                           dev(np.array(h["coo_vals"])), b, nrhs, c, nrhs, al, be)
    assert np.array_equal(host(c), expect), "hybrid"


# ---- random matrices against the oracle -------------------------------------

MATS = {
    "rand532x231": lambda: (532, 231) + matgen.random_csr(532, 231, 0, 40, seed=42),
    "rand_unsorted": lambda: (300, 300) + matgen.random_csr(300, 300, 1, 30, seed=3, sort=False),
    "heavy_tail": lambda: (2000, 3000) + matgen.random_csr(2000, 3000, 1, 700, seed=5, dist="tail"),
    "poisson": lambda: (lambda t: (t[0], t[0], t[1], t[2], t[3]))(matgen.poisson_2d_5pt(61, 47)),
    "empty_rows": lambda: (64 * 3 + 5, 50) + matgen.random_csr(64 * 3 + 5, 50, 0, 2, seed=8),
}


@pytest.mark.parametrize("name", sorted(MATS))
def test_conversions_bitexact_vs_oracle(gk, oracle, name):
    n, nc, rp, ci, v = MATS[name]()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    nnz = int(rp[-1])
    # csr -> coo rows, and back
    idxs = torch.full((max(nnz, 1),), -5, dtype=torch.int32, device="cuda:0")
    gk.convert_ptrs_to_idxs_i32(stream_ptr(), rpd, n, idxs)
    e = np.zeros(max(nnz, 1), np.int32)
    oracle.ref_convert_ptrs_to_idxs(rp, n, e)
    assert np.array_equal(host(idxs)[:nnz], e[:nnz])
    back = torch.full((n + 1,), 7, dtype=torch.int32, device="cuda:0")
    ws, nb = ws_for(gk, n + 1)
    gk.convert_idxs_to_ptrs_i32(stream_ptr(), idxs, nnz, n, back, ws, nb)
    assert np.array_equal(host(back), rp)
    sizes = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    gk.convert_ptrs_to_sizes_i32(stream_ptr(), rpd, n, sizes)
    assert np.array_equal(host(sizes), np.diff(rp))
    # ell
    k, st, cols, vals = dev_to_ell(gk, n, rpd, cid, vd)
    ek, est, ecols, evals = fu.oracle_to_ell(oracle, n, rp, ci, v)
    assert (k, st) == (ek, est) and np.array_equal(host(cols), ecols) and np.array_equal(host(vals), evals)
    # sellp, two geometries
    for ss, sf in ((64, 1), (32, 4), (7, 3)):
        sets, lens, cols, vals = dev_to_sellp(gk, n, rpd, cid, vd, ss, sf)
        es, el, ecols, evals = fu.oracle_to_sellp(oracle, n, rp, ci, v, ss, sf)
        assert np.array_equal(host(sets).astype(np.uint64), es) and np.array_equal(host(lens).astype(np.uint64)[:len(el)], el)
        assert np.array_equal(host(cols), ecols) and np.array_equal(host(vals), evals)
    # hybrid, every strategy
    for kw in (dict(kind=4), dict(kind=0, num_columns=3), dict(kind=1, percent=0.8), dict(kind=2, percent=0.5, ratio=0.01), dict(kind=3)):
        h = dev_to_hybrid(gk, n, nc, rpd, cid, vd, **kw)
        e = fu.oracle_to_hybrid(oracle, n, nc, rp, ci, v, **kw)
        assert h["ell_lim"] == e["ell_lim"] and h["coo_nnz"] == e["coo_nnz"], kw
        assert np.array_equal(host(h["coo_row_ptrs"]), e["coo_row_ptrs"])
        for key in ("ell_cols", "ell_vals", "coo_rows", "coo_cols", "coo_vals"):
            assert np.array_equal(host(h[key]), e[key]), (kw, key)


@pytest.mark.parametrize("nrhs", [1, 3, 4, 15])
@pytest.mark.parametrize("advanced", [False, True], ids=["simple", "advanced"])
@pytest.mark.parametrize("name", sorted(MATS))
def test_spmv_all_formats_vs_oracle(gk, oracle, name, advanced, nrhs):
    n, nc, rp, ci, v = MATS[name]()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    nnz = int(rp[-1])
    rng = np.random.default_rng(15)
    b = rng.standard_normal((nc, nrhs))
    c0 = rng.standard_normal((n, nrhs))
    bd = dev(b)
    alpha, beta = (2.0, -1.0) if advanced else (None, None)
    al, be = scal(alpha), scal(beta)
    s = stream_ptr()

    def start():
        return dev(c0) if advanced else torch.full((n, nrhs), float("nan"), dtype=torch.float64, device="cuda:0")
    # ELL: bit-exact
    k, st, cols, vals = dev_to_ell(gk, n, rpd, cid, vd)
    c = start()
    gk.ell_spmv_f64_i32(s, n, nc, nrhs, k, st, cols, vals, bd, nrhs, c, nrhs, al, be)
    e = c0.copy()
    if advanced:
        oracle.ref_ell_advanced_spmv(n, nrhs, alpha, k, st, host(cols), host(vals), b, nrhs, beta, e, nrhs)
    else:
        oracle.ref_ell_spmv(n, nrhs, k, st, host(cols), host(vals), b, nrhs, e, nrhs)
    assert np.array_equal(host(c), e), "ell"
    # SELL-P: bit-exact
    for ss, sf in ((64, 1), (16, 2)):
        sets, lens, cols, vals = dev_to_sellp(gk, n, rpd, cid, vd, ss, sf)
        c = start()
        gk.sellp_spmv_f64_i32(s, n, nc, nrhs, ss, sets, lens, cols, vals, bd, nrhs, c, nrhs, al, be)
        e = c0.copy()
        hs, hl = host(sets).astype(np.uint64), host(lens).astype(np.uint64)
        if advanced:
            oracle.ref_sellp_advanced_spmv(n, nrhs, alpha, ss, hs, hl, host(cols), host(vals), b, nrhs, beta, e, nrhs)
        else:
            oracle.ref_sellp_spmv(n, nrhs, ss, hs, hl, host(cols), host(vals), b, nrhs, e, nrhs)
        assert np.array_equal(host(c), e), f"sellp {ss}"
    # COO: r<double>
    rows = torch.zeros(max(nnz, 1), dtype=torch.int32, device="cuda:0")
    gk.convert_ptrs_to_idxs_i32(s, rpd, n, rows)
    c = start()
    gk.coo_spmv_f64_i32(s, n, nc, nrhs, nnz, rows, cid, vd, bd, nrhs, c, nrhs, al, be)
    e = c0.copy()
    if advanced:
        oracle.ref_coo_advanced_spmv(n, nnz, nrhs, alpha, host(rows), ci, v, b, nrhs, beta, e, nrhs)
    else:
        oracle.ref_coo_spmv(n, nnz, nrhs, host(rows), ci, v, b, nrhs, e, nrhs)
    assert matgen.rel_err(host(c), e) <= 1e-14, "coo"
    # Hybrid (automatic and a forced split): ELL part exact, COO part atomics
    for kw in (dict(kind=4), dict(kind=0, num_columns=2)):
        h = dev_to_hybrid(gk, n, nc, rpd, cid, vd, **kw)
        c = start()
        gk.hybrid_spmv_f64_i32(s, n, nc, nrhs, h["ell_lim"], h["ell_stride"], h["ell_cols"], h["ell_vals"], h["coo_nnz"],
                               h["coo_rows"], h["coo_cols"], h["coo_vals"], bd, nrhs, c, nrhs, al, be)
        e = c0.copy()
        if advanced:
            oracle.ref_csr_advanced_spmv(n, nrhs, alpha, rp, ci, v, b, nrhs, beta, e, nrhs)
        else:
            oracle.ref_csr_spmv(n, nrhs, rp, ci, v, b, nrhs, e, nrhs)
        assert matgen.rel_err(host(c), e) <= 1e-14, f"hybrid {kw}"


def test_coo_exact_when_rows_fit_a_tile_and_unsorted_ok(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(300, 300)  # 1536-nonzero tiles cut rows: most rows inside one tile
    nnz = int(rp[-1])
    rows = np.zeros(nnz, np.int32)
    oracle.ref_convert_ptrs_to_idxs(rp, n, rows)
    b = np.random.default_rng(1).standard_normal((n, 1))
    c = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    gk.coo_spmv_f64_i32(stream_ptr(), n, n, 1, nnz, dev(rows), dev(ci), dev(v), dev(b), 1, c, 1, None, None)
    e = np.zeros((n, 1))
    oracle.ref_coo_spmv(n, nnz, 1, rows, ci, v, b, 1, e, 1)
    got = host(c)
    cut_rows = set(rows[np.arange(1536, nnz, 1536)]) | set(rows[np.arange(1535, nnz, 1536)])
    inside = np.array([r not in cut_rows for r in range(n)])
    assert np.array_equal(got[inside], e[inside])      # same order as the reference
    assert matgen.rel_err(got, e) <= 1e-15
    # a random permutation of the entries (unsorted COO) still gives the right answer
    perm = np.random.default_rng(2).permutation(nnz)
    c.zero_()
    gk.coo_spmv_f64_i32(stream_ptr(), n, n, 1, nnz, dev(rows[perm]), dev(ci[perm]), dev(v[perm]), dev(b), 1, c, 1, None, None)
    assert matgen.rel_err(host(c), e) <= 1e-14


@pytest.mark.parametrize("n", [1, 1023, 1024, 1025, 1 << 20, (1 << 20) + 7, 3_000_001])
def test_prefix_sum_sizes(gk, n):
    rng = np.random.default_rng(n)
    v = rng.integers(0, 9, size=n)
    for dt, fn, npdt in ((torch.int32, gk.prefix_sum_i32, np.int32), (torch.int64, gk.prefix_sum_i64, np.int64)):
        d = dev(v.astype(npdt))
        ws, nb = ws_for(gk, n)
        fn(stream_ptr(), d, n, ws, nb)
        e = np.concatenate([[0], np.cumsum(v)[:-1]]).astype(npdt)
        assert np.array_equal(host(d), e)


def test_full_size_p2_formats(gk, oracle):
    """1M-row Poisson: conversions bit-exact against the oracle, every format's
    SpMV equal to the CSR result (ELL/SELL-P exactly: same per-row order)."""
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
    xd = dev(x)
    e = np.empty((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, x, 1, e, 1)
    s = stream_ptr()
    k, st, cols, vals = dev_to_ell(gk, n, rpd, cid, vd)
    ek, est, ecols, evals = fu.oracle_to_ell(oracle, n, rp, ci, v)
    assert k == 5 and np.array_equal(host(cols), ecols) and np.array_equal(host(vals), evals)
    c = torch.empty((n, 1), dtype=torch.float64, device="cuda:0")
    gk.ell_spmv_f64_i32(s, n, n, 1, k, st, cols, vals, xd, 1, c, 1, None, None)
    assert np.array_equal(host(c), e)
    sets, lens, cols, vals = dev_to_sellp(gk, n, rpd, cid, vd)
    es, el, ecols, evals = fu.oracle_to_sellp(oracle, n, rp, ci, v)
    assert np.array_equal(host(cols), ecols) and np.array_equal(host(vals), evals)
    gk.sellp_spmv_f64_i32(s, n, n, 1, 64, sets, lens, cols, vals, xd, 1, c, 1, None, None)
    assert np.array_equal(host(c), e)
    rows = torch.zeros(int(rp[-1]), dtype=torch.int32, device="cuda:0")
    gk.convert_ptrs_to_idxs_i32(s, rpd, n, rows)
    gk.coo_spmv_f64_i32(s, n, n, 1, int(rp[-1]), rows, cid, vd, xd, 1, c, 1, None, None)
    assert matgen.rel_err(host(c), e) <= 1e-15
    h = dev_to_hybrid(gk, n, n, rpd, cid, vd, kind=0, num_columns=4)
    assert h["coo_nnz"] == int(np.sum(np.maximum(np.diff(rp) - 4, 0)))
    gk.hybrid_spmv_f64_i32(s, n, n, 1, 4, n, h["ell_cols"], h["ell_vals"], h["coo_nnz"], h["coo_rows"], h["coo_cols"],
                           h["coo_vals"], xd, 1, c, 1, None, None)
    assert matgen.rel_err(host(c), e) <= 1e-15


@pytest.mark.parametrize("bits", [32, 64])
def test_format_conversion_known_answers(gk, bits):
    """reference/test/components/format_conversion_kernels.cpp:62-128, both index types, through the C ABI"""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats.json")))["format_conversion"]
    npt, tt = (np.int32, torch.int32) if bits == 32 else (np.int64, torch.int64)
    sfx = "_i32" if bits == 32 else "_i64"
    ptrs = dev(np.array(g["ptrs"], npt))
    idxs = torch.full((5,), -1, dtype=tt, device="cuda:0")
    getattr(gk, "convert_ptrs_to_idxs" + sfx)(stream_ptr(), ptrs, 4, idxs)
    assert list(host(idxs)) == g["idxs_of_ptrs"]
    sizes = torch.full((4,), 99, dtype=torch.int64, device="cuda:0")
    getattr(gk, "convert_ptrs_to_sizes" + sfx)(stream_ptr(), ptrs, 4, sizes)
    assert list(host(sizes)) == g["sizes_of_ptrs"]
    nb = gk.prefix_sum_workspace_bytes(g["empty_num_blocks"] + 1)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda:0")
    out = torch.full((6,), -1, dtype=tt, device="cuda:0")
    getattr(gk, "convert_idxs_to_ptrs" + sfx)(stream_ptr(), dev(np.array(g["idxs"], npt)), 6, g["num_blocks"], out, ws, nb)
    assert list(host(out)) == g["ptrs_of_idxs"]
    out = torch.full((g["empty_num_blocks"] + 1,), -1, dtype=tt, device="cuda:0")
    getattr(gk, "convert_idxs_to_ptrs" + sfx)(stream_ptr(), torch.zeros(1, dtype=tt, device="cuda:0"), 0, g["empty_num_blocks"], out, ws, nb)
    assert not host(out).any()
    getattr(gk, "convert_ptrs_to_idxs" + sfx)(stream_ptr(), torch.zeros(10, dtype=tt, device="cuda:0"), 9, None)   # empty: must not fault
    torch.cuda.synchronize()
