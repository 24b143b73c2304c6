"""Pins the oracle's ELL/SELL-P/COO/Hybrid kernels, conversions and index
components against the reference's known answers (tests/golden/formats.json)."""
import json
import os

import numpy as np
import pytest

import formats_util as fu

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats.json")))


def csr(name):
    m = G["csr"][name]
    return m["nrows"], m["ncols"], np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32), np.array(m["vals"])


def test_prefix_sum_and_components(oracle):
    v = np.array(G["prefix_sum"]["vals"], np.int32)
    oracle.ref_prefix_sum_i32(v, len(v))
    assert list(v) == G["prefix_sum"]["expected"]
    v = np.array(G["prefix_sum"]["vals"], np.int64)
    oracle.ref_prefix_sum_i64(v, len(v))
    assert list(v) == G["prefix_sum"]["expected"]
    n, _, rp, ci, vals = csr("mtx")
    idxs = np.zeros(4, np.int32)
    oracle.ref_convert_ptrs_to_idxs(rp, n, idxs)
    assert list(idxs) == G["to_coo"]["row_idxs"]
    back = np.full(3, 9, np.int32)
    oracle.ref_convert_idxs_to_ptrs(idxs, 4, n, back)
    assert list(back) == list(rp)
    sizes = np.zeros(2, np.uint64)
    oracle.ref_convert_ptrs_to_sizes(rp, n, sizes)
    assert list(sizes) == [3, 1]


def test_csr_to_ell(oracle):
    n, _, rp, ci, v = csr("mtx")
    k, stride, cols, vals = fu.oracle_to_ell(oracle, n, rp, ci, v)
    g = G["to_ell"]
    assert k == g["num_stored_per_row"] and stride == g["stride"]
    assert list(cols) == g["col_idxs"] and list(vals) == g["vals"]


def test_csr_to_sellp(oracle):
    n, _, rp, ci, v = csr("mtx")
    g = G["to_sellp"]
    sets, lens, cols, vals = fu.oracle_to_sellp(oracle, n, rp, ci, v, g["slice_size"], g["stride_factor"])
    assert list(sets) == g["slice_sets"] and list(lens) == g["slice_lengths"]
    for k, e in g["checks"]["col_idxs"].items():
        assert cols[int(k)] == e
    for k, e in g["checks"]["vals"].items():
        assert vals[int(k)] == e


@pytest.mark.parametrize("key", ["to_hybrid_automatic", "to_hybrid_column2"])
def test_csr_to_hybrid(oracle, key):
    g = G[key]
    n, nc, rp, ci, v = csr(g["matrix"])
    st = g["strategy"]
    h = fu.oracle_to_hybrid(oracle, n, nc, rp, ci, v, kind=st["kind"], num_columns=st.get("num_columns", 0))
    assert h["ell_lim"] == g["ell_num_stored_per_row"] and h["ell_stride"] == g["ell_stride"]
    assert h["coo_nnz"] == len(g["coo_vals"])
    assert list(h["coo_rows"][:h["coo_nnz"]]) == g["coo_row_idxs"]
    assert list(h["coo_cols"][:h["coo_nnz"]]) == g["coo_col_idxs"]
    assert list(h["coo_vals"][:h["coo_nnz"]]) == g["coo_vals"]
    if "ell_vals" in g:
        assert list(h["ell_vals"]) == g["ell_vals"]
        # the reference test expects col 0 for the explicit zero of mtx2 (a stored entry)
        assert list(h["ell_cols"]) == g["ell_col_idxs"]


def _apply_all_formats(oracle, case):
    """Yields (format name, result) of the oracle for every format/layout."""
    n, nc, rp, ci, v = csr("mtx")
    b = np.array(case["b"], np.float64)
    nrhs = b.shape[1]
    adv = "alpha" in case
    c0 = np.array(case["c"], np.float64) if adv else np.full((n, nrhs), np.nan)

    def run(simple, advanced):
        c = c0.copy()
        if adv:
            advanced(c)
        else:
            simple(c)
        return c

    for stride in (None, G["apply_layouts"]["ell_stride16"]["stride"]):
        k, st, cols, vals = fu.oracle_to_ell(oracle, n, rp, ci, v, stride)
        yield f"ell_stride{st}", run(
            lambda c: oracle.ref_ell_spmv(n, nrhs, k, st, cols, vals, b, nrhs, c, nrhs),
            lambda c: oracle.ref_ell_advanced_spmv(n, nrhs, case["alpha"], k, st, cols, vals, b, nrhs, case["beta"], c, nrhs))
    for ss, sf in ((64, 1), (2, 2)):
        sets, lens, cols, vals = fu.oracle_to_sellp(oracle, n, rp, ci, v, ss, sf)
        yield f"sellp_{ss}_{sf}", run(
            lambda c: oracle.ref_sellp_spmv(n, nrhs, ss, sets, lens, cols, vals, b, nrhs, c, nrhs),
            lambda c: oracle.ref_sellp_advanced_spmv(n, nrhs, case["alpha"], ss, sets, lens, cols, vals, b, nrhs, case["beta"], c, nrhs))
    rows = np.array(G["to_coo"]["row_idxs"], np.int32)
    yield "coo", run(
        lambda c: oracle.ref_coo_spmv(n, 4, nrhs, rows, ci, v, b, nrhs, c, nrhs),
        lambda c: oracle.ref_coo_advanced_spmv(n, 4, nrhs, case["alpha"], rows, ci, v, b, nrhs, case["beta"], c, nrhs))
    h = G["apply_layouts"]["hybrid_mtx3"]
    ev, ec = np.array(h["ell_vals"]), np.array(h["ell_col_idxs"], np.int32)
    cr, cc, cv = (np.array(h["coo_row_idxs"], np.int32), np.array(h["coo_col_idxs"], np.int32), np.array(h["coo_vals"]))

    def hyb_simple(c):
        oracle.ref_ell_spmv(n, nrhs, 2, 2, ec, ev, b, nrhs, c, nrhs)
        oracle.ref_coo_spmv2(1, nrhs, cr, cc, cv, b, nrhs, c, nrhs)

    def hyb_adv(c):
        oracle.ref_ell_advanced_spmv(n, nrhs, case["alpha"], 2, 2, ec, ev, b, nrhs, case["beta"], c, nrhs)
        oracle.ref_coo_advanced_spmv2(1, nrhs, case["alpha"], cr, cc, cv, b, nrhs, c, nrhs)

    yield "hybrid_mtx3", run(hyb_simple, hyb_adv)


@pytest.mark.parametrize("case", G["applies"], ids=lambda c: c["name"])
def test_applies_known_answers(oracle, case):
    for name, got in _apply_all_formats(oracle, case):
        assert np.array_equal(got, np.array(case["expect"])), name


def test_format_conversion_known_answers(oracle):
    """reference/test/components/format_conversion_kernels.cpp:62-128"""
    g = G["format_conversion"]
    ptrs = np.array(g["ptrs"], np.int32)
    idxs = np.full(5, -1, np.int32)
    oracle.ref_convert_ptrs_to_idxs(ptrs, 4, idxs)
    assert list(idxs) == g["idxs_of_ptrs"]
    sizes = np.full(4, 99, np.uint64)
    oracle.ref_convert_ptrs_to_sizes(ptrs, 4, sizes)
    assert list(sizes) == g["sizes_of_ptrs"]
    out = np.full(6, -1, np.int32)
    oracle.ref_convert_idxs_to_ptrs(np.array(g["idxs"], np.int32), 6, g["num_blocks"], out)
    assert list(out) == g["ptrs_of_idxs"]
    out = np.full(g["empty_num_blocks"] + 1, -1, np.int32)   # ConvertsEmptyIdxsToPtrs
    oracle.ref_convert_idxs_to_ptrs(np.zeros(1, np.int32), 0, g["empty_num_blocks"], out)
    assert not out.any()
    oracle.ref_convert_ptrs_to_idxs(np.zeros(10, np.int32), 9, np.zeros(1, np.int32))   # ConvertsEmptyPtrsToIdxs: no write
