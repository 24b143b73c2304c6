"""Pins the oracle's block-Jacobi (find_blocks / generate / apply, scalar
variant) against reference/test/preconditioner/jacobi_kernels.cpp."""
import json
import os

import numpy as np
import pytest

import matgen

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobi.json")))
R = G["tol"]


def mtx():
    m = G["mtx"]
    return m["n"], np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32), np.array(m["vals"])


def scheme(oracle, max_bs, stride=32):
    s = np.zeros(3, np.int64)
    oracle.ref_jacobi_storage_scheme(max_bs, stride, s)
    return s


def generate(oracle, n, rp, ci, v, ptrs, max_bs, stride=32, cond=False):
    s = scheme(oracle, max_bs, stride)
    nb = len(ptrs) - 1
    blocks = np.full(int(oracle.ref_jacobi_storage_space(s, nb)), np.nan)
    c = np.zeros(nb) if cond else None
    oracle.ref_jacobi_generate(n, rp, ci, v, nb, s, np.array(ptrs, np.int32), c, blocks)
    return s, blocks, c


def block_of(s, blocks, b, bs):
    gs = 1 << int(s[2])
    off = int(s[1]) * (b >> int(s[2])) + int(s[0]) * (b & (gs - 1))
    p = int(s[0]) << int(s[2])
    return np.array([[blocks[off + r + c * p] for c in range(bs)] for r in range(bs)])


@pytest.mark.parametrize("case", G["find_blocks"], ids=lambda c: c["name"])
def test_find_blocks(oracle, case):
    if case.get("use_mtx"):
        n, rp, ci, _ = mtx()
    else:
        n, rp, ci = case["n"], np.array(case["row_ptrs"], np.int32), np.array(case["col_idxs"], np.int32)
    ptrs = np.zeros(n + 1, np.int32)
    nb = oracle.ref_jacobi_find_blocks(n, rp, ci, case["max_block_size"], ptrs)
    assert list(ptrs[:nb + 1]) == case["expect"]


@pytest.mark.parametrize("stride", [32, 64])
def test_inverts_diagonal_blocks_and_condition_numbers(oracle, stride):
    n, rp, ci, v = mtx()
    s, blocks, cond = generate(oracle, n, rp, ci, v, G["block_pointers"], G["max_block_size"], stride, cond=True)
    assert np.allclose(block_of(s, blocks, 0, 2), G["inverse_blocks"]["b1"], rtol=0, atol=R)
    assert np.allclose(block_of(s, blocks, 1, 3), G["inverse_blocks"]["b2"], rtol=0, atol=R)
    assert np.allclose(cond, G["conditioning"]["expect"], rtol=0, atol=G["conditioning"]["tol"])


def test_pivots_when_inverting(oracle):
    p = G["pivoting"]
    s, blocks, _ = generate(oracle, 3, np.array(p["row_ptrs"], np.int32), np.array(p["col_idxs"], np.int32),
                            np.array(p["vals"]), p["block_pointers"], 3)
    assert np.allclose(block_of(s, blocks, 0, 3), p["inverse"], rtol=0, atol=R)


def _strided(a, stride):
    a = np.array(a, np.float64)
    out = np.full((a.shape[0], stride or a.shape[1]), -9.0)
    out[:, :a.shape[1]] = a
    return out


@pytest.mark.parametrize("case", G["applies"], ids=lambda c: c["name"])
def test_applies(oracle, case):
    n, rp, ci, v = mtx()
    s, blocks, _ = generate(oracle, n, rp, ci, v, G["block_pointers"], G["max_block_size"])
    ptrs = np.array(G["block_pointers"], np.int32)
    st = case.get("stride")
    x, b = _strided(case["x"], st), _strided(case["b"], st)
    nrhs = np.array(case["x"]).shape[1]
    if "alpha" in case:
        oracle.ref_jacobi_apply(2, s, ptrs, blocks, nrhs, case["alpha"], b, b.shape[1], case["beta"], x, x.shape[1])
    else:
        oracle.ref_jacobi_simple_apply(2, s, ptrs, blocks, nrhs, b, b.shape[1], x, x.shape[1])
    assert matgen.rel_err(x[:, :nrhs], case["expect"]) <= R
    assert np.all(x[:, nrhs:] == -9.0)


@pytest.mark.parametrize("case", G["scalar_applies"], ids=lambda c: c["name"])
def test_scalar_jacobi(oracle, case):
    n, rp, ci, v = mtx()
    d = np.zeros(n)
    oracle.ref_csr_extract_diagonal(n, rp, ci, v, d)
    inv = np.zeros(n)
    oracle.ref_jacobi_invert_diagonal(n, d, inv)
    assert np.array_equal(inv, np.full(n, 0.25))
    st = case.get("stride")
    x, b = _strided(case["x"], st), _strided(case["b"], st)
    nrhs = np.array(case["x"]).shape[1]
    if "alpha" in case:
        oracle.ref_jacobi_scalar_apply(n, nrhs, inv, case["alpha"], b, b.shape[1], case["beta"], x, x.shape[1])
    else:
        oracle.ref_jacobi_simple_scalar_apply(n, nrhs, inv, b, b.shape[1], x, x.shape[1])
    assert matgen.rel_err(x[:, :nrhs], case["expect"]) <= R


# ---- adaptive precision block storage --------------------------------------------------
A = G["adaptive"]
PREC_DTYPE = {0x00: np.float64, 0x01: np.float32, 0x02: np.uint16, 0x10: np.uint32, 0x11: np.uint16, 0x20: np.uint16}


def generate_adaptive(oracle, n, rp, ci, v, ptrs, max_bs, precisions, accuracy=1e-1, stride=32):
    s = scheme(oracle, max_bs, stride)
    nb = len(ptrs) - 1
    blocks = np.full(int(oracle.ref_jacobi_storage_space(s, nb)), np.nan)
    cond = np.zeros(nb)
    prec = np.array([precisions[i % len(precisions)] for i in range(nb)], np.uint8)  # initialize_precisions :485-493
    oracle.ref_jacobi_generate_adaptive(n, rp, ci, v, nb, s, np.array(ptrs, np.int32), accuracy, cond, prec, blocks)
    return s, blocks, cond, prec


def reduced_block_of(s, blocks, b, bs, prec):
    """the block as stored: reinterpret the group's memory in the reduced type"""
    gs = 1 << int(s[2])
    group = blocks[int(s[1]) * (b >> int(s[2])):].view(PREC_DTYPE[prec])
    off = int(s[0]) * (b & (gs - 1))
    p = int(s[0]) << int(s[2])
    return np.array([[group[off + r + c * p] for c in range(bs)] for r in range(bs)])


def test_reduced_storage_types(oracle):
    """core/base/extended_float.hpp: float rounds to nearest, half truncates the
    float's significand and flushes its subnormal range, truncated<> keeps the
    upper bits of the IEEE pattern"""
    rt = oracle.ref_jacobi_round_to_precision
    v = 1.0 + 2.0 ** -30 + 2.0 ** -12
    assert rt(0x00, v) == v
    assert rt(0x01, v) == float(np.float32(v))
    assert rt(0x10, v) == 1.0 + 2.0 ** -12          # 20 significand bits kept
    assert rt(0x02, v) == 1.0                       # 10 bits, truncated (nearest would stay 1.0 too)
    assert rt(0x02, 1.0 + 2.0 ** -10 + 2.0 ** -11) == 1.0 + 2.0 ** -10   # truncation, not rounding to nearest-even (1 + 2^-9)
    assert rt(0x11, 1.0 + 2.0 ** -7 + 2.0 ** -8) == 1.0 + 2.0 ** -7     # 7 bits of the float
    assert rt(0x20, 1.0 + 2.0 ** -4 + 2.0 ** -5) == 1.0 + 2.0 ** -4     # 4 bits of the double
    assert rt(0x02, 1e-6) == 0.0 and rt(0x02, -1e-6) == 0.0 and np.signbit(rt(0x02, -1e-6))  # below 2^-14: flushed
    assert rt(0x02, 1e6) == np.inf and rt(0x02, -1e6) == -np.inf       # above 65504: infinity
    assert rt(0x02, 65504.0) == 65504.0 and rt(0x02, 2.0 ** -14) == 2.0 ** -14
    assert np.isnan(rt(0x02, np.nan)) and rt(0x01, np.inf) == np.inf
    assert rt(0x10, -3.75) == -3.75 and rt(0x20, -3.75) == -3.75 and rt(0x11, -3.75) == -3.75


def test_inverts_diagonal_blocks_with_adaptive_precision(oracle):
    n, rp, ci, v = mtx()
    s, blocks, cond, prec = generate_adaptive(oracle, n, rp, ci, v, G["block_pointers"],
                                              A["max_block_size_group_of_one"], A["block_precisions"])
    assert list(prec) == [1, 0]
    b1 = reduced_block_of(s, blocks, 0, 2, 1)
    assert b1.dtype == np.float32 and np.allclose(b1, G["inverse_blocks"]["b1"], rtol=0, atol=A["half_tol"])
    # exactly the fp64 inverse rounded to float
    s0, plain, _ = generate(oracle, n, rp, ci, v, G["block_pointers"], A["max_block_size_group_of_one"])
    assert np.array_equal(b1, block_of(s0, plain, 0, 2).astype(np.float32))
    assert np.allclose(block_of(s, blocks, 1, 3), G["inverse_blocks"]["b2"], rtol=0, atol=R)
    assert np.allclose(cond, G["conditioning"]["expect"], rtol=0, atol=G["conditioning"]["tol"])


def test_small_blocks_share_the_group_precision(oracle):
    # :450-470: both blocks in one group, requested (0,1) and (0,0) -> common (0,0)
    n, rp, ci, v = mtx()
    s, blocks, _, prec = generate_adaptive(oracle, n, rp, ci, v, G["block_pointers"],
                                           A["max_block_size_small_blocks"], A["block_precisions"])
    assert list(prec) == [0, 0]
    assert np.allclose(block_of(s, blocks, 0, 2), G["inverse_blocks"]["b1"], rtol=0, atol=R)
    assert np.allclose(block_of(s, blocks, 1, 3), G["inverse_blocks"]["b2"], rtol=0, atol=R)


def test_pivots_with_adaptive_precision(oracle):
    p = G["pivoting"]
    s, blocks, _, prec = generate_adaptive(oracle, 3, np.array(p["row_ptrs"], np.int32), np.array(p["col_idxs"], np.int32),
                                           np.array(p["vals"]), p["block_pointers"], 3, A["block_precisions"])
    assert list(prec) == [1]
    assert np.allclose(reduced_block_of(s, blocks, 0, 3, 1), p["inverse"], rtol=0, atol=A["half_tol"])


def test_selects_correct_block_precisions(oracle):
    n, rp, ci, v = mtx()
    c = A["selects"]
    _, _, cond, prec = generate_adaptive(oracle, n, rp, ci, v, G["block_pointers"], c["max_block_size"], [255], c["accuracy"])
    assert list(prec) == c["expect"]        # u*cond ~1.2e-3 -> half; ~2.0e-3 -> float
    assert 2.0 ** -11 * cond[0] < c["accuracy"] < 2.0 ** -11 * cond[1]


def test_avoids_precisions_that_overflow(oracle):
    c = A["overflow"]
    rows, cols, vals = [], [], []
    for k, blk in enumerate(c["diag_blocks"]):
        for i in range(2):
            for j in range(2):
                rows.append(2 * k + i); cols.append(2 * k + j); vals.append(blk[i][j])
    rp = np.zeros(5, np.int32)
    np.add.at(rp, np.array(rows) + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    _, _, _, prec = generate_adaptive(oracle, 4, rp, np.array(cols, np.int32), np.array(vals), c["block_pointers"],
                                      c["max_block_size"], [255], c["accuracy"])
    assert list(prec) == c["expect"]        # both blocks in one group, both need truncated<float,2> = (1,1)


@pytest.mark.parametrize("case", A["applies"], ids=lambda c: c["name"])
def test_applies_with_adaptive_precision(oracle, case):
    n, rp, ci, v = mtx()
    s, blocks, _, prec = generate_adaptive(oracle, n, rp, ci, v, G["block_pointers"], case["max_block_size"],
                                           A["block_precisions"])
    ptrs = np.array(G["block_pointers"], np.int32)
    st = case.get("stride")
    x, b = _strided(case["x"], st), _strided(case["b"], st)
    nrhs = np.array(case["x"]).shape[1]
    oracle.ref_jacobi_apply_adaptive(2, s, ptrs, prec, blocks, nrhs, case.get("alpha", 1.0), b, b.shape[1],
                                     case.get("beta", 0.0), x, x.shape[1])
    assert matgen.rel_err(x[:, :nrhs], case["expect"]) <= A["half_tol"]
    assert np.all(x[:, nrhs:] == -9.0)


def test_adaptive_with_full_precision_equals_plain(oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(12)
    ptrs = np.zeros(n + 1, np.int32)
    nb = oracle.ref_jacobi_find_blocks(n, rp, ci, 8, ptrs)
    ptrs = ptrs[:nb + 1]
    s, blocks, cond = generate(oracle, n, rp, ci, v, list(ptrs), 8, 64, cond=True)
    s2, blocks2, cond2, prec = generate_adaptive(oracle, n, rp, ci, v, list(ptrs), 8, [0], stride=64)
    assert np.array_equal(np.nan_to_num(blocks), np.nan_to_num(blocks2)) and np.array_equal(cond, cond2) and not prec.any()


@pytest.mark.parametrize("stride", [32, 64])
def test_transposes_diagonal_blocks(oracle, stride):
    """jacobi_kernels.cpp:331-392: CanTransposeDiagonalBlocks and
    ...WithAdaptivePrecision (transpose == conj_transpose for real values)"""
    n, rp, ci, v = mtx()
    T = G["transposed_blocks"]
    ptrs = np.array(G["block_pointers"], np.int32)
    s, blocks, _ = generate(oracle, n, rp, ci, v, G["block_pointers"], G["max_block_size"], stride)
    out = np.full_like(blocks, np.nan)
    oracle.ref_jacobi_transpose(2, s, ptrs, None, blocks, out)
    assert np.allclose(block_of(s, out, 0, 2), T["b1"], rtol=0, atol=R)
    assert np.allclose(block_of(s, out, 1, 3), T["b2"], rtol=0, atol=R)
    assert np.array_equal(block_of(s, out, 1, 3), block_of(s, blocks, 1, 3).T)   # a move, not arithmetic
    back = np.full_like(blocks, np.nan)
    oracle.ref_jacobi_transpose(2, s, ptrs, None, out, back)
    assert np.array_equal(block_of(s, back, 0, 2), block_of(s, blocks, 0, 2))
    if stride != 32:
        return   # with 64 lanes per group both blocks share a group, hence one precision
    # adaptive: block 0 stored in float, block 1 in double
    s, blocks, cond, prec = generate_adaptive(oracle, n, rp, ci, v, G["block_pointers"],
                                              A["max_block_size_group_of_one"], A["block_precisions"], stride=stride)
    out = np.full_like(blocks, np.nan)
    oracle.ref_jacobi_transpose(2, s, ptrs, prec, blocks, out)
    b1 = reduced_block_of(s, out, 0, 2, 1)
    assert b1.dtype == np.float32 and np.allclose(b1, T["b1"], rtol=0, atol=A["half_tol"])
    assert np.array_equal(b1, reduced_block_of(s, blocks, 0, 2, 1).T)
    assert np.allclose(block_of(s, out, 1, 3), T["b2"], rtol=0, atol=R)
