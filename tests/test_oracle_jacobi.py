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
