"""GPU parity tests (C ABI) of block-Jacobi against the oracle: known answers
of reference/test/preconditioner/jacobi_kernels.cpp, then random block-diagonal
dominant matrices (the reference's cross-executor tests,
test/preconditioner/jacobi_kernels.cpp).  find_blocks bit-exact; generate and
apply bit-exact (same operations in the same order per element)."""
import json
import os

import numpy as np
import pytest
import torch

import matgen
from gpu_util import dev, host, stream_ptr
from test_oracle_jacobi import _strided, block_of

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobi.json")))
R = G["tol"]


def mtx():
    m = G["mtx"]
    return m["n"], np.array(m["row_ptrs"], np.int32), np.array(m["col_idxs"], np.int32), np.array(m["vals"])


def gpu_scheme(gk, max_bs):
    import ctypes
    out = (ctypes.c_int64 * 4)()
    gk.jacobi_storage_scheme(max_bs, ctypes.addressof(out))
    return np.array(list(out), np.int64)


def gpu_find_blocks(gk, n, rpd, cid, max_bs):
    import ctypes
    ptrs = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda:0")
    nbd = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    nws = gk.jacobi_find_blocks_workspace_bytes(n)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda:0")
    hn = ctypes.c_int64(-1)
    gk.jacobi_find_blocks_i32(stream_ptr(), n, rpd, cid, max_bs, ptrs, nbd, ws, nws, ctypes.addressof(hn))
    return int(hn.value), ptrs


def gpu_generate(gk, n, rpd, cid, vd, ptrs_d, nb, max_bs, cond=False):
    nel = gk.jacobi_storage_elements(max_bs, nb)
    blocks = torch.full((max(nel, 1),), float("nan"), dtype=torch.float64, device="cuda:0")
    c = torch.zeros(max(nb, 1), dtype=torch.float64, device="cuda:0") if cond else None
    gk.jacobi_generate_f64_i32(stream_ptr(), n, rpd, cid, vd, nb, max_bs, ptrs_d, c, blocks)
    return blocks, c


def test_storage_scheme_matches_reference_formula(gk, oracle):
    for max_bs in range(1, 33):
        e = np.zeros(3, np.int64)
        oracle.ref_jacobi_storage_scheme(max_bs, 64, e)  # HIP: warp size 64 (jacobi.hpp:581-586)
        s = gpu_scheme(gk, max_bs)
        assert list(s[:3]) == list(e) and s[3] == e[0] << e[2]
        for nb in (0, 1, 5, 64, 1000):
            assert gk.jacobi_storage_elements(max_bs, nb) == oracle.ref_jacobi_storage_space(e, nb)


@pytest.mark.parametrize("case", G["find_blocks"], ids=lambda c: c["name"])
def test_find_blocks_known_answers(gk, case):
    if case.get("use_mtx"):
        n, rp, ci, _ = mtx()
    else:
        n, rp, ci = case["n"], np.array(case["row_ptrs"], np.int32), np.array(case["col_idxs"], np.int32)
    nb, ptrs = gpu_find_blocks(gk, n, dev(rp), dev(ci), case["max_block_size"])
    assert list(host(ptrs)[:nb + 1]) == case["expect"]


def test_known_inverses_condition_numbers_and_applies(gk):
    n, rp, ci, v = mtx()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    ptrs = dev(np.array(G["block_pointers"], np.int32))
    blocks, cond = gpu_generate(gk, n, rpd, cid, vd, ptrs, 2, 3, cond=True)
    s = gpu_scheme(gk, 3)
    hb = host(blocks)
    assert np.allclose(block_of(s, hb, 0, 2), G["inverse_blocks"]["b1"], rtol=0, atol=R)
    assert np.allclose(block_of(s, hb, 1, 3), G["inverse_blocks"]["b2"], rtol=0, atol=R)
    assert np.allclose(host(cond), G["conditioning"]["expect"], rtol=0, atol=G["conditioning"]["tol"])
    for case in G["applies"]:
        st = case.get("stride")
        x, b = dev(_strided(case["x"], st)), dev(_strided(case["b"], st))
        nrhs = np.array(case["x"]).shape[1]
        al = dev(np.array([case["alpha"]])) if "alpha" in case else None
        be = dev(np.array([case["beta"]])) if "alpha" in case else None
        gk.jacobi_apply_f64_i32(stream_ptr(), 2, 3, ptrs, blocks, nrhs, al, b, b.shape[1], be, x, x.shape[1])
        assert matgen.rel_err(host(x)[:, :nrhs], case["expect"]) <= R, case["name"]
        assert np.all(host(x)[:, nrhs:] == -9.0)
    p = G["pivoting"]
    blocks, _ = gpu_generate(gk, 3, dev(np.array(p["row_ptrs"], np.int32)), dev(np.array(p["col_idxs"], np.int32)),
                             dev(np.array(p["vals"])), dev(np.array(p["block_pointers"], np.int32)), 1, 3)
    assert np.allclose(block_of(gpu_scheme(gk, 3), host(blocks), 0, 3), p["inverse"], rtol=0, atol=R)


@pytest.mark.parametrize("case", G["scalar_applies"], ids=lambda c: c["name"])
def test_scalar_jacobi_known_answers(gk, case):
    n, rp, ci, v = mtx()
    d = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    gk.csr_extract_diagonal_f64_i32(stream_ptr(), n, dev(rp), dev(ci), dev(v), d)
    inv = torch.zeros_like(d)
    gk.jacobi_invert_diagonal_f64(stream_ptr(), n, d, inv)
    assert np.array_equal(host(inv), np.full(n, 0.25))
    st = case.get("stride")
    x, b = dev(_strided(case["x"], st)), dev(_strided(case["b"], st))
    nrhs = np.array(case["x"]).shape[1]
    al = dev(np.array([case["alpha"]])) if "alpha" in case else None
    be = dev(np.array([case["beta"]])) if "alpha" in case else None
    gk.jacobi_scalar_apply_f64(stream_ptr(), n, nrhs, inv, al, b, b.shape[1], be, x, x.shape[1])
    assert matgen.rel_err(host(x)[:, :nrhs], case["expect"]) <= R


def block_structured_matrix(nblocks, max_bs, seed, fill=0.6):
    """Random matrix with dense-ish diagonal blocks of random sizes <= max_bs
    (rows inside a block share their pattern so that find_blocks detects them),
    diagonally dominant, plus off-block coupling."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, max_bs + 1, size=nblocks)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    n = int(starts[-1])
    rp, ci, v = [0], [], []
    for b in range(nblocks):
        s0, bs = int(starts[b]), int(sizes[b])
        cols_in = [c for c in range(s0, s0 + bs) if rng.random() < fill]
        extra = sorted(set(int(c) for c in rng.integers(0, n, size=2)) - set(range(s0, s0 + bs)))
        for r in range(s0, s0 + bs):
            cols = sorted(set(cols_in) | {r} | set(extra)) if bs > 1 else sorted({r} | set(extra))
            for c in cols:
                ci.append(c)
                v.append(rng.standard_normal() + (bs + 3.0 if c == r else 0.0))
            rp.append(len(ci))
    return n, np.array(rp, np.int32), np.array(ci, np.int32), np.array(v), starts.astype(np.int32)


@pytest.mark.parametrize("max_bs", [1, 2, 3, 4, 7, 8, 13, 16, 17, 31, 32])
def test_find_generate_apply_bitexact_vs_oracle(gk, oracle, max_bs):
    n, rp, ci, v, _ = block_structured_matrix(257, max_bs, seed=max_bs)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    eptrs = np.zeros(n + 1, np.int32)
    enb = oracle.ref_jacobi_find_blocks(n, rp, ci, max_bs, eptrs)
    nb, ptrs = gpu_find_blocks(gk, n, rpd, cid, max_bs)
    assert nb == enb and np.array_equal(host(ptrs)[:nb + 1], eptrs[:nb + 1])
    es = np.zeros(3, np.int64)
    oracle.ref_jacobi_storage_scheme(max_bs, 64, es)
    eblocks = np.zeros(int(oracle.ref_jacobi_storage_space(es, nb)))
    econd = np.zeros(nb)
    oracle.ref_jacobi_generate(n, rp, ci, v, nb, es, eptrs, econd, eblocks)
    blocks, cond = gpu_generate(gk, n, rpd, cid, vd, ptrs, nb, max_bs, cond=True)
    assert np.array_equal(host(blocks), eblocks)
    assert np.array_equal(host(cond)[:nb], econd)
    rng = np.random.default_rng(99)
    for nrhs in (1, 3):
        b = rng.standard_normal((n, nrhs))
        x0 = rng.standard_normal((n, nrhs))
        ex = x0.copy()
        oracle.ref_jacobi_simple_apply(nb, es, eptrs, eblocks, nrhs, b, nrhs, ex, nrhs)
        x = dev(x0)
        gk.jacobi_apply_f64_i32(stream_ptr(), nb, max_bs, ptrs, blocks, nrhs, None, dev(b), nrhs, None, x, nrhs)
        assert np.array_equal(host(x), ex)
        for beta in (-1.0, 0.0):
            ex = x0.copy()
            oracle.ref_jacobi_apply(nb, es, eptrs, eblocks, nrhs, 2.0, b, nrhs, beta, ex, nrhs)
            x = dev(x0)
            gk.jacobi_apply_f64_i32(stream_ptr(), nb, max_bs, ptrs, blocks, nrhs, dev(np.array([2.0])), dev(b), nrhs,
                                    dev(np.array([beta])), x, nrhs)
            assert np.array_equal(host(x), ex)


def test_singular_block_stops_like_the_reference(gk, oracle):
    # a zero pivot: the reference's invert_block returns early and stores the
    # partially transformed block; the kernel must do the same (bit-exact)
    rp = np.array([0, 2, 4, 5], np.int32)
    ci = np.array([0, 1, 0, 1, 2], np.int32)
    v = np.array([1.0, 2.0, 2.0, 4.0, 3.0])
    ptrs = np.array([0, 2, 3], np.int32)
    es = np.zeros(3, np.int64)
    oracle.ref_jacobi_storage_scheme(2, 64, es)
    eb = np.zeros(int(oracle.ref_jacobi_storage_space(es, 2)))
    oracle.ref_jacobi_generate(3, rp, ci, v, 2, es, ptrs, None, eb)
    blocks, _ = gpu_generate(gk, 3, dev(rp), dev(ci), dev(v), dev(ptrs), 2, 2)
    assert np.array_equal(host(blocks), eb, equal_nan=True)


def test_find_blocks_large_poisson_and_identity_patterns(gk, oracle):
    # 1M rows: no two consecutive rows share a pattern -> blocks of max_bs rows
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    eptrs = np.zeros(n + 1, np.int32)
    enb = oracle.ref_jacobi_find_blocks(n, rp, ci, 32, eptrs)
    nb, ptrs = gpu_find_blocks(gk, n, dev(rp), dev(ci), 32)
    assert nb == enb == 31250 and np.array_equal(host(ptrs)[:nb + 1], eptrs[:nb + 1])
    # generate + apply at that size agree with the oracle bit for bit
    es = np.zeros(3, np.int64)
    oracle.ref_jacobi_storage_scheme(32, 64, es)
    eblocks = np.zeros(int(oracle.ref_jacobi_storage_space(es, nb)))
    oracle.ref_jacobi_generate(n, rp, ci, v, nb, es, eptrs, None, eblocks)
    blocks, _ = gpu_generate(gk, n, dev(rp), dev(ci), dev(v), ptrs, nb, 32)
    assert np.array_equal(host(blocks), eblocks)
    b = np.sin(0.01 * np.arange(n)).reshape(n, 1)
    ex = np.zeros((n, 1))
    oracle.ref_jacobi_simple_apply(nb, es, eptrs, eblocks, 1, b, 1, ex, 1)
    x = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    gk.jacobi_apply_f64_i32(stream_ptr(), nb, 32, ptrs, blocks, 1, None, dev(b), 1, None, x, 1)
    assert np.array_equal(host(x), ex)


# ---- adaptive precision block storage --------------------------------------------------
A = G["adaptive"]


def gpu_generate_adaptive(gk, n, rpd, cid, vd, ptrs_d, nb, max_bs, precisions, accuracy=1e-1):
    nel = gk.jacobi_storage_elements(max_bs, nb)
    blocks = torch.full((max(nel, 1),), float("nan"), dtype=torch.float64, device="cuda:0")
    cond = torch.zeros(max(nb, 1), dtype=torch.float64, device="cuda:0")
    prec = dev(np.resize(np.asarray(precisions, np.uint8), max(nb, 1)))
    gk.jacobi_generate_adaptive_f64_i32(stream_ptr(), n, rpd, cid, vd, nb, max_bs, ptrs_d, accuracy, cond, prec, blocks)
    return blocks, cond, prec


def oracle_generate_adaptive(oracle, n, rp, ci, v, ptrs, nb, max_bs, precisions, accuracy=1e-1):
    es = np.zeros(3, np.int64)
    oracle.ref_jacobi_storage_scheme(max_bs, 64, es)
    blocks = np.zeros(int(oracle.ref_jacobi_storage_space(es, nb)))
    cond = np.zeros(max(nb, 1))
    prec = np.resize(np.asarray(precisions, np.uint8), max(nb, 1)).copy()
    oracle.ref_jacobi_generate_adaptive(n, rp, ci, v, nb, es, np.asarray(ptrs, np.int32), accuracy, cond, prec, blocks)
    return es, blocks, cond, prec


def test_adaptive_known_answers(gk, oracle):
    n, rp, ci, v = mtx()
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    ptrs = np.array(G["block_pointers"], np.int32)
    # requested {(0,1), (0,0)}: with the HIP stride of 64 both blocks share a
    # group at every max_block_size <= 32 -> common precision (0,0), :450-470
    for max_bs in (17, 3):
        blocks, cond, prec = gpu_generate_adaptive(gk, n, rpd, cid, vd, dev(ptrs), 2, max_bs, A["block_precisions"])
        es, eb, ec, ep = oracle_generate_adaptive(oracle, n, rp, ci, v, ptrs, 2, max_bs, A["block_precisions"])
        assert list(host(prec)) == list(ep) == [0, 0]
        assert np.array_equal(host(blocks), eb) and np.array_equal(host(cond), ec)
        assert np.allclose(host(cond), G["conditioning"]["expect"], rtol=0, atol=G["conditioning"]["tol"])
    # SelectsCorrectBlockPrecisions (:582-602): alone in their groups the
    # blocks pick half and float; sharing one group they settle on float
    c = A["selects"]
    for b, expect in ((0, 2), (1, 1)):
        one = ptrs[b:b + 2]
        lo, hi = int(one[0]), int(one[1])
        sub_rows = range(lo, hi)
        srp, sci, sv = [0], [], []
        for r in sub_rows:
            for k in range(rp[r], rp[r + 1]):
                if lo <= ci[k] < hi:
                    sci.append(ci[k] - lo); sv.append(v[k])
            srp.append(len(sci))
        srp, sci, sv = np.array(srp, np.int32), np.array(sci, np.int32), np.array(sv)
        p1 = np.array([0, hi - lo], np.int32)
        _, _, prec = gpu_generate_adaptive(gk, hi - lo, dev(srp), dev(sci), dev(sv), dev(p1), 1, c["max_block_size"],
                                           [255], c["accuracy"])
        assert list(host(prec)) == [expect]
    _, _, prec = gpu_generate_adaptive(gk, n, rpd, cid, vd, dev(ptrs), 2, c["max_block_size"], [255], c["accuracy"])
    _, _, _, ep = oracle_generate_adaptive(oracle, n, rp, ci, v, ptrs, 2, c["max_block_size"], [255], c["accuracy"])
    assert list(host(prec)) == list(ep) == [1, 1]
    # AvoidsPrecisionsThatOverflow (:605-645)
    c = A["overflow"]
    rows, cols, vals = [], [], []
    for k, blk in enumerate(c["diag_blocks"]):
        for i in range(2):
            for j in range(2):
                rows.append(2 * k + i); cols.append(2 * k + j); vals.append(blk[i][j])
    orp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=4))]).astype(np.int32)
    _, _, prec = gpu_generate_adaptive(gk, 4, dev(orp), dev(np.array(cols, np.int32)), dev(np.array(vals)),
                                       dev(np.array(c["block_pointers"], np.int32)), 2, c["max_block_size"], [255],
                                       c["accuracy"])
    assert list(host(prec)) == c["expect"]


@pytest.mark.parametrize("case", A["applies"], ids=lambda c: c["name"])
def test_adaptive_applies_known_answers(gk, case):
    # at stride 64 both blocks share a group; here BOTH are stored as float (the
    # reference's fixture keeps the second in fp64), hence 2 x its half_tol
    n, rp, ci, v = mtx()
    ptrs = dev(np.array(G["block_pointers"], np.int32))
    blocks, _, prec = gpu_generate_adaptive(gk, n, dev(rp), dev(ci), dev(v), ptrs, 2, case["max_block_size"], [1])
    assert list(host(prec)) == [1, 1]   # float storage for the whole group
    st = case.get("stride")
    x, b = dev(_strided(case["x"], st)), dev(_strided(case["b"], st))
    nrhs = np.array(case["x"]).shape[1]
    al = dev(np.array([case["alpha"]])) if "alpha" in case else None
    be = dev(np.array([case["beta"]])) if "alpha" in case else None
    gk.jacobi_apply_adaptive_f64_i32(stream_ptr(), 2, case["max_block_size"], ptrs, prec, blocks, nrhs, al, b,
                                     b.shape[1], be, x, x.shape[1])
    assert matgen.rel_err(host(x)[:, :nrhs], case["expect"]) <= 2 * A["half_tol"]
    assert np.all(host(x)[:, nrhs:] == -9.0)


@pytest.mark.parametrize("max_bs", [2, 4, 7, 13, 16, 32])
@pytest.mark.parametrize("storage", [0x00, 0x01, 0x02, 0x10, 0x11, 0x20, "mixed", "auto"])
def test_adaptive_generate_apply_bitexact_vs_oracle(gk, oracle, max_bs, storage):
    n, rp, ci, v, _ = block_structured_matrix(203, max_bs, seed=100 + max_bs)
    if storage == "auto":   # spread the condition numbers so that several precisions get chosen
        scale = np.repeat(10.0 ** np.random.default_rng(3).integers(-3, 4, n), np.diff(rp))
        v = v * scale
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    eptrs = np.zeros(n + 1, np.int32)
    nb = oracle.ref_jacobi_find_blocks(n, rp, ci, max_bs, eptrs)
    ptrs = dev(eptrs)
    req = {"mixed": [0x01, 0x00, 0x02, 0x10, 0x11, 0x20, 0xff], "auto": [0xff]}.get(storage, [storage])
    for accuracy in ((1e-1, 1e-3, 1e-6) if storage in ("auto", "mixed") else (1e-1,)):
        blocks, cond, prec = gpu_generate_adaptive(gk, n, rpd, cid, vd, ptrs, nb, max_bs, req, accuracy)
        es, eb, ec, ep = oracle_generate_adaptive(oracle, n, rp, ci, v, eptrs[:nb + 1], nb, max_bs, req, accuracy)
        assert np.array_equal(host(prec)[:nb], ep[:nb])
        assert np.array_equal(host(cond)[:nb], ec[:nb])
        assert host(blocks).tobytes() == eb.tobytes()
        if storage == "auto" and accuracy == 1e-1:
            assert len(set(ep[:nb].tolist())) >= 2   # the test data exercises more than one storage type
        rng = np.random.default_rng(7)
        for nrhs in (1, 3):
            b = rng.standard_normal((n, nrhs))
            x0 = rng.standard_normal((n, nrhs))
            for alpha, beta in ((None, None), (2.0, -1.0), (0.5, 0.0)):
                ex = x0.copy()
                oracle.ref_jacobi_apply_adaptive(nb, es, eptrs, ep, eb, nrhs, 1.0 if alpha is None else alpha, b, nrhs,
                                                 0.0 if beta is None else beta, ex, nrhs)
                x = dev(x0)
                gk.jacobi_apply_adaptive_f64_i32(stream_ptr(), nb, max_bs, ptrs, prec, blocks, nrhs,
                                                 None if alpha is None else dev(np.array([alpha])), dev(b), nrhs,
                                                 None if beta is None else dev(np.array([beta])), x, nrhs)
                assert np.array_equal(host(x), ex)


def test_adaptive_storage_speeds_up_block_jacobi_cg(gk, oracle):
    """Adaptive storage is a preconditioner-quality trade: CG with the reduced
    blocks still converges to the same tolerance (benchmark/solver use)."""
    from gkomi import solvers
    n, rp, ci, v = matgen.poisson_2d_5pt(64)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    b = torch.ones(n, dtype=torch.float64, device="cuda:0")
    its = {}
    for name, storage in (("fp64", None), ("adaptive", solvers.AUTODETECT)):
        pc = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=8, storage_optimization=storage)
        res = solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=2000, reduction=1e-10, precond=pc)
        assert res["converged"] and res["rel_residual"] < 1e-10
        its[name] = res["iterations"]
        if storage is not None:
            assert set(host(pc.block_precisions).tolist()) - {0}   # something was actually reduced
    assert abs(its["adaptive"] - its["fp64"]) <= max(3, its["fp64"] // 10)


def _rows_with_runs(run_lengths, seed):
    """CSR pattern whose consecutive rows are identical inside each run and differ across runs"""
    rng = np.random.default_rng(seed)
    n = int(np.sum(run_lengths))
    rp, ci = [0], []
    prev = None
    for L in run_lengths:
        while True:
            cols = sorted(set(int(c) for c in rng.integers(0, n, size=int(rng.integers(1, 4)))))
            if cols != prev:
                break
        prev = cols
        for _ in range(int(L)):
            ci.extend(cols)
            rp.append(len(ci))
    return n, np.array(rp, np.int32), np.array(ci, np.int32)


@pytest.mark.parametrize("max_bs", [1, 2, 3, 5, 8, 13, 32])
@pytest.mark.parametrize("shape", ["short", "long", "mixed", "chunk_edges"])
def test_find_blocks_parallel_chain_vs_oracle(gk, oracle, max_bs, shape):
    """the data-parallel restatement of the two greedy passes: runs of identical
    rows of every length, crossing the 2048-row chunks of the chain walk"""
    rng = np.random.default_rng(17 + max_bs)
    if shape == "short":
        runs = rng.integers(1, 4, size=4000)
    elif shape == "long":
        runs = rng.integers(20, 400, size=60)
    elif shape == "mixed":
        runs = np.where(rng.random(1500) < 0.9, rng.integers(1, 6, size=1500), rng.integers(30, 120, size=1500))
    else:  # boundaries placed around multiples of 2048
        runs = []
        total = 0
        for k in range(1, 6):
            target = 2048 * k + int(rng.integers(-3, 4))
            while total < target - 40:
                r = int(rng.integers(1, 40))
                runs.append(r); total += r
            runs.append(target - total); total = target
        runs = np.array([r for r in runs if r > 0])
    n, rp, ci = _rows_with_runs(runs, 5)
    eptrs = np.zeros(n + 1, np.int32)
    enb = oracle.ref_jacobi_find_blocks(n, rp, ci, max_bs, eptrs)
    nb, ptrs = gpu_find_blocks(gk, n, dev(rp), dev(ci), max_bs)
    assert nb == enb and np.array_equal(host(ptrs)[:nb + 1], eptrs[:nb + 1])


def test_find_blocks_degenerate_sizes(gk, oracle):
    for n in (1, 2, 31, 32, 33, 2047, 2048, 2049):
        rp = np.arange(n + 1, dtype=np.int32)          # diagonal: no two rows alike
        ci = np.arange(n, dtype=np.int32)
        for max_bs in (1, 4, 32):
            eptrs = np.zeros(n + 1, np.int32)
            enb = oracle.ref_jacobi_find_blocks(n, rp, ci, max_bs, eptrs)
            nb, ptrs = gpu_find_blocks(gk, n, dev(rp), dev(ci), max_bs)
            assert nb == enb and np.array_equal(host(ptrs)[:nb + 1], eptrs[:nb + 1])
        ci2 = np.zeros(n, dtype=np.int32)               # every row = {0}: one run of n rows
        eptrs = np.zeros(n + 1, np.int32)
        enb = oracle.ref_jacobi_find_blocks(n, rp, ci2, 5, eptrs)
        nb, ptrs = gpu_find_blocks(gk, n, dev(rp), dev(ci2), 5)
        assert nb == enb and np.array_equal(host(ptrs)[:nb + 1], eptrs[:nb + 1])


@pytest.mark.parametrize("max_bs", [1, 3, 8, 13, 32])
@pytest.mark.parametrize("storage", [0x00, "mixed"])
def test_transpose_bitexact_vs_oracle_and_known_answers(gk, oracle, max_bs, storage):
    """jacobi::transpose_jacobi: every block transposed in its storage
    precision -- byte for byte the oracle's result (NaN padding included: the
    slots outside the blocks are never written)."""
    n, rp, ci, v, _ = block_structured_matrix(157, max_bs, seed=40 + max_bs)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    eptrs = np.zeros(n + 1, np.int32)
    nb = oracle.ref_jacobi_find_blocks(n, rp, ci, max_bs, eptrs)
    ptrs = dev(eptrs)
    req = [0x01, 0x00, 0x02, 0x10, 0x11, 0x20, 0xff] if storage == "mixed" else [0x00]
    blocks, cond, prec = gpu_generate_adaptive(gk, n, rpd, cid, vd, ptrs, nb, max_bs, req, 1e-2)
    es, eb, ec, ep = oracle_generate_adaptive(oracle, n, rp, ci, v, eptrs[:nb + 1], nb, max_bs, req, 1e-2)
    eout = np.full_like(eb, np.nan)
    oracle.ref_jacobi_transpose(nb, es, eptrs, ep, eb, eout)
    out = torch.full_like(blocks, float("nan"))
    gk.jacobi_transpose_f64_i32(stream_ptr(), nb, max_bs, ptrs, prec, blocks, out)
    assert host(out).tobytes() == eout.tobytes()
    back = torch.full_like(blocks, float("nan"))
    gk.jacobi_transpose_f64_i32(stream_ptr(), nb, max_bs, ptrs, prec, out, back)
    gb, bb = host(blocks), host(back)
    written = ~np.isnan(bb)
    assert np.array_equal(bb[written], gb[written])   # transposing twice gives the blocks back
    # applying the transposed preconditioner = multiplying with the transposed inverse
    rng = np.random.default_rng(3)
    b = rng.standard_normal((n, 1))
    x = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    gk.jacobi_apply_adaptive_f64_i32(stream_ptr(), nb, max_bs, ptrs, prec, out, 1, None, dev(b), 1, None, x, 1)
    ex = np.zeros((n, 1))
    oracle.ref_jacobi_apply_adaptive(nb, es, eptrs, ep, eout, 1, 1.0, b, 1, 0.0, ex, 1)
    assert np.array_equal(host(x), ex)


def test_transposed_known_blocks(gk):
    # reference/test/preconditioner/jacobi_kernels.cpp:331-392
    n, rp, ci, v = mtx()
    T = G["transposed_blocks"]
    ptrs = dev(np.array(G["block_pointers"], np.int32))
    blocks, _ = gpu_generate(gk, n, dev(rp), dev(ci), dev(v), ptrs, 2, G["max_block_size"])
    out = torch.full_like(blocks, float("nan"))
    gk.jacobi_transpose_f64_i32(stream_ptr(), 2, G["max_block_size"], ptrs, None, blocks, out)
    s = gpu_scheme(gk, G["max_block_size"])
    ob = host(out)
    p = int(s[3])

    def blk(b, bs):
        off = int(s[1]) * (b >> int(s[2])) + int(s[0]) * (b & ((1 << int(s[2])) - 1))
        return np.array([[ob[off + r + c * p] for c in range(bs)] for r in range(bs)])
    assert np.allclose(blk(0, 2), T["b1"], rtol=0, atol=G["tol"])
    assert np.allclose(blk(1, 3), T["b2"], rtol=0, atol=G["tol"])
