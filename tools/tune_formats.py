#!/usr/bin/env python3
"""Warm/cold timing of every SpMV format on the 5-pt Poisson matrix plus the
XCD chunk-size sweep of the CSR stream kernel and the CG kernel breakdown."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gkomi
import gkomi.solvers as solvers
import matgen

gk = gkomi.lib()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, rp, ci, v = matgen.poisson_2d_5pt(grid)
nnz = int(rp[-1])
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s = torch.cuda.current_stream().cuda_stream
x = d(np.sin(0.01 * np.arange(n)).reshape(n, 1))
y = torch.empty((n, 1), dtype=torch.float64, device="cuda")
rpd, cid, vd = d(rp), d(ci), d(v)
# big scratch to flush the Infinity Cache between cold launches
flush = torch.empty(80_000_000, dtype=torch.float64, device="cuda")


def timeit(fn, reps=100, cold=False):
    ts = []
    for _ in range(5):
        fn()
    if cold:
        for _ in range(12):
            flush.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def report(name, fn, nbytes):
    w, c = timeit(fn), timeit(fn, cold=True)
    print(f"{name:34s} warm {w:8.2f} us {nbytes/w/1e3:7.0f} GB/s | cold(single, event) {c:8.2f} us {nbytes/c/1e3:7.0f} GB/s")


csr_bytes = 12 * nnz + 4 * (n + 1) + 16 * n
print(f"grid {grid}: n={n} nnz={nnz}")
report("csr auto", lambda: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, 0, 5), csr_bytes)
for code in (3, 4, 5, 6, 7, 8, 9):
    st = 1 | (5 << 8) | (code << 17)
    report(f"csr v5 xcd-chunk {1 << (code - 1)}", lambda st=st: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, st, 5), csr_bytes)
report("csr v5 no swizzle", lambda: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, 1 | (5 << 8) | (1 << 16), 5), csr_bytes)
# ELL
k = 5
ecols = torch.full((n * k,), -1, dtype=torch.int32, device="cuda")
evals = torch.zeros(n * k, dtype=torch.float64, device="cuda")
gk.csr_convert_to_ell_f64_i32(s, n, rpd, cid, vd, k, n, ecols, evals)
report("ell", lambda: gk.ell_spmv_f64_i32(s, n, n, 1, k, n, ecols, evals, x, 1, y, 1, None, None), 12 * n * k + 16 * n)
# SELL-P
nsl = (n + 63) // 64
sets = torch.zeros(nsl + 1, dtype=torch.int64, device="cuda")
lens = torch.zeros(nsl, dtype=torch.int64, device="cuda")
nb = gk.prefix_sum_workspace_bytes(nsl + 1)
ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda")
gk.sellp_compute_slice_sets_i32(s, rpd, n, 64, 1, sets, lens, ws, nb)
total = int(sets[nsl].item()) * 64
scols = torch.full((total,), -1, dtype=torch.int32, device="cuda")
svals = torch.zeros(total, dtype=torch.float64, device="cuda")
gk.csr_convert_to_sellp_f64_i32(s, n, rpd, cid, vd, 64, sets, lens, scols, svals)
report("sellp(64)", lambda: gk.sellp_spmv_f64_i32(s, n, n, 1, 64, sets, lens, scols, svals, x, 1, y, 1, None, None), 12 * total + 16 * nsl + 16 * n)
# COO
rows = torch.zeros(nnz, dtype=torch.int32, device="cuda")
gk.convert_ptrs_to_idxs_i32(s, rpd, n, rows)
report("coo (fill + spmv2)", lambda: gk.coo_spmv_f64_i32(s, n, n, 1, nnz, rows, cid, vd, x, 1, y, 1, None, None), 16 * nnz + 24 * n)
cws_bytes = gk.coo_sorted_workspace_bytes(nnz, 8)
cws = torch.empty(cws_bytes, dtype=torch.uint8, device="cuda")
report("coo sorted, carries (any row length)", lambda: gk.coo_spmv_sorted_f64_i32(s, n, n, 1, nnz, rows, cid, vd, x, 1, y, 1, None, None, -1, cws, cws_bytes), 16 * nnz + 16 * n)
report("coo sorted, halo (rows <= 64)", lambda: gk.coo_spmv_sorted_f64_i32(s, n, n, 1, nnz, rows, cid, vd, x, 1, y, 1, None, None, 5, cws, cws_bytes), 16 * nnz + 16 * n)
for k_rhs in (4, 8):
    xb = d(np.sin(0.01 * np.arange(n * k_rhs)).reshape(n, k_rhs))
    yb = torch.empty((n, k_rhs), dtype=torch.float64, device="cuda")
    report(f"coo {k_rhs} rhs (fill + tile atomics)", lambda: gk.coo_spmv_f64_i32(s, n, n, k_rhs, nnz, rows, cid, vd, xb, k_rhs, yb, k_rhs, None, None), 16 * nnz + 24 * n * k_rhs)
    report(f"coo {k_rhs} rhs sorted, carries", lambda: gk.coo_spmv_sorted_f64_i32(s, n, n, k_rhs, nnz, rows, cid, vd, xb, k_rhs, yb, k_rhs, None, None, -1, cws, cws_bytes), 16 * nnz + 16 * n * k_rhs)
    report(f"coo {k_rhs} rhs sorted, halo", lambda: gk.coo_spmv_sorted_f64_i32(s, n, n, k_rhs, nnz, rows, cid, vd, xb, k_rhs, yb, k_rhs, None, None, 5, cws, cws_bytes), 16 * nnz + 16 * n * k_rhs)
# several right-hand sides: one pass over the matrix per 4 columns vs one per column
for k_rhs in (2, 4, 8):
    xb = d(np.sin(0.01 * np.arange(n * k_rhs)).reshape(n, k_rhs))
    yb = torch.empty((n, k_rhs), dtype=torch.float64, device="cuda")
    mb = 12 * nnz + 4 * (n + 1) + 16 * n * k_rhs
    report(f"csr {k_rhs} rhs, multi-rhs kernel", lambda: gk.csr_spmv_f64_i32(s, n, n, k_rhs, nnz, rpd, cid, vd, xb, k_rhs, yb, k_rhs, None, None, 0, 5), mb)
    report(f"csr {k_rhs} rhs, one grid row per rhs", lambda: gk.csr_spmv_f64_i32(s, n, n, k_rhs, nnz, rpd, cid, vd, xb, k_rhs, yb, k_rhs, None, None, 1 | (5 << 8), 5), mb)
for k_rhs in (4, 8):
    xb = d(np.sin(0.01 * np.arange(n * k_rhs)).reshape(n, k_rhs))
    yb = torch.empty((n, k_rhs), dtype=torch.float64, device="cuda")
    report(f"ell {k_rhs} rhs", lambda: gk.ell_spmv_f64_i32(s, n, n, k_rhs, k, n, ecols, evals, xb, k_rhs, yb, k_rhs, None, None), 12 * n * k + 16 * n * k_rhs)
    report(f"sellp {k_rhs} rhs", lambda: gk.sellp_spmv_f64_i32(s, n, n, k_rhs, 64, sets, lens, scols, svals, xb, k_rhs, yb, k_rhs, None, None), 12 * total + 16 * n * k_rhs)
if os.environ.get("FORMATS_ONLY"):
    sys.exit(0)
# Jacobi apply (max block size 32)
pre = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=32)
jb = 8 * pre.blocks.numel() + 16 * n
report("block-jacobi apply (bs 32)", lambda: gk.jacobi_apply_f64_i32(s, pre.num_blocks, 32, pre.block_ptrs, pre.blocks, 1, None, x, 1, None, y, 1), jb)
# BLAS-1
ws2 = torch.empty(gk.dense_reduction_workspace_bytes(n, 1) + 8, dtype=torch.uint8, device="cuda")
res = torch.zeros(1, dtype=torch.float64, device="cuda")
one = d(np.array([0.5]))
report("dot", lambda: gk.dense_compute_dot_f64(s, n, 1, x, 1, y, 1, res, ws2, ws2.numel()), 16 * n)
report("norm2", lambda: gk.dense_compute_norm2_f64(s, n, 1, x, 1, res, ws2, ws2.numel()), 8 * n)
report("axpy", lambda: gk.dense_add_scaled_f64(s, n, 1, one, 1, x, 1, y, 1), 24 * n)
# CG
sv = np.sin(np.arange(n, dtype=np.float64)); sv /= np.linalg.norm(sv)
b = torch.empty((n, 1), dtype=torch.float64, device="cuda")
gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, d(sv.reshape(n, 1)), 1, b, 1, None, None, 0, 5)
for mode in (0, 1):
    for ce in ((1,) if mode == 0 else (4, 16, 64)):
        solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=20000, reduction=1e-10, mode=mode, check_every=ce)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=20000, reduction=1e-10, mode=mode, check_every=ce)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"cg mode {mode} check_every {ce:3d}: {r['iterations']} iters {el*1e3:8.3f} ms  {r['iterations']/el:9.0f} it/s  {el/r['iterations']*1e6:6.2f} us/it")
# same with b = ones (long solve)
b.fill_(1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=20000, reduction=1e-10, mode=1, check_every=32)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"cg mode 1 b=1: {r['iterations']} iters {el*1e3:8.3f} ms  {r['iterations']/el:9.0f} it/s  {el/r['iterations']*1e6:6.2f} us/it conv={r['converged']}")
