#!/usr/bin/env python3
"""A/B of the padded LDS tile of the CSR stream kernel (variants 15/16 vs 5/14)
on rows of fixed length L (even lengths are the bank-conflict case) and the
stencils.  Warm, interleaved, median of 5 batches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gkomi
import matgen

gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s = torch.cuda.current_stream().cuda_stream


def fixed_rows(n, k):
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    cols = np.clip(rows + np.tile((np.arange(k) - k // 2) * 3, n), 0, n - 1)
    keep = np.ones(len(cols), dtype=bool)
    keep[1:] = (cols[1:] != cols[:-1]) | (rows[1:] != rows[:-1])
    rows, cols = rows[keep], cols[keep]
    rp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rp[1:])
    return n, rp, cols.astype(np.int32), np.random.default_rng(1).standard_normal(len(cols))


def bench(name, n, rp, ci, v):
    nnz = len(v)
    rpd, cid, vd = d(rp), d(ci), d(v)
    x = d(np.sin(0.01 * np.arange(n)).reshape(n, 1))
    y = torch.empty((n, 1), dtype=torch.float64, device="cuda")
    hint = int(np.diff(rp).max())
    codes = {"v5": 1 | (5 << 8), "v15 pad": 1 | (15 << 8), "v14 nt": 1 | (14 << 8) | (1 << 16), "v16 nt pad": 1 | (16 << 8) | (1 << 16)}
    ref = None
    times = {k: [] for k in codes}
    for rep in range(5):
        for k, code in codes.items():
            for _ in range(3):
                gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, code, hint)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, code, hint)
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3 / 50)
            if ref is None:
                ref = y.clone()
            assert torch.equal(ref, y), (name, k)   # bit-identical across variants
    nbytes = 12 * nnz + 20 * n
    print(f"{name:22s} nnz {nnz:9d} " + "  ".join(f"{k} {np.median(t):7.2f} us ({nbytes/np.median(t)/1e3:5.0f} GB/s)" for k, t in times.items()))


bench("5pt 1000^2", *matgen.poisson_2d_5pt(1000))
bench("7pt 108^3", *matgen.poisson_3d_7pt(108))
for L in (6, 7, 8, 12, 15, 16, 24, 32, 64):
    bench(f"fixed rows L={L}", *fixed_rows(1000000 if L <= 32 else 500000, L))
bench("27pt 100^3", *matgen.stencil_3d_27pt(100))
bench("5pt 2000^2", *matgen.poisson_2d_5pt(2000))
