#!/usr/bin/env python3
"""<double, int64> SpMV: the 256^3 7-point matrix in both index types, and the 700^3 one (2.4 G nonzeros, int64 only).
Algorithmic bytes: (8 + index bytes) nnz + index bytes (n + 1) + 16 n."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import torch
import gkomi
from gkomi import formats
gk = gkomi.lib()


def timed(f, reps):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for g in (256, 700):
    free, _ = torch.cuda.mem_get_info()
    if g == 700 and free < 60 * (1 << 30):
        print("700^3 skipped: needs 60 GB"); continue
    A64 = formats.Csr64.poisson_3d_7pt(gk, g)
    n, nnz = A64.nrows, A64.nnz
    x = torch.sin(0.01 * torch.arange(n, dtype=torch.float64, device="cuda")).reshape(n, 1)
    y = torch.empty_like(x)
    t64 = timed(lambda: A64.apply(x, y), 20 if g == 256 else 5)
    b64 = 16 * nnz + 8 * (n + 1) + 16 * n
    print(f"{g}^3 7-pt, n = {n}, nnz = {nnz}{' (> 2^31)' if nnz > 2**31 else ''}: int64 {t64:9.1f} us  {b64 / t64 / 1e6:5.2f} TB/s = {b64 / t64 / 8e6:.3f} of 8 TB/s "
          f"(tile {A64.srow_tile}, split kernel)", flush=True)
    if g == 256:
        A32 = formats.Csr(gk, n, n, A64.row_ptrs.to(torch.int32), A64.col_idxs.to(torch.int32), A64.vals)
        t32 = timed(lambda: A32.apply(x, y), 20)
        b32 = 12 * nnz + 4 * (n + 1) + 16 * n
        print(f"{' ' * 46}int32 {t32:9.1f} us  {b32 / t32 / 1e6:5.2f} TB/s = {b32 / t32 / 8e6:.3f} of 8 TB/s", flush=True)
    del A64, x, y
    torch.cuda.empty_cache()
