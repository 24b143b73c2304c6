#!/usr/bin/env python3
"""P3 (256^3 7-pt, 16.7M rows) SpMV: XCD chunking x nontemporal streams for the nonzero-split
kernel and the row-block kernel.  Usage: python tools/p3_probe.py [grid]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s = torch.cuda.current_stream().cuda_stream
n, rp, ci, v = matgen.poisson_3d_7pt(g)
nnz = len(v)
rpd, cid, vd = d(rp), d(ci), d(v)
del rp, ci, v
x = d(np.sin(0.01 * np.arange(n)).reshape(n, 1)); y = torch.empty_like(x)
bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
SPLIT, STREAM = 4, 1
def run(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"{g}^3: n={n} nnz={nnz} bytes={bytes_}")
for tile in (2048, 3072):
    cnt = int(gk.csr_srow_entries(nnz, tile)); srow = torch.empty(cnt, dtype=torch.int32, device="cuda")
    gk.csr_make_srow_i32(s, n, nnz, rpd, tile, srow, cnt)
    for name, bits, code in (("noswz nt", 258, 0), ("swz nt", 2, 0), ("chunk 1 nt", 2, 1), ("chunk 4 nt", 2, 3),
                             ("chunk 16 nt", 2, 5), ("chunk 64 nt", 2, 7), ("chunk 256 nt", 2, 9)):
        st = SPLIT | ((bits & 0xff) << 8) | ((bits >> 8) << 16) | (code << 17)
        t = run(lambda: gk.csr_spmv_srow_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, st, 7, srow, tile))
        print(f"split tile {tile} {name:9s}: {t:8.1f} us {bytes_/t/1e3:7.0f} GB/s {bytes_/t/1e3/8000:.3f}")
t = run(lambda: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, 0, 7))
print(f"automatic (no srow)          : {t:8.1f} us {bytes_/t/1e3:7.0f} GB/s {bytes_/t/1e3/8000:.3f}")
for name, st in (("stream v5 swz", STREAM | (5 << 8)), ("stream v5 noswz", STREAM | (5 << 8) | (1 << 16)),
                 ("stream v14 nt swz", STREAM | (14 << 8)), ("stream v14 nt noswz", STREAM | (14 << 8) | (1 << 16))):
    t = run(lambda: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, st, 7))
    print(f"{name:29s}: {t:8.1f} us {bytes_/t/1e3:7.0f} GB/s {bytes_/t/1e3/8000:.3f}")
