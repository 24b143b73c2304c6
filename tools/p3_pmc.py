#!/usr/bin/env python3
"""One variant of the P3 SpMV, 10 launches, for rocprofv3 --pmc FETCH_SIZE (fabric -> L2 bytes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 258
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s = torch.cuda.current_stream().cuda_stream
n, rp, ci, v = matgen.poisson_3d_7pt(g)
nnz = len(v)
rpd, cid, vd = d(rp), d(ci), d(v)
x = d(np.sin(0.01 * np.arange(n)).reshape(n, 1)); y = torch.empty_like(x)
cnt = int(gk.csr_srow_entries(nnz, tile)); srow = torch.empty(cnt, dtype=torch.int32, device="cuda")
gk.csr_make_srow_i32(s, n, nnz, rpd, tile, srow, cnt)
st = 4 | ((bits & 0xff) << 8) | ((bits >> 8) << 16)
for _ in range(10):
    gk.csr_spmv_srow_f64_i32(s, n, n, 1, nnz, rpd, cid, vd, x, 1, y, 1, None, None, st, 7, srow, tile)
torch.cuda.synchronize()
print("algorithmic bytes", 12 * nnz + 4 * (n + 1) + 16 * n)
