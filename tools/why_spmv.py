#!/usr/bin/env python3
"""Target of tools/profile_why.sh: N cold (rotating copies) then N warm launches
of one CSR SpMV strategy on the 1M-row 5-pt matrix, nothing else, so that a
rocprofv3 --pmc pass sees exactly 2N dispatches of one kernel (first N cold).
usage: why_spmv.py <strategy word | 'split[:bits]'> [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
what = sys.argv[1] if len(sys.argv) > 1 else "0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n, rp, ci, v = matgen.poisson_2d_5pt(1000)
nnz = int(rp[-1])
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
copies = [(d(rp), d(ci), d(v), d(x), torch.empty((n, 1), dtype=torch.float64, device="cuda")) for _ in range(8)]
s = torch.cuda.current_stream().cuda_stream
tile = int(gk.csr_srow_tile())
srows = []
for c in copies:
    t = torch.empty(int(gk.csr_srow_entries(nnz, tile)), dtype=torch.int32, device="cuda")
    gk.csr_make_srow_i32(s, n, nnz, c[0], tile, t, t.numel()); srows.append(t)
torch.cuda.synchronize()
if what.startswith("split"):
    bits = int(what.split(":")[1], 0) if ":" in what else 0
    strategy = 4 | ((bits & 0xff) << 8) | ((bits >> 8) << 16)
    run = lambda j: gk.csr_spmv_srow_f64_i32(s, n, n, 1, nnz, copies[j][0], copies[j][1], copies[j][2], copies[j][3], 1,
                                             copies[j][4], 1, None, None, strategy, 5, srows[j], tile)
else:
    strategy = int(what, 0)
    run = lambda j: gk.csr_spmv_f64_i32(s, n, n, 1, nnz, copies[j][0], copies[j][1], copies[j][2], copies[j][3], 1,
                                        copies[j][4], 1, None, None, strategy, 5)
for i in range(N):
    run(i % 8)
torch.cuda.synchronize()
for i in range(N):
    run(0)
torch.cuda.synchronize()
