#!/usr/bin/env python3
"""One fused CG solve on the 1M-row Poisson matrix (b = ones, ~2100 iterations)
for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, gkomi.solvers as solvers, matgen
gk = gkomi.lib()
n, rp, ci, v = matgen.poisson_2d_5pt(1000)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rpd, cid, vd = d(rp), d(ci), d(v)
b = torch.ones((n, 1), dtype=torch.float64, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=20000, reduction=1e-10, mode=1, check_every=32)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{r['iterations']} iters {el*1e3:.2f} ms {el/r['iterations']*1e6:.2f} us/it")
