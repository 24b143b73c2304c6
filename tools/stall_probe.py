#!/usr/bin/env python3
"""Where do the occasional ~70 ms stalls of one solver call come from?  Times 60
CG solves and 60 batches of SpMV launches one by one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, gkomi, gkomi.solvers as solvers, matgen
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n, rp, ci, v = matgen.poisson_2d_5pt(1000)
rpd, cid, vd = d(rp), d(ci), d(v)
b = torch.ones((n, 1), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
y = torch.empty_like(b)
for name, fn in (("cg solve (162+ its)", lambda: solvers.cg_solve(gk, n, rpd, cid, vd, b, max_iters=200, reduction=1e-30, check_every=32)),
                 ("400 spmv launches", lambda: [gk.csr_spmv_f64_i32(s, n, n, 1, len(v), rpd, cid, vd, b, 1, y, 1, None, None, 0, 5) for _ in range(400)])):
    ts = []
    for i in range(60):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(f"{name}: median {np.median(ts):.2f} ms, max {ts.max():.2f} ms, calls > 3x median: {np.nonzero(ts > 3 * np.median(ts))[0].tolist()}")
    print("   ", " ".join(f"{t:.1f}" for t in ts))
