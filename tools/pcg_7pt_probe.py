import os, sys, time
sys.path.insert(0, "repo-8852-ginkgo_amd"); sys.path.insert(0, "tests")
import numpy as np, torch
import gkomi, gkomi.solvers as solvers, matgen
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for g in (80, 100):
    n, rp, ci, v = matgen.poisson_3d_7pt(g)
    rpd, cid, vd = d(rp), d(ci), d(v)
    b = d(np.cos(0.01 * np.arange(n)))
    for hint in (-1, 7):
        solvers.cg_solve(gk, n, rpd, cid, vd, b, mode=1, max_iters=5000, reduction=1e-10, max_row_nnz=hint)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            before = gk.cg_persistent_solves()
            r = solvers.cg_solve(gk, n, rpd, cid, vd, b, mode=1, max_iters=5000, reduction=1e-10, max_row_nnz=hint)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"7-pt {g}^3 n={n} hint={hint}: {r['iterations']} iters {best*1e3:.2f} ms {best/r['iterations']*1e6:.1f} us/it persistent={gk.cg_persistent_solves()-before} conv={r['converged']}")
