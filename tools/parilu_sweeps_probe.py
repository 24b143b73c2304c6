#!/usr/bin/env python3
"""GMRES(30) on the AT-like 108^3 system: iterations and time against the number of ParILU sweeps
(hip/factorization/par_ilu_kernels.hip.cpp:72 of the reference: 0 = 10 sweeps).  Diagnostic only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
import gkomi.solvers as solvers
from gkomi.formats import Csr
gk = gkomi.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n, rp, ci, v = matgen.at_like(108)
a = [dev(rp), dev(ci), dev(v)]
A = Csr(gk, n, n, *a)
b = dev(np.cos(0.3 * np.arange(n)).reshape(n, 1))


def solve(pc):
    best, its = None, []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = solvers.solve_op(gk, "gmres", A, b, krylov_dim=30, max_iters=3000, reduction=1e-10, precond=pc)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        best = el if best is None else min(best, el); its.append(r["iterations"])
    return best * 1e3, its


ms, its = solve(None)
print(f"no preconditioner: {ms:.2f} ms, iterations {its}")
for sweeps in (3, 5, 10, 20):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pre = solvers.par_ilu_generate(gk, n, a[0].clone(), a[1], a[2], iterations=sweeps)
        torch.cuda.synchronize(); gen = (time.perf_counter() - t0) * 1e3
        ms, its = solve(pre)
        print(f"ParILU sweeps {sweeps} (generate {gen:.1f} ms): {ms:.2f} ms, iterations {its}, {ms / its[-1] * 1e3:.0f} us/iteration")
        del pre
