#!/usr/bin/env python3
"""GMRES(30) + ParILU (reference default: 10 sweeps) on the AT-like 108^3 system, twice; the program
tools/profile_ilu_gmres.sh traces.  Diagnostic only."""
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
pre = None if "--plain" in sys.argv else solvers.par_ilu_generate(gk, n, a[0].clone(), a[1], a[2], iterations=0)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = solvers.solve_op(gk, "gmres", A, b, krylov_dim=30, max_iters=3000, reduction=1e-10, precond=pre)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) * 1e3:.2f} ms, {r['iterations']} iterations")
