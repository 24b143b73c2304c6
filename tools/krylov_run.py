#!/usr/bin/env python3
"""One fused Krylov solver (argv[1]: bicgstab | cgs | fcg | cg) on the AT-like 108^3 system (cg / fcg: the symmetric 7-point
matrix of the same size), twice; the program tools/profile_krylov.sh traces.  Diagnostic only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
import gkomi.solvers as solvers
from gkomi.formats import Csr
gk = gkomi.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
which = sys.argv[1] if len(sys.argv) > 1 else "bicgstab"
n, rp, ci, v = matgen.poisson_3d_7pt(108) if which in ("cg", "fcg") else matgen.at_like(108)
A = Csr(gk, n, n, dev(rp), dev(ci), dev(v))
b = dev(np.cos(0.3 * np.arange(n)).reshape(n, 1))
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = solvers.solve_op(gk, which, A, b, max_iters=400, reduction=1e-30, fused="--unfused" not in sys.argv)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{which}: {el * 1e3:.2f} ms, {r['iterations']} iterations, {el / max(r['iterations'], 1) * 1e6:.1f} us per iteration")
