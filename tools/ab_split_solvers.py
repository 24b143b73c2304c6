#!/usr/bin/env python3
"""A/B: the fused drivers on a Csr with / without its srow (nonzero-split vs row-cut SpMV + dot), P2 and
the AT-like 108^3 system.  us per iteration, median of 5 solves."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, matgen
from gkomi import formats, solvers
gk = gkomi.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return r, statistics.median(ts)

n, rp, ci, v = matgen.poisson_2d_5pt(1000)
b = torch.ones(n, dtype=torch.float64, device="cuda")
gk.cg_persistent_enable(0)
for split in (False, True):
    A = formats.Csr.from_host(gk, n, n, rp, ci, v, split=split)
    for solver in ("cg", "fcg", "bicgstab", "cgs"):
        kw = dict(max_iters=400, reduction=1e-30, fused=True, check_every=50)
        r, t = timed(lambda: solvers.solve_op(gk, solver, A, b, **kw))
        print(f"P2 {solver:8s} split={split}: {t / r['iterations'] * 1e6:7.2f} us/iteration ({r['iterations']} iterations)", flush=True)
n, rp, ci, v = matgen.at_like(108)
b = dev(np.cos(0.3 * np.arange(n)))
for split in (False, True):
    A = formats.Csr.from_host(gk, n, n, rp, ci, v, split=split)
    r, t = timed(lambda: solvers.solve_op(gk, "gmres", A, b, max_iters=3000, reduction=1e-10, krylov_dim=30), 3)
    print(f"AT108 gmres30 split={split}: {t / r['iterations'] * 1e6:7.2f} us/iteration ({r['iterations']} iterations, {t*1e3:.2f} ms)", flush=True)
    r, t = timed(lambda: solvers.solve_op(gk, "bicgstab", A, b, max_iters=300, reduction=1e-30, fused=True, check_every=50), 3)
    print(f"AT108 bicgstab split={split}: {t / r['iterations'] * 1e6:7.2f} us/iteration", flush=True)
