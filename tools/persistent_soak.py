#!/usr/bin/env python3
"""Soak test of the single-launch solvers: many solves in a row, every one must take the
single-launch path, give the same iteration count and the same bits.  Usage: persistent_soak.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, gkomi.solvers as solvers, matgen
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = 0
for grid, count in (((256, 256), reps), ((1000, 1000), max(reps // 10, 10))):
    n, rp, ci, v = matgen.poisson_2d_5pt(*grid)
    rpd, cid, vd = d(rp), d(ci), d(v)
    b = d(np.sin(0.1 * np.arange(n)))
    ref = None
    t0 = time.perf_counter(); before = gk.cg_persistent_solves()
    for i in range(count):
        r = solvers.cg_solve(gk, n, rpd, cid, vd, b, mode=1, max_iters=5000, reduction=1e-10, max_row_nnz=5)
        key = (r["iterations"], r["x"].cpu().numpy().tobytes())
        if ref is None:
            ref = key
        elif key != ref:
            bad += 1
            print("MISMATCH at solve", i, r["iterations"], ref[0])
    took = gk.cg_persistent_solves() - before
    print(f"cg {grid}: {count} solves, {took} single-launch, {ref[0]} iterations each, {time.perf_counter()-t0:.1f} s, mismatches {bad}")
    if took != count:
        bad += 1
# GMRES with the single-launch Arnoldi sweep
n, rp, ci, v = matgen.poisson_3d_7pt(40)
v = v.copy(); rows = np.repeat(np.arange(n), np.diff(rp)); v[ci == rows - 1] -= 0.5; v[ci == rows] += 0.5
rpd, cid, vd = d(rp), d(ci), d(v); b = d(np.cos(0.3 * np.arange(n)))
ref = None; t0 = time.perf_counter()
for i in range(max(reps // 10, 10)):
    r = solvers.gmres_solve(gk, n, rpd, cid, vd, b, krylov_dim=30, max_iters=2000, reduction=1e-10)
    key = (r["iterations"], r["x"].cpu().numpy().tobytes())
    if ref is None:
        ref = key
    elif key != ref:
        bad += 1
        print("GMRES MISMATCH at solve", i, r["iterations"], ref[0])
print(f"gmres 40^3: {max(reps // 10, 10)} solves, {ref[0]} iterations each, {time.perf_counter()-t0:.1f} s, mismatches {bad}")
sys.exit(1 if bad else 0)
