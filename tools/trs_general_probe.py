#!/usr/bin/env python3
"""Triangular solves on the ILU(0) factors of the config-3 stand-ins (not box-grid numberings): levels, plan,
time per solve (level plan vs analysis-free kernel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, matgen
from gkomi import solvers
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def ani(name):
    kind, nr, nc, rows, cols, vals = matgen.read_mtx(os.path.join(ROOT, "tests", "golden", name + ".mtx"))
    rp, ci, v = matgen.coo_to_csr(nr, rows, cols, vals)
    return nr, rp, ci, v

CASES = (("ani4 (3081 rows: one workgroup, x in LDS)", lambda: ani("ani4")), ("ani1", lambda: ani("ani1")),
         ("poisson_2d_60 (3600 rows)", lambda: matgen.poisson_2d_5pt(60)))
if "--small-only" not in sys.argv:
    CASES += (("t2_like_permuted_1108", lambda: matgen.t2_like_permuted(1108)), ("diffusion_patch_ordered_1104", lambda: matgen.diffusion_2d_patch_ordered(1104)),
              ("poisson_2d_1000 (bricks)", lambda: matgen.poisson_2d_5pt(1000)))
for name, gen in CASES:
    n, rp, ci, v = gen()
    a = [d(rp), d(ci), d(v)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pre = solvers.par_ilu_generate(gk, n, a[0].clone(), a[1], a[2], iterations=5)
    torch.cuda.synchronize(); gen_ms = (time.perf_counter() - t0) * 1e3
    b = d(np.sin(0.1 * np.arange(n)) + 2.0).reshape(n, 1)
    x = torch.zeros_like(b)
    print(f"{name}: n {n}, generate {gen_ms:.1f} ms, plans L/U: " + "/".join("bricks" if bk is not None else ("levels" if pl is not None else "analysis-free")
          for bk, pl in ((pre.l_bricks, pre.l_plan), (pre.u_bricks, pre.u_plan))), flush=True)
    for lower, f in ((True, pre.L), (False, pre.U)):
        pl = solvers.TrsPlan(gk, n, f[0], f[1], f[2], lower)
        t_plan = timed(lambda: pl.solve(b, x))
        nnz = int(f[2].numel())
        bytes_ = 12 * nnz + 4 * (n + 1) + 16 * n
        print(f"   {'lower' if lower else 'upper'}: nnz {nnz}, levels {pl.nlevels}, rows per level {n / max(pl.nlevels, 1):.0f}, level plan {t_plan:8.1f} us = "
              f"{t_plan / max(pl.nlevels, 1):.2f} us per level, {bytes_ / t_plan / 1e6:.2f} TB/s on SURVEY 8(d) bytes", flush=True)
    for lower, f, bk in ((True, pre.L, pre.l_bricks), (False, pre.U, pre.u_bricks)):
        if bk is not None:
            t_b = timed(lambda: bk.solve(b, x))
            print(f"   {'lower' if lower else 'upper'}: brick plan {t_b:8.1f} us ({bk.nbricks} bricks, {bk.coarse_levels} brick levels)", flush=True)
    y = torch.zeros_like(b)
    t_apply = timed(lambda: pre.apply(b, y))
    print(f"   Ilu apply (what the solvers call): {t_apply:8.1f} us", flush=True)
