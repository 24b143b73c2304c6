#!/usr/bin/env python3
"""Triangular-solve timing on dependency structures of increasing difficulty."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
s = torch.cuda.current_stream().cuda_stream
nb = gk.trs_workspace_bytes(); tws = torch.zeros(nb, dtype=torch.uint8, device="cuda")


def lower_of(n, rp, ci, v):
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = ci <= rows
    rp2 = np.zeros(n + 1, np.int32); np.add.at(rp2, rows[keep] + 1, 1); np.cumsum(rp2, out=rp2)
    return rp2, ci[keep].copy(), v[keep].copy()


def run(name, n, rp, ci, v, reps=5):
    rpd, cid, vd = d(rp), d(ci), d(v)
    b = torch.ones((n, 1), dtype=torch.float64, device="cuda"); x = torch.zeros_like(b)
    f = lambda: gk.lower_trs_solve_f64_i32(s, n, 1, rpd, cid, vd, 0, b, 1, x, 1, tws, nb)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / reps
    flag = ctypes.c_int(0); gk.trs_check_overrun(s, tws, ctypes.addressof(flag))
    xe = x.clone()
    # the analysed solve (LowerTrs::generate = level analysis, then the level-ordered kernel)
    import time, gkomi.solvers as solvers
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = solvers.TrsPlan(gk, n, rpd, cid, vd, True)
    torch.cuda.synchronize(); t_an = (time.perf_counter() - t0) * 1e3
    fp = lambda: plan.solve(b, x)
    fp(); torch.cuda.synchronize()
    same = bool(torch.equal(x, xe))
    e0.record()
    for _ in range(reps): fp()
    e1.record(); torch.cuda.synchronize()
    tp = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:34s} n={n:8d} nnz={len(v):9d}  {t:10.1f} us  {t/n*1e3:8.2f} ns/row  overrun={flag.value} | "
          f"analysed: {tp:10.1f} us  levels {plan.nlevels:6d}  slots {plan.entries:9d}  analysis {t_an:7.1f} ms  "
          f"identical={same} overrun={int(plan.overrun())}")


if os.environ.get("TRS_SWEEP"):
    # scout distances / poll cadence of the analysed solve on the config-4 factor shape
    import gkomi.solvers as solvers
    nn, rp, ci, v = matgen.poisson_3d_7pt(int(os.environ.get("TRS_SWEEP_GRID", "108")))
    lrp, lci, lv = lower_of(nn, rp, ci, v)
    rpd, cid, vd = d(lrp), d(lci), d(lv)
    b = torch.ones((nn, 1), dtype=torch.float64, device="cuda"); x = torch.zeros_like(b)
    for near in (1, 2, 3, 4):
        for far in (8, 16, 32):
            for nap in (0, 1, 4):
                os.environ.update(GKOMI_TRS_NEAR=str(near), GKOMI_TRS_FAR=str(far), GKOMI_TRS_NAP=str(nap))
                plan = solvers.TrsPlan(gk, nn, rpd, cid, vd, True)
                plan.solve(b, x); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): plan.solve(b, x)
                e1.record(); torch.cuda.synchronize()
                print(f"near {near} far {far:2d} nap {nap}: {e0.elapsed_time(e1) * 1e3 / 5:8.1f} us  overrun={int(plan.overrun())}", flush=True)
    sys.exit(0)
quick = os.environ.get("TRS_QUICK")
n = 1 << 20
run("diagonal (no dependencies)", n, np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.full(n, 2.0))
def chain(m, group=None):
    """bidiagonal lower matrix; the sub-diagonal is dropped at multiples of `group`"""
    rows = np.arange(m)
    has_sub = rows > 0 if group is None else (rows % group) != 0
    cnt = 1 + has_sub.astype(np.int32)
    rp = np.zeros(m + 1, np.int32); np.cumsum(cnt, out=rp[1:])
    ci = np.empty(rp[-1], np.int32); v = np.empty(rp[-1])
    ci[rp[1:] - 1], v[rp[1:] - 1] = rows, 2.0
    ci[rp[:-1][has_sub]], v[rp[:-1][has_sub]] = rows[has_sub] - 1, -1.0
    return rp, ci, v


for m, g in ((64, None), (4096, None)) if quick else ((64, None), (256, None), (256, 64), (1024, None), (4096, None), (100000, None), (1 << 20, 64), (1 << 20, 256)):
    run(f"chain, groups of {g}", m, *chain(m, g))
if os.environ.get("TRS_PLANES"):  # slope = per-plane hand-off, intercept = in-plane time
    for nx in (1, 4, 16, 48, 108, 216):
        nn, rp, ci, v = matgen.poisson_3d_7pt(nx, 108, 108)
        run(f"3-D 7-pt lower, {nx}x108x108", nn, *lower_of(nn, rp, ci, v))
    for ny in (1, 4, 16, 108, 432):
        nn, rp, ci, v = matgen.poisson_3d_7pt(1, ny, 108)
        run(f"2-D lines of 108, {ny} lines", nn, *lower_of(nn, rp, ci, v))
    for ny in (1, 4, 16, 108):
        nn, rp, ci, v = matgen.poisson_3d_7pt(1, ny, 1000)
        run(f"2-D lines of 1000, {ny} lines", nn, *lower_of(nn, rp, ci, v))
    sys.exit(0)
for g in (200, 1000):
    nn, rp, ci, v = matgen.poisson_2d_5pt(g)
    run(f"2-D 5-pt lower, {g}^2", nn, *lower_of(nn, rp, ci, v))
for g in (48, 108):
    nn, rp, ci, v = matgen.poisson_3d_7pt(g)
    run(f"3-D 7-pt lower, {g}^3", nn, *lower_of(nn, rp, ci, v))
