#!/usr/bin/env python3
"""Brick plan of the triangular solves (csrc/trs_bricks.hip) against the level plan on the ILU(0)-shaped
factors of the configs: time per solve, identical bits, analysis time, over brick sizes / workgroup sizes.
usage: trs_bricks_probe.py [3d|2d|both|chain|sweep] [grid]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen
import gkomi.solvers as solvers
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def tri(n, rp, ci, v, lower):
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = (ci <= rows) if lower else (ci >= rows)
    rp2 = np.zeros(n + 1, np.int32); np.add.at(rp2, rows[keep] + 1, 1); np.cumsum(rp2, out=rp2)
    return rp2, ci[keep].copy(), v[keep].copy()


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(name, n, rp, ci, v, sizes, threads_list, psizes):
    for lower in (True, False):
        trp, tci, tv = tri(n, rp, ci, v, lower)
        rpd, cid, vd = d(trp), d(tci), d(tv)
        b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
        x = torch.zeros_like(b)
        t0 = time.perf_counter(); plan = solvers.TrsPlan(gk, n, rpd, cid, vd, lower); torch.cuda.synchronize()
        t_an = (time.perf_counter() - t0) * 1e3
        t_lvl = timed(lambda: plan.solve(b, x))
        ref = x.clone()
        print(f"{name} {'lower' if lower else 'upper'} n={n}: level plan {t_lvl:8.1f} us  levels {plan.nlevels}  analysis {t_an:.1f} ms", flush=True)
        for rows, threads, mode in [(r, t, 1) for r in sizes for t in threads_list] + [(r, 0, 2) for r in psizes]:
            if True:
                t0 = time.perf_counter()
                try:
                    bk = solvers.TrsBricks(gk, n, rpd, cid, vd, lower, rows, threads, mode)
                except gkomi.GkomiError as e:
                    print(f"   bricks {rows}: {e}"); continue
                torch.cuda.synchronize(); t_an = (time.perf_counter() - t0) * 1e3
                x.fill_(7.0)
                t = timed(lambda: bk.solve(b, x))
                same = bool(torch.equal(x, ref))
                print(f"   mode {bk.mode} brick_rows {rows:6d} threads {bk.threads:3d}: {t:8.1f} us  identical={same} overrun={int(bk.overrun())}  "
                      f"bricks {bk.nbricks:6d} brick levels {bk.coarse_levels:4d} critical steps {bk.critical_steps:6d} "
                      f"lds {bk.lds_bytes // 1024:3d} KiB  estimate {bk.estimate_us():7.1f} us  analysis {t_an:6.1f} ms", flush=True)


def sweep(name, n, rp, ci, v, rows_list):
    """pump cadence of the pipelined solve"""
    trp, tci, tv = tri(n, rp, ci, v, True)
    rpd, cid, vd = d(trp), d(tci), d(tv)
    b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
    x = torch.zeros_like(b)
    for rows in rows_list:
        bk = solvers.TrsBricks(gk, n, rpd, cid, vd, True, rows, 0, 2)
        for nap in (1, 2, 4, 8, 16, 32):
            os.environ["GKOMI_TRS_BRICK_NAP"] = str(nap)
            print(f"{name} brick_rows {rows} nap {nap:2d}: {timed(lambda: bk.solve(b, x)):8.1f} us  overrun={int(bk.overrun())}", flush=True)
    os.environ.pop("GKOMI_TRS_BRICK_NAP")


what = sys.argv[1] if len(sys.argv) > 1 else "both"
if what == "chain":
    # grids that are one chain of bricks: time per level = LDS step + (hand-off latency) / (levels per brick)
    for (nx, ny, rows) in ((32, 3200, 1024), (64, 1600, 4096), (16, 6400, 256), (1024, 64, 1024)):
        n, rp, ci, v = matgen.poisson_2d_5pt(nx, ny)
        trp, tci, tv = tri(n, rp, ci, v, True)
        rpd, cid, vd = d(trp), d(tci), d(tv)
        b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
        x = torch.zeros_like(b)
        for mode in (1, 2):
            bk = solvers.TrsBricks(gk, n, rpd, cid, vd, True, rows, 64 if mode == 1 else 0, mode)
            t = timed(lambda: bk.solve(b, x))
            lv = nx + ny - 1
            print(f"{nx}x{ny} mode {mode} bricks {bk.nbricks} brick levels {bk.coarse_levels} steps {bk.critical_steps} "
                  f"levels {lv}: {t:8.1f} us = {t / lv * 1e3:6.1f} ns/level", flush=True)
    sys.exit(0)
if what == "sweep":
    n, rp, ci, v = matgen.poisson_3d_7pt(108)
    sweep("7pt 108^3", n, rp, ci, v, (512, 1000, 1728))
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    sweep("5pt 1000^2", n, rp, ci, v, (1024, 2304))
    sys.exit(0)
quick = os.environ.get("TRS_QUICK")
if what in ("3d", "both"):
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 108
    n, rp, ci, v = matgen.poisson_3d_7pt(g)
    run(f"7pt {g}^3", n, rp, ci, v, (4096,) if quick else (1728, 4096), (0,), (0,) if quick else (0, 216, 512, 1000, 1728))
if what in ("2d", "both"):
    g = int(sys.argv[2]) if len(sys.argv) > 2 and what == "2d" else 1000
    n, rp, ci, v = matgen.poisson_2d_5pt(g)
    run(f"5pt {g}^2", n, rp, ci, v, (4096,) if quick else (1024, 4096), (0,), (0,) if quick else (0, 256, 1024, 2304))
