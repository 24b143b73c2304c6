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
                      f"lds {bk.lds_bytes // 1024:3d} KiB  estimate {bk.estimate_us(plan.nlevels):7.1f} us  analysis {t_an:6.1f} ms", flush=True)


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
if what == "edges":
    # brick shapes: cubes against bricks that are long along one axis (every level fits the 64-lane wave)
    n, rp, ci, v = matgen.poisson_3d_7pt(108)
    trp, tci, tv = tri(n, rp, ci, v, True)
    rpd, cid, vd = d(trp), d(tci), d(tv)
    b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
    x = torch.zeros_like(b)
    ref_plan = solvers.TrsPlan(gk, n, rpd, cid, vd, True); ref = torch.zeros_like(b); ref_plan.solve(b, ref)
    for edges in ("12,12,12", "10,10,10", "8,8,27", "8,27,8", "27,8,8", "7,9,27", "4,16,27", "16,4,27", "5,12,27", "8,8,13", "8,8,18", "6,10,27", "7,7,36", "6,8,36", "4,8,54", "8,4,54", "8,8,20"):
        os.environ["GKOMI_TRS_BRICK_EDGES"] = edges
        try:
            bk = solvers.TrsBricks(gk, n, rpd, cid, vd, True, 0, 0, 2)
        except gkomi.GkomiError as e:
            print(f"edges {edges}: {e}"); continue
        x.fill_(3.0)
        t = timed(lambda: bk.solve(b, x))
        print(f"edges {edges:10s}: {t:8.1f} us  identical={bool(torch.equal(x, ref))} bricks {bk.nbricks:5d} brick levels {bk.coarse_levels:3d} "
              f"steps {bk.nsteps:6d} lds {bk.lds_bytes // 1024:3d} KiB", flush=True)
    n, rp, ci, v = matgen.poisson_2d_5pt(1000)
    trp, tci, tv = tri(n, rp, ci, v, True)
    rpd, cid, vd = d(trp), d(tci), d(tv)
    b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
    x = torch.zeros_like(b)
    ref_plan = solvers.TrsPlan(gk, n, rpd, cid, vd, True); ref = torch.zeros_like(b); ref_plan.solve(b, ref)
    for edges in ("37,37", "32,32", "45,45", "64,31", "31,64", "50,40", "64,16", "16,64", "40,50"):
        os.environ["GKOMI_TRS_BRICK_EDGES"] = edges
        try:
            bk = solvers.TrsBricks(gk, n, rpd, cid, vd, True, 0, 0, 2)
        except gkomi.GkomiError as e:
            print(f"2-D edges {edges}: {e}"); continue
        x.fill_(3.0)
        t = timed(lambda: bk.solve(b, x))
        print(f"2-D edges {edges:10s}: {t:8.1f} us  identical={bool(torch.equal(x, ref))} bricks {bk.nbricks:5d} brick levels {bk.coarse_levels:3d} "
              f"steps {bk.nsteps:6d} lds {bk.lds_bytes // 1024:3d} KiB", flush=True)
    os.environ.pop("GKOMI_TRS_BRICK_EDGES")
    sys.exit(0)
if what == "timeline":
    # every brick stamps start / in LDS / first step done / last step done (shader clock): who waits for what
    import ctypes
    fn = gk._cdll.gkomi_trs_bricks_debug_stamps
    fn.argtypes = [ctypes.c_void_p] * 4
    n, rp, ci, v = matgen.poisson_3d_7pt(108)
    trp, tci, tv = tri(n, rp, ci, v, True)
    rpd, cid, vd = d(trp), d(tci), d(tv)
    b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
    x = torch.zeros_like(b)
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bk = solvers.TrsBricks(gk, n, rpd, cid, vd, True, rows, 0, 2)
    os.environ["GKOMI_TRS_BRICK_STAMPS"] = "-1"
    for _ in range(3): bk.solve(b, x)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * (8 * bk.nbricks))()
    fn(None, bk.handle.value, bk.plan.data_ptr(), ctypes.addressof(out))
    t = np.array(out, dtype=np.int64).reshape(-1, 8)
    us = (t - t[:, 0].min()) / 100.0   # s_memrealtime: 100 MHz, one clock for the whole chip
    def harr(which):
        data = ctypes.POINTER(ctypes.c_int32)(); count = ctypes.c_int64(0)
        gk.trs_bricks_host_array(bk.handle.value, which, ctypes.addressof(data), ctypes.addressof(count))
        return np.ctypeslib.as_array(data, shape=(count.value,)).copy()
    pred_ptr, pred_idx = harr(6), harr(7)
    print(f"bricks {bk.nbricks}, brick levels {bk.coarse_levels}; first start to last end {us[:, 3].max():.1f} us")
    load, wait, run = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    # hand-off: my first step done - the latest first-step-done among the bricks I depend on
    hop = np.array([us[r, 2] - us[pred_idx[pred_ptr[r]:pred_ptr[r + 1]], 2].max() for r in range(bk.nbricks) if pred_ptr[r + 1] > pred_ptr[r]])
    lead = np.array([us[pred_idx[pred_ptr[r]:pred_ptr[r + 1]], 2].max() - us[r, 1] for r in range(bk.nbricks) if pred_ptr[r + 1] > pred_ptr[r]])
    print(f"medians [us]: image + rhs into LDS {np.median(load):.1f}, then until the first step is done {np.median(wait):.1f}, the other steps {np.median(run):.1f}")
    print(f"first step done after the slowest predecessor's first step: median {np.median(hop):.2f} us, 10% {np.percentile(hop, 10):.2f}, 90% {np.percentile(hop, 90):.2f}")
    print(f"a brick is in LDS before its slowest predecessor's first step is done by: median {np.median(lead):.1f} us, 10% {np.percentile(lead, 10):.1f} (negative = the brick was late)")
    # anatomy of a hand-off: the slowest predecessor's steps 0..9 done (its level 9 = what my first row needs, 10^3
    # bricks) -> my pump publishes its first inflow -> my first step done
    has = np.array([r for r in range(bk.nbricks) if pred_ptr[r + 1] > pred_ptr[r]])
    p9 = np.array([us[pred_idx[pred_ptr[r]:pred_ptr[r + 1]], 5].max() for r in has])
    print(f"hand-off anatomy [us, medians]: predecessor's first step -> its step 9: {np.median(p9 - np.array([us[pred_idx[pred_ptr[r]:pred_ptr[r + 1]], 2].max() for r in has])):.2f}; "
          f"its step 9 -> my pump's first publication: {np.median(us[has, 4] - p9):.2f}; publication -> my first step done: {np.median(us[has, 2] - us[has, 4]):.2f}")
    print(f"resident at once (max over time): {max(int(((us[:, 0] <= tt) & (us[:, 3] >= tt)).sum()) for tt in np.linspace(0, us[:, 3].max(), 400))}")
    # the critical chain backwards from the last brick to finish
    r = int(np.argmax(us[:, 3])); chain = []
    while True:
        chain.append(r)
        ps = pred_idx[pred_ptr[r]:pred_ptr[r + 1]]
        if len(ps) == 0: break
        r = int(ps[np.argmax(us[ps, 2])])
    print("critical chain (rank: started, in LDS, first step done, last step done):")
    for r in chain[::-1][::max(1, len(chain) // 16)]:
        print(f"  {r:5d}: {us[r, 0]:7.1f} {us[r, 1]:7.1f} {us[r, 2]:7.1f} {us[r, 3]:7.1f}")
    sys.exit(0)
if what == "stamps":
    # shader-clock stamps at the top of every second step of one brick (GKOMI_TRS_BRICK_STAMPS build path)
    import ctypes
    fn = gk._cdll.gkomi_trs_bricks_debug_stamps
    fn.argtypes = [ctypes.c_void_p] * 4
    for (nx, ny, rows, brick) in ((8, 128, 1024, 0), (32, 3200, 1024, 50), (108, 108 * 108, 1024, 600)):
        n, rp, ci, v = matgen.poisson_2d_5pt(nx, ny) if ny != 108 * 108 else matgen.poisson_3d_7pt(108)
        trp, tci, tv = tri(n, rp, ci, v, True)
        rpd, cid, vd = d(trp), d(tci), d(tv)
        b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
        x = torch.zeros_like(b)
        bk = solvers.TrsBricks(gk, n, rpd, cid, vd, True, rows, 0, 2)
        os.environ["GKOMI_TRS_BRICK_STAMPS"] = str(min(brick, bk.nbricks - 1))
        for _ in range(3): bk.solve(b, x)
        torch.cuda.synchronize()
        out = (ctypes.c_longlong * 1024)()
        fn(None, bk.handle.value, bk.plan.data_ptr(), ctypes.addressof(out))
        os.environ.pop("GKOMI_TRS_BRICK_STAMPS")
        ns = int(out[0]); st = np.array(out[1:1 + (ns + 1) // 2], dtype=np.int64)
        dt = np.diff(st) / 2.0
        print(f"{nx}x{ny} brick {brick} of {bk.nbricks}: {ns} steps; shader ticks per step: median {np.median(dt):.0f} min {dt.min():.0f} "
              f"max {dt.max():.0f} mean {dt.mean():.0f}; first 40: {' '.join(str(int(v)) for v in dt[:40])}", flush=True)
    sys.exit(0)
if what == "chain":
    # grids that are one chain of bricks: time per level = LDS step + (hand-off latency) / (levels per brick)
    # the first two are ONE brick each: their difference = 128 levels without any hand-off
    for (nx, ny, rows) in ((8, 128, 1024), (8, 256, 2048), (32, 3200, 1024), (64, 1600, 4096), (16, 6400, 256), (1024, 64, 1024)):
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
    run(f"7pt {g}^3", n, rp, ci, v, (4096,) if quick else (1728, 4096), (0,), (0,) if quick else (0, 512, 600, 729, 850, 1000, 1331))
if what in ("2d", "both"):
    g = int(sys.argv[2]) if len(sys.argv) > 2 and what == "2d" else 1000
    n, rp, ci, v = matgen.poisson_2d_5pt(g)
    run(f"5pt {g}^2", n, rp, ci, v, (4096,) if quick else (1024, 4096), (0,), (0,) if quick else (0, 256, 1024, 2304))
