#!/usr/bin/env python3
"""Brick shapes of the pipelined triangular solve on the 108^3 ILU(0)-shaped lower factor: time per solve for
edges (k, j, i) forced through GKOMI_TRS_BRICK_EDGES.  A level of a brick is one step while the product of the two
smaller edges is <= 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, matgen
import gkomi.solvers as solvers
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 108
n, rp, ci, v = matgen.at_like(g)
rows = np.repeat(np.arange(n), np.diff(rp))
for lower in (True, False):
    keep = (ci <= rows) if lower else (ci >= rows)
    trp = np.zeros(n + 1, np.int32); np.add.at(trp, rows[keep] + 1, 1); np.cumsum(trp, out=trp)
    rpd, cid, vd = d(trp), d(ci[keep].copy()), d(v[keep].copy())
    b = torch.from_numpy(np.sin(0.1 * np.arange(n)) + 2.0).cuda().reshape(n, 1)
    x = torch.zeros_like(b)
    ref = None
    for edges in (sys.argv[2].split(";") if len(sys.argv) > 2 else ("27,8,8", "36,8,8", "18,8,8", "14,8,8", "54,8,8", "27,6,10", "27,4,16", "27,16,4", "27,9,7", "12,12,12", "27,5,12")):
        os.environ["GKOMI_TRS_BRICK_EDGES"] = edges
        try:
            bk = solvers.TrsBricks(gk, n, rpd, cid, vd, lower, 0, 0, 2)
        except gkomi.GkomiError as e:
            print(f"{'lower' if lower else 'upper'} edges {edges:10s}: {e}", flush=True)
            continue
        ts = []
        for _ in range(5):
            x.fill_(7.0)
            bk.solve(b, x); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                bk.solve(b, x)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        if ref is None:
            ref = x.clone()
        print(f"{'lower' if lower else 'upper'} edges {edges:10s}: {sorted(ts)[2]:7.1f} us  bricks {bk.nbricks:5d} brick levels {bk.coarse_levels:3d} "
              f"lds {bk.lds_bytes // 1024:3d} KiB identical={bool(torch.equal(x, ref))} overrun={int(bk.overrun())}", flush=True)
        del bk
