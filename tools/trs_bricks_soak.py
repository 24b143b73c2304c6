#!/usr/bin/env python3
"""Soak of the brick plan of the triangular solves: many solves of the full-size factors, every result compared
bit for bit with the level plan's (which the test suite pins to the oracle); both hand-off modes, both triangles,
several brick sizes, fresh right-hand sides.  A race shows up as a mismatch or an overrun flag."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen, gkomi.solvers as solvers
from test_trs_bricks_analysis import triangle
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for name, (n, rp, ci, v) in (("7pt 108^3", matgen.at_like(108)), ("5pt 1000^2", matgen.poisson_2d_5pt(1000))):
    for lower in (True, False):
        trp, tci, tv = triangle(n, rp, ci, v, lower)
        rpd, cid, vd = d(trp), d(tci), d(tv)
        ref_plan = solvers.TrsPlan(gk, n, rpd, cid, vd, lower)
        plans = [(mode, rows, solvers.TrsBricks(gk, n, rpd, cid, vd, lower, rows, 0, mode))
                 for mode, rows in ((2, 0), (2, 600), (1, 0), (1, 1500))]
        gen = torch.Generator(device="cuda"); gen.manual_seed(n + int(lower))
        ref, x = torch.zeros((n, 1), dtype=torch.float64, device="cuda"), torch.zeros((n, 1), dtype=torch.float64, device="cuda")
        for rep in range(reps):
            b = torch.randn((n, 1), dtype=torch.float64, device="cuda", generator=gen)
            ref_plan.solve(b, ref)
            for mode, rows, bk in plans:
                x.fill_(float(rep))
                bk.solve(b, x)
                if not torch.equal(x, ref):
                    bad += 1
                    print(f"MISMATCH {name} lower={lower} mode {mode} brick_rows {rows} rep {rep}: {int((x != ref).sum())} rows differ", flush=True)
        flags = [int(bk.overrun()) for _, _, bk in plans] + [int(ref_plan.overrun())]
        bad += sum(flags)
        print(f"{name} {'lower' if lower else 'upper'}: {reps} right-hand sides x {len(plans)} brick plans identical to the level plan; overrun flags {flags}", flush=True)
print("soak", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
