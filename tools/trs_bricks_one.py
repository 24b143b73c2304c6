#!/usr/bin/env python3
"""One brick, no inflow: 50 pipelined solves of an 8 x 128 grid factor (135 levels) -- for rocprofv3 --pmc
passes over the compute wave's step loop (tools/profile_bricks.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gkomi, matgen, gkomi.solvers as solvers
from test_trs_bricks_analysis import triangle
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, rp, ci, v = matgen.poisson_2d_5pt(8, 128)
rp, ci, v = triangle(n, rp, ci, v, True)
bk = solvers.TrsBricks(gk, n, d(rp), d(ci), d(v), True, 1024, 64 if mode == 1 else 0, mode)
assert bk.nbricks == 1
b = torch.ones((n, 1), dtype=torch.float64, device="cuda"); x = torch.zeros_like(b)
for _ in range(50):
    bk.solve(b, x)
torch.cuda.synchronize()
print("levels", 8 + 128 - 1, "steps", bk.nsteps)
