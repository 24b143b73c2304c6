#!/usr/bin/env python3
"""Where the ILU generate of the AT-like 108^3 system spends its time: ParILU chain, level analysis, brick
analysis (GKOMI_ANALYSIS_TRACE=1 prints the host phases)."""
import os, sys, time
os.environ["GKOMI_ANALYSIS_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, matgen
from gkomi import solvers
gk = gkomi.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n, rp, ci, v = matgen.at_like(108)
a = [dev(rp), dev(ci), dev(v)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pre = solvers.par_ilu_generate(gk, n, a[0].clone(), a[1], a[2], iterations=5)
    torch.cuda.synchronize()
    print(f"par_ilu_generate total {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
L = pre.L
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pl = solvers.TrsPlan(gk, n, L[0], L[1], L[2], True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    bk = solvers.TrsBricks(gk, n, L[0], L[1], L[2], True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"level analysis {1e3 * (t1 - t0):.2f} ms, brick analysis + numeric {1e3 * (t2 - t1):.2f} ms", flush=True)
print("cpus", len(os.sched_getaffinity(0)))
