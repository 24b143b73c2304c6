#!/usr/bin/env python3
"""Set-up paths that run the library's own radix sort / scans: Csr::read(device_matrix_data) of the 1M-row Poisson
triplets (sort + sum_duplicates), the level analysis of a 108^3 factor, Jacobi(32) generate on the permuted 1108^2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, matgen
from gkomi import formats, solvers
gk = gkomi.lib()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3

n, rp, ci, v = matgen.poisson_2d_5pt(1000)
rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
perm = np.random.default_rng(1).permutation(len(v))
r, c, vv = d(rows[perm]), d(ci[perm]), d(v[perm])
print(f"Csr::read of {len(v)} shuffled triplets (sort_row_major + sum_duplicates + idxs -> ptrs): {timed(lambda: formats.Csr.from_triplets(gk, n, n, r, c, vv)):.2f} ms", flush=True)
n, rp, ci, v = matgen.at_like(108)
rws = np.repeat(np.arange(n), np.diff(rp)); keep = ci <= rws
trp = np.zeros(n + 1, np.int32); np.add.at(trp, rws[keep] + 1, 1); np.cumsum(trp, out=trp)
L = [d(trp), d(ci[keep].copy()), d(v[keep].copy())]
print(f"level analysis (symbolic + numeric) of the 108^3 lower factor: {timed(lambda: solvers.TrsPlan(gk, n, L[0], L[1], L[2], True)):.2f} ms", flush=True)
n, rp, ci, v = matgen.t2_like_permuted(1108)
a = [d(rp), d(ci), d(v)]
print(f"Jacobi(32) find_blocks + generate on the permuted 1108^2 matrix: {timed(lambda: solvers.jacobi_generate(gk, n, a[0], a[1], a[2], max_block_size=32)):.2f} ms", flush=True)
