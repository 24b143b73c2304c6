"""The T2-like stand-in of config 3 (randomly permuted 2-D FEM-like matrix, ~5 per row on 1.2 M columns) with the
column-partitioned strategy: does the timed analysis keep a copy, and what does the SpMV take?"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, gkomi, matgen
from gkomi import formats
gk = gkomi.lib()
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
n, rp, ci, v = matgen.t2_like_permuted(1108)
b = torch.from_numpy(np.cos(0.001 * np.arange(n))).cuda().reshape(n, 1)
y = torch.zeros(n, 1, dtype=torch.float64, device="cuda")
M = formats.Csr.from_host(gk, n, n, rp, ci, v)
print(f"t2_like_permuted_1108: n {n}, nnz {M.nnz} ({M.nnz / n:.1f} per row), automatic {timed(lambda: M.apply(b, y)):.1f} us")
for nb in (None, 2, 4):
    P = formats.Csr.from_host(gk, n, n, rp, ci, v, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
    if P.colpart(nb) is None:
        print("   ", "analysis" if nb is None else f"nb {nb}", ": no copy (blocks_for", gk.csr_colpart_blocks_for(n, n, M.nnz), ")")
        continue
    info = (ctypes.c_int64 * 8)()
    gk.csr_colpart_info(P._colpart[0], ctypes.addressof(info))
    print(f"    {'analysis kept' if nb is None else 'forced'} nb {info[0]}: {timed(lambda: P.apply(b, y)):.1f} us")
