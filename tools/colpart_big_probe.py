"""Column-partitioned copy on matrices beyond 1 M columns (uniformly random columns): the library's automatic kernel vs
the copy with the block count the timed analysis keeps and with forced counts."""
import os, sys
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np, torch, gkomi
from gkomi import formats
import benchmark_spmv as bs
gk = gkomi.lib()
def timed(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows, per in ((2000000, 16), (4000000, 16), (3000000, 8)):
    M = bs.random_matrix(gk, {"random": "uniform", "rows": rows, "nnz_per_row": per}, 5)
    n = M.nrows
    b = torch.from_numpy(np.cos(0.001 * np.arange(n))).cuda().reshape(n, 1)
    y = torch.zeros(n, 1, dtype=torch.float64, device="cuda")
    t0 = timed(lambda: M.apply(b, y))
    alg = 12 * M.nnz + 4 * (n + 1) + 16 * n
    print(f"rows {rows}, {per} per row, nnz {M.nnz}: library {t0:8.1f} us = {alg / t0 / 1e6:.2f} TB/s", flush=True)
    import ctypes
    for nb in (None, 4, 8):
        P = formats.Csr(gk, M.nrows, M.ncols, M.row_ptrs, M.col_idxs, M.vals, strategy=formats.Csr.CSR_STRATEGIES["csrp"])
        if P.colpart(nb) is None:
            print("   nb", nb, "not built"); continue
        info = (ctypes.c_int64 * 8)()
        gk.csr_colpart_info(P._colpart[0], ctypes.addressof(info))
        t1 = timed(lambda: P.apply(b, y))
        print(f"   {'analysis chose' if nb is None else 'forced'} nb {info[0]} (slice {8 * n / info[0] / 2**20:.1f} MiB): {t1:8.1f} us = {alg / t1 / 1e6:.2f} TB/s", flush=True)
        del P
        torch.cuda.empty_cache()
