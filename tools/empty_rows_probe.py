import sys, os
sys.path[:0] = ["/root/repo/repo-8852-ginkgo_amd", "/root/repo/tests"]
import numpy as np, torch, gkomi, matgen
from gpu_util import DevCsr, csr_apply_srow, csr_apply, dev, make_srow
gk = gkomi.lib()
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rng = np.random.default_rng(1)
# 1: split kernel, empty runs
nrows, ncols = 400003, 9001
counts = rng.integers(0, 7, size=nrows); counts[5000:260000] = 0; counts[300000:300900] = 1; counts[380000:] = 0
rp, ci, v = matgen.random_rows_csr(nrows, ncols, counts, 5)
A = DevCsr(nrows, ncols, rp, ci, v); srow, tile = make_srow(gk, A, 1536)
b = dev(rng.standard_normal((ncols, 1))); out = csr_apply_srow(gk, A, b, srow, tile, strategy=4)
print("split empty runs: %.4f ms (bound 0.2)" % t(lambda: csr_apply_srow(gk, A, b, srow, tile, out, strategy=4)))
# 2: balanced
n, ncols = 420000, 50000
counts = rng.integers(0, 5, size=n); counts[2000:302000] = 0; counts[302000] = 9000; counts[310000:311500] = 1; counts[350000] = 4000; counts[400000:] = 0
rp, ci, v = matgen.random_rows_csr(n, ncols, counts, seed=6)
A = DevCsr(n, ncols, rp, ci, v); srow, tile = make_srow(gk, A)
b = dev(rng.standard_normal((ncols, 1))); out = csr_apply_srow(gk, A, b, srow, tile, None, None, None, 3)
print("balanced empty runs (srow): %.4f ms (bound 0.3)" % t(lambda: csr_apply_srow(gk, A, b, srow, tile, out, None, None, 3)))
out2 = csr_apply(gk, A, b, None, None, None, 3)
print("balanced empty runs (search): %.4f ms (bound 0.3)" % t(lambda: csr_apply(gk, A, b, out2, None, None, 3)))
# 3: non-local block
n, halo = 2097152, 131072
counts = np.zeros(n, np.int64); counts[:65536] = 1; counts[-65536:] = 1
rp = np.zeros(n + 1, np.int32); np.cumsum(counts, out=rp[1:]); ci = np.arange(halo, dtype=np.int32); v = rng.standard_normal(halo)
A = DevCsr(n, halo, rp, ci, v); srow, tile = make_srow(gk, A)
b = dev(rng.standard_normal((halo, 1))); out = dev(rng.standard_normal((n, 1)))
print("non-local block, automatic: %.4f ms (bound 0.1)" % t(lambda: csr_apply_srow(gk, A, b, srow, tile, out, 1.0, 1.0, 0)))
print("non-local block, explicit split (sparse-rows mode): %.4f ms" % t(lambda: csr_apply_srow(gk, A, b, srow, tile, out, 1.0, 1.0, 4)))
