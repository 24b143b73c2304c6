"""Which kernel runs the virtual matrix of the column-partitioned copy, and what would the other one take?  Uniform random
columns (virtual rows of ~8: nonzero-split kernel by the automatic strategy) forced through the load-balanced kernel, to price
what the power-law class (virtual rows of up to 55 k: load-balanced kernel) pays for its hubs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import colpart_probe as cp
from gkomi import formats
import benchmark_spmv as bs
gk = cp.gk
for kind, nb in (("uniform", 2), ("uniform", 4), ("powerlaw", 4)):
    M = bs.random_matrix(gk, {"random": kind, "rows": 1000000, "nnz_per_row": 8 if kind == "powerlaw" else 16}, 7)
    n = M.nrows
    b = torch.from_numpy(np.cos(0.001 * np.arange(n))).cuda().reshape(n, 1)
    V = cp.partition(M, nb)
    part = torch.zeros(nb * n, 1, dtype=torch.float64, device="cuda")
    print(f"{kind}, nb {nb}: longest virtual row {V.max_row_nnz()}", flush=True)
    for name in ("csr", "csri", "csrm"):
        W = V.to(name)
        print(f"   virtual SpMV as {name:5s}: {cp.timed(lambda: W.apply(b, part)):7.1f} us", flush=True)
