#!/usr/bin/env python3
"""One matrix class of tools/spmv_classes.json, its CSR SpMV (automatic strategy) 10 times, for rocprofv3 --pmc
passes on the gather of b (tools/gather_pmc.sh): L2 hits / misses and the L2's fabric-side read requests."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import gkomi
import benchmark_spmv as bs
gk = gkomi.lib()
case = json.loads(sys.argv[1])
fmt = sys.argv[2] if len(sys.argv) > 2 else "csr"
A = bs.random_matrix(gk, case, 42) if "random" in case else bs.stencil_matrix(gk, case["stencil"], int(case["size"]))
M = A.to(fmt)
b = torch.from_numpy(np.random.default_rng(42).uniform(-1.0, 1.0, (A.ncols, 1))).cuda()
x = torch.zeros((A.nrows, 1), dtype=torch.float64, device="cuda")
for _ in range(10):
    M.apply(b, x)
torch.cuda.synchronize()
print(json.dumps({"rows": A.nrows, "nnz": A.nnz, "algorithmic_bytes": M.storage_bytes() + 8 * (A.ncols + A.nrows)}))
