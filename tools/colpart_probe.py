#!/usr/bin/env python3
"""Would a column-partitioned copy of a gather-bound CSR matrix pay?  Prototype with torch ops for the build
(the SpMV itself is the library's kernel on the partitioned arrays): the matrix becomes a CSR of nb * n "virtual"
rows -- virtual row k * n + r holds row r's nonzeros of column block k -- so that the workgroups resident at any
moment gather from one block of b (an L2-sized slice); y[r] = sum_k part[k * n + r] afterwards.
usage: python tools/colpart_probe.py [uniform|powerlaw|local] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import gkomi
from gkomi import formats
import benchmark_spmv as bs

gk = gkomi.lib()


def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def partition(M, nb):
    n, nc = M.nrows, M.ncols
    width = -(-nc // nb)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), (M.row_ptrs[1:] - M.row_ptrs[:-1]).long())
    key = (M.col_idxs.long() // width) * n + rows
    order = torch.argsort(key, stable=True)
    vkey = key[order]
    counts = torch.bincount(vkey, minlength=nb * n)
    vrp = torch.zeros(nb * n + 1, dtype=torch.int32, device="cuda")
    vrp[1:] = torch.cumsum(counts, 0).int()
    return formats.Csr(gk, nb * n, nc, vrp, M.col_idxs[order].contiguous(), M.vals[order].contiguous())


def main():
  for kind in (sys.argv[1:] or ["uniform", "powerlaw", "local"]):
      case = {"random": kind, "rows": 1000000, "nnz_per_row": 8 if kind == "powerlaw" else 16, "bandwidth": 2000}
      M = bs.random_matrix(gk, case, 7)
      n = M.nrows
      b = torch.from_numpy(np.cos(0.001 * np.arange(n))).cuda().reshape(n, 1)
      y = torch.zeros(n, 1, dtype=torch.float64, device="cuda")
      t0 = timed(lambda: M.apply(b, y))
      alg = 12 * M.nnz + 4 * (n + 1) + 16 * n
      print(f"{kind}: n {n}, nnz {M.nnz}, library apply {t0:7.1f} us = {alg / t0 / 1e6:.2f} TB/s", flush=True)
      for nb in (2, 4, 8, 16):
          V = partition(M, nb)
          part = torch.zeros(nb * n, 1, dtype=torch.float64, device="cuda")
          t1 = timed(lambda: V.apply(b, part))
          out = torch.empty(n, dtype=torch.float64, device="cuda")
          t2 = timed(lambda: torch.sum(part.view(nb, n), 0, out=out))
          err = float((out - y[:, 0]).abs().max() / y.abs().max())
          print(f"   nb {nb:2d} (slice {8 * n / nb / 2**20:.1f} MiB): virtual SpMV {t1:7.1f} us + reduce {t2:5.1f} us = {t1 + t2:7.1f} us "
                f"-> {alg / (t1 + t2) / 1e6:.2f} TB/s on the ORIGINAL matrix's bytes; max rel diff {err:.1e}", flush=True)
          del V, part


if __name__ == "__main__":
    main()
