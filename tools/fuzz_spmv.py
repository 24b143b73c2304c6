#!/usr/bin/env python3
"""Differential fuzz of the SpMV entry points against the oracle on small odd shapes (empty matrices, nnz 0 / 1 / 2 / 3,
single rows and columns, empty rows everywhere, rows longer than a tile, nnz on tile boundaries, several right-hand
sides, advanced applies): every CSR strategy (with and without srow), COO (sorted / unsorted), ELL, SELL-P, Hybrid.
usage: python tools/fuzz_spmv.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gkomi, oracle_lib
from gkomi import formats

gk, orc = gkomi.lib(), oracle_lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BITEXACT = {"csr", "csrm", "csrs"}
bad = 0
for case in range(cases):
    shape = rng.integers(0, 8)
    nrows = int(rng.choice([1, 2, 3, 7, 64, 65, 255, 256, 257, 1000, 5000]))
    ncols = int(rng.choice([1, 2, 5, 63, 64, 300, 4097]))
    if shape == 0:
        counts = np.zeros(nrows, np.int64)                                  # no nonzeros at all
    elif shape == 1:
        counts = (rng.random(nrows) < 0.1).astype(np.int64)                 # nearly empty
    elif shape == 2:
        counts = rng.integers(0, min(ncols, 9) + 1, size=nrows)
    elif shape == 3:
        counts = np.zeros(nrows, np.int64); counts[rng.integers(0, nrows)] = min(ncols, 4000)   # one long row
    elif shape == 4:
        counts = np.full(nrows, min(ncols, 3))
    elif shape == 5:
        counts = rng.integers(0, min(ncols, 70) + 1, size=nrows); counts[: nrows // 2] = 0
    elif shape == 6:
        target = int(rng.choice([1, 2, 3, 1535, 1536, 1537, 3072]))         # nnz on the tile boundaries
        counts = np.zeros(nrows, np.int64)
        left = target
        for r in range(nrows):
            k = min(left, ncols, int(rng.integers(0, 9)))
            counts[r] = k; left -= k
    else:
        counts = np.minimum(ncols, (3 * rng.pareto(1.2, size=nrows)).astype(np.int64))
    counts = np.minimum(counts, ncols)
    rp = np.zeros(nrows + 1, np.int32); np.cumsum(counts, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(ncols, size=int(k), replace=False)) for k in counts] + [np.zeros(0, np.int64)]).astype(np.int32)
    if rng.random() < 0.3 and len(ci) > 1:                                   # unsorted rows
        for r in range(nrows):
            rng.shuffle(ci[rp[r]:rp[r + 1]])
    v = rng.standard_normal(len(ci))
    nrhs = int(rng.choice([1, 1, 1, 2, 3, 5]))
    b = rng.standard_normal((ncols, nrhs)); c0 = rng.standard_normal((nrows, nrhs))
    adv = rng.random() < 0.5
    alpha, beta = (float(rng.standard_normal()), float(rng.standard_normal())) if adv else (None, None)
    expect = c0.copy() if adv else np.full((nrows, nrhs), np.nan)
    z32, zf = np.zeros(1, np.int32), np.zeros(1)
    if adv:
        orc.ref_csr_advanced_spmv(nrows, nrhs, alpha, rp, ci if len(ci) else z32, v if len(v) else zf, b, nrhs, beta, expect, nrhs)
    else:
        orc.ref_csr_spmv(nrows, nrhs, rp, ci if len(ci) else z32, v if len(v) else zf, b, nrhs, expect, nrhs)
    scale = np.abs(expect).max() + 1.0
    M = formats.Csr.from_host(gk, nrows, ncols, rp, ci if len(ci) else z32[:0], v)
    for fmt in ("csr", "csrm", "csrc", "csri", "csrp", "coo", "ell", "sellp", "hybrid"):
        for split in ((True, False) if fmt.startswith("csr") else (True,)):
            try:
                A = formats.Csr(gk, nrows, ncols, M.row_ptrs, M.col_idxs, M.vals, formats.Csr.CSR_STRATEGIES[fmt], split) if fmt.startswith("csr") else M.to(fmt)
                x = torch.from_numpy(c0.copy()).cuda() if adv else torch.full((nrows, nrhs), float("nan"), dtype=torch.float64, device="cuda")
                A.apply(torch.from_numpy(b).cuda(), x, alpha, beta)
                got = x.cpu().numpy()
            except Exception as e:  # noqa: BLE001
                print(f"case {case} shape {shape} {nrows}x{ncols} nnz {len(ci)} nrhs {nrhs} adv {adv} {fmt} split {split}: EXCEPTION {e!r}", flush=True)
                bad += 1
                continue
            # bit-exact: the stream / split kernels (what the automatic strategy runs for rows of up to 256), ELL, SELL-P;
            # the sub-wave, load-balanced, partitioned and COO kernels re-associate the sums
            exact = fmt == "csrm" or fmt in ("ell", "sellp") or (fmt in BITEXACT and (counts.max() if nrows else 0) <= 256)
            ok = np.array_equal(got, expect) if exact else np.all(np.abs(got - expect) <= 1e-12 * scale)
            if not ok:
                print(f"case {case} shape {shape} {nrows}x{ncols} nnz {len(ci)} nrhs {nrhs} adv {adv} {fmt} split {split}: MISMATCH max {np.nanmax(np.abs(got - expect)):.3e}", flush=True)
                bad += 1
print(f"{cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
