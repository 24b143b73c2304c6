#!/usr/bin/env python3
"""benchmark/spmv of the reference (benchmark/spmv/spmv.cpp:66-290) over the C
ABI: reads the same JSON test-case list on stdin ([{"filename": "A.mtx"}, ...];
additionally {"stencil": "5pt"|"7pt"|"27pt", "size": N} and {"random": "uniform"|"local"|
"powerlaw", "rows": N, "nnz_per_row": k} for generated matrices),
writes the same result layout on stdout: per case "spmv": {format: {"storage",
"max_relative_norm2", "time" [s], "repetitions", "completed"}}, "optimal":
{"spmv": best format}; the answer every format is checked against is COO's,
as in the reference (--detailed).  SURVEY 8(f) rank 4."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
import numpy as np
import torch

import gkomi
from gkomi import formats


def stencil_matrix(gk, kind, size):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import matgen
    gen = {"5pt": matgen.poisson_2d_5pt, "7pt": matgen.poisson_3d_7pt, "27pt": matgen.stencil_3d_27pt}[kind]
    n, rp, ci, v = gen(size)
    return formats.Csr.from_host(gk, n, n, rp, ci, v)


def random_matrix(gk, case, seed):
    """{"random": "uniform"|"local"|"powerlaw", "rows": N, "nnz_per_row": k}: stand-ins for
    the irregular SuiteSparse classes (no SuiteSparse files offline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import matgen
    n, k = int(case["rows"]), int(case["nnz_per_row"])
    rng = np.random.default_rng(seed)
    kind = case["random"]
    if kind == "powerlaw":  # most rows short, a heavy tail of long ones (web / circuit graphs)
        counts = np.minimum(n // 4, np.maximum(1, (k * 0.5 * rng.pareto(1.3, size=n)).astype(np.int64)))
    elif kind == "fixed":   # every row exactly k entries near the diagonal (LDS bank-conflict probe)
        counts = np.full(n, k)
    else:
        counts = rng.integers(max(1, k // 2), k + k // 2 + 1, size=n)
    if kind == "fixed":
        # distinct columns by construction: a strided window around the diagonal
        rows = np.repeat(np.arange(n, dtype=np.int64), k)
        offs = np.tile((np.arange(k) - k // 2) * 3, n)
        cols = np.clip(rows + offs, 0, n - 1)
        keep = np.ones(len(cols), dtype=bool)
        keep[1:] = (cols[1:] != cols[:-1]) | (rows[1:] != rows[:-1])
        rows, cols = rows[keep], cols[keep]
        rp = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=n), out=rp[1:])
        return formats.Csr.from_host(gk, n, n, rp, cols.astype(np.int32), rng.standard_normal(len(cols)))
    rp, ci, v = matgen.random_rows_csr(n, n, counts, seed, local=int(case.get("bandwidth", 2000)) if kind == "local" else None)
    return formats.Csr.from_host(gk, n, n, rp, ci, v)


def timed(fn, warmup, min_reps, min_seconds):
    """IterationControl (benchmark/utils/general.hpp:96-117): warm-up runs, then
    at least min_reps repetitions and min_seconds of work, GPU-timed"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    reps, total = 0, 0.0
    while reps < min_reps or total < min_seconds:
        batch = max(min_reps - reps, 10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(batch):
            fn()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1) * 1e-3
        reps += batch
    return total / reps, reps


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--formats", default="csr,coo,ell,sellp,hybrid",
                    help="as benchmark/utils/formats.hpp: csr (automatical), csri (load_balance), csrm (merge_path), "
                         "csrc (classical), csrs, coo, ell, sellp, hybrid; csrp = column-partitioned copy where it pays (gkomi_csr_colpart_*); csri_serial = load_balance with every row "
                         "segment added by one thread (the kernel of rounds 1-3, for A/B timings)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--min_repetitions", type=int, default=10)
    ap.add_argument("--min_runtime", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    gk = gkomi.lib()
    cases = json.load(sys.stdin)
    assert isinstance(cases, list), 'expected a JSON list of {"filename": ...} objects'
    rng = np.random.default_rng(args.seed)
    for case in cases:
        try:
            if "filename" in case:
                A = formats.read_matrix(gk, case["filename"])
            elif "random" in case:
                A = random_matrix(gk, case, args.seed)
            else:
                A = stencil_matrix(gk, case["stencil"], int(case["size"]))
        except Exception as e:  # keep going like the reference (--keep_errors)
            case["error"] = str(e)
            continue
        case["problem"] = {"rows": A.nrows, "cols": A.ncols, "nonzeros": A.nnz}
        b = torch.from_numpy(rng.uniform(-1.0, 1.0, (A.ncols, args.nrhs))).cuda()
        x = torch.zeros((A.nrows, args.nrhs), dtype=torch.float64, device="cuda")
        answer = formats.Coo.from_csr(A).apply(b, torch.zeros_like(x))
        spmv = case.setdefault("spmv", {})
        best = None
        for fmt in args.formats.split(","):
            entry = spmv.setdefault(fmt, {})
            try:
                M = A.to(fmt)
                entry["storage"] = M.storage_bytes()
                got = M.apply(b, torch.zeros_like(x))
                num = torch.linalg.vector_norm(got - answer, dim=0)
                den = torch.linalg.vector_norm(answer, dim=0)
                entry["max_relative_norm2"] = float(torch.max(num / torch.where(den == 0, torch.ones_like(den), den)))
                t, reps = timed(lambda: M.apply(b, x), args.warmup, args.min_repetitions, args.min_runtime)
                # not in the reference's schema: algorithmic GB/s (storage + b + x once)
                gbs = (entry["storage"] + 8 * args.nrhs * (A.ncols + A.nrows)) / t / 1e9
                entry.update(time=t, repetitions=reps, completed=True, bandwidth_gbs=gbs, frac_of_8tbs=gbs / 8000.0)
                if best is None or t < best[1]:
                    best = (fmt, t)
            except Exception as e:
                entry.update(completed=False, error=str(e))
        case.setdefault("optimal", {})["spmv"] = best[0] if best else "none"
    json.dump(cases, sys.stdout, indent=4)
    print()


if __name__ == "__main__":
    main()
