#!/usr/bin/env python3
"""benchmark/spmv of the reference (benchmark/spmv/spmv.cpp:66-290) over the C
ABI: reads the same JSON test-case list on stdin ([{"filename": "A.mtx"}, ...];
additionally {"stencil": "5pt"|"7pt", "size": N} for generated matrices),
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
    n, rp, ci, v = matgen.poisson_2d_5pt(size) if kind == "5pt" else matgen.poisson_3d_7pt(size)
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
    ap.add_argument("--formats", default="csr,coo,ell,sellp,hybrid")
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
            A = formats.read_matrix(gk, case["filename"]) if "filename" in case else stencil_matrix(gk, case["stencil"], int(case["size"]))
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
                entry.update(time=t, repetitions=reps, completed=True)
                if best is None or t < best[1]:
                    best = (fmt, t)
            except Exception as e:
                entry.update(completed=False, error=str(e))
        case.setdefault("optimal", {})["spmv"] = best[0] if best else "none"
    json.dump(cases, sys.stdout, indent=4)
    print()


if __name__ == "__main__":
    main()
