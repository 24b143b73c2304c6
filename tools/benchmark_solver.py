#!/usr/bin/env python3
"""benchmark/solver of the reference (benchmark/solver/solver.cpp) over the C
ABI.  stdin: the same JSON list ([{"filename": "A.mtx", "optimal": {"spmv":
"csr"}}, ...] or {"stencil": ..., "size": ...}); stdout: per case "solver":
{"<solver>[-<preconditioner>]": {"generate": {"time"}, "apply": {"iterations",
"time"}, "residual_norm", "rhs_norm", "completed"}}.  Right-hand side "sinus"
(b = A s/|s|, s_i = sin(i), solver.cpp:145-162) or "1"; x0 = 0; criterion
Combined(Iteration(max_iters), ResidualNorm(rel_res_goal, rhs_norm)).
SURVEY 8(f) rank 4."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
import numpy as np
import torch

import gkomi
from gkomi import formats, solvers
from benchmark_spmv import stencil_matrix


def make_preconditioner(gk, name, A, args):
    n = A.nrows
    if name == "none":
        return None
    if name == "jacobi":
        return solvers.jacobi_generate(gk, n, A.row_ptrs, A.col_idxs, A.vals, max_block_size=args.jacobi_max_block_size,
                                       storage_optimization=solvers.AUTODETECT if args.jacobi_storage == "autodetect" else None)
    if name in ("parilu", "ilu"):
        return solvers.par_ilu_generate(gk, n, A.row_ptrs, A.col_idxs, A.vals, iterations=args.parilu_iterations)
    if name in ("paric", "ic"):
        return solvers.par_ic_generate(gk, n, A.row_ptrs, A.col_idxs, A.vals, iterations=args.parilu_iterations)
    raise ValueError("unknown preconditioner " + name)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--solvers", default="cg,bicgstab,cgs,fcg,gmres")
    ap.add_argument("--preconditioners", default="none")
    ap.add_argument("--max_iters", type=int, default=1000)
    ap.add_argument("--rel_res_goal", type=float, default=1e-6)
    ap.add_argument("--gmres_restart", type=int, default=100)
    ap.add_argument("--rhs_generation", default="sinus", choices=["sinus", "1"])
    ap.add_argument("--jacobi_max_block_size", type=int, default=32)
    ap.add_argument("--jacobi_storage", default="0,0", choices=["0,0", "autodetect"])
    ap.add_argument("--parilu_iterations", type=int, default=5)
    ap.add_argument("--repetitions", type=int, default=1)
    args = ap.parse_args()
    gk = gkomi.lib()
    cases = json.load(sys.stdin)
    for case in cases:
        try:
            A = formats.read_matrix(gk, case["filename"]) if "filename" in case else stencil_matrix(gk, case["stencil"], int(case["size"]))
        except Exception as e:
            case["error"] = str(e)
            continue
        n = A.nrows
        case["problem"] = {"rows": n, "cols": A.ncols, "nonzeros": A.nnz}
        if args.rhs_generation == "sinus":
            s = torch.sin(torch.arange(n, dtype=torch.float64, device="cuda")).reshape(n, 1)
            s /= torch.linalg.vector_norm(s)
            b = A.apply(s, torch.zeros_like(s)).reshape(n)
        else:
            b = torch.ones(n, dtype=torch.float64, device="cuda")
        out = case.setdefault("solver", {})
        for sname in args.solvers.split(","):
            for pname in args.preconditioners.split(","):
                key = sname if pname == "none" else f"{sname}-{pname}"
                entry = out.setdefault(key, {})
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    pc = make_preconditioner(gk, pname, A, args)
                    torch.cuda.synchronize()
                    entry["generate"] = {"time": time.perf_counter() - t0}
                    best = None
                    for _ in range(args.repetitions):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        if sname == "cg":
                            r = solvers.cg_solve(gk, n, A.row_ptrs, A.col_idxs, A.vals, b, max_iters=args.max_iters,
                                                 reduction=args.rel_res_goal, precond=pc)
                        elif sname == "gmres":
                            r = solvers.gmres_solve(gk, n, A.row_ptrs, A.col_idxs, A.vals, b, krylov_dim=args.gmres_restart,
                                                    max_iters=args.max_iters, reduction=args.rel_res_goal, precond=pc)
                        else:
                            r = solvers.krylov_solve(gk, sname, n, A.row_ptrs, A.col_idxs, A.vals, b, max_iters=args.max_iters,
                                                     reduction=args.rel_res_goal, precond=pc, fused=True, check_every=16)
                        torch.cuda.synchronize()
                        el = time.perf_counter() - t0
                        best = el if best is None else min(best, el)
                    x = r["x"].reshape(n, 1)
                    res = b.reshape(n, 1) - A.apply(x, torch.zeros_like(x))
                    entry["apply"] = {"iterations": r["iterations"], "time": best}
                    entry["residual_norm"] = float(torch.linalg.vector_norm(res))
                    entry["rhs_norm"] = float(torch.linalg.vector_norm(b))
                    entry["converged"] = bool(r["converged"])
                    entry["completed"] = True
                except Exception as e:
                    entry.update(completed=False, error=str(e))
    json.dump(cases, sys.stdout, indent=4)
    print()


if __name__ == "__main__":
    main()
