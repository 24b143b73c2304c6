#!/usr/bin/env python3
"""Per-iteration time of the Krylov drivers on the 1M-row problems: BiCGSTAB
reference sequence vs the fused 6-launch driver, FCG, CGS, CG (fused) for
comparison; `formats` as second argument: the fused drivers on CSR / ELL /
SELL-P / COO system matrices.  Usage: python tools/tune_krylov.py [grid] [fused|formats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gkomi
import gkomi.solvers as solvers
import matgen

gk = gkomi.lib()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def timed(label, fn):
    fn()
    el = float("inf")
    for _ in range(3):   # best of 3: the caching allocator occasionally stalls a call for ~70 ms
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); el = min(el, time.perf_counter() - t0)
    it = max(r["iterations"], 1)
    print(f"  {label:44s} {r['iterations']:5d} iters {el*1e3:9.2f} ms {el/it*1e6:8.1f} us/it conv={r['converged']} rel_res={r['rel_residual']:.2e}")
    return r


for name in ("poisson2d", "convection3d"):
    if name == "poisson2d":
        n, rp, ci, v = matgen.poisson_2d_5pt(grid)
    else:
        g3 = max(8, int(round((grid * grid) ** (1 / 3))))
        n, rp, ci, v = matgen.poisson_3d_7pt(g3)
        v = v.copy(); rows = np.repeat(np.arange(n), np.diff(rp))
        v[ci == rows - 1] -= 0.5; v[ci == rows] += 0.5
    rpd, cid, vd = d(rp), d(ci), d(v)
    sv = np.sin(np.arange(n, dtype=np.float64)); sv /= np.linalg.norm(sv)
    b = torch.empty((n, 1), dtype=torch.float64, device="cuda")
    gk.csr_spmv_f64_i32(torch.cuda.current_stream().cuda_stream, n, n, 1, len(v), rpd, cid, vd, d(sv.reshape(n, 1)), 1, b, 1,
                        None, None, 0, 7)
    bv = b[:, 0].contiguous()
    print(f"{name}: n={n} nnz={len(v)}")
    kw = dict(max_iters=5000, reduction=1e-10)
    if len(sys.argv) > 2 and sys.argv[2] == "formats":
        from gkomi import formats
        A = formats.Csr(gk, n, n, rpd, cid, vd)
        spmv_x = d(sv.reshape(n, 1)); spmv_y = torch.zeros_like(spmv_x)
        for fmt in ("csr", "ell", "sellp", "coo"):
            M = A if fmt == "csr" else A.to(fmt)
            for _ in range(3):
                M.apply(spmv_x, spmv_y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                M.apply(spmv_x, spmv_y)
            e1.record(); torch.cuda.synchronize()
            print(f"  {fmt:6s} spmv (warm) {e0.elapsed_time(e1) * 20:7.1f} us")
            solvers_here = ("cg", "fcg", "bicgstab", "cgs") if name == "poisson2d" else ("bicgstab", "cgs")
            for sol in solvers_here:
                if sol == "cg":
                    timed(f"{fmt:6s} cg reference sequence (op)", lambda: solvers.solve_op(gk, "cg", M, bv, **kw))
                timed(f"{fmt:6s} {sol} fused (op)", lambda: solvers.solve_op(gk, sol, M, bv, fused=True, check_every=16, **kw))
        timed("csr    cg fused (CSR entry)", lambda: solvers.cg_solve(gk, n, rpd, cid, vd, b, mode=1, check_every=16, **kw))
        timed("csr    cg fused (CSR entry, max_row_nnz hint)", lambda: solvers.cg_solve(
            gk, n, rpd, cid, vd, b, mode=1, check_every=16, max_row_nnz=A.max_row_nnz(), **kw))
        continue
    if len(sys.argv) > 2 and sys.argv[2] == "fused":   # for rocprofv3: only the fused driver
        timed("bicgstab fused, check_every 32", lambda: solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bv, check_every=32, fused=True, **kw))
        continue
    timed("bicgstab reference sequence, check_every 8", lambda: solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bv, check_every=8, **kw))
    timed("bicgstab fused, check_every 8", lambda: solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bv, check_every=8, fused=True, **kw))
    timed("bicgstab fused, check_every 32", lambda: solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bv, check_every=32, fused=True, **kw))
    jac = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=4)
    timed("bicgstab reference sequence + jacobi(4)", lambda: solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bv, precond=jac, **kw))
    timed("bicgstab fused + jacobi(4)", lambda: solvers.krylov_solve(gk, "bicgstab", n, rpd, cid, vd, bv, precond=jac, fused=True, **kw))
    timed("cgs reference sequence", lambda: solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bv, **kw))
    timed("cgs fused", lambda: solvers.krylov_solve(gk, "cgs", n, rpd, cid, vd, bv, fused=True, check_every=16, **kw))
    if name == "poisson2d":
        timed("fcg reference sequence", lambda: solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bv, **kw))
        timed("fcg fused", lambda: solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bv, fused=True, check_every=16, **kw))
        timed("fcg fused + jacobi(4)", lambda: solvers.krylov_solve(gk, "fcg", n, rpd, cid, vd, bv, fused=True, precond=jac, check_every=16, **kw))
        timed("cg fused", lambda: solvers.cg_solve(gk, n, rpd, cid, vd, b, mode=1, check_every=16, **kw))
