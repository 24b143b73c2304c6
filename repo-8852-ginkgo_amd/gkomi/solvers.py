"""Thin Python plumbing over the native solver drivers of the C ABI (device
buffers are torch tensors).  Mirrors the factory parameters of
gko::solver::Cg (include/ginkgo/core/solver/cg.hpp) that the hot path uses."""
import ctypes

import numpy as np
import torch

APPLY_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
BASELINES = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}


def cg_solve(gk, n, row_ptrs, col_idxs, vals, b, x=None, max_iters=1000, reduction=1e-10,
             baseline="rhs_norm", mode=1, check_every=16, strategy=0, max_row_nnz=-1,
             precond=None, precond_ctx=None):
    """Cg with Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)).

    b: (n,) or (n, nrhs) float64 device tensor.  precond: None (Identity) or an
    integer address / ctypes function pointer of a gkomi_apply_fn.
    Returns dict(x, iterations, converged, residual_norm, baseline_norm, rel_residual)."""
    b2 = b.reshape(n, -1)
    nrhs = b2.shape[1]
    if mode == 1 and nrhs != 1:
        mode = 0
    if x is None:
        x = torch.zeros_like(b2)
    x2 = x.reshape(n, nrhs)
    assert b2.is_contiguous() and x2.is_contiguous()
    nnz = int(vals.numel())
    nbytes = gk.cg_workspace_bytes(n, nrhs)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = np.zeros(2 + 2 * nrhs, dtype=np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    pc = None
    if precond is not None:
        pc = ctypes.cast(precond, ctypes.c_void_p).value if not isinstance(precond, int) else precond
    gk.cg_solve_f64_i32(stream, n, nrhs, nnz, row_ptrs, col_idxs, vals, strategy, max_row_nnz,
                        pc, precond_ctx, b2, x2, max_iters, reduction, BASELINES[baseline], mode,
                        check_every, ws, nbytes, info)
    res = info[2::2].copy()
    base = info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]),
            "converged": bool(info[1]), "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}
