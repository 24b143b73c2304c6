"""Thin Python plumbing over the native solver drivers of the C ABI (device
buffers are torch tensors).  Mirrors the factory parameters of
gko::solver::Cg (include/ginkgo/core/solver/cg.hpp) that the hot path uses."""
import ctypes

import numpy as np
import torch

from ._lib import GkomiError

GKOMI_ENOTSUPPORTED = -2

APPLY_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
BASELINES = {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}


def _check_precond(precond, nrhs):
    """The native preconditioner contexts carry the column count of the vectors they
    are applied to (it is also their stride): a context generated for another count
    would read the n x nrhs row-major vectors as n x ctx.nrhs."""
    ctx = getattr(precond, "ctx", None)
    have = getattr(ctx, "nrhs", None)
    if have is not None and int(have) != int(nrhs):
        raise ValueError(f"preconditioner generated for nrhs = {int(have)}, applied to {int(nrhs)} right-hand sides: "
                         f"pass nrhs={int(nrhs)} to its generate call")


def cg_solve(gk, n, row_ptrs, col_idxs, vals, b, x=None, max_iters=1000, reduction=1e-10,
             baseline="rhs_norm", mode=1, check_every=16, strategy=0, max_row_nnz=-1,
             precond=None, precond_ctx=None):
    """Cg with Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)).

    b: (n,) or (n, nrhs) float64 device tensor.  precond: None (Identity) or an
    integer address / ctypes function pointer of a gkomi_apply_fn.
    Returns dict(x, iterations, converged, residual_norm, baseline_norm, rel_residual)."""
    b2 = b.reshape(n, -1) if n > 0 else b.reshape(0, b.shape[1] if b.dim() > 1 else 1)
    nrhs = b2.shape[1]
    _check_precond(precond, nrhs)
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
    if precond is not None and hasattr(precond, "ctx_ptr"):
        pc, precond_ctx = precond.fn, precond.ctx_ptr
    elif precond is not None:
        pc = ctypes.cast(precond, ctypes.c_void_p).value if not isinstance(precond, int) else precond
    gk.cg_solve_f64_i32(stream, n, nrhs, nnz, row_ptrs, col_idxs, vals, strategy, max_row_nnz,
                        pc, precond_ctx, b2, x2, max_iters, reduction, BASELINES[baseline], mode,
                        check_every, ws, nbytes, info)
    res = info[2::2].copy()
    base = info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]),
            "converged": bool(info[1]), "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}


def krylov_solve(gk, solver, n, row_ptrs, col_idxs, vals, b, x=None, max_iters=1000, reduction=1e-10,
                 baseline="rhs_norm", strategy=0, max_row_nnz=-1, precond=None, check_every=8, fused=False):
    """solver in {"bicgstab", "fcg", "cgs"}: {Bicgstab,Fcg,Cgs}::apply with
    Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)); precond: None or a Preconditioner.
    fused (bicgstab, one right-hand side): the 6-launch driver instead of the reference kernel sequence."""
    assert solver in ("bicgstab", "fcg", "cgs")
    b2 = b.reshape(n, -1) if n > 0 else b.reshape(0, b.shape[1] if b.dim() > 1 else 1)
    nrhs = b2.shape[1]
    _check_precond(precond, nrhs)
    if x is None:
        x = torch.zeros_like(b2)
    x2 = x.reshape(n, nrhs)
    assert b2.is_contiguous() and x2.is_contiguous()
    nnz = int(vals.numel())
    nbytes = gk.krylov_workspace_bytes(n, nrhs)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = np.zeros(2 + 2 * nrhs, dtype=np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    fn = precond.fn if precond is not None else None
    ctx = precond.ctx_ptr if precond is not None else None
    if fused:
        assert solver in ("bicgstab", "fcg", "cgs") and nrhs == 1
        getattr(gk, solver + "_solve_fused_f64_i32")(stream, n, nnz, row_ptrs, col_idxs, vals, strategy, max_row_nnz, fn, ctx, b2, x2,
                                        max_iters, reduction, BASELINES[baseline], check_every, ws, nbytes, info)
    else:
        getattr(gk, solver + "_solve_f64_i32")(stream, n, nrhs, nnz, row_ptrs, col_idxs, vals, strategy, max_row_nnz, fn,
                                               ctx, b2, x2, max_iters, reduction, BASELINES[baseline], check_every, ws,
                                               nbytes, info)
    res, base = info[2::2].copy(), info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]), "converged": bool(info[1]),
            "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}


def bicg_solve(gk, n, row_ptrs, col_idxs, vals, b, x=None, max_iters=1000, reduction=1e-10, baseline="rhs_norm",
               strategy=0, max_row_nnz=-1, precond=None, precond_t=None, check_every=8, transposed=None):
    """Bicg::apply: the transposed system matrix is built once here (csr::transpose)
    unless `transposed` = (row_ptrs, col_idxs, vals) is given; precond_t is the
    transposed preconditioner (pass the same object for a symmetric one)."""
    b2 = b.reshape(n, -1) if n > 0 else b.reshape(0, b.shape[1] if b.dim() > 1 else 1)
    nrhs = b2.shape[1]
    _check_precond(precond, nrhs)
    if x is None:
        x = torch.zeros_like(b2)
    x2 = x.reshape(n, nrhs)
    nnz = int(vals.numel())
    trp, tci, tv = transposed if transposed is not None else _transpose(gk, n, row_ptrs, col_idxs, vals)
    nbytes = gk.krylov_workspace_bytes(n, nrhs)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = np.zeros(2 + 2 * nrhs, dtype=np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    args = [p.fn if p is not None else None for p in (precond, precond_t)]
    ctxs = [p.ctx_ptr if p is not None else None for p in (precond, precond_t)]
    gk.bicg_solve_f64_i32(stream, n, nrhs, nnz, row_ptrs, col_idxs, vals, trp, tci, tv, strategy, max_row_nnz, args[0],
                          ctxs[0], args[1], ctxs[1], b2, x2, max_iters, reduction, BASELINES[baseline], check_every, ws,
                          nbytes, info)
    res, base = info[2::2].copy(), info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]), "converged": bool(info[1]),
            "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}


def ir_solve(gk, n, row_ptrs, col_idxs, vals, b, x=None, relaxation_factor=1.0, inner=None, max_iters=1000,
             reduction=1e-10, baseline="rhs_norm", strategy=0, max_row_nnz=-1):
    """Ir::apply with x as the initial guess; inner: None (Richardson) or a
    Preconditioner-like object whose apply approximates A^-1."""
    b2 = b.reshape(n, -1) if n > 0 else b.reshape(0, b.shape[1] if b.dim() > 1 else 1)
    nrhs = b2.shape[1]
    _check_precond(inner, nrhs)
    if x is None:
        x = torch.zeros_like(b2)
    x2 = x.reshape(n, nrhs)
    nnz = int(vals.numel())
    nbytes = gk.krylov_workspace_bytes(n, nrhs)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = np.zeros(2 + 2 * nrhs, dtype=np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    gk.ir_solve_f64_i32(stream, n, nrhs, nnz, row_ptrs, col_idxs, vals, strategy, max_row_nnz,
                        inner.fn if inner is not None else None, inner.ctx_ptr if inner is not None else None,
                        relaxation_factor, b2, x2, max_iters, reduction, BASELINES[baseline], ws, nbytes, info)
    res, base = info[2::2].copy(), info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]), "converged": bool(info[1]),
            "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}


def solve_op(gk, solver, matrix, b, x=None, max_iters=1000, reduction=1e-10, baseline="rhs_norm", precond=None,
             krylov_dim=100, check_every=8, fused=False):
    """Any solver in {"cg", "gmres", "bicgstab", "fcg", "cgs"} on a system matrix in
    any format (a gkomi.formats object): the *_solve_op_f64 drivers."""
    n = matrix.nrows
    b2 = b.reshape(n, -1) if n > 0 else b.reshape(0, b.shape[1] if b.dim() > 1 else 1)
    nrhs = b2.shape[1]
    _check_precond(precond, nrhs)
    if x is None:
        x = torch.zeros_like(b2)
    x2 = x.reshape(n, nrhs)
    cb = matrix.callback()
    info = np.zeros(2 + 2 * nrhs, dtype=np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    fn = precond.fn if precond is not None else None
    ctx = precond.ctx_ptr if precond is not None else None
    if solver == "cg" and fused:
        assert nrhs == 1
        nbytes = gk.cg_workspace_bytes(n, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        gk.cg_solve_fused_op_f64(stream, n, cb.fn, cb.ctx_ptr, fn, ctx, b2, x2, max_iters, reduction,
                                 BASELINES[baseline], check_every, ws, nbytes, info)
    elif solver == "cg":
        nbytes = gk.cg_workspace_bytes(n, nrhs)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        gk.cg_solve_op_f64(stream, n, nrhs, cb.fn, cb.ctx_ptr, fn, ctx, b2, x2, max_iters, reduction, BASELINES[baseline],
                           ws, nbytes, info)
    elif solver == "gmres":
        nbytes = gk.gmres_workspace_bytes(n, nrhs, krylov_dim)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        gk.gmres_solve_op_f64(stream, n, nrhs, cb.fn, cb.ctx_ptr, fn, ctx, b2, x2, krylov_dim, max_iters, reduction,
                              BASELINES[baseline], ws, nbytes, info)
    elif fused:
        assert solver in ("bicgstab", "fcg", "cgs") and nrhs == 1
        nbytes = gk.krylov_workspace_bytes(n, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        getattr(gk, solver + "_solve_fused_op_f64")(stream, n, cb.fn, cb.ctx_ptr, fn, ctx, b2, x2, max_iters, reduction,
                                       BASELINES[baseline], check_every, ws, nbytes, info)
    else:
        nbytes = gk.krylov_workspace_bytes(n, nrhs)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        getattr(gk, solver + "_solve_op_f64")(stream, n, nrhs, cb.fn, cb.ctx_ptr, fn, ctx, b2, x2, max_iters, reduction,
                                              BASELINES[baseline], check_every, ws, nbytes, info)
    res, base = info[2::2].copy(), info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]), "converged": bool(info[1]),
            "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}


class JacobiCtx(ctypes.Structure):
    """gkomi_jacobi_ctx (include/gkomi.h)."""
    _fields_ = [("n", ctypes.c_int64), ("nrhs", ctypes.c_int64), ("num_blocks", ctypes.c_int64),
                ("max_block_size", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("block_ptrs", ctypes.c_void_p), ("blocks", ctypes.c_void_p),
                ("block_precisions", ctypes.c_void_p)]


class IluCtx(ctypes.Structure):
    """gkomi_ilu_ctx (include/gkomi.h)."""
    _fields_ = [("n", ctypes.c_int64), ("nrhs", ctypes.c_int64),
                ("l_row_ptrs", ctypes.c_void_p), ("l_col_idxs", ctypes.c_void_p), ("l_vals", ctypes.c_void_p),
                ("u_row_ptrs", ctypes.c_void_p), ("u_col_idxs", ctypes.c_void_p), ("u_vals", ctypes.c_void_p),
                ("intermediate", ctypes.c_void_p), ("trs_workspace", ctypes.c_void_p),
                ("trs_workspace_bytes", ctypes.c_size_t), ("l_unit_diag", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("l_plan", ctypes.c_void_p), ("l_nslices", ctypes.c_int64), ("l_entries", ctypes.c_int64),
                ("l_max_deps", ctypes.c_int64),
                ("u_plan", ctypes.c_void_p), ("u_nslices", ctypes.c_int64), ("u_entries", ctypes.c_int64),
                ("u_max_deps", ctypes.c_int64),
                ("l_bricks", ctypes.c_void_p), ("l_bricks_plan", ctypes.c_void_p),
                ("u_bricks", ctypes.c_void_p), ("u_bricks_plan", ctypes.c_void_p)]


class Preconditioner:
    """A native gkomi_apply_fn + its context; keeps the device buffers alive."""

    def __init__(self, gk, name, ctx, keep):
        self.fn = ctypes.cast(getattr(gk._cdll, name), ctypes.c_void_p).value
        self.ctx = ctx
        self.ctx_ptr = ctypes.addressof(ctx)
        self.keep = keep

    def apply(self, b, x):
        """x = M^-1 b through the callback itself (what the solver drivers call); b, x: n x nrhs device tensors"""
        rc = APPLY_FN(self.fn)(self.ctx_ptr, torch.cuda.current_stream().cuda_stream, b.data_ptr(), x.data_ptr())
        if rc != 0:
            raise GkomiError("gkomi_apply_fn", rc, "preconditioner apply failed")
        return x


AUTODETECT = 0xff  # gko::precision_reduction::autodetect()


def jacobi_generate(gk, n, row_ptrs, col_idxs, vals, max_block_size=32, nrhs=1, storage_optimization=None,
                    accuracy=1e-1):
    """preconditioner::Jacobi::generate (core/preconditioner/jacobi.cpp:300-370):
    detect_blocks + generate, or extract/invert the diagonal for max_block_size 1.
    storage_optimization: None (fp64 blocks), a precision_reduction byte for all
    blocks (AUTODETECT = adaptive) or a sequence repeated over the blocks
    (jacobi::initialize_precisions, jacobi_kernels.cpp:485-493)."""
    dv = vals.device
    s = torch.cuda.current_stream().cuda_stream
    if max_block_size == 1:
        diag = torch.zeros(n, dtype=torch.float64, device=dv)
        gk.csr_extract_diagonal_f64_i32(s, n, row_ptrs, col_idxs, vals, diag)
        inv = torch.zeros_like(diag)
        gk.jacobi_invert_diagonal_f64(s, n, diag, inv)
        ctx = JacobiCtx(n, nrhs, n, 1, 0, 0, inv.data_ptr(), 0)
        return Preconditioner(gk, "gkomi_jacobi_apply_cb", ctx, (inv,))
    ptrs = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    nbd = torch.zeros(1, dtype=torch.int64, device=dv)
    nws = gk.jacobi_find_blocks_workspace_bytes(n)
    ws = torch.empty(nws, dtype=torch.uint8, device=dv)
    hn = ctypes.c_int64(0)
    gk.jacobi_find_blocks_i32(s, n, row_ptrs, col_idxs, max_block_size, ptrs, nbd, ws, nws, ctypes.addressof(hn))
    nb = int(hn.value)
    blocks = torch.zeros(max(gk.jacobi_storage_elements(max_block_size, nb), 1), dtype=torch.float64, device=dv)
    prec = cond = None
    if storage_optimization is None:
        gk.jacobi_generate_f64_i32(s, n, row_ptrs, col_idxs, vals, nb, max_block_size, ptrs, None, blocks)
    else:
        src = np.atleast_1d(np.asarray(storage_optimization, np.uint8))
        prec = torch.from_numpy(np.resize(src, max(nb, 1))).to(dv)
        cond = torch.zeros(max(nb, 1), dtype=torch.float64, device=dv)
        gk.jacobi_generate_adaptive_f64_i32(s, n, row_ptrs, col_idxs, vals, nb, max_block_size, ptrs, accuracy,
                                            cond, prec, blocks)
    ctx = JacobiCtx(n, nrhs, nb, max_block_size, 0, ptrs.data_ptr(), blocks.data_ptr(),
                    prec.data_ptr() if prec is not None else 0)
    p = Preconditioner(gk, "gkomi_jacobi_apply_cb", ctx, (ptrs, blocks, prec, cond))
    p.num_blocks, p.block_ptrs, p.blocks, p.block_precisions, p.conditioning = nb, ptrs, blocks, prec, cond
    return p


def jacobi_transpose(gk, pre):
    """Jacobi::transpose (core/preconditioner/jacobi.cpp): the preconditioner of A^T,
    every stored block transposed in its storage precision."""
    if not hasattr(pre, "blocks"):
        return pre   # scalar Jacobi (a diagonal) is its own transpose
    s = torch.cuda.current_stream().cuda_stream
    out = torch.zeros_like(pre.blocks)
    c = pre.ctx
    gk.jacobi_transpose_f64_i32(s, pre.num_blocks, c.max_block_size, pre.block_ptrs, pre.block_precisions, pre.blocks, out)
    ctx = JacobiCtx(c.n, c.nrhs, c.num_blocks, c.max_block_size, 0, pre.block_ptrs.data_ptr(), out.data_ptr(),
                    pre.block_precisions.data_ptr() if pre.block_precisions is not None else 0)
    t = Preconditioner(gk, "gkomi_jacobi_apply_cb", ctx, (pre.block_ptrs, out, pre.block_precisions, pre.conditioning))
    t.num_blocks, t.block_ptrs, t.blocks, t.block_precisions, t.conditioning = (
        pre.num_blocks, pre.block_ptrs, out, pre.block_precisions, pre.conditioning)
    return t


class TrsPlan:
    """solver::LowerTrs / UpperTrs after generate(): the analysed factor
    (gkomi_trs_analyse_{symbolic,numeric}) and its solve."""

    def __init__(self, gk, n, row_ptrs, col_idxs, vals, lower):
        self.gk, self.n, self.lower = gk, int(n), bool(lower)
        self.row_ptrs, self.col_idxs = row_ptrs, col_idxs
        dv = vals.device
        s = torch.cuda.current_stream().cuda_stream
        nb = gk.trs_symbolic_workspace_bytes(n)
        self.symbolic = torch.empty(max(nb, 8), dtype=torch.uint8, device=dv)
        out = (ctypes.c_int64 * 4)()
        gk.trs_analyse_symbolic_i32(s, n, row_ptrs, col_idxs, int(self.lower), self.symbolic, nb, ctypes.addressof(out))
        self.nslices, self.entries, self.nlevels, self.max_deps = (int(v) for v in out)
        self.plan_bytes = gk.trs_plan_bytes(self.nslices, self.entries)
        self.plan = torch.empty(max(self.plan_bytes, 8), dtype=torch.uint8, device=dv)
        self.refresh(vals)

    def refresh(self, vals):
        """new values, same sparsity pattern: numeric phase only"""
        s = torch.cuda.current_stream().cuda_stream
        self.vals = vals
        self.gk.trs_analyse_numeric_f64_i32(s, self.n, self.row_ptrs, self.col_idxs, vals, int(self.lower),
                                            self.symbolic, self.nslices, self.entries, self.nlevels, self.plan,
                                            self.plan_bytes)

    def solve(self, b, x, unit_diag=False):
        s = torch.cuda.current_stream().cuda_stream
        b2, x2 = b.reshape(self.n, -1), x.reshape(self.n, -1)
        self.gk.trs_solve_plan_f64(s, self.n, b2.shape[1], self.plan, self.nslices, self.entries, self.max_deps, int(unit_diag),
                                   b2, b2.stride(0), x2, x2.stride(0))
        return x

    def overrun(self):
        flag = ctypes.c_int(0)
        self.gk.trs_plan_check_overrun(torch.cuda.current_stream().cuda_stream, self.plan, ctypes.addressof(flag))
        return bool(flag.value)


class TrsBricks:
    """solver::LowerTrs / UpperTrs after generate() for factors of grid problems: the brick
    plan (gkomi_trs_bricks_*, csrc/trs_bricks.hip).  Raises GkomiError(GKOMI_ENOTSUPPORTED)
    when the factor is not stencil-shaped; callers then keep TrsPlan."""

    def __init__(self, gk, n, row_ptrs, col_idxs, vals, lower, brick_rows=0, threads=0, mode=0, handle=None):
        self.gk, self.n, self.lower = gk, int(n), bool(lower)
        self.row_ptrs, self.col_idxs = row_ptrs, col_idxs
        self.handle = ctypes.c_void_p(0)
        s = torch.cuda.current_stream().cuda_stream
        if handle is not None:
            self.handle = ctypes.c_void_p(handle)
        else:
            gk.trs_bricks_create_i32(s, n, row_ptrs, col_idxs, int(self.lower), int(brick_rows), int(threads), int(mode),
                                     ctypes.addressof(self.handle))
        self.levels_estimate = int(gk.trs_bricks_levels_estimate(self.handle.value))
        info = (ctypes.c_int64 * 8)()
        gk.trs_bricks_info(self.handle.value, ctypes.addressof(info))
        (self.nbricks, self.coarse_levels, self.nsteps, self.critical_steps, self.lds_bytes, self.width,
         self.threads, self.mode) = (int(v) for v in info)
        self.plan_bytes = gk.trs_bricks_plan_bytes(self.handle.value)
        self.plan = torch.empty(max(self.plan_bytes, 8), dtype=torch.uint8, device=vals.device)
        self.refresh(vals)

    def estimate_us(self, nlevels):
        """critical path of the pipelined solve: one LDS step per level of the factor + a memory hand-off per brick level"""
        return TRS_BRICK_STEP_US * nlevels + TRS_BRICK_HOP_US * self.coarse_levels

    def refresh(self, vals):
        s = torch.cuda.current_stream().cuda_stream
        self.vals = vals
        self.gk.trs_bricks_numeric_f64_i32(s, self.handle.value, self.row_ptrs, self.col_idxs, vals, self.plan,
                                           self.plan_bytes)

    def solve(self, b, x, unit_diag=False):
        s = torch.cuda.current_stream().cuda_stream
        b2, x2 = b.reshape(self.n, -1), x.reshape(self.n, -1)
        self.gk.trs_bricks_solve_f64(s, self.handle.value, self.plan, b2.shape[1], int(unit_diag), b2, b2.stride(0),
                                     x2, x2.stride(0))
        return x

    def overrun(self):
        flag = ctypes.c_int(0)
        self.gk.trs_bricks_check_overrun(torch.cuda.current_stream().cuda_stream, self.plan, ctypes.addressof(flag))
        return bool(flag.value)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h is not None and h.value:
            self.gk.trs_bricks_destroy(h.value)


# a level must hold this many rows on average for the level-scheduled solve to pay: below it
# (chains, narrow bands) the analysis-free kernel with its in-workgroup LDS hand-offs is faster
TRS_PLAN_MIN_ROWS_PER_LEVEL = 64


# per dependency level of the level plan / per level + per brick level of the pipelined brick plan (us,
# measured on the 108^3 and 1000^2 factors, profiles/r02_trs_bricks.md): what `ilu_from_factors` compares
TRS_LEVEL_US = 1.7
TRS_BRICK_STEP_US = 0.17
TRS_BRICK_HOP_US = 5.0


def ilu_from_factors(gk, n, L, U, nrhs=1, l_unit_diag=False, analyse=True, bricks=True, brick_rows=0):
    """preconditioner::Ilu over given CSR factors L = (row_ptrs, col_idxs, vals), U likewise.
    analyse: LowerTrs / UpperTrs::generate -- the dependency analysis of both factors (True / False / "force");
    a factor of a grid problem gets the brick plan (bricks=True; csrc/trs_bricks.hip), one whose levels are
    wide enough the level-scheduled kernel, a small one (<= 4096 rows) the single-workgroup solve of the same plan,
    anything else keeps the analysis-free solve.
    brick_rows: rows per brick of the brick plan (0: the library's default); bricks="force" takes the
    brick plan whenever the factor admits one, whatever the cost model says.
    The brick analysis runs on the device; a factor that takes the brick plan skips the level analysis (its
    level count is estimated from the box geometry the bricks found)."""
    dv = L[2].device
    inter = torch.zeros((n, nrhs), dtype=torch.float64, device=dv)
    nb = gk.trs_workspace_bytes()
    tws = torch.zeros(nb, dtype=torch.uint8, device=dv)
    plans = [None, None]
    brick_plans = [None, None]
    if analyse and n > 0:
        for i, (f, lower) in enumerate(((L, True), (U, False))):
            bk = None
            if bricks and n >= 2:
                try:
                    bk = TrsBricks(gk, n, f[0], f[1], f[2], lower, brick_rows=brick_rows)
                except GkomiError as e:
                    if e.code != GKOMI_ENOTSUPPORTED:
                        raise
            if bk is not None:
                nlevels = bk.levels_estimate
                # pipelined: about one step per level of the factor
                if nlevels > 16 and (bricks == "force" or gk.trs_prefer_bricks(n, nlevels, bk.coarse_levels)):
                    brick_plans[i] = bk
                    continue
                del bk
            plan = TrsPlan(gk, n, f[0], f[1], f[2], lower)
            # wide levels, or a small factor (one workgroup, x in LDS): gkomi_trs_use_plan, the rule of the shims and the mirror
            if analyse == "force" or gk.trs_use_plan(n, plan.nlevels, plan.max_deps):
                plans[i] = plan
    pl, pu = plans
    bl, bu = brick_plans
    ctx = IluCtx(n, nrhs, L[0].data_ptr(), L[1].data_ptr(), L[2].data_ptr(), U[0].data_ptr(), U[1].data_ptr(),
                 U[2].data_ptr(), inter.data_ptr(), tws.data_ptr(), nb, int(l_unit_diag), 0,
                 pl.plan.data_ptr() if pl else None, pl.nslices if pl else 0, pl.entries if pl else 0,
                 pl.max_deps if pl else -1,
                 pu.plan.data_ptr() if pu else None, pu.nslices if pu else 0, pu.entries if pu else 0,
                 pu.max_deps if pu else -1,
                 bl.handle.value if bl else None, bl.plan.data_ptr() if bl else None,
                 bu.handle.value if bu else None, bu.plan.data_ptr() if bu else None)
    p = Preconditioner(gk, "gkomi_ilu_apply_cb", ctx, (L, U, inter, tws, pl, pu, bl, bu))
    p.l_plan, p.u_plan, p.l_bricks, p.u_bricks = pl, pu, bl, bu
    return p


def gmres_solve(gk, n, row_ptrs, col_idxs, vals, b, x=None, krylov_dim=100, max_iters=1000, reduction=1e-10,
                baseline="rhs_norm", strategy=0, max_row_nnz=-1, precond=None):
    """Gmres with Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)).
    precond: None or a Preconditioner."""
    b2 = b.reshape(n, -1) if n > 0 else b.reshape(0, b.shape[1] if b.dim() > 1 else 1)
    nrhs = b2.shape[1]
    _check_precond(precond, nrhs)
    if x is None:
        x = torch.zeros_like(b2)
    x2 = x.reshape(n, nrhs)
    nnz = int(vals.numel())
    nbytes = gk.gmres_workspace_bytes(n, nrhs, krylov_dim)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = np.zeros(2 + 2 * nrhs, dtype=np.float64)
    stream = torch.cuda.current_stream().cuda_stream
    fn = precond.fn if precond is not None else None
    ctx = precond.ctx_ptr if precond is not None else None
    gk.gmres_solve_f64_i32(stream, n, nrhs, nnz, row_ptrs, col_idxs, vals, strategy, max_row_nnz, fn, ctx, b2, x2,
                           krylov_dim, max_iters, reduction, BASELINES[baseline], ws, nbytes, info)
    res, base = info[2::2].copy(), info[3::2].copy()
    return {"x": x2 if b.dim() > 1 else x2.reshape(n), "iterations": int(info[0]), "converged": bool(info[1]),
            "residual_norm": res, "baseline_norm": base,
            "rel_residual": float(np.max(res / np.where(base == 0, 1.0, base)))}


def _with_diagonal(gk, n, row_ptrs, col_idxs, vals):
    """factorization::add_diagonal_elements on a working copy (par_ilu.cpp:93-95)"""
    s = torch.cuda.current_stream().cuda_stream
    dv = vals.device
    rp = row_ptrs.clone()
    nb = gk.factorization_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dv)
    missing = ctypes.c_int64(0)
    gk.factorization_count_missing_diagonal_i32(s, n, n, rp, col_idxs, ws, nb, ctypes.addressof(missing))
    if not missing.value:
        return rp, col_idxs, vals
    nnz = int(vals.numel()) + missing.value
    nc = torch.zeros(nnz, dtype=torch.int32, device=dv)
    nv = torch.zeros(nnz, dtype=torch.float64, device=dv)
    gk.factorization_add_diagonal_elements_f64_i32(s, n, n, rp, col_idxs, vals, nc, nv, ws)
    return rp, nc, nv


def _transpose(gk, n, rp, ci, v):
    s = torch.cuda.current_stream().cuda_stream
    tb = gk.csr_transpose_workspace_bytes(n)
    tws = torch.empty(tb, dtype=torch.uint8, device=v.device)
    trp = torch.zeros(n + 1, dtype=torch.int32, device=v.device)
    tc, tv = torch.zeros_like(ci), torch.zeros_like(v)
    gk.csr_transpose_f64_i32(s, n, n, int(v.numel()), rp, ci, v, trp, tc, tv, tws, tb)
    return trp, tc, tv


def par_ilu_generate(gk, n, row_ptrs, col_idxs, vals, iterations=0, nrhs=1):
    """preconditioner::Ilu over factorization::ParIlu (core/factorization/par_ilu.cpp:74-163):
    returns the Preconditioner (L^-1 then U^-1); .L / .U hold the factors."""
    s = torch.cuda.current_stream().cuda_stream
    dv = vals.device
    rp, ci, v = _with_diagonal(gk, n, row_ptrs, col_idxs, vals)
    nnz = int(v.numel())
    lrp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    urp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    sb = gk.prefix_sum_workspace_bytes(n + 1)
    sws = torch.empty(max(sb, 8), dtype=torch.uint8, device=dv)
    gk.factorization_initialize_row_ptrs_l_u_i32(s, n, rp, ci, lrp, urp, sws, sb)
    lnnz, unnz = int(lrp[n].item()), int(urp[n].item())
    lc, lv = torch.zeros(lnnz, dtype=torch.int32, device=dv), torch.zeros(lnnz, dtype=torch.float64, device=dv)
    uc, uv = torch.zeros(unnz, dtype=torch.int32, device=dv), torch.zeros(unnz, dtype=torch.float64, device=dv)
    gk.factorization_initialize_l_u_f64_i32(s, n, rp, ci, v, lrp, lc, lv, urp, uc, uv)
    utrp, utc, utv = _transpose(gk, n, urp, uc, uv)
    rows = torch.zeros(max(nnz, 1), dtype=torch.int32, device=dv)
    gk.convert_ptrs_to_idxs_i32(s, rp, n, rows)
    gk.par_ilu_compute_l_u_factors_f64_i32(s, iterations, nnz, rows, ci, v, lrp, lc, lv, utrp, utc, utv)
    U = _transpose(gk, n, utrp, utc, utv)
    p = ilu_from_factors(gk, n, (lrp, lc, lv), U, nrhs=nrhs)
    p.L, p.U = (lrp, lc, lv), U
    return p


def par_ic_generate(gk, n, row_ptrs, col_idxs, vals, iterations=0, nrhs=1):
    """preconditioner::Ic over factorization::ParIc (core/factorization/par_ic.cpp:70-145):
    L^-1 then L^-T through the triangular solves; .L / .Lt hold the factors."""
    s = torch.cuda.current_stream().cuda_stream
    dv = vals.device
    rp, ci, v = _with_diagonal(gk, n, row_ptrs, col_idxs, vals)
    lrp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    sb = gk.prefix_sum_workspace_bytes(n + 1)
    sws = torch.empty(max(sb, 8), dtype=torch.uint8, device=dv)
    gk.factorization_initialize_row_ptrs_l_i32(s, n, rp, ci, lrp, sws, sb)
    lnnz = int(lrp[n].item())
    lc, lv = torch.zeros(lnnz, dtype=torch.int32, device=dv), torch.zeros(lnnz, dtype=torch.float64, device=dv)
    gk.factorization_initialize_l_f64_i32(s, n, rp, ci, v, lrp, lc, lv, 0)
    a_vals = lv.clone()
    rows = torch.zeros(max(lnnz, 1), dtype=torch.int32, device=dv)
    gk.convert_ptrs_to_idxs_i32(s, lrp, n, rows)
    gk.par_ic_init_factor_f64_i32(s, n, lrp, lc, lv)
    gk.par_ic_compute_factor_f64_i32(s, iterations, lnnz, rows, a_vals, lrp, lc, lv)
    Lt = _transpose(gk, n, lrp, lc, lv)
    p = ilu_from_factors(gk, n, (lrp, lc, lv), Lt, nrhs=nrhs)
    p.L, p.Lt = (lrp, lc, lv), Lt
    return p
