"""Helpers for the GPU parity tests: torch is only device memory + streams."""
import numpy as np
import torch


def dev(a, device="cuda:0"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def host(t):
    return t.detach().cpu().numpy()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def sync():
    torch.cuda.synchronize()


class DevCsr:
    def __init__(self, nrows, ncols, row_ptrs, col_idxs, vals):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.row_ptrs = dev(np.asarray(row_ptrs, np.int32))
        self.col_idxs = dev(np.asarray(col_idxs, np.int32))
        self.vals = dev(np.asarray(vals, np.float64))
        self.nnz = int(np.asarray(row_ptrs)[-1]) if nrows > 0 else 0
        self.max_row_nnz = int(np.max(np.diff(row_ptrs))) if nrows > 0 else 0


def csr_apply(gk, A, b, c=None, alpha=None, beta=None, strategy=0, hint=None):
    """c = A b  or  c = alpha A b + beta c through the C ABI.  b, c: 2-D torch."""
    nrhs = b.shape[1]
    if c is None:
        c = torch.full((A.nrows, nrhs), float("nan"), dtype=torch.float64, device=b.device)
    al = dev(np.array([alpha], np.float64)) if alpha is not None else None
    be = dev(np.array([beta], np.float64)) if beta is not None else None
    gk.csr_spmv_f64_i32(stream_ptr(), A.nrows, A.ncols, nrhs, A.nnz, A.row_ptrs, A.col_idxs, A.vals,
                        b, b.stride(0), c, c.stride(0), al, be, strategy,
                        A.max_row_nnz if hint is None else hint)
    return c


def make_srow(gk, A, tile=None):
    """Csr::make_srow: the tile start rows of the nonzero-split kernel."""
    tile = int(gk.csr_srow_tile()) if tile is None else tile
    n = int(gk.csr_srow_entries(A.nnz, tile))
    srow = torch.full((n,), -1, dtype=torch.int32, device=A.vals.device)
    gk.csr_make_srow_i32(stream_ptr(), A.nrows, A.nnz, A.row_ptrs, tile, srow, n)
    return srow, tile


def csr_apply_srow(gk, A, b, srow, tile, c=None, alpha=None, beta=None, strategy=0, hint=None):
    """gkomi_csr_spmv_srow_f64_i32: the apply of a matrix that carries its srow."""
    nrhs = b.shape[1]
    if c is None:
        c = torch.full((A.nrows, nrhs), float("nan"), dtype=torch.float64, device=b.device)
    al = dev(np.array([alpha], np.float64)) if alpha is not None else None
    be = dev(np.array([beta], np.float64)) if beta is not None else None
    gk.csr_spmv_srow_f64_i32(stream_ptr(), A.nrows, A.ncols, nrhs, A.nnz, A.row_ptrs, A.col_idxs, A.vals,
                             b, b.stride(0), c, c.stride(0), al, be, strategy,
                             A.max_row_nnz if hint is None else hint, srow, tile)
    return c
