"""gko::matrix::{Csr, Coo, Ell, Sellp, Hybrid} over the C ABI for Python callers
(tests, bench, tools/benchmark_*.py): device arrays are torch tensors, every
conversion and apply is a gkomi_* kernel -- no CPU fallback.  Conversions
follow core/matrix/csr.cpp:257-405 (the sizes the reference reads back with
exec->copy_val_to_host are read back here too)."""
import ctypes

import numpy as np
import torch

from ._lib import GkomiError

ENOTSUPPORTED = -2  # GKOMI_ENOTSUPPORTED of include/gkomi.h
I32, I64, F64, U8 = torch.int32, torch.int64, torch.float64, torch.uint8


def _ctx_type(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": fields})


_i64, _ptr = ctypes.c_int64, ctypes.c_void_p
Csr64Ctx = _ctx_type("Csr64Ctx", [("nrows", _i64), ("ncols", _i64), ("nnz", _i64), ("row_ptrs", _ptr), ("col_idxs", _ptr),
                                  ("vals", _ptr), ("strategy", _i64), ("max_row_nnz_hint", _i64), ("srow", _ptr),
                                  ("srow_tile", _i64)])
CsrCtx = _ctx_type("CsrCtx", [("nrows", _i64), ("ncols", _i64), ("nnz", _i64), ("row_ptrs", _ptr), ("col_idxs", _ptr),
                              ("vals", _ptr), ("strategy", _i64), ("max_row_nnz_hint", _i64), ("srow", _ptr),
                              ("srow_tile", _i64)])
EllCtx = _ctx_type("EllCtx", [("nrows", _i64), ("ncols", _i64), ("num_stored_per_row", _i64), ("stride", _i64),
                              ("col_idxs", _ptr), ("vals", _ptr)])
SellpCtx = _ctx_type("SellpCtx", [("nrows", _i64), ("ncols", _i64), ("slice_size", _i64), ("slice_sets", _ptr),
                                  ("slice_lengths", _ptr), ("col_idxs", _ptr), ("vals", _ptr)])
CooCtx = _ctx_type("CooCtx", [("nrows", _i64), ("ncols", _i64), ("nnz", _i64), ("row_idxs", _ptr), ("col_idxs", _ptr),
                              ("vals", _ptr)])
HybridCtx = _ctx_type("HybridCtx", [("nrows", _i64), ("ncols", _i64), ("ell_num_stored_per_row", _i64),
                                    ("ell_stride", _i64), ("ell_col_idxs", _ptr), ("ell_vals", _ptr), ("coo_nnz", _i64),
                                    ("coo_row_idxs", _ptr), ("coo_col_idxs", _ptr), ("coo_vals", _ptr)])


class MatrixCallback:
    """(gkomi_matrix_apply_fn, context) of a format object for the *_solve_op_f64 drivers"""

    def __init__(self, gk, symbol, ctx, owner):
        self.fn = ctypes.cast(getattr(gk._cdll, symbol), ctypes.c_void_p).value
        self.ctx = ctx
        self.ctx_ptr = ctypes.addressof(ctx)
        self.owner = owner  # keeps the device arrays alive


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _scalar(dev, x):
    return None if x is None else torch.tensor([float(x)], dtype=F64, device=dev)


class Csr:
    name = "csr"

    def __init__(self, gk, nrows, ncols, row_ptrs, col_idxs, vals, strategy=0, split=True):
        """split=False: no srow -- the matrix is cut by rows (the kernel of a Csr without srow_)"""
        self.gk, self.nrows, self.ncols = gk, int(nrows), int(ncols)
        self.split = bool(split)
        self.row_ptrs, self.col_idxs, self.vals = row_ptrs, col_idxs, vals
        self.nnz = int(vals.numel())
        self.strategy = strategy
        self._max_row_nnz = None
        self._srow = None
        self.srow_tile = 0
        self._gather_flags = None
        self.gather_footprint = None
        self._colpart = None       # (handle, plan tensor) of the column-partitioned copy, False = not worthwhile

    def __del__(self):
        if getattr(self, "_colpart", None):
            try:
                self.gk.csr_colpart_destroy(self._colpart[0])
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

    @classmethod
    def from_host(cls, gk, nrows, ncols, row_ptrs, col_idxs, vals, device="cuda:0", strategy=0, split=True):
        d = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(device)
        return cls(gk, nrows, ncols, d(row_ptrs, np.int32), d(col_idxs, np.int32), d(vals, np.float64), strategy, split)

    @classmethod
    def from_triplets(cls, gk, nrows, ncols, rows, cols, vals, sum_duplicates=True, strategy=0):
        """Csr::read(device_matrix_data): sort (+ sum duplicates) on the device, then idxs -> ptrs"""
        s = _stream(vals)
        rows, cols, vals = rows.clone(), cols.clone(), vals.clone()
        nnz = int(vals.numel())
        nb = gk.matrix_data_workspace_bytes(nnz)
        ws = torch.empty(max(nb, 8), dtype=U8, device=vals.device)
        gk.matrix_data_sort_row_major_f64_i32(s, nnz, rows, cols, vals, ws, nb)
        if sum_duplicates and nnz:
            orr, oc, ov = torch.empty_like(rows), torch.empty_like(cols), torch.empty_like(vals)
            kept = ctypes.c_int64(0)
            gk.matrix_data_sum_duplicates_f64_i32(s, nnz, rows, cols, vals, orr, oc, ov, ws, nb, ctypes.addressof(kept))
            rows, cols, vals = orr[:kept.value], oc[:kept.value], ov[:kept.value]
            nnz = kept.value
        ptrs = torch.zeros(nrows + 1, dtype=I32, device=vals.device)
        pb = gk.prefix_sum_workspace_bytes(nrows + 1)
        pws = torch.empty(max(pb, 8), dtype=U8, device=vals.device)
        gk.convert_idxs_to_ptrs_i32(s, rows, nnz, nrows, ptrs, pws, pb)
        return cls(gk, nrows, ncols, ptrs, cols.contiguous(), vals.contiguous(), strategy)

    def storage_bytes(self):
        return 4 * (self.nrows + 1) + 12 * self.nnz

    def max_row_nnz(self):
        if self._max_row_nnz is None:
            mx = torch.zeros(1, dtype=I32, device=self.vals.device)
            self.gk.csr_max_row_nnz_i32(_stream(self.vals), self.nrows, self.row_ptrs, mx)
            self._max_row_nnz = int(mx.item())
        return self._max_row_nnz

    def srow(self):
        """Csr::make_srow (csr.hpp:1139-1157): the tile start rows of the nonzero-split kernel, built once"""
        if not self.split or self.nnz < 2:
            return None
        if self._srow is None:
            gk = self.gk
            self.srow_tile = int(gk.csr_srow_tile_for(self.nnz))
            ne = int(gk.csr_srow_entries(self.nnz, self.srow_tile))
            self._srow = torch.zeros(max(ne, 1), dtype=I32, device=self.vals.device)
            try:
                gk.csr_make_srow_i32(_stream(self.vals), self.nrows, self.nnz, self.row_ptrs, self.srow_tile,
                                     self._srow, ne)
            except GkomiError as e:  # too large for the split kernel: the row-cut kernels serve it
                if e.code != ENOTSUPPORTED:
                    raise
                self._srow, self.srow_tile = False, 0
        return self._srow if self._srow is not False else None

    def gather_flags(self):
        """the column-pattern statistic of the strategy object (one-time, like max_row_nnz and srow):
        GKOMI_CSR_COLBLOCK when the gathers of b overflow an XCD's L2 (gkomi_csr_analyse_gather_i32)"""
        if self._gather_flags is None:
            self._gather_flags = 0
            if self.nnz > 0 and (self.strategy & 0xff) in (0, 3) and not (self.strategy >> 30) & 1 and hasattr(self.gk, "csr_analyse_gather_i32"):
                scratch = torch.zeros(2, dtype=F64, device=self.vals.device)
                flags, foot = ctypes.c_int(0), ctypes.c_int64(0)
                self.gk.csr_analyse_gather_i32(_stream(self.vals), self.ncols, self.nnz, self.col_idxs, scratch,
                                               ctypes.addressof(flags), ctypes.addressof(foot))
                self._gather_flags, self.gather_footprint = int(flags.value), int(foot.value)
        return self._gather_flags

    PARTITIONED = 1 << 29   # strategy bit of this mirror: apply through the column-partitioned copy ("csrp")

    def colpart(self, nb=None):
        """The column-partitioned copy (gkomi_csr_colpart_*, the analysis of the "csrp" strategy): built once, like
        srow; None when the shape does not pay (gkomi_csr_colpart_blocks_for) and no block count is forced.  The copy
        holds VALUES: call values_changed() after writing to self.vals."""
        if self._colpart is None:
            gk = self.gk
            auto = nb is None
            if auto:
                # the shape must fit (blocks_for) AND the column pattern must be scattered (the one-time gather statistic:
                # a stencil or banded matrix gathers from L2 already and would only pay for the partial sums)
                self.gather_flags()
                scattered = (self.gather_footprint or 0) > (3 << 20)   # a tile's gathers range over more of b than an L2 keeps
                nb = int(gk.csr_colpart_blocks_for(self.nrows, self.ncols, self.nnz)) if scattered else 0
            nb = int(nb)
            if nb == 0:
                self._colpart = False
            else:
                ask = 0 if auto else nb     # 0: the analysis times blocks_for's count and half of it, keeps the faster
                nbytes = int(gk.csr_colpart_plan_bytes(self.nrows, self.nnz, ask))
                plan = torch.empty(nbytes, dtype=U8, device=self.vals.device)
                h = ctypes.c_void_p(0)
                try:
                    gk.csr_colpart_create_f64_i32(_stream(self.vals), self.nrows, self.ncols, self.nnz, self.row_ptrs, self.col_idxs,
                                                  self.vals, ask, plan, nbytes, ctypes.addressof(h))
                except GkomiError as e:   # the analysis timed the copy against the matrix's own kernel: it does not pay
                    if e.code != ENOTSUPPORTED or not auto:
                        raise
                    self._colpart = False
                    return None
                self._colpart = (h.value, plan)
        return self._colpart or None

    def values_changed(self):
        if self._colpart:
            self.gk.csr_colpart_refresh_f64(_stream(self.vals), self._colpart[0], self.vals)

    def apply(self, b, x, alpha=None, beta=None):
        dv = self.vals.device
        if (self.strategy & self.PARTITIONED) and b.shape[1] == 1 and self.colpart() is not None:
            self.gk.csr_colpart_spmv_f64(_stream(self.vals), self._colpart[0], b, b.stride(0), x, x.stride(0),
                                         _scalar(dv, alpha), _scalar(dv, beta))
            return x
        self.gk.csr_spmv_srow_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.nnz, self.row_ptrs,
                                      self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0), _scalar(dv, alpha),
                                      _scalar(dv, beta), (self.strategy & ~(3 << 29)) | self.gather_flags(), self.max_row_nnz(),
                                      self.srow(), self.srow_tile)
        return x

    def callback(self):
        if (self.strategy & self.PARTITIONED) and self.colpart() is not None:
            # the copy as the system matrix of the solver drivers (the matrix is const while a solve runs)
            cb = MatrixCallback.__new__(MatrixCallback)
            cb.fn = ctypes.cast(getattr(self.gk._cdll, "gkomi_csr_colpart_matrix_apply_cb"), ctypes.c_void_p).value
            cb.ctx, cb.ctx_ptr, cb.owner = None, self._colpart[0], self
            return cb
        srow = self.srow()
        ctx = CsrCtx(self.nrows, self.ncols, self.nnz, self.row_ptrs.data_ptr(), self.col_idxs.data_ptr(),
                     self.vals.data_ptr(), self.strategy & ~(3 << 29), self.max_row_nnz(),
                     srow.data_ptr() if srow is not None else None, self.srow_tile)
        return MatrixCallback(self.gk, "gkomi_csr_matrix_apply_cb", ctx, self)

    def row_idxs(self):
        rows = torch.zeros(max(self.nnz, 1), dtype=I32, device=self.vals.device)
        self.gk.convert_ptrs_to_idxs_i32(_stream(self.vals), self.row_ptrs, self.nrows, rows)
        return rows

    # benchmark/utils/formats.hpp:272-290: "csr" = automatical, "csri" = load_balance,
    # "csrm" = merge_path, "csrc" = classical, "csrs" = sparselib (served by the automatic
    # kernel here, like Csr::sparselib in the C++ mirror); GKOMI_CSR_* codes of include/gkomi.h
    CSR_STRATEGIES = {"csr": 0, "csrs": 0, "csrm": 1, "csrc": 2, "csri": 3, "csri_serial": 3 | (1 << 8),
                      "csr_1pass": 0 | (1 << 30), "csri_1pass": 3 | (1 << 30),   # bit 30: no column-pattern analysis (A/B)
                      "csrp": 0 | (1 << 29)}   # bit 29: the column-partitioned copy where it pays (else the automatic kernels)

    def to(self, fmt, **kw):
        if fmt in self.CSR_STRATEGIES:
            code = self.CSR_STRATEGIES[fmt]
            if code == self.strategy:
                return self
            return Csr(self.gk, self.nrows, self.ncols, self.row_ptrs, self.col_idxs, self.vals, code, self.split)
        return {"coo": Coo, "ell": Ell, "sellp": Sellp, "hybrid": Hybrid}[fmt].from_csr(self, **kw)


class Csr64:
    """gko::matrix::Csr<double, int64>: the index type of matrices with more than 2^31 nonzeros
    (gkomi_csr_*_i64).  Same object model as Csr: carries its srow and its longest row; apply and the
    solver callback run the nonzero-split kernel (automatic strategy)."""
    name = "csr64"

    def __init__(self, gk, nrows, ncols, row_ptrs, col_idxs, vals, strategy=0, split=True):
        assert row_ptrs.dtype == I64 and col_idxs.dtype == I64
        self.gk, self.nrows, self.ncols = gk, int(nrows), int(ncols)
        self.row_ptrs, self.col_idxs, self.vals = row_ptrs, col_idxs, vals
        self.nnz = int(vals.numel())
        self.strategy, self.split = strategy, bool(split)
        self._max_row_nnz = None
        self._srow = None
        self.srow_tile = 0

    @classmethod
    def from_host(cls, gk, nrows, ncols, row_ptrs, col_idxs, vals, device="cuda:0", strategy=0, split=True):
        d = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(device)
        return cls(gk, nrows, ncols, d(row_ptrs, np.int64), d(col_idxs, np.int64), d(vals, np.float64), strategy, split)

    @classmethod
    def poisson_3d_7pt(cls, gk, g, device="cuda:0"):
        """BASELINE config 5's matrix for a g^3 grid, written on the device (gkomi_diag_poisson3d_7pt_f64_i64)"""
        n, nnz = g ** 3, 7 * g ** 3 - 6 * g * g
        rp = torch.empty(n + 1, dtype=I64, device=device)
        ci = torch.empty(nnz, dtype=I64, device=device)
        v = torch.empty(nnz, dtype=F64, device=device)
        gk.diag_poisson3d_7pt_f64_i64(_stream(v), g, rp, ci, v)
        return cls(gk, n, n, rp, ci, v)

    def storage_bytes(self):
        return 8 * (self.nrows + 1) + 16 * self.nnz

    def max_row_nnz(self):
        if self._max_row_nnz is None:
            mx = torch.zeros(1, dtype=I64, device=self.vals.device)
            self.gk.csr_max_row_nnz_i64(_stream(self.vals), self.nrows, self.row_ptrs, mx)
            self._max_row_nnz = int(mx.item())
        return self._max_row_nnz

    def srow(self):
        if not self.split or self.nnz < 2:
            return None
        if self._srow is None:
            gk = self.gk
            self.srow_tile = int(gk.csr_srow_tile_for(self.nnz))
            ne = int(gk.csr_srow_entries(self.nnz, self.srow_tile))
            self._srow = torch.zeros(max(ne, 1), dtype=I64, device=self.vals.device)
            gk.csr_make_srow_i64(_stream(self.vals), self.nrows, self.nnz, self.row_ptrs, self.srow_tile, self._srow, ne)
        return self._srow

    def apply(self, b, x, alpha=None, beta=None):
        dv = self.vals.device
        self.gk.csr_spmv_srow_f64_i64(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.nnz, self.row_ptrs,
                                      self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0), _scalar(dv, alpha),
                                      _scalar(dv, beta), self.strategy, self.max_row_nnz(), self.srow(), self.srow_tile)
        return x

    def callback(self):
        srow = self.srow()
        ctx = Csr64Ctx(self.nrows, self.ncols, self.nnz, self.row_ptrs.data_ptr(), self.col_idxs.data_ptr(),
                       self.vals.data_ptr(), self.strategy, self.max_row_nnz(),
                       srow.data_ptr() if srow is not None else None, self.srow_tile)
        return MatrixCallback(self.gk, "gkomi_csr64_matrix_apply_cb", ctx, self)


class Coo:
    name = "coo"

    def __init__(self, csr, row_idxs):
        self.gk, self.nrows, self.ncols, self.nnz = csr.gk, csr.nrows, csr.ncols, csr.nnz
        self.row_idxs, self.col_idxs, self.vals = row_idxs, csr.col_idxs, csr.vals

    @classmethod
    def from_csr(cls, csr):
        return cls(csr, csr.row_idxs())

    def storage_bytes(self):
        return 16 * self.nnz

    def callback(self):
        ctx = CooCtx(self.nrows, self.ncols, self.nnz, self.row_idxs.data_ptr(), self.col_idxs.data_ptr(),
                     self.vals.data_ptr())
        return MatrixCallback(self.gk, "gkomi_coo_matrix_apply_cb", ctx, self)

    def _sorted_workspace(self, nrhs):
        """workspace of the atomic-free kernels when row_idxs is sorted (analysed once), else None"""
        gk = self.gk
        nb = gk.coo_sorted_workspace_bytes(self.nnz, nrhs)
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < nb:
            ws = self._ws = torch.empty(max(nb, 16), dtype=U8, device=self.vals.device)
        if getattr(self, "_sorted", None) is None:
            flag, longest = ctypes.c_int(0), ctypes.c_int64(0)
            gk.coo_analyse_rows_i32(_stream(self.vals), self.nnz, self.row_idxs, ws, ws.numel(), ctypes.addressof(flag),
                                    ctypes.addressof(longest))
            aligned = self.vals.data_ptr() % 16 == 0 and self.row_idxs.data_ptr() % 8 == 0 and \
                self.col_idxs.data_ptr() % 8 == 0
            self._sorted = bool(flag.value) and aligned
            self.max_row_nnz = int(longest.value)   # capped at 65
        return ws if self._sorted else None

    def apply(self, b, x, alpha=None, beta=None, sorted_rows=None, hint=None):
        """sorted_rows: None = use the atomic-free kernels when the rows are sorted, False = never;
        hint: max_row_nnz_hint of the sorted entry (default: what the analysis found)"""
        dv = self.vals.device
        ws = self._sorted_workspace(b.shape[1]) if sorted_rows is not False else None
        if ws is not None:
            self.gk.coo_spmv_sorted_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.nnz,
                                            self.row_idxs, self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0),
                                            _scalar(dv, alpha), _scalar(dv, beta),
                                            self.max_row_nnz if hint is None else hint, ws, ws.numel())
            return x
        self.gk.coo_spmv_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.nnz, self.row_idxs,
                                 self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0), _scalar(dv, alpha),
                                 _scalar(dv, beta))
        return x

    def apply2(self, b, x, alpha=None, sorted_rows=None, hint=None):
        """x += [alpha] A b (Coo::apply2)"""
        dv = self.vals.device
        ws = self._sorted_workspace(b.shape[1]) if sorted_rows is not False else None
        if ws is not None:
            self.gk.coo_spmv2_sorted_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.nnz,
                                             self.row_idxs, self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0),
                                             _scalar(dv, alpha), self.max_row_nnz if hint is None else hint, ws,
                                             ws.numel())
            return x
        self.gk.coo_spmv2_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.nnz, self.row_idxs,
                                  self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0), _scalar(dv, alpha))
        return x


class Ell:
    name = "ell"

    def __init__(self, gk, nrows, ncols, k, stride, col_idxs, vals):
        self.gk, self.nrows, self.ncols, self.k, self.stride = gk, nrows, ncols, k, stride
        self.col_idxs, self.vals = col_idxs, vals

    @classmethod
    def from_csr(cls, csr, stride=None):
        k = csr.max_row_nnz()
        stride = csr.nrows if stride is None else stride
        dv = csr.vals.device
        cols = torch.full((max(stride * k, 1),), -1, dtype=I32, device=dv)
        vals = torch.zeros(max(stride * k, 1), dtype=F64, device=dv)
        csr.gk.csr_convert_to_ell_f64_i32(_stream(vals), csr.nrows, csr.row_ptrs, csr.col_idxs, csr.vals, k, stride,
                                          cols, vals)
        return cls(csr.gk, csr.nrows, csr.ncols, k, stride, cols, vals)

    def storage_bytes(self):
        return 12 * self.stride * self.k

    def callback(self):
        ctx = EllCtx(self.nrows, self.ncols, self.k, self.stride, self.col_idxs.data_ptr(), self.vals.data_ptr())
        return MatrixCallback(self.gk, "gkomi_ell_matrix_apply_cb", ctx, self)

    def apply(self, b, x, alpha=None, beta=None):
        dv = self.vals.device
        self.gk.ell_spmv_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.k, self.stride,
                                 self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0), _scalar(dv, alpha),
                                 _scalar(dv, beta))
        return x


class Sellp:
    name = "sellp"

    def __init__(self, gk, nrows, ncols, slice_size, sets, lens, col_idxs, vals):
        self.gk, self.nrows, self.ncols, self.slice_size = gk, nrows, ncols, slice_size
        self.sets, self.lens, self.col_idxs, self.vals = sets, lens, col_idxs, vals

    @classmethod
    def from_csr(cls, csr, slice_size=64, stride_factor=1):
        gk, dv = csr.gk, csr.vals.device
        s = _stream(csr.vals)
        nsl = (csr.nrows + slice_size - 1) // slice_size
        sets = torch.zeros(nsl + 1, dtype=I64, device=dv)
        lens = torch.zeros(max(nsl, 1), dtype=I64, device=dv)
        nb = gk.prefix_sum_workspace_bytes(nsl + 1)
        ws = torch.empty(max(nb, 8), dtype=U8, device=dv)
        gk.sellp_compute_slice_sets_i32(s, csr.row_ptrs, csr.nrows, slice_size, stride_factor, sets, lens, ws, nb)
        total = int(sets[nsl].item()) * slice_size
        cols = torch.full((max(total, 1),), -1, dtype=I32, device=dv)
        vals = torch.zeros(max(total, 1), dtype=F64, device=dv)
        gk.csr_convert_to_sellp_f64_i32(s, csr.nrows, csr.row_ptrs, csr.col_idxs, csr.vals, slice_size, sets, lens,
                                        cols, vals)
        return cls(gk, csr.nrows, csr.ncols, slice_size, sets, lens, cols, vals)

    def storage_bytes(self):
        return 12 * int(self.vals.numel()) + 16 * int(self.lens.numel()) + 8

    def callback(self):
        ctx = SellpCtx(self.nrows, self.ncols, self.slice_size, self.sets.data_ptr(), self.lens.data_ptr(),
                       self.col_idxs.data_ptr(), self.vals.data_ptr())
        return MatrixCallback(self.gk, "gkomi_sellp_matrix_apply_cb", ctx, self)

    def apply(self, b, x, alpha=None, beta=None):
        dv = self.vals.device
        self.gk.sellp_spmv_f64_i32(_stream(self.vals), self.nrows, self.ncols, b.shape[1], self.slice_size, self.sets,
                                   self.lens, self.col_idxs, self.vals, b, b.stride(0), x, x.stride(0),
                                   _scalar(dv, alpha), _scalar(dv, beta))
        return x


class Hybrid:
    """strategy kinds as GKOMI_HYBRID_* (include/gkomi.h); 4 = automatic"""
    name = "hybrid"

    def __init__(self, gk, nrows, ncols, ell_lim, ell_cols, ell_vals, coo_nnz, coo_rows, coo_cols, coo_vals):
        self.gk, self.nrows, self.ncols, self.ell_lim = gk, nrows, ncols, ell_lim
        self.ell_cols, self.ell_vals = ell_cols, ell_vals
        self.coo_nnz, self.coo_rows, self.coo_cols, self.coo_vals = coo_nnz, coo_rows, coo_cols, coo_vals

    @classmethod
    def from_csr(cls, csr, kind=4, percent=0.8, ratio=1e-4, num_columns=0):
        gk, dv = csr.gk, csr.vals.device
        s = _stream(csr.vals)
        n = csr.nrows
        res = ctypes.c_int64(0)
        gk.hybrid_ell_width_i32(s, csr.row_ptrs, n, kind, percent, ratio, num_columns, ctypes.addressof(res))
        ell_lim = min(int(res.value), csr.ncols)
        crp = torch.zeros(n + 1, dtype=I64, device=dv)
        nb = gk.prefix_sum_workspace_bytes(n + 1)
        ws = torch.empty(max(nb, 8), dtype=U8, device=dv)
        gk.hybrid_compute_coo_row_ptrs_i32(s, csr.row_ptrs, n, ell_lim, crp, ws, nb)
        coo_nnz = int(crp[n].item())
        ell_cols = torch.full((max(ell_lim * n, 1),), -1, dtype=I32, device=dv)
        ell_vals = torch.zeros(max(ell_lim * n, 1), dtype=F64, device=dv)
        cr = torch.zeros(max(coo_nnz, 1), dtype=I32, device=dv)
        cc = torch.zeros(max(coo_nnz, 1), dtype=I32, device=dv)
        cv = torch.zeros(max(coo_nnz, 1), dtype=F64, device=dv)
        gk.csr_convert_to_hybrid_f64_i32(s, n, csr.row_ptrs, csr.col_idxs, csr.vals, crp, ell_lim, n, ell_cols,
                                         ell_vals, cr, cc, cv)
        return cls(gk, n, csr.ncols, ell_lim, ell_cols, ell_vals, coo_nnz, cr, cc, cv)

    def storage_bytes(self):
        return 12 * self.ell_lim * self.nrows + 16 * self.coo_nnz

    def callback(self):
        ctx = HybridCtx(self.nrows, self.ncols, self.ell_lim, self.nrows, self.ell_cols.data_ptr(),
                        self.ell_vals.data_ptr(), self.coo_nnz, self.coo_rows.data_ptr(), self.coo_cols.data_ptr(),
                        self.coo_vals.data_ptr())
        return MatrixCallback(self.gk, "gkomi_hybrid_matrix_apply_cb", ctx, self)

    def apply(self, b, x, alpha=None, beta=None):
        dv = self.ell_vals.device
        self.gk.hybrid_spmv_f64_i32(_stream(self.ell_vals), self.nrows, self.ncols, b.shape[1], self.ell_lim,
                                    self.nrows, self.ell_cols, self.ell_vals, self.coo_nnz, self.coo_rows,
                                    self.coo_cols, self.coo_vals, b, b.stride(0), x, x.stride(0), _scalar(dv, alpha),
                                    _scalar(dv, beta))
        return x


FORMATS = ("csr", "coo", "ell", "sellp", "hybrid")


_BINARY_TYPES = {b"D": np.float64, b"S": np.float32, b"I": np.int32, b"L": np.int64}


def read_binary(gk, path, device="cuda:0"):
    """the reference's binary format (core/base/mtx_io.cpp:768-960): header
    "GINKGO" + value type + index type, rows, cols, entries, then (row, col,
    value) records -> Csr, assembled on the device"""
    with open(path, "rb") as f:
        header = f.read(32)
        assert len(header) == 32 and header[:6] == b"GINKGO", "not a Ginkgo binary matrix file"
        vt, it = header[6:7], header[7:8]
        if vt in (b"Z", b"C"):
            raise ValueError("cannot read into this format, would assign complex to real")
        rows_n, cols_n, entries = (int(x) for x in np.frombuffer(header[8:], dtype=np.uint64))
        rec = np.dtype([("row", _BINARY_TYPES[it]), ("col", _BINARY_TYPES[it]), ("val", _BINARY_TYPES[vt])])
        data = np.frombuffer(f.read(rec.itemsize * entries), dtype=rec)
        assert len(data) == entries, "truncated file"
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
    return Csr.from_triplets(gk, rows_n, cols_n, t(data["row"], np.int32), t(data["col"], np.int32),
                             t(data["val"], np.float64), sum_duplicates=False)


def read_matrix(gk, path, device="cuda:0"):
    """read_generic: MatrixMarket text if the file starts with '%', binary otherwise"""
    with open(path, "rb") as f:
        first = f.read(1)
    return read_mtx(gk, path, device) if first == b"%" else read_binary(gk, path, device)


def read_mtx(gk, path, device="cuda:0"):
    """MatrixMarket coordinate file -> Csr, assembled on the device (symmetric
    storage expanded, duplicates summed, row-major sorted)."""
    with open(path) as f:
        header = f.readline().lower().split()
        assert header[:3] == ["%%matrixmarket", "matrix", "coordinate"], "coordinate MatrixMarket files only"
        field, symmetry = header[3], header[4]
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nrows, ncols, nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, ndmin=2) if nnz else np.zeros((0, 3))
    rows = data[:, 0].astype(np.int32) - 1
    cols = data[:, 1].astype(np.int32) - 1
    vals = data[:, 2].astype(np.float64) if field != "pattern" else np.ones(len(rows))
    if symmetry in ("symmetric", "skew-symmetric"):
        off = rows != cols
        sign = -1.0 if symmetry == "skew-symmetric" else 1.0
        rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, sign * vals[off]]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return Csr.from_triplets(gk, nrows, ncols, t(rows), t(cols), t(vals))
