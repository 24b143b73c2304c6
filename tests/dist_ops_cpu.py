"""CPU stand-in for gkomi.distributed.GpuOps, backed by the ORACLE: lets the
gloo tests exercise the partition / halo-exchange / reduction logic of
gkomi.distributed without a GPU.  Test infrastructure only."""
import numpy as np
import torch


class OracleOps:
    def __init__(self, oracle):
        self.o = oracle
        self.device = torch.device("cpu")

    def tensor(self, a, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(a)).clone()
        return t if dtype is None else t.to(dtype)

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    @staticmethod
    def _np(t):
        return t.numpy()

    def build_local_nonlocal(self, rows, cols, vals, rp, cp, local_part):
        nnz = int(rows.numel())
        i32, f64 = np.int32, np.float64
        m = max(nnz, 1)
        out = dict(l_rows=np.zeros(m, i32), l_cols=np.zeros(m, i32), l_vals=np.zeros(m, f64), nl_rows=np.zeros(m, i32),
                   nl_cols=np.zeros(m, i32), nl_vals=np.zeros(m, f64), gather_idxs=np.zeros(m, i32),
                   recv_sizes=np.zeros(rp.num_parts, i32), non_local_to_global=np.zeros(m, np.int64))
        sizes = np.zeros(3, np.int64)
        self.o.ref_dist_build_local_nonlocal(
            nnz, self._np(rows), self._np(cols), self._np(vals), rp.range_bounds, rp.part_ids, rp.starts, rp.num_ranges,
            cp.range_bounds, cp.part_ids, cp.starts, cp.num_ranges, rp.num_parts, local_part, out["l_rows"], out["l_cols"],
            out["l_vals"], out["nl_rows"], out["nl_cols"], out["nl_vals"], out["gather_idxs"], out["recv_sizes"],
            out["non_local_to_global"], sizes)
        res = {k: torch.from_numpy(v) for k, v in out.items()}
        res.update(num_local=int(sizes[0]), num_non_local=int(sizes[1]), num_unique=int(sizes[2]))
        return res

    def coo_to_csr(self, nrows, row_idxs, nnz):
        ptrs = np.zeros(nrows + 1, np.int32)
        self.o.ref_convert_idxs_to_ptrs(self._np(row_idxs), nnz, nrows, ptrs)
        return torch.from_numpy(ptrs)

    def spmv(self, csr, b, x, alpha=None, beta=None):
        nrows, ncols, nnz, rp, ci, v = csr
        bb, xx = self._np(b), self._np(x)
        if alpha is None:
            self.o.ref_csr_spmv(nrows, b.shape[1], self._np(rp), self._np(ci), self._np(v), bb, b.stride(0), xx,
                                x.stride(0))
        else:
            self.o.ref_csr_advanced_spmv(nrows, b.shape[1], float(alpha[0]), self._np(rp), self._np(ci), self._np(v), bb,
                                         b.stride(0), float(beta[0]), xx, x.stride(0))

    def row_gather(self, idxs, count, src, out):
        self.o.ref_dense_row_gather(count, src.shape[1], self._np(idxs), self._np(src), src.stride(0), self._np(out),
                                    out.stride(0))

    def reduction_workspace(self, nrows, ncols):
        return torch.zeros(8, dtype=torch.uint8)

    def local_dot(self, x, y, result, ws):
        self.o.ref_dense_compute_dot(x.shape[0], x.shape[1], self._np(x), x.stride(0), self._np(y), y.stride(0),
                                     self._np(result))

    def local_squared_norm2(self, x, result, ws):
        self.o.ref_dense_compute_squared_norm2(x.shape[0], x.shape[1], self._np(x), x.stride(0), self._np(result))

    def sqrt_(self, t):
        self.o.ref_dense_compute_sqrt(1, t.numel(), self._np(t), t.numel())

    def cg_initialize(self, b, r, z, p, q, prev_rho, rho, stop):
        n, k = b.shape
        a = self._np
        self.o.ref_cg_initialize(n, k, a(b), k, a(r), k, a(z), k, a(p), k, a(q), k, a(prev_rho), a(rho), a(stop))

    def cg_step_1(self, p, z, rho, prev_rho, stop):
        n, k = p.shape
        a = self._np
        self.o.ref_cg_step_1(n, k, a(p), k, a(z), k, a(rho), a(prev_rho), a(stop))

    def cg_step_2(self, x, r, p, q, beta, rho, stop):
        n, k = x.shape
        a = self._np
        self.o.ref_cg_step_2(n, k, a(x), k, a(r), k, a(p), k, a(q), k, a(beta), a(rho), a(stop))

    def copy(self, src, dst):
        self.o.ref_dense_copy(src.shape[0], src.shape[1], self._np(src), src.stride(0), self._np(dst), dst.stride(0))

    def sub_scaled(self, alpha, x, y):
        n, k = x.shape
        self.o.ref_dense_sub_scaled(n, k, self._np(alpha), alpha.numel(), self._np(x), x.stride(0), self._np(y),
                                    y.stride(0))

    def fill(self, x, value):
        self.o.ref_dense_fill(x.shape[0], x.shape[1], self._np(x), x.stride(0), value)

    def copy_scalar(self, src, dst):
        self.o.ref_dense_copy(1, src.numel(), self._np(src), src.numel(), self._np(dst), dst.numel())

    def residual_check_device(self, tau, orig_tau, reduction, stop, flags):
        self.o.ref_residual_norm(tau.numel(), self._np(tau), self._np(orig_tau), reduction, 2, 1, self._np(stop),
                                 self._np(flags))

    def residual_check(self, tau, orig_tau, reduction, stop, flags):
        self.o.ref_residual_norm(tau.numel(), self._np(tau), self._np(orig_tau), reduction, 2, 1, self._np(stop),
                                 self._np(flags))
        return bool(flags[0])
