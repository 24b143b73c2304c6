// Cross-file host entry points inside the library (not part of the C ABI).
#pragma once
#include "common.hpp"

namespace gkomi {

int csr_spmv_dot_launch(hipStream_t stream, int nrows, int64_t nnz,
                        const int32_t* row_ptrs, const int32_t* col_idxs,
                        const double* vals, const double* p, double* q,
                        double* partial, const uint8_t* stop_status,
                        bool swizzle, const double* dot_w = nullptr,
                        double* partial2 = nullptr);
int csr_spmv_dot_num_partials(int nrows);
// the nonzero-split kernel with the dot epilogue, for matrices with an srow that stream from HBM
int csr_split_dot_num_partials(int64_t nnz, int64_t tile);
int csr_split_dot_launch(hipStream_t stream, int nrows, int64_t nnz, const int32_t* row_ptrs,
                         const int32_t* col_idxs, const double* vals, const double* p, double* q,
                         double* partial, const uint8_t* stop_status, const int32_t* srow, int64_t tile,
                         int over);
bool csr_auto_swizzle(int64_t nrows, int64_t nnz);

// One single-launch ("persistent") solver kernel at a time per process: two of them would each
// hold some CUs and wait for the rest (runtime.hip).  try_acquire is non-blocking.
bool persistent_try_acquire();
void persistent_release();
int device_cu_count();  // CUs of the current device, 0 = unknown

// end-of-solve health check of a preconditioner callback (blocking): 0 or an error such as
// GKOMI_ETRS_OVERRUN when one of the ILU's triangular solves gave up (precond.hip)
int precond_status(gkomi_apply_fn precond, void* ctx, gkomi_stream_t s);

// The system matrix of a solver driver: CSR arrays (op == nullptr) or any
// format behind a gkomi_matrix_apply_fn (Ell, Sellp, Coo, Hybrid, a user LinOp).
struct sysmat {
    int64_t n = 0, nnz = 0;
    const int32_t* row_ptrs = nullptr;
    const int32_t* col_idxs = nullptr;
    const double* vals = nullptr;
    int strategy = 0;
    int64_t hint = -1;
    gkomi_matrix_apply_fn op = nullptr;
    void* ctx = nullptr;
    bool is_csr() const { return op == nullptr; }
    // out = A in (alpha == beta == nullptr) or out = alpha A in + beta out; stride == nrhs
    int apply(gkomi_stream_t s, int64_t nrhs, const double* alpha, const double* in,
              const double* beta, double* out) const
    {
        if (op != nullptr) return op(ctx, s, nrhs, alpha, in, nrhs, beta, out, nrhs);
        return gkomi_csr_spmv_f64_i32(s, n, n, nrhs, nnz, row_ptrs, col_idxs, vals, in, nrhs, out,
                                      nrhs, alpha, beta, strategy, hint);
    }
};

inline sysmat make_csr_sysmat(int64_t n, int64_t nnz, const int32_t* row_ptrs,
                              const int32_t* col_idxs, const double* vals, int strategy,
                              int64_t hint)
{
    sysmat A;
    A.n = n; A.nnz = nnz; A.row_ptrs = row_ptrs; A.col_idxs = col_idxs; A.vals = vals;
    A.strategy = strategy; A.hint = hint;
    return A;
}

inline sysmat make_op_sysmat(int64_t n, gkomi_matrix_apply_fn op, void* ctx)
{
    sysmat A;
    A.n = n; A.op = op; A.ctx = ctx;
    return A;
}

// ELL / SELL-P behind the library's own callbacks (formats.hip): partials per
// launch of the SpMV + dot epilogue, 0 for any other operator
int op_spmv_dot_num_partials(gkomi_matrix_apply_fn op, const void* ctx);
int op_spmv_dot_launch(hipStream_t stream, gkomi_matrix_apply_fn op, const void* ctx,
                       const double* in, double* out, double* partial,
                       const uint8_t* stop_status, const double* dot_w, double* partial2);

// How a fused single-rhs driver gets out = A in together with the per-block
// partials of w . out (w = in unless given) and, on request, of out . out in
// ONE launch: the CSR stream kernel's epilogue, the ELL / SELL-P kernels'
// epilogue, or not at all (num_partials == 0: any other operator -- the driver
// follows A.apply with its own partials kernel).  A CSR matrix behind
// gkomi_csr_matrix_apply_cb counts as CSR.
struct spmv_dot_plan {
    int num_partials = 0;
    bool csr = false, swizzle = false;
    sysmat A;
    explicit spmv_dot_plan(const sysmat& A_) : A(A_)
    {
        if (!A.is_csr() && A.op == &gkomi_csr_matrix_apply_cb && A.ctx != nullptr) {
            const auto* m = static_cast<const gkomi_csr_ctx*>(A.ctx);
            if (m->nrows == A.n && m->ncols == A.n) {
                A = make_csr_sysmat(A.n, m->nnz, m->row_ptrs, m->col_idxs, m->vals,
                                    static_cast<int>(m->strategy), m->max_row_nnz_hint);
            }
        }
        if (A.n <= 0 || A.n > INT32_MAX) return;
        if (A.is_csr()) {
            if (reinterpret_cast<uintptr_t>(A.vals) % 16 != 0 ||
                reinterpret_cast<uintptr_t>(A.col_idxs) % 8 != 0) {
                return;
            }
            csr = true;
            swizzle = csr_auto_swizzle(A.n, A.nnz);
            num_partials = csr_spmv_dot_num_partials(static_cast<int>(A.n));
        } else if (A.ctx != nullptr) {
            num_partials = op_spmv_dot_num_partials(A.op, A.ctx);
        }
    }
    bool fused() const { return num_partials > 0; }
    int launch(hipStream_t stream, const double* in, double* out, double* partial,
               const uint8_t* stop_status, const double* dot_w = nullptr,
               double* partial2 = nullptr) const
    {
        if (csr) {
            return csr_spmv_dot_launch(stream, static_cast<int>(A.n), A.nnz, A.row_ptrs, A.col_idxs,
                                       A.vals, in, out, partial, stop_status, swizzle, dot_w, partial2);
        }
        return op_spmv_dot_launch(stream, A.op, A.ctx, in, out, partial, stop_status, dot_w, partial2);
    }
};

}  // namespace gkomi
