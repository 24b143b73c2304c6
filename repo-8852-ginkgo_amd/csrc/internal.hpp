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
bool csr_auto_swizzle(int64_t nrows, int64_t nnz);

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

}  // namespace gkomi
