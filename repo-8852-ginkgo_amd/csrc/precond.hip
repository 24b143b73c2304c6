// Preconditioner apply callbacks (gkomi_apply_fn) for the native solver
// drivers: the LinOp::apply of preconditioner::Jacobi
// (core/preconditioner/jacobi.cpp apply_impl -> jacobi::simple_apply) and of
// preconditioner::Ilu (include/ginkgo/core/preconditioner/ilu.hpp:265-286:
// L^-1 into a cached intermediate, then U^-1).
#include "internal.hpp"

extern "C" int gkomi_jacobi_apply_cb(void* ctx_, gkomi_stream_t s, const double* in, double* out)
{
    const gkomi_jacobi_ctx* c = static_cast<const gkomi_jacobi_ctx*>(ctx_);
    if (c == nullptr) return GKOMI_EINVAL;
    if (c->max_block_size == 1) {
        return gkomi_jacobi_scalar_apply_f64(s, c->n, c->nrhs, c->blocks, nullptr, in, c->nrhs,
                                             nullptr, out, c->nrhs);
    }
    if (c->block_precisions != nullptr) {
        return gkomi_jacobi_apply_adaptive_f64_i32(s, c->num_blocks, c->max_block_size,
                                                   c->block_ptrs, c->block_precisions, c->blocks,
                                                   c->nrhs, nullptr, in, c->nrhs, nullptr, out,
                                                   c->nrhs);
    }
    return gkomi_jacobi_apply_f64_i32(s, c->num_blocks, c->max_block_size, c->block_ptrs,
                                      c->blocks, c->nrhs, nullptr, in, c->nrhs, nullptr, out,
                                      c->nrhs);
}

extern "C" int gkomi_ilu_apply_cb(void* ctx_, gkomi_stream_t s, const double* in, double* out)
{
    const gkomi_ilu_ctx* c = static_cast<const gkomi_ilu_ctx*>(ctx_);
    if (c == nullptr) return GKOMI_EINVAL;
    // factors analysed at generate (LowerTrs / UpperTrs::generate): the brick plan where the factor
    // has one, else the level-scheduled solve, else the analysis-free kernel
    int err;
    if (c->l_bricks != nullptr && c->u_bricks != nullptr && c->nrhs == 1 && in != out) {
        // both factors on the brick plan, one column: the two solves are a chain -- the lower solve's bricks arm
        // `out` for the upper solve, the upper solve's bricks re-arm the intermediate vector for the next apply,
        // and neither needs a launch that pre-fills its output with the sentinel (2 x ~6 us per apply).  The
        // intermediate vector is armed the ordinary way the first time (state in the context record).
        gkomi_ilu_ctx* state = static_cast<gkomi_ilu_ctx*>(ctx_);
        const bool armed = state->pad_ == 1;
        state->pad_ = 0;
        err = gkomi::trs_bricks_solve_chained(s, static_cast<gkomi_trs_bricks*>(c->l_bricks), c->l_bricks_plan, c->l_unit_diag, in,
                                              c->intermediate, armed, out, nullptr);
        if (err == GKOMI_SUCCESS) {
            err = gkomi::trs_bricks_solve_chained(s, static_cast<gkomi_trs_bricks*>(c->u_bricks), c->u_bricks_plan, 0, c->intermediate,
                                                  out, true, nullptr, c->intermediate);
            if (err == GKOMI_SUCCESS) state->pad_ = 1;
            if (err != GKOMI_ENOTSUPPORTED) return err;
            return gkomi_trs_bricks_solve_f64(s, c->u_bricks, c->u_bricks_plan, 1, 0, c->intermediate, 1, out, 1);
        }
        if (err != GKOMI_ENOTSUPPORTED) return err;
    }
    if (c->l_bricks != nullptr) {
        err = gkomi_trs_bricks_solve_f64(s, c->l_bricks, c->l_bricks_plan, c->nrhs, c->l_unit_diag, in, c->nrhs,
                                         c->intermediate, c->nrhs);
    } else if (c->l_plan != nullptr) {
        err = gkomi_trs_solve_plan_f64(s, c->n, c->nrhs, c->l_plan, c->l_nslices, c->l_entries, c->l_max_deps,
                                       c->l_unit_diag, in, c->nrhs, c->intermediate, c->nrhs);
    } else {
        err = gkomi_lower_trs_solve_f64_i32(s, c->n, c->nrhs, c->l_row_ptrs, c->l_col_idxs, c->l_vals,
                                            c->l_unit_diag, in, c->nrhs, c->intermediate, c->nrhs, c->trs_workspace,
                                            c->trs_workspace_bytes);
    }
    if (err) return err;
    if (c->u_bricks != nullptr) {
        return gkomi_trs_bricks_solve_f64(s, c->u_bricks, c->u_bricks_plan, c->nrhs, 0, c->intermediate, c->nrhs, out,
                                          c->nrhs);
    }
    if (c->u_plan != nullptr) {
        return gkomi_trs_solve_plan_f64(s, c->n, c->nrhs, c->u_plan, c->u_nslices, c->u_entries, c->u_max_deps, 0,
                                        c->intermediate, c->nrhs, out, c->nrhs);
    }
    return gkomi_upper_trs_solve_f64_i32(s, c->n, c->nrhs, c->u_row_ptrs, c->u_col_idxs, c->u_vals,
                                         0, c->intermediate, c->nrhs, out, c->nrhs,
                                         c->trs_workspace, c->trs_workspace_bytes);
}

// A triangular solve that gave up leaves NaNs in z and the solver then never converges: the
// drivers ask once, at the end of a solve, whether that is what happened (the flags are sticky).
int gkomi::precond_status(gkomi_apply_fn precond, void* ctx_, gkomi_stream_t s)
{
    if (precond != gkomi_ilu_apply_cb || ctx_ == nullptr) return GKOMI_SUCCESS;
    const gkomi_ilu_ctx* c = static_cast<const gkomi_ilu_ctx*>(ctx_);
    int flag = 0, any = 0;
    const bool l_analysed = c->l_plan != nullptr || c->l_bricks != nullptr;
    const bool u_analysed = c->u_plan != nullptr || c->u_bricks != nullptr;
    if (c->trs_workspace != nullptr && !(l_analysed && u_analysed)) {
        const int err = gkomi_trs_check_overrun(s, c->trs_workspace, &flag);
        if (err) return err;
        any |= flag;
    }
    for (const void* plan : {static_cast<const void*>(c->l_plan), static_cast<const void*>(c->u_plan)}) {
        if (plan == nullptr) continue;
        const int err = gkomi_trs_plan_check_overrun(s, plan, &flag);
        if (err) return err;
        any |= flag;
    }
    for (const void* plan : {c->l_bricks != nullptr ? static_cast<const void*>(c->l_bricks_plan) : nullptr,
                             c->u_bricks != nullptr ? static_cast<const void*>(c->u_bricks_plan) : nullptr}) {
        if (plan == nullptr) continue;
        const int err = gkomi_trs_bricks_check_overrun(s, plan, &flag);
        if (err) return err;
        any |= flag;
    }
    return any ? GKOMI_ETRS_OVERRUN : GKOMI_SUCCESS;
}

// ---- system matrices as callbacks (gkomi_matrix_apply_fn) --------------------------------
extern "C" int gkomi_csr_matrix_apply_cb(void* ctx_, gkomi_stream_t s, int64_t nrhs,
                                         const double* alpha, const double* b, int64_t b_stride,
                                         const double* beta, double* c, int64_t c_stride)
{
    const gkomi_csr_ctx* m = static_cast<const gkomi_csr_ctx*>(ctx_);
    if (m == nullptr) return GKOMI_EINVAL;
    return gkomi_csr_spmv_srow_f64_i32(s, m->nrows, m->ncols, nrhs, m->nnz, m->row_ptrs, m->col_idxs,
                                       m->vals, b, b_stride, c, c_stride, alpha, beta,
                                       static_cast<int>(m->strategy), m->max_row_nnz_hint, m->srow,
                                       m->srow != nullptr ? m->srow_tile : 0);
}

// Csr<double, int64> as a system matrix of the *_solve_op_f64 drivers
extern "C" int gkomi_csr64_matrix_apply_cb(void* ctx_, gkomi_stream_t s, int64_t nrhs, const double* alpha, const double* b,
                                           int64_t b_stride, const double* beta, double* c, int64_t c_stride)
{
    const gkomi_csr64_ctx* m = static_cast<const gkomi_csr64_ctx*>(ctx_);
    if (m == nullptr) return GKOMI_EINVAL;
    return gkomi_csr_spmv_srow_f64_i64(s, m->nrows, m->ncols, nrhs, m->nnz, m->row_ptrs, m->col_idxs, m->vals, b, b_stride,
                                       c, c_stride, alpha, beta, static_cast<int>(m->strategy), m->max_row_nnz_hint,
                                       m->srow, m->srow != nullptr ? m->srow_tile : 0);
}

extern "C" int gkomi_ell_matrix_apply_cb(void* ctx_, gkomi_stream_t s, int64_t nrhs,
                                         const double* alpha, const double* b, int64_t b_stride,
                                         const double* beta, double* c, int64_t c_stride)
{
    const gkomi_ell_ctx* m = static_cast<const gkomi_ell_ctx*>(ctx_);
    if (m == nullptr) return GKOMI_EINVAL;
    return gkomi_ell_spmv_f64_i32(s, m->nrows, m->ncols, nrhs, m->num_stored_per_row, m->stride,
                                  m->col_idxs, m->vals, b, b_stride, c, c_stride, alpha, beta);
}

extern "C" int gkomi_sellp_matrix_apply_cb(void* ctx_, gkomi_stream_t s, int64_t nrhs,
                                           const double* alpha, const double* b, int64_t b_stride,
                                           const double* beta, double* c, int64_t c_stride)
{
    const gkomi_sellp_ctx* m = static_cast<const gkomi_sellp_ctx*>(ctx_);
    if (m == nullptr) return GKOMI_EINVAL;
    return gkomi_sellp_spmv_f64_i32(s, m->nrows, m->ncols, nrhs, m->slice_size, m->slice_sets,
                                    m->slice_lengths, m->col_idxs, m->vals, b, b_stride, c, c_stride,
                                    alpha, beta);
}

extern "C" int gkomi_coo_matrix_apply_cb(void* ctx_, gkomi_stream_t s, int64_t nrhs,
                                         const double* alpha, const double* b, int64_t b_stride,
                                         const double* beta, double* c, int64_t c_stride)
{
    const gkomi_coo_ctx* m = static_cast<const gkomi_coo_ctx*>(ctx_);
    if (m == nullptr) return GKOMI_EINVAL;
    return gkomi_coo_spmv_f64_i32(s, m->nrows, m->ncols, nrhs, m->nnz, m->row_idxs, m->col_idxs,
                                  m->vals, b, b_stride, c, c_stride, alpha, beta);
}

extern "C" int gkomi_hybrid_matrix_apply_cb(void* ctx_, gkomi_stream_t s, int64_t nrhs,
                                            const double* alpha, const double* b, int64_t b_stride,
                                            const double* beta, double* c, int64_t c_stride)
{
    const gkomi_hybrid_ctx* m = static_cast<const gkomi_hybrid_ctx*>(ctx_);
    if (m == nullptr) return GKOMI_EINVAL;
    return gkomi_hybrid_spmv_f64_i32(s, m->nrows, m->ncols, nrhs, m->ell_num_stored_per_row,
                                     m->ell_stride, m->ell_col_idxs, m->ell_vals, m->coo_nnz,
                                     m->coo_row_idxs, m->coo_col_idxs, m->coo_vals, b, b_stride, c,
                                     c_stride, alpha, beta);
}
