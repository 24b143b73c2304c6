/* ORACLE (test infrastructure).  The <float, int32> instantiation of the core of the path -- csr::spmv /
 * advanced_spmv, the dense BLAS-1 kernels, the CG kernels, residual_norm and the Cg driver -- restated in
 * single precision: the reference instantiates every kernel of the path for float too
 * (GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE, include/ginkgo/core/base/types.hpp:544-560), with the same loop
 * bodies; every intermediate is a float (no promotion to double), -ffp-contract=off as everywhere in oracle/.
 * Pinned by the same known-answer vectors as the double functions: the reference's tests are TYPED_TESTs over
 * float and double with the same numbers (tests/test_oracle_f32.py). */
#include "oracle_common.h"

/* reference/matrix/csr_kernels.cpp:75-96 */
ORACLE_API void ref_csr_spmv_f32(i64 nrows, i64 nrhs, const i32* row_ptrs, const i32* col_idxs, const float* vals,
                                 const float* b, i64 b_stride, float* c, i64 c_stride)
{
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 j = 0; j < nrhs; ++j) c[row * c_stride + j] = 0.0f;
        for (i64 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const float val = vals[k];
            const i64 col = col_idxs[k];
            for (i64 j = 0; j < nrhs; ++j) {
                const float prod = val * b[col * b_stride + j];
                c[row * c_stride + j] = c[row * c_stride + j] + prod;
            }
        }
    }
}

/* reference/matrix/csr_kernels.cpp:102-128 */
ORACLE_API void ref_csr_advanced_spmv_f32(i64 nrows, i64 nrhs, float alpha, const i32* row_ptrs, const i32* col_idxs,
                                          const float* vals, const float* b, i64 b_stride, float beta, float* c, i64 c_stride)
{
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 j = 0; j < nrhs; ++j) c[row * c_stride + j] = c[row * c_stride + j] * beta;
        for (i64 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const float val = vals[k];
            const i64 col = col_idxs[k];
            for (i64 j = 0; j < nrhs; ++j) {
                const float av = alpha * val;
                const float prod = av * b[col * b_stride + j];
                c[row * c_stride + j] = c[row * c_stride + j] + prod;
            }
        }
    }
}

/* reference/matrix/dense_kernels.cpp: copy :127-141, fill :144-155, scale :158-177, inv_scale :180-201,
 * add_scaled :204-225, sub_scaled :228-249 (alpha: one entry, or one per column) */
ORACLE_API void ref_dense_fill_f32(i64 nrows, i64 ncols, float* x, i64 stride, float value)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) x[i * stride + j] = value;
}

ORACLE_API void ref_dense_copy_f32(i64 nrows, i64 ncols, const float* in, i64 in_stride, float* out, i64 out_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) out[i * out_stride + j] = in[i * in_stride + j];
}

ORACLE_API void ref_dense_scale_f32(i64 nrows, i64 ncols, const float* alpha, i64 alpha_ncols, float* x, i64 stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) x[i * stride + j] = x[i * stride + j] * alpha[alpha_ncols == 1 ? 0 : j];
}

ORACLE_API void ref_dense_inv_scale_f32(i64 nrows, i64 ncols, const float* alpha, i64 alpha_ncols, float* x, i64 stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) x[i * stride + j] = x[i * stride + j] / alpha[alpha_ncols == 1 ? 0 : j];
}

ORACLE_API void ref_dense_add_scaled_f32(i64 nrows, i64 ncols, const float* alpha, i64 alpha_ncols, const float* x,
                                         i64 x_stride, float* y, i64 y_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) {
            const float prod = alpha[alpha_ncols == 1 ? 0 : j] * x[i * x_stride + j];
            y[i * y_stride + j] = y[i * y_stride + j] + prod;
        }
}

ORACLE_API void ref_dense_sub_scaled_f32(i64 nrows, i64 ncols, const float* alpha, i64 alpha_ncols, const float* x,
                                         i64 x_stride, float* y, i64 y_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) {
            const float prod = alpha[alpha_ncols == 1 ? 0 : j] * x[i * x_stride + j];
            y[i * y_stride + j] = y[i * y_stride + j] - prod;
        }
}

/* :282-297 compute_dot, :347-364 compute_norm2 */
ORACLE_API void ref_dense_compute_dot_f32(i64 nrows, i64 ncols, const float* x, i64 x_stride, const float* y, i64 y_stride,
                                          float* result)
{
    for (i64 j = 0; j < ncols; ++j) result[j] = 0.0f;
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) {
            const float prod = x[i * x_stride + j] * y[i * y_stride + j];
            result[j] = result[j] + prod;
        }
}

ORACLE_API void ref_dense_compute_norm2_f32(i64 nrows, i64 ncols, const float* x, i64 x_stride, float* result)
{
    for (i64 j = 0; j < ncols; ++j) result[j] = 0.0f;
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) {
            const float sq = x[i * x_stride + j] * x[i * x_stride + j];
            result[j] = result[j] + sq;
        }
    for (i64 j = 0; j < ncols; ++j) result[j] = sqrtf(result[j]);
}

/* reference/solver/cg_kernels.cpp:53-72, 77-97, 102-123 */
ORACLE_API void ref_cg_initialize_f32(i64 nrows, i64 nrhs, const float* b, i64 b_stride, float* r, i64 r_stride, float* z,
                                      i64 z_stride, float* p, i64 p_stride, float* q, i64 q_stride, float* prev_rho,
                                      float* rho, u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rho[j] = 0.0f;
        prev_rho[j] = 1.0f;
        stop_status[j] = 0;
    }
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            r[i * r_stride + j] = b[i * b_stride + j];
            z[i * z_stride + j] = p[i * p_stride + j] = q[i * q_stride + j] = 0.0f;
        }
}

ORACLE_API void ref_cg_step_1_f32(i64 nrows, i64 nrhs, float* p, i64 p_stride, const float* z, i64 z_stride,
                                  const float* rho, const float* prev_rho, const u8* stop_status)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (prev_rho[j] == 0.0f) {
                p[i * p_stride + j] = z[i * z_stride + j];
            } else {
                const float tmp = rho[j] / prev_rho[j];
                const float prod = tmp * p[i * p_stride + j];
                p[i * p_stride + j] = z[i * z_stride + j] + prod;
            }
        }
}

ORACLE_API void ref_cg_step_2_f32(i64 nrows, i64 nrhs, float* x, i64 x_stride, float* r, i64 r_stride, const float* p,
                                  i64 p_stride, const float* q, i64 q_stride, const float* beta, const float* rho,
                                  const u8* stop_status)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (beta[j] != 0.0f) {
                const float tmp = rho[j] / beta[j];
                const float px = tmp * p[i * p_stride + j];
                const float qx = tmp * q[i * q_stride + j];
                x[i * x_stride + j] = x[i * x_stride + j] + px;
                r[i * r_stride + j] = r[i * r_stride + j] - qx;
            }
        }
}

/* reference/stop/residual_norm_kernels.cpp:57-83; flags = {all_converged, one_changed} */
ORACLE_API void ref_residual_norm_f32(i64 nrhs, const float* tau, const float* orig_tau, float goal, u8 id,
                                      int set_finalized, u8* stop_status, u8* flags)
{
    flags[0] = 1;
    flags[1] = 0;
    for (i64 i = 0; i < nrhs; ++i) {
        const float bound = goal * orig_tau[i];
        if (tau[i] < bound) {
            stop_status[i] = st_converge(stop_status[i], id, set_finalized);
            flags[1] = 1;
        }
    }
    for (i64 i = 0; i < nrhs; ++i) {
        if (!st_has_stopped(stop_status[i])) {
            flags[0] = 0;
            break;
        }
    }
}

/* Cg::apply_dense_impl, Identity preconditioner, Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)):
 * core/solver/cg.cpp:107-193 (see ref_cg_solve in cg.c).  Single right-hand side; returns the iteration count. */
ORACLE_API i64 ref_cg_solve_f32(i64 n, const i32* row_ptrs, const i32* col_idxs, const float* vals, const float* b, float* x,
                                i64 max_iters, float reduction, int baseline)
{
    float* r = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* z = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* p = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* q = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float rho, prev_rho, beta, tau, orig_tau;
    u8 status, flags[2];
    ref_cg_initialize_f32(n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, &prev_rho, &rho, &status);
    ref_csr_advanced_spmv_f32(n, 1, -1.0f, row_ptrs, col_idxs, vals, x, 1, 1.0f, r, 1);
    if (baseline == 0) {
        ref_dense_compute_norm2_f32(n, 1, b, 1, &orig_tau);
    } else if (baseline == 1) {
        ref_dense_compute_norm2_f32(n, 1, r, 1, &orig_tau);
    } else {
        orig_tau = 1.0f;
    }
    i64 iter = -1;
    while (1) {
        memcpy(z, r, sizeof(float) * (size_t)n);
        ref_dense_compute_dot_f32(n, 1, r, 1, z, 1, &rho);
        ++iter;
        int stop = 0;
        if (iter >= max_iters) {
            status = st_stop(status, 1, 1);
            stop = 1;
        }
        ref_dense_compute_norm2_f32(n, 1, r, 1, &tau);
        if (!stop) {
            ref_residual_norm_f32(1, &tau, &orig_tau, reduction, 1, 1, &status, flags);
            stop = flags[0];
        }
        if (stop) break;
        ref_cg_step_1_f32(n, 1, p, 1, z, 1, &rho, &prev_rho, &status);
        ref_csr_spmv_f32(n, 1, row_ptrs, col_idxs, vals, p, 1, q, 1);
        ref_dense_compute_dot_f32(n, 1, p, 1, q, 1, &beta);
        ref_cg_step_2_f32(n, 1, x, 1, r, 1, p, 1, q, 1, &beta, &rho, &status);
        const float t = prev_rho;
        prev_rho = rho;
        rho = t;
    }
    free(r);
    free(z);
    free(p);
    free(q);
    return iter;
}
