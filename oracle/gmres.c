/* ORACLE (test infrastructure).  GMRES kernels and Gmres::apply_dense_impl. */
#include "oracle_common.h"

typedef uint64_t u64;
typedef int (*oracle_apply_fn)(void* ctx, const double* in, double* out);

void ref_csr_spmv(i64, i64, const i32*, const i32*, const double*, const double*, i64, double*, i64);
void ref_csr_advanced_spmv(i64, i64, double, const i32*, const i32*, const double*, const double*,
                           i64, double, double*, i64);
void ref_dense_compute_dot(i64, i64, const double*, i64, const double*, i64, double*);
void ref_dense_compute_norm2(i64, i64, const double*, i64, double*);
void ref_dense_sub_scaled(i64, i64, const double*, i64, const double*, i64, double*, i64);
void ref_dense_inv_scale(i64, i64, const double*, i64, double*, i64);
void ref_dense_add_scaled(i64, i64, const double*, i64, const double*, i64, double*, i64);
void ref_residual_norm(i64, const double*, const double*, double, u8, int, u8*, u8*);
void ref_set_all_statuses(i64, u8, int, u8*);

/* reference/solver/common_gmres_kernels.cpp:140-160 (initialize) */
ORACLE_API void ref_gmres_initialize(i64 n, i64 nrhs, i64 krylov_dim, const double* b,
                                     i64 b_stride, double* residual, i64 r_stride,
                                     double* givens_sin, double* givens_cos, u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        for (i64 i = 0; i < n; ++i) residual[i * r_stride + j] = b[i * b_stride + j];
        for (i64 i = 0; i < krylov_dim; ++i) {
            givens_sin[i * nrhs + j] = 0.0;
            givens_cos[i * nrhs + j] = 0.0;
        }
        stop_status[j] = 0;
    }
}

/* reference/solver/gmres_kernels.cpp:55-70 (restart) */
ORACLE_API void ref_gmres_restart(i64 n, i64 nrhs, const double* residual, i64 r_stride,
                                  const double* residual_norm, double* rnc,
                                  double* krylov_bases, i64 kb_stride, u64* final_iter_nums)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rnc[j] = residual_norm[j];
        for (i64 i = 0; i < n; ++i)
            krylov_bases[i * kb_stride + j] = residual[i * r_stride + j] / residual_norm[j];
        final_iter_nums[j] = 0;
    }
}

/* common_gmres_kernels.cpp:60-138, :165-185 (hessenberg_qr).  hess points at
 * column block `iter` of the Hessenberg matrix: entry (row, rhs) = hess[row*h_stride + rhs];
 * sin/cos/rnc rows have stride nrhs. */
ORACLE_API void ref_gmres_hessenberg_qr(i64 nrhs, double* gsin, double* gcos,
                                        double* residual_norm, double* rnc, double* hess,
                                        i64 h_stride, i64 iter, u64* final_iter_nums,
                                        const u8* stop_status)
{
    for (i64 i = 0; i < nrhs; ++i)
        if (!st_has_stopped(stop_status[i])) final_iter_nums[i]++;
    for (i64 i = 0; i < nrhs; ++i) {
        if (st_has_stopped(stop_status[i])) continue;
        for (i64 j = 0; j < iter; ++j) {
            const double temp = gcos[j * nrhs + i] * hess[j * h_stride + i] +
                                gsin[j * nrhs + i] * hess[(j + 1) * h_stride + i];
            hess[(j + 1) * h_stride + i] = -gsin[j * nrhs + i] * hess[j * h_stride + i] +
                                           gcos[j * nrhs + i] * hess[(j + 1) * h_stride + i];
            hess[j * h_stride + i] = temp;
        }
        /* calculate_sin_and_cos */
        if (hess[iter * h_stride + i] == 0.0) {
            gcos[iter * nrhs + i] = 0.0;
            gsin[iter * nrhs + i] = 1.0;
        } else {
            const double this_h = hess[iter * h_stride + i];
            const double next_h = hess[(iter + 1) * h_stride + i];
            const double scale = fabs(this_h) + fabs(next_h);
            const double hyp = scale * sqrt(fabs(this_h / scale) * fabs(this_h / scale) +
                                            fabs(next_h / scale) * fabs(next_h / scale));
            gcos[iter * nrhs + i] = this_h / hyp;
            gsin[iter * nrhs + i] = next_h / hyp;
        }
        hess[iter * h_stride + i] = gcos[iter * nrhs + i] * hess[iter * h_stride + i] +
                                    gsin[iter * nrhs + i] * hess[(iter + 1) * h_stride + i];
        hess[(iter + 1) * h_stride + i] = 0.0;
    }
    for (i64 i = 0; i < nrhs; ++i) {
        if (st_has_stopped(stop_status[i])) continue;
        rnc[(iter + 1) * nrhs + i] = -gsin[iter * nrhs + i] * rnc[iter * nrhs + i];
        rnc[iter * nrhs + i] = gcos[iter * nrhs + i] * rnc[iter * nrhs + i];
        residual_norm[i] = fabs(rnc[(iter + 1) * nrhs + i]);
    }
}

/* common_gmres_kernels.cpp:192-217 (solve_krylov): hessenberg entry (i, j*nrhs + k) */
ORACLE_API void ref_gmres_solve_krylov(i64 nrhs, const double* rnc, const double* hessenberg,
                                       i64 h_stride, double* y, const u64* final_iter_nums,
                                       const u8* stop_status)
{
    for (i64 k = 0; k < nrhs; ++k) {
        if (stop_status[k] & ST_FINALIZED) continue;
        for (i64 i = (i64)final_iter_nums[k] - 1; i >= 0; --i) {
            double temp = rnc[i * nrhs + k];
            for (i64 j = i + 1; j < (i64)final_iter_nums[k]; ++j)
                temp -= hessenberg[i * h_stride + j * nrhs + k] * y[j * nrhs + k];
            y[i * nrhs + k] = temp / hessenberg[i * h_stride + i * nrhs + k];
        }
    }
}

/* gmres_kernels.cpp:74-100 (multi_axpy) */
ORACLE_API void ref_gmres_multi_axpy(i64 n, i64 nrhs, const double* krylov_bases, i64 kb_stride,
                                     const double* y, double* before_precond, i64 bp_stride,
                                     const u64* final_iter_nums, u8* stop_status)
{
    for (i64 k = 0; k < nrhs; ++k) {
        if (stop_status[k] & ST_FINALIZED) continue;
        for (i64 i = 0; i < n; ++i) {
            before_precond[i * bp_stride + k] = 0.0;
            for (i64 j = 0; j < (i64)final_iter_nums[k]; ++j)
                before_precond[i * bp_stride + k] +=
                    krylov_bases[(i + j * n) * kb_stride + k] * y[j * nrhs + k];
        }
        if (st_has_stopped(stop_status[k])) stop_status[k] |= ST_FINALIZED;
    }
}

/* Gmres::apply_dense_impl (core/solver/gmres.cpp:139-372), single rhs, CSR
 * matrix, criteria Combined(Iteration(max_iters) [id 1], ResidualNorm(reduction,
 * baseline) [id 2]); precond == NULL is Identity (a copy).  Returns total_iter. */
ORACLE_API i64 ref_gmres_solve(i64 n, const i32* row_ptrs, const i32* col_idxs,
                               const double* vals, oracle_apply_fn precond, void* precond_ctx,
                               const double* b, double* x, i64 krylov_dim, i64 max_iters,
                               double reduction, int baseline, double* final_res_norm)
{
    const size_t vn = sizeof(double) * (size_t)n;
    double* residual = (double*)malloc(vn);
    double* pv = (double*)malloc(vn);
    double* before = (double*)malloc(vn);
    double* after = (double*)malloc(vn);
    double* kb = (double*)malloc(vn * (size_t)(krylov_dim + 1));
    double* hess = (double*)calloc((size_t)((krylov_dim + 1) * krylov_dim), sizeof(double));
    double* gsin = (double*)malloc(sizeof(double) * (size_t)krylov_dim);
    double* gcos = (double*)malloc(sizeof(double) * (size_t)krylov_dim);
    double* rnc = (double*)malloc(sizeof(double) * (size_t)(krylov_dim + 1));
    double* y = (double*)malloc(sizeof(double) * (size_t)krylov_dim);
    double residual_norm, orig_tau, one = 1.0;
    u64 final_iter = 0;
    u8 status, flags[2];
    ref_gmres_initialize(n, 1, krylov_dim, b, 1, residual, 1, gsin, gcos, &status);
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, residual, 1);
    ref_dense_compute_norm2(n, 1, residual, 1, &residual_norm);
    ref_gmres_restart(n, 1, residual, 1, &residual_norm, rnc, kb, 1, &final_iter);
    if (baseline == 0) {
        ref_dense_compute_norm2(n, 1, b, 1, &orig_tau);
    } else if (baseline == 1) {
        orig_tau = residual_norm;
    } else {
        orig_tau = 1.0;
    }
    i64 total_iter = -1, restart_iter = 0;
    while (1) {
        ++total_iter;
        int stop = 0;
        if (total_iter >= max_iters) {
            ref_set_all_statuses(1, 1, 0, &status);
            stop = 1;
        } else {
            ref_residual_norm(1, &residual_norm, &orig_tau, reduction, 2, 0, &status, flags);
            stop = flags[0];
        }
        if (stop) break;
        if (restart_iter == krylov_dim) {
            ref_gmres_solve_krylov(1, rnc, hess, krylov_dim, y, &final_iter, &status);
            ref_gmres_multi_axpy(n, 1, kb, 1, y, before, 1, &final_iter, &status);
            if (precond) precond(precond_ctx, before, after); else memcpy(after, before, vn);
            ref_dense_add_scaled(n, 1, &one, 1, after, 1, x, 1);
            memcpy(residual, b, vn);
            ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, residual, 1);
            ref_dense_compute_norm2(n, 1, residual, 1, &residual_norm);
            ref_gmres_restart(n, 1, residual, 1, &residual_norm, rnc, kb, 1, &final_iter);
            restart_iter = 0;
        }
        double* this_k = kb + n * restart_iter;
        double* next_k = kb + n * (restart_iter + 1);
        if (precond) precond(precond_ctx, this_k, pv); else memcpy(pv, this_k, vn);
        double* hess_iter = hess + restart_iter; /* column block restart_iter, nrhs == 1 */
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, pv, 1, next_k, 1);
        for (i64 i = 0; i <= restart_iter; ++i) {
            double* h = hess_iter + i * krylov_dim;
            ref_dense_compute_dot(n, 1, next_k, 1, kb + n * i, 1, h);
            ref_dense_sub_scaled(n, 1, h, 1, kb + n * i, 1, next_k, 1);
        }
        double* hn = hess_iter + (restart_iter + 1) * krylov_dim;
        ref_dense_compute_norm2(n, 1, next_k, 1, hn);
        ref_dense_inv_scale(n, 1, hn, 1, next_k, 1);
        ref_gmres_hessenberg_qr(1, gsin, gcos, &residual_norm, rnc, hess_iter, krylov_dim,
                                restart_iter, &final_iter, &status);
        restart_iter++;
    }
    ref_gmres_solve_krylov(1, rnc, hess, krylov_dim, y, &final_iter, &status);
    ref_gmres_multi_axpy(n, 1, kb, 1, y, before, 1, &final_iter, &status);
    if (precond) precond(precond_ctx, before, after); else memcpy(after, before, vn);
    ref_dense_add_scaled(n, 1, &one, 1, after, 1, x, 1);
    if (final_res_norm) *final_res_norm = residual_norm;
    free(residual); free(pv); free(before); free(after); free(kb); free(hess);
    free(gsin); free(gcos); free(rnc); free(y);
    return total_iter;
}
