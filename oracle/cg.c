/* ORACLE (test infrastructure).  CG kernels + stop criteria + the Cg driver. */
#include "oracle_common.h"

void ref_csr_spmv(i64, i64, const i32*, const i32*, const double*,
                  const double*, i64, double*, i64);
void ref_csr_advanced_spmv(i64, i64, double, const i32*, const i32*,
                           const double*, const double*, i64, double, double*,
                           i64);
void ref_dense_compute_dot(i64, i64, const double*, i64, const double*, i64,
                           double*);
void ref_dense_compute_norm2(i64, i64, const double*, i64, double*);

/* reference/solver/cg_kernels.cpp:53-72 */
ORACLE_API void ref_cg_initialize(i64 nrows, i64 nrhs, const double* b,
                                  i64 b_stride, double* r, i64 r_stride,
                                  double* z, i64 z_stride, double* p,
                                  i64 p_stride, double* q, i64 q_stride,
                                  double* prev_rho, double* rho,
                                  u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rho[j] = 0.0;
        prev_rho[j] = 1.0;
        stop_status[j] = 0;
    }
    for (i64 i = 0; i < nrows; ++i) {
        for (i64 j = 0; j < nrhs; ++j) {
            r[i * r_stride + j] = b[i * b_stride + j];
            z[i * z_stride + j] = p[i * p_stride + j] = q[i * q_stride + j] = 0.0;
        }
    }
}

/* reference/solver/cg_kernels.cpp:77-97 */
ORACLE_API void ref_cg_step_1(i64 nrows, i64 nrhs, double* p, i64 p_stride,
                              const double* z, i64 z_stride, const double* rho,
                              const double* prev_rho, const u8* stop_status)
{
    for (i64 i = 0; i < nrows; ++i) {
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (prev_rho[j] == 0.0) {
                p[i * p_stride + j] = z[i * z_stride + j];
            } else {
                const double tmp = rho[j] / prev_rho[j];
                p[i * p_stride + j] = z[i * z_stride + j] + tmp * p[i * p_stride + j];
            }
        }
    }
}

/* reference/solver/cg_kernels.cpp:102-123 */
ORACLE_API void ref_cg_step_2(i64 nrows, i64 nrhs, double* x, i64 x_stride,
                              double* r, i64 r_stride, const double* p,
                              i64 p_stride, const double* q, i64 q_stride,
                              const double* beta, const double* rho,
                              const u8* stop_status)
{
    for (i64 i = 0; i < nrows; ++i) {
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (beta[j] != 0.0) {
                const double tmp = rho[j] / beta[j];
                x[i * x_stride + j] += tmp * p[i * p_stride + j];
                r[i * r_stride + j] -= tmp * q[i * q_stride + j];
            }
        }
    }
}

/* reference/stop/residual_norm_kernels.cpp:57-83; flags = {all_converged, one_changed} */
ORACLE_API void ref_residual_norm(i64 nrhs, const double* tau,
                                  const double* orig_tau, double goal, u8 id,
                                  int set_finalized, u8* stop_status, u8* flags)
{
    flags[0] = 1;
    flags[1] = 0;
    for (i64 i = 0; i < nrhs; ++i) {
        if (tau[i] < goal * orig_tau[i]) {
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

/* reference/stop/residual_norm_kernels.cpp:100-122 */
ORACLE_API void ref_implicit_residual_norm(i64 nrhs, const double* tau,
                                           const double* orig_tau, double goal,
                                           u8 id, int set_finalized,
                                           u8* stop_status, u8* flags)
{
    flags[0] = 1;
    flags[1] = 0;
    for (i64 i = 0; i < nrhs; ++i) {
        if (sqrt(fabs(tau[i])) < goal * orig_tau[i]) {
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

/* reference/stop/criterion_kernels.cpp:50-60 */
ORACLE_API void ref_set_all_statuses(i64 nrhs, u8 id, int set_finalized,
                                     u8* stop_status)
{
    for (i64 i = 0; i < nrhs; ++i)
        stop_status[i] = st_stop(stop_status[i], id, set_finalized);
}

/*
 * Cg::apply_dense_impl with an Identity preconditioner and the criteria
 * Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)):
 * core/solver/cg.cpp:107-193, core/stop/residual_norm.cpp:119-228,
 * core/stop/iteration.cpp:40, core/stop/combined.cpp:40.
 * baseline: 0 = rhs_norm, 1 = initial_resnorm, 2 = absolute.
 * Single right-hand side.  Returns the number of iterations (the `iter` at
 * which the criterion fired).  res_hist (may be NULL) receives ||r|| per
 * checked iteration, up to hist_cap entries.
 */
ORACLE_API i64 ref_cg_solve(i64 n, const i32* row_ptrs, const i32* col_idxs,
                            const double* vals, const double* b, double* x,
                            i64 max_iters, double reduction, int baseline,
                            double* res_hist, i64 hist_cap)
{
    double* r = (double*)malloc(sizeof(double) * (size_t)n);
    double* z = (double*)malloc(sizeof(double) * (size_t)n);
    double* p = (double*)malloc(sizeof(double) * (size_t)n);
    double* q = (double*)malloc(sizeof(double) * (size_t)n);
    double rho, prev_rho, beta, tau, orig_tau;
    u8 status, flags[2];
    ref_cg_initialize(n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, &prev_rho, &rho,
                      &status);
    /* r = b - A x  (advanced apply, alpha=-1, beta=1), cg.cpp:142 */
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, r, 1);
    /* criterion generate: baseline norm (residual_norm.cpp:119-189) */
    if (baseline == 0) {
        ref_dense_compute_norm2(n, 1, b, 1, &orig_tau);
    } else if (baseline == 1) {
        ref_dense_compute_norm2(n, 1, r, 1, &orig_tau);
    } else {
        orig_tau = 1.0;
    }
    i64 iter = -1;
    while (1) {
        memcpy(z, r, sizeof(double) * (size_t)n); /* Identity::apply: z = r */
        ref_dense_compute_dot(n, 1, r, 1, z, 1, &rho);
        ++iter;
        /* Combined: Iteration first (as built by with_criteria(iter, res)),
         * then ResidualNorm (residual_norm.cpp:193-228 recomputes ||r||) */
        int stop = 0;
        if (iter >= max_iters) {
            ref_set_all_statuses(1, 1, 1, &status);
            stop = 1;
        }
        ref_dense_compute_norm2(n, 1, r, 1, &tau);
        if (res_hist && iter < hist_cap) res_hist[iter] = tau;
        if (!stop) {
            ref_residual_norm(1, &tau, &orig_tau, reduction, 1, 1, &status, flags);
            stop = flags[0];
        }
        if (stop) break;
        ref_cg_step_1(n, 1, p, 1, z, 1, &rho, &prev_rho, &status);
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, p, 1, q, 1);
        ref_dense_compute_dot(n, 1, p, 1, q, 1, &beta);
        ref_cg_step_2(n, 1, x, 1, r, 1, p, 1, q, 1, &beta, &rho, &status);
        double t = prev_rho;
        prev_rho = rho;
        rho = t;
    }
    free(r);
    free(z);
    free(p);
    free(q);
    return iter;
}

/* ---- CPU baseline of the CG leg of bench.py (never the checker) ---------------
 * The omp/ path of Cg::apply_dense_impl (core/solver/cg.cpp:107-193, Identity
 * preconditioner, one right-hand side): omp csr::spmv
 * (omp/matrix/csr_kernels.cpp:76-99), the unified cg::step_1 / step_2 loops
 * (common/unified/solver/cg_kernels.cpp:53-131) as `omp parallel for`, and the
 * omp reductions (omp/base/kernel_launch_reduction.hpp:65-86: one partial per
 * thread, added in thread order).  Returns the iteration count; the result
 * depends on the thread count to rounding, so it is compared with nothing. */
void omp_csr_spmv(i64, i64, const i32*, const i32*, const double*,
                  const double*, i64, double*, i64);

static double omp_dot(i64 n, const double* a, const double* b)
{
    double total = 0.0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (i64 i = 0; i < n; ++i) total += a[i] * b[i];
    return total;
}

ORACLE_API i64 omp_cg_solve(i64 n, const i32* row_ptrs, const i32* col_idxs,
                            const double* vals, const double* b, double* x,
                            i64 max_iters, double reduction, double* final_rel)
{
    double* r = (double*)malloc(sizeof(double) * (size_t)n);
    double* p = (double*)calloc((size_t)n, sizeof(double));
    double* q = (double*)malloc(sizeof(double) * (size_t)n);
    /* r = b - A x */
    omp_csr_spmv(n, 1, row_ptrs, col_idxs, vals, x, 1, q, 1);
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) r[i] = b[i] - q[i];
    const double orig_tau = sqrt(omp_dot(n, b, b));
    double prev_rho = 1.0, rho = 0.0, tau = 0.0;
    i64 iter = -1;
    while (1) {
        rho = omp_dot(n, r, r); /* z = r */
        ++iter;
        tau = sqrt(rho);
        if (iter >= max_iters || tau < reduction * orig_tau) break;
        const double tmp = prev_rho == 0.0 ? 0.0 : rho / prev_rho;
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < n; ++i) p[i] = prev_rho == 0.0 ? r[i] : r[i] + tmp * p[i];
        omp_csr_spmv(n, 1, row_ptrs, col_idxs, vals, p, 1, q, 1);
        const double beta = omp_dot(n, p, q);
        if (beta != 0.0) {
            const double a = rho / beta;
#pragma omp parallel for schedule(static)
            for (i64 i = 0; i < n; ++i) {
                x[i] += a * p[i];
                r[i] -= a * q[i];
            }
        }
        prev_rho = rho;
    }
    if (final_rel) *final_rel = orig_tau == 0.0 ? tau : tau / orig_tau;
    free(r);
    free(p);
    free(q);
    return iter;
}
