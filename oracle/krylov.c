/* ORACLE (test infrastructure).  BiCGSTAB, FCG and CGS: step kernels
 * (reference/solver/{bicgstab,fcg,cgs}_kernels.cpp) and drivers
 * (core/solver/{bicgstab,fcg,cgs}.cpp), SURVEY 8(f) rank 3.  Vectors are
 * row-major n x nrhs with stride nrhs in the drivers; the criterion is
 * Combined(Iteration(max_iters), ResidualNorm(reduction, baseline)) as in
 * ref_cg_solve; the preconditioner is the Identity. */
#include "oracle_common.h"

void ref_csr_spmv(i64, i64, const i32*, const i32*, const double*, const double*, i64, double*,
                  i64);
void ref_csr_advanced_spmv(i64, i64, double, const i32*, const i32*, const double*,
                           const double*, i64, double, double*, i64);
void ref_dense_compute_dot(i64, i64, const double*, i64, const double*, i64, double*);
void ref_dense_compute_norm2(i64, i64, const double*, i64, double*);
void ref_residual_norm(i64, const double*, const double*, double, u8, int, u8*, u8*);
void ref_set_all_statuses(i64, u8, int, u8*);

#define AT(v, i, j) v[(i) * v##_stride + (j)]

/* ---- BiCGSTAB (bicgstab_kernels.cpp:57-232) -------------------------------- */
ORACLE_API void ref_bicgstab_initialize(i64 n, i64 nrhs, const double* b, i64 b_stride, double* r,
                                        i64 r_stride, double* rr, i64 rr_stride, double* y,
                                        i64 y_stride, double* s, i64 s_stride, double* t,
                                        i64 t_stride, double* z, i64 z_stride, double* v,
                                        i64 v_stride, double* p, i64 p_stride, double* prev_rho,
                                        double* rho, double* alpha, double* beta, double* gamma,
                                        double* omega, u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rho[j] = prev_rho[j] = alpha[j] = beta[j] = gamma[j] = omega[j] = 1.0;
        stop_status[j] = 0;
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            AT(r, i, j) = AT(b, i, j);
            AT(rr, i, j) = AT(z, i, j) = AT(v, i, j) = AT(s, i, j) = AT(t, i, j) = AT(y, i, j) =
                AT(p, i, j) = 0.0;
        }
}

ORACLE_API void ref_bicgstab_step_1(i64 n, i64 nrhs, const double* r, i64 r_stride, double* p,
                                    i64 p_stride, const double* v, i64 v_stride,
                                    const double* rho, const double* prev_rho,
                                    const double* alpha, const double* omega,
                                    const u8* stop_status)
{
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (prev_rho[j] * omega[j] != 0.0) {
                const double tmp = rho[j] / prev_rho[j] * alpha[j] / omega[j];
                AT(p, i, j) = AT(r, i, j) + tmp * (AT(p, i, j) - omega[j] * AT(v, i, j));
            } else {
                AT(p, i, j) = AT(r, i, j);
            }
        }
}

ORACLE_API void ref_bicgstab_step_2(i64 n, i64 nrhs, const double* r, i64 r_stride, double* s,
                                    i64 s_stride, const double* v, i64 v_stride,
                                    const double* rho, double* alpha, const double* beta,
                                    const u8* stop_status)
{
    /* alpha is assigned inside the row loop in the reference; with n == 0 it
     * stays untouched, reproduced by looping the same way */
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (beta[j] != 0.0) {
                alpha[j] = rho[j] / beta[j];
                AT(s, i, j) = AT(r, i, j) - alpha[j] * AT(v, i, j);
            } else {
                alpha[j] = 0.0;
                AT(s, i, j) = AT(r, i, j);
            }
        }
}

ORACLE_API void ref_bicgstab_step_3(i64 n, i64 nrhs, double* x, i64 x_stride, double* r,
                                    i64 r_stride, const double* s, i64 s_stride, const double* t,
                                    i64 t_stride, const double* y, i64 y_stride, const double* z,
                                    i64 z_stride, const double* alpha, const double* beta,
                                    const double* gamma, double* omega, const u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        if (st_has_stopped(stop_status[j])) continue;
        omega[j] = beta[j] != 0.0 ? gamma[j] / beta[j] : 0.0;
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            AT(x, i, j) += alpha[j] * AT(y, i, j) + omega[j] * AT(z, i, j);
            AT(r, i, j) = AT(s, i, j) - omega[j] * AT(t, i, j);
        }
}

ORACLE_API void ref_bicgstab_finalize(i64 n, i64 nrhs, double* x, i64 x_stride, const double* y,
                                      i64 y_stride, const double* alpha, u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        if (st_has_stopped(stop_status[j]) && !(stop_status[j] & ST_FINALIZED)) {
            for (i64 i = 0; i < n; ++i) {
                AT(x, i, j) += alpha[j] * AT(y, i, j);
                stop_status[j] |= ST_FINALIZED;
            }
        }
    }
}

/* ---- FCG (fcg_kernels.cpp:55-145) ----------------------------------------- */
ORACLE_API void ref_fcg_initialize(i64 n, i64 nrhs, const double* b, i64 b_stride, double* r,
                                   i64 r_stride, double* z, i64 z_stride, double* p, i64 p_stride,
                                   double* q, i64 q_stride, double* t, i64 t_stride,
                                   double* prev_rho, double* rho, double* rho_t, u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rho[j] = 0.0;
        prev_rho[j] = rho_t[j] = 1.0;
        stop_status[j] = 0;
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            AT(t, i, j) = AT(r, i, j) = AT(b, i, j);
            AT(z, i, j) = AT(p, i, j) = AT(q, i, j) = 0.0;
        }
}

ORACLE_API void ref_fcg_step_1(i64 n, i64 nrhs, double* p, i64 p_stride, const double* z,
                               i64 z_stride, const double* rho_t, const double* prev_rho,
                               const u8* stop_status)
{
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (prev_rho[j] == 0.0) {
                AT(p, i, j) = AT(z, i, j);
            } else {
                const double tmp = rho_t[j] / prev_rho[j];
                AT(p, i, j) = AT(z, i, j) + tmp * AT(p, i, j);
            }
        }
}

ORACLE_API void ref_fcg_step_2(i64 n, i64 nrhs, double* x, i64 x_stride, double* r, i64 r_stride,
                               double* t, i64 t_stride, const double* p, i64 p_stride,
                               const double* q, i64 q_stride, const double* beta,
                               const double* rho, const u8* stop_status)
{
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (beta[j] != 0.0) {
                const double tmp = rho[j] / beta[j];
                const double prev_r = AT(r, i, j);
                AT(x, i, j) += tmp * AT(p, i, j);
                AT(r, i, j) -= tmp * AT(q, i, j);
                AT(t, i, j) = AT(r, i, j) - prev_r;
            }
        }
}

/* ---- CGS (cgs_kernels.cpp:55-185) ----------------------------------------- */
ORACLE_API void ref_cgs_initialize(i64 n, i64 nrhs, const double* b, i64 b_stride, double* r,
                                   i64 r_stride, double* r_tld, i64 r_tld_stride, double* p,
                                   i64 p_stride, double* q, i64 q_stride, double* u, i64 u_stride,
                                   double* u_hat, i64 u_hat_stride, double* v_hat,
                                   i64 v_hat_stride, double* t, i64 t_stride, double* alpha,
                                   double* beta, double* gamma, double* prev_rho, double* rho,
                                   u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rho[j] = 0.0;
        prev_rho[j] = alpha[j] = beta[j] = gamma[j] = 1.0;
        stop_status[j] = 0;
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            AT(r, i, j) = AT(r_tld, i, j) = AT(b, i, j);
            AT(u, i, j) = AT(u_hat, i, j) = AT(p, i, j) = AT(q, i, j) = AT(v_hat, i, j) =
                AT(t, i, j) = 0.0;
        }
}

ORACLE_API void ref_cgs_step_1(i64 n, i64 nrhs, const double* r, i64 r_stride, double* u,
                               i64 u_stride, double* p, i64 p_stride, const double* q,
                               i64 q_stride, double* beta, const double* rho,
                               const double* prev_rho, const u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        if (st_has_stopped(stop_status[j])) continue;
        if (prev_rho[j] != 0.0) beta[j] = rho[j] / prev_rho[j];
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            AT(u, i, j) = AT(r, i, j) + beta[j] * AT(q, i, j);
            AT(p, i, j) = AT(u, i, j) + beta[j] * (AT(q, i, j) + beta[j] * AT(p, i, j));
        }
}

ORACLE_API void ref_cgs_step_2(i64 n, i64 nrhs, const double* u, i64 u_stride,
                               const double* v_hat, i64 v_hat_stride, double* q, i64 q_stride,
                               double* t, i64 t_stride, double* alpha, const double* rho,
                               const double* gamma, const u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        if (st_has_stopped(stop_status[j])) continue;
        if (gamma[j] != 0.0) alpha[j] = rho[j] / gamma[j];
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            AT(q, i, j) = AT(u, i, j) - alpha[j] * AT(v_hat, i, j);
            AT(t, i, j) = AT(u, i, j) + AT(q, i, j);
        }
}

ORACLE_API void ref_cgs_step_3(i64 n, i64 nrhs, const double* t, i64 t_stride, const double* u_hat,
                               i64 u_hat_stride, double* r, i64 r_stride, double* x, i64 x_stride,
                               const double* alpha, const u8* stop_status)
{
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            AT(x, i, j) += alpha[j] * AT(u_hat, i, j);
            AT(r, i, j) -= alpha[j] * AT(t, i, j);
        }
}

/* ---- drivers (single right-hand side) -------------------------------------- */
static double baseline_norm(int baseline, i64 n, const double* b, const double* r)
{
    double t = 1.0;
    if (baseline == 0) ref_dense_compute_norm2(n, 1, b, 1, &t);
    if (baseline == 1) ref_dense_compute_norm2(n, 1, r, 1, &t);
    return t;
}

/* Combined(Iteration, ResidualNorm) on residual `res`; returns 1 to stop */
static int check(i64 iter, i64 max_iters, i64 n, const double* res, double orig_tau,
                 double reduction, int set_finalized, u8* status, u8* one_changed)
{
    u8 flags[2] = {0, 0};
    double tau;
    *one_changed = 0;
    if (iter >= max_iters) {
        const u8 before = *status;
        ref_set_all_statuses(1, 1, set_finalized, status);
        *one_changed = before != *status;
        return 1;
    }
    ref_dense_compute_norm2(n, 1, res, 1, &tau);
    ref_residual_norm(1, &tau, &orig_tau, reduction, 1, set_finalized, status, flags);
    *one_changed = flags[1];
    return flags[0];
}

static double* vec(i64 n) { return (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* core/solver/bicgstab.cpp:107-234; returns iterations */
ORACLE_API i64 ref_bicgstab_solve(i64 n, const i32* row_ptrs, const i32* col_idxs,
                                  const double* vals, const double* b, double* x, i64 max_iters,
                                  double reduction, int baseline)
{
    double *r = vec(n), *z = vec(n), *y = vec(n), *v = vec(n), *s = vec(n), *t = vec(n),
           *p = vec(n), *rr = vec(n);
    double alpha, beta, gamma, prev_rho, rho, omega;
    u8 status, one_changed;
    ref_bicgstab_initialize(n, 1, b, 1, r, 1, rr, 1, y, 1, s, 1, t, 1, z, 1, v, 1, p, 1, &prev_rho,
                            &rho, &alpha, &beta, &gamma, &omega, &status);
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, r, 1);
    const double orig_tau = baseline_norm(baseline, n, b, r);
    memcpy(rr, r, sizeof(double) * (size_t)n);
    i64 iter = -1;
    while (1) {
        ++iter;
        ref_dense_compute_dot(n, 1, rr, 1, r, 1, &rho);
        if (check(iter, max_iters, n, r, orig_tau, reduction, 1, &status, &one_changed)) break;
        ref_bicgstab_step_1(n, 1, r, 1, p, 1, v, 1, &rho, &prev_rho, &alpha, &omega, &status);
        memcpy(y, p, sizeof(double) * (size_t)n); /* Identity preconditioner */
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, y, 1, v, 1);
        ref_dense_compute_dot(n, 1, rr, 1, v, 1, &beta);
        ref_bicgstab_step_2(n, 1, r, 1, s, 1, v, 1, &rho, &alpha, &beta, &status);
        const int all = check(iter, max_iters, n, s, orig_tau, reduction, 0, &status, &one_changed);
        if (one_changed) ref_bicgstab_finalize(n, 1, x, 1, y, 1, &alpha, &status);
        if (all) break;
        memcpy(z, s, sizeof(double) * (size_t)n);
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, z, 1, t, 1);
        ref_dense_compute_dot(n, 1, s, 1, t, 1, &gamma);
        ref_dense_compute_dot(n, 1, t, 1, t, 1, &beta);
        ref_bicgstab_step_3(n, 1, x, 1, r, 1, s, 1, t, 1, y, 1, z, 1, &alpha, &beta, &gamma,
                            &omega, &status);
        double sw = prev_rho;
        prev_rho = rho;
        rho = sw;
    }
    free(r); free(z); free(y); free(v); free(s); free(t); free(p); free(rr);
    return iter;
}

/* core/solver/fcg.cpp:104-196 */
ORACLE_API i64 ref_fcg_solve(i64 n, const i32* row_ptrs, const i32* col_idxs, const double* vals,
                             const double* b, double* x, i64 max_iters, double reduction,
                             int baseline)
{
    double *r = vec(n), *z = vec(n), *p = vec(n), *q = vec(n), *t = vec(n);
    double beta, prev_rho, rho, rho_t;
    u8 status, one_changed;
    ref_fcg_initialize(n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, t, 1, &prev_rho, &rho, &rho_t, &status);
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, r, 1);
    const double orig_tau = baseline_norm(baseline, n, b, r);
    i64 iter = -1;
    while (1) {
        memcpy(z, r, sizeof(double) * (size_t)n);
        ref_dense_compute_dot(n, 1, r, 1, z, 1, &rho);
        ref_dense_compute_dot(n, 1, t, 1, z, 1, &rho_t);
        ++iter;
        if (check(iter, max_iters, n, r, orig_tau, reduction, 1, &status, &one_changed)) break;
        ref_fcg_step_1(n, 1, p, 1, z, 1, &rho_t, &prev_rho, &status);
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, p, 1, q, 1);
        ref_dense_compute_dot(n, 1, p, 1, q, 1, &beta);
        ref_fcg_step_2(n, 1, x, 1, r, 1, t, 1, p, 1, q, 1, &beta, &rho, &status);
        double sw = prev_rho;
        prev_rho = rho;
        rho = sw;
    }
    free(r); free(z); free(p); free(q); free(t);
    return iter;
}

/* core/solver/cgs.cpp:107-205 */
ORACLE_API i64 ref_cgs_solve(i64 n, const i32* row_ptrs, const i32* col_idxs, const double* vals,
                             const double* b, double* x, i64 max_iters, double reduction,
                             int baseline)
{
    double *r = vec(n), *r_tld = vec(n), *p = vec(n), *q = vec(n), *u = vec(n), *u_hat = vec(n),
           *v_hat = vec(n), *t = vec(n);
    double alpha, beta, gamma, prev_rho, rho;
    u8 status, one_changed;
    ref_cgs_initialize(n, 1, b, 1, r, 1, r_tld, 1, p, 1, q, 1, u, 1, u_hat, 1, v_hat, 1, t, 1,
                       &alpha, &beta, &gamma, &prev_rho, &rho, &status);
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, r, 1);
    const double orig_tau = baseline_norm(baseline, n, b, r);
    memcpy(r_tld, r, sizeof(double) * (size_t)n);
    i64 iter = -1;
    while (1) {
        ref_dense_compute_dot(n, 1, r, 1, r_tld, 1, &rho);
        ++iter;
        if (check(iter, max_iters, n, r, orig_tau, reduction, 1, &status, &one_changed)) break;
        ref_cgs_step_1(n, 1, r, 1, u, 1, p, 1, q, 1, &beta, &rho, &prev_rho, &status);
        memcpy(t, p, sizeof(double) * (size_t)n);
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, t, 1, v_hat, 1);
        ref_dense_compute_dot(n, 1, r_tld, 1, v_hat, 1, &gamma);
        ref_cgs_step_2(n, 1, u, 1, v_hat, 1, q, 1, t, 1, &alpha, &rho, &gamma, &status);
        memcpy(u_hat, t, sizeof(double) * (size_t)n);
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, u_hat, 1, t, 1);
        ref_cgs_step_3(n, 1, t, 1, u_hat, 1, r, 1, x, 1, &alpha, &status);
        double sw = prev_rho;
        prev_rho = rho;
        rho = sw;
    }
    free(r); free(r_tld); free(p); free(q); free(u); free(u_hat); free(v_hat); free(t);
    return iter;
}

/* ---- BiCG (bicg_kernels.cpp:55-145, core/solver/bicg.cpp:117-232) ------------- */
ORACLE_API void ref_bicg_initialize(i64 n, i64 nrhs, const double* b, i64 b_stride, double* r,
                                    i64 r_stride, double* z, i64 z_stride, double* p, i64 p_stride,
                                    double* q, i64 q_stride, double* prev_rho, double* rho,
                                    double* r2, i64 r2_stride, double* z2, i64 z2_stride,
                                    double* p2, i64 p2_stride, double* q2, i64 q2_stride,
                                    u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) {
        rho[j] = 0.0;
        prev_rho[j] = 1.0;
        stop_status[j] = 0;
    }
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            AT(r, i, j) = AT(r2, i, j) = AT(b, i, j);
            AT(z, i, j) = AT(p, i, j) = AT(q, i, j) = 0.0;
            AT(z2, i, j) = AT(p2, i, j) = AT(q2, i, j) = 0.0;
        }
}

ORACLE_API void ref_bicg_step_1(i64 n, i64 nrhs, double* p, i64 p_stride, const double* z,
                                i64 z_stride, double* p2, i64 p2_stride, const double* z2,
                                i64 z2_stride, const double* rho, const double* prev_rho,
                                const u8* stop_status)
{
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (prev_rho[j] == 0.0) {
                AT(p, i, j) = AT(z, i, j);
                AT(p2, i, j) = AT(z2, i, j);
            } else {
                const double tmp = rho[j] / prev_rho[j];
                AT(p, i, j) = AT(z, i, j) + tmp * AT(p, i, j);
                AT(p2, i, j) = AT(z2, i, j) + tmp * AT(p2, i, j);
            }
        }
}

ORACLE_API void ref_bicg_step_2(i64 n, i64 nrhs, double* x, i64 x_stride, double* r, i64 r_stride,
                                double* r2, i64 r2_stride, const double* p, i64 p_stride,
                                const double* q, i64 q_stride, const double* q2, i64 q2_stride,
                                const double* beta, const double* rho, const u8* stop_status)
{
    for (i64 i = 0; i < n; ++i)
        for (i64 j = 0; j < nrhs; ++j) {
            if (st_has_stopped(stop_status[j])) continue;
            if (beta[j] != 0.0) {
                const double tmp = rho[j] / beta[j];
                AT(x, i, j) += tmp * AT(p, i, j);
                AT(r, i, j) -= tmp * AT(q, i, j);
                AT(r2, i, j) -= tmp * AT(q2, i, j);
            }
        }
}

void ref_csr_transpose(i64, i64, const i32*, const i32*, const double*, i32*, i32*, double*);

ORACLE_API i64 ref_bicg_solve(i64 n, const i32* row_ptrs, const i32* col_idxs, const double* vals,
                              const double* b, double* x, i64 max_iters, double reduction,
                              int baseline)
{
    const i64 nnz = row_ptrs[n];
    i32* trp = (i32*)calloc((size_t)n + 1, sizeof(i32));
    i32* tci = (i32*)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(i32));
    double* tv = vec(nnz);
    ref_csr_transpose(n, n, row_ptrs, col_idxs, vals, trp, tci, tv);
    double *r = vec(n), *z = vec(n), *p = vec(n), *q = vec(n), *r2 = vec(n), *z2 = vec(n),
           *p2 = vec(n), *q2 = vec(n);
    double beta, prev_rho, rho;
    u8 status, one_changed;
    ref_bicg_initialize(n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, &prev_rho, &rho, r2, 1, z2, 1, p2, 1, q2,
                        1, &status);
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, r, 1);
    memcpy(r2, r, sizeof(double) * (size_t)n);
    const double orig_tau = baseline_norm(baseline, n, b, r);
    i64 iter = -1;
    while (1) {
        memcpy(z, r, sizeof(double) * (size_t)n);
        memcpy(z2, r2, sizeof(double) * (size_t)n);
        ref_dense_compute_dot(n, 1, z, 1, r2, 1, &rho);
        ++iter;
        if (check(iter, max_iters, n, r, orig_tau, reduction, 1, &status, &one_changed)) break;
        ref_bicg_step_1(n, 1, p, 1, z, 1, p2, 1, z2, 1, &rho, &prev_rho, &status);
        ref_csr_spmv(n, 1, row_ptrs, col_idxs, vals, p, 1, q, 1);
        ref_csr_spmv(n, 1, trp, tci, tv, p2, 1, q2, 1);
        ref_dense_compute_dot(n, 1, p2, 1, q, 1, &beta);
        ref_bicg_step_2(n, 1, x, 1, r, 1, r2, 1, p, 1, q, 1, q2, 1, &beta, &rho, &status);
        double sw = prev_rho;
        prev_rho = rho;
        rho = sw;
    }
    free(r); free(z); free(p); free(q); free(r2); free(z2); free(p2); free(q2);
    free(trp); free(tci); free(tv);
    return iter;
}

/* ---- IR (core/solver/ir.cpp:186-277; ir_kernels.cpp:48-56) with the Identity as
 * inner solver = Richardson: x += relaxation_factor * (b - A x) -------------------- */
ORACLE_API void ref_ir_initialize(i64 nrhs, u8* stop_status)
{
    for (i64 j = 0; j < nrhs; ++j) stop_status[j] = 0;
}

ORACLE_API i64 ref_ir_solve(i64 n, const i32* row_ptrs, const i32* col_idxs, const double* vals,
                            double relaxation_factor, const double* b, double* x, i64 max_iters,
                            double reduction, int baseline)
{
    double* residual = vec(n);
    u8 status, one_changed;
    ref_ir_initialize(1, &status);
    memcpy(residual, b, sizeof(double) * (size_t)n);
    ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, residual, 1);
    const double orig_tau = baseline_norm(baseline, n, b, residual);
    i64 iter = -1;
    while (1) {
        ++iter;
        if (iter > 0) { /* residual = b - A x (the reference's two checks see the same vector) */
            memcpy(residual, b, sizeof(double) * (size_t)n);
            ref_csr_advanced_spmv(n, 1, -1.0, row_ptrs, col_idxs, vals, x, 1, 1.0, residual, 1);
        }
        if (check(iter, max_iters, n, residual, orig_tau, reduction, 1, &status, &one_changed)) break;
        for (i64 i = 0; i < n; ++i) x[i] += relaxation_factor * residual[i];
    }
    free(residual);
    return iter;
}
