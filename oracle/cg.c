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

/* ---- CPU baseline of bench.py (never the checker) -----------------------------
 * The reference's omp/ executor on the host cores of the GPU box, restated:
 *   - csr::spmv            omp/matrix/csr_kernels.cpp:76-99 (`parallel for` over rows)
 *   - Cg::apply_dense_impl core/solver/cg.cpp:107-193 with the Identity
 *     preconditioner: every iteration copies r into z (matrix::Identity::apply),
 *     rho = dot(r, z), norm2(r) for the criterion, step_1, SpMV, dot(p, q),
 *     step_2 -- each its own pass over memory, as the omp executor runs them
 *     (common/unified/solver/cg_kernels.cpp:53-131 through
 *     omp/base/kernel_launch.hpp; reductions omp/base/kernel_launch_reduction.hpp:65-86:
 *     one contiguous chunk per thread, partials added in thread order).
 * Memory is FIRST TOUCHED by the thread that will stream it (static row
 * partition), so that on a multi-socket host every NUMA node serves its own
 * share (OMP_PROC_BIND=true; bench.py sets it before this library
 * loads).  The results depend on the thread count to rounding and are compared
 * with nothing. */
#ifdef _OPENMP
#include <omp.h>
#endif

void omp_csr_spmv(i64, i64, const i32*, const i32*, const double*,
                  const double*, i64, double*, i64);

typedef struct {
    i64 n, nnz;
    i32* rp;
    i32* ci;
    double* v;
    double* x;
    double* y;
} omp_bench_ctx;

static int omp_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* sum_i a[i] * b[i], the omp executor's reduction shape */
static double omp_ref_dot(i64 n, const double* a, const double* b, double* partial, int nt)
{
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        const i64 per = (n + nt - 1) / nt;
        const i64 lo = tid * per, hi = lo + per < n ? lo + per : n;
        double acc = 0.0;
        for (i64 i = lo; i < hi; ++i) acc += a[i] * b[i];
        partial[tid * 8] = acc;
    }
    double total = 0.0;
    for (int t = 0; t < nt; ++t) total += partial[t * 8];
    return total;
}

ORACLE_API u64 omp_bench_create(i64 n, const i32* rp, const i32* ci, const double* v,
                                const double* x)
{
    omp_bench_ctx* c = (omp_bench_ctx*)malloc(sizeof(omp_bench_ctx));
    c->n = n;
    c->nnz = rp[n];
    c->rp = (i32*)malloc(sizeof(i32) * (size_t)(n + 1));
    c->ci = (i32*)malloc(sizeof(i32) * (size_t)(c->nnz > 0 ? c->nnz : 1));
    c->v = (double*)malloc(sizeof(double) * (size_t)(c->nnz > 0 ? c->nnz : 1));
    c->x = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    c->y = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    /* same static schedule as the SpMV: a thread touches the rows it will own */
#pragma omp parallel for schedule(static)
    for (i64 row = 0; row < n; ++row) {
        c->rp[row] = rp[row];
        for (i64 k = rp[row]; k < rp[row + 1]; ++k) {
            c->ci[k] = ci[k];
            c->v[k] = v[k];
        }
        c->x[row] = x[row];
        c->y[row] = 0.0;
    }
    c->rp[n] = rp[n];
    return (u64)(size_t)c;
}

ORACLE_API void omp_bench_destroy(u64 h)
{
    omp_bench_ctx* c = (omp_bench_ctx*)(size_t)h;
    free(c->rp);
    free(c->ci);
    free(c->v);
    free(c->x);
    free(c->y);
    free(c);
}

ORACLE_API int omp_bench_threads(void) { return omp_threads(); }

/* reps x (y = A x) */
ORACLE_API void omp_bench_spmv(u64 h, i64 reps)
{
    omp_bench_ctx* c = (omp_bench_ctx*)(size_t)h;
    for (i64 r = 0; r < reps; ++r) omp_csr_spmv(c->n, 1, c->rp, c->ci, c->v, c->x, 1, c->y, 1);
}

ORACLE_API void omp_bench_get_y(u64 h, double* out)
{
    omp_bench_ctx* c = (omp_bench_ctx*)(size_t)h;
    for (i64 i = 0; i < c->n; ++i) out[i] = c->y[i];
}

/* Cg::apply_dense_impl, omp kernels, x0 = 0, criterion ResidualNorm(reduction, rhs_norm)
 * + Iteration(max_iters); returns the iteration count */
ORACLE_API i64 omp_bench_cg(u64 h, const double* b_in, double* x_out, i64 max_iters,
                            double reduction, double* final_rel)
{
    omp_bench_ctx* c = (omp_bench_ctx*)(size_t)h;
    const i64 n = c->n;
    const int nt = omp_threads();
    double* partial = (double*)calloc((size_t)nt * 8, sizeof(double));
    double *b = (double*)malloc(sizeof(double) * (size_t)n), *x = (double*)malloc(sizeof(double) * (size_t)n);
    double *r = (double*)malloc(sizeof(double) * (size_t)n), *z = (double*)malloc(sizeof(double) * (size_t)n);
    double *p = (double*)malloc(sizeof(double) * (size_t)n), *q = (double*)malloc(sizeof(double) * (size_t)n);
    /* cg::initialize: r = b, z = p = q = 0 (first touch with the partition of the loops below) */
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) {
        b[i] = b_in[i];
        x[i] = 0.0;
        r[i] = b_in[i];
        z[i] = p[i] = q[i] = 0.0;
    }
    /* r = -1 * A x + 1 * r  (x = 0: the advanced apply leaves r = b, still a pass) */
    omp_csr_spmv(n, 1, c->rp, c->ci, c->v, x, 1, q, 1);
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) r[i] = r[i] - q[i];
    const double orig_tau = sqrt(omp_ref_dot(n, b, b, partial, nt));
    double prev_rho = 1.0, rho = 0.0, tau = 0.0;
    i64 iter = -1;
    while (1) {
        /* z = Identity * r */
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < n; ++i) z[i] = r[i];
        rho = omp_ref_dot(n, r, z, partial, nt);
        ++iter;
        tau = sqrt(omp_ref_dot(n, r, r, partial, nt)); /* ResidualNorm: norm2(r) */
        if (iter >= max_iters || tau < reduction * orig_tau) break;
        /* step_1 */
        const double tmp = prev_rho == 0.0 ? 0.0 : rho / prev_rho;
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < n; ++i) p[i] = prev_rho == 0.0 ? z[i] : z[i] + tmp * p[i];
        omp_csr_spmv(n, 1, c->rp, c->ci, c->v, p, 1, q, 1);
        const double beta = omp_ref_dot(n, p, q, partial, nt);
        /* step_2 */
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
    for (i64 i = 0; i < n; ++i) x_out[i] = x[i];
    free(partial); free(b); free(x); free(r); free(z); free(p); free(q);
    return iter;
}
