/* ORACLE (test infrastructure).  Dense BLAS-1 kernels,
 * reference/matrix/dense_kernels.cpp. */
#include "oracle_common.h"

/* :143-153 fill */
ORACLE_API void ref_dense_fill(i64 nrows, i64 ncols, double* x, i64 stride,
                               double value)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) x[i * stride + j] = value;
}

/* :126-138 copy */
ORACLE_API void ref_dense_copy(i64 nrows, i64 ncols, const double* in,
                               i64 in_stride, double* out, i64 out_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            out[i * out_stride + j] = in[i * in_stride + j];
}

/* :157-176 scale (alpha 1x1 or 1xncols) */
ORACLE_API void ref_dense_scale(i64 nrows, i64 ncols, const double* alpha,
                                i64 alpha_ncols, double* x, i64 stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            x[i * stride + j] *= alpha[alpha_ncols == 1 ? 0 : j];
}

/* :181-201 inv_scale */
ORACLE_API void ref_dense_inv_scale(i64 nrows, i64 ncols, const double* alpha,
                                    i64 alpha_ncols, double* x, i64 stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            x[i * stride + j] /= alpha[alpha_ncols == 1 ? 0 : j];
}

/* :207-226 add_scaled */
ORACLE_API void ref_dense_add_scaled(i64 nrows, i64 ncols, const double* alpha,
                                     i64 alpha_ncols, const double* x,
                                     i64 x_stride, double* y, i64 y_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            y[i * y_stride + j] +=
                alpha[alpha_ncols == 1 ? 0 : j] * x[i * x_stride + j];
}

/* :232-251 sub_scaled */
ORACLE_API void ref_dense_sub_scaled(i64 nrows, i64 ncols, const double* alpha,
                                     i64 alpha_ncols, const double* x,
                                     i64 x_stride, double* y, i64 y_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            y[i * y_stride + j] -=
                alpha[alpha_ncols == 1 ? 0 : j] * x[i * x_stride + j];
}

/* :282-297 compute_dot (== compute_conj_dot :313-328 for real values) */
ORACLE_API void ref_dense_compute_dot(i64 nrows, i64 ncols, const double* x,
                                      i64 x_stride, const double* y,
                                      i64 y_stride, double* result)
{
    for (i64 j = 0; j < ncols; ++j) result[j] = 0.0;
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            result[j] += x[i * x_stride + j] * y[i * y_stride + j];
}

/* :347-364 compute_norm2 */
ORACLE_API void ref_dense_compute_norm2(i64 nrows, i64 ncols, const double* x,
                                        i64 x_stride, double* result)
{
    for (i64 j = 0; j < ncols; ++j) result[j] = 0.0;
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            result[j] += x[i * x_stride + j] * x[i * x_stride + j];
    for (i64 j = 0; j < ncols; ++j) result[j] = sqrt(result[j]);
}

/* :416-431 compute_squared_norm2 */
ORACLE_API void ref_dense_compute_squared_norm2(i64 nrows, i64 ncols,
                                                const double* x, i64 x_stride,
                                                double* result)
{
    for (i64 j = 0; j < ncols; ++j) result[j] = 0.0;
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j)
            result[j] += x[i * x_stride + j] * x[i * x_stride + j];
}

/* :381-396 compute_norm1 */
ORACLE_API void ref_dense_compute_norm1(i64 nrows, i64 ncols, const double* x,
                                        i64 x_stride, double* result)
{
    for (i64 j = 0; j < ncols; ++j) result[j] = 0.0;
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) result[j] += fabs(x[i * x_stride + j]);
}

/* :435-447 compute_sqrt */
ORACLE_API void ref_dense_compute_sqrt(i64 nrows, i64 ncols, double* x,
                                       i64 stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < ncols; ++j) x[i * stride + j] = sqrt(x[i * stride + j]);
}

/* row_gather (reference/matrix/dense_kernels.cpp, dense::row_gather) */
ORACLE_API void ref_dense_row_gather(i64 nout, i64 ncols, const i32* rows,
                                     const double* in, i64 in_stride,
                                     double* out, i64 out_stride)
{
    for (i64 i = 0; i < nout; ++i)
        for (i64 j = 0; j < ncols; ++j)
            out[i * out_stride + j] = in[(i64)rows[i] * in_stride + j];
}

/* OpenMP baseline for the reductions: one contiguous chunk per thread ->
 * partial[tid] -> sequential sum over threads
 * (omp/base/kernel_launch_reduction.hpp:65-86). Vector (ncols==1) case. */
#ifdef _OPENMP
#include <omp.h>
#endif
ORACLE_API double omp_dense_dot_vec(i64 n, const double* x, const double* y)
{
    double total = 0.0;
#ifdef _OPENMP
    int nt = omp_get_max_threads();
    double* partial = (double*)calloc((size_t)nt * 8, sizeof(double));
#pragma omp parallel num_threads(nt)
    {
        int tid = omp_get_thread_num();
        i64 per = (n + nt - 1) / nt;
        i64 lo = tid * per, hi = lo + per < n ? lo + per : n;
        double acc = 0.0;
        for (i64 i = lo; i < hi; ++i) acc += x[i] * y[i];
        partial[tid * 8] = acc;
    }
    for (int t = 0; t < nt; ++t) total += partial[t * 8];
    free(partial);
#else
    for (i64 i = 0; i < n; ++i) total += x[i] * y[i];
#endif
    return total;
}

/* y += alpha x, omp (common/unified/matrix/dense_kernels.cpp add_scaled via
 * omp/base/kernel_launch.hpp: parallel for over elements) */
ORACLE_API void omp_dense_axpy_vec(i64 n, double alpha, const double* x,
                                   double* y)
{
#pragma omp parallel for
    for (i64 i = 0; i < n; ++i) y[i] += alpha * x[i];
}

ORACLE_API int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
