/* ORACLE (test infrastructure).  CSR kernels. */
#include "oracle_common.h"

/* reference/matrix/csr_kernels.cpp:75-96 (csr::spmv) */
ORACLE_API void ref_csr_spmv(i64 nrows, i64 nrhs, const i32* row_ptrs,
                             const i32* col_idxs, const double* vals,
                             const double* b, i64 b_stride, double* c,
                             i64 c_stride)
{
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 j = 0; j < nrhs; ++j) c[row * c_stride + j] = 0.0;
        for (i64 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const double val = vals[k];
            const i64 col = col_idxs[k];
            for (i64 j = 0; j < nrhs; ++j) {
                c[row * c_stride + j] += val * b[col * b_stride + j];
            }
        }
    }
}

/* reference/matrix/csr_kernels.cpp:102-128 (csr::advanced_spmv) */
ORACLE_API void ref_csr_advanced_spmv(i64 nrows, i64 nrhs, double alpha,
                                      const i32* row_ptrs, const i32* col_idxs,
                                      const double* vals, const double* b,
                                      i64 b_stride, double beta, double* c,
                                      i64 c_stride)
{
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 j = 0; j < nrhs; ++j) c[row * c_stride + j] *= beta;
        for (i64 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const double val = vals[k];
            const i64 col = col_idxs[k];
            for (i64 j = 0; j < nrhs; ++j) {
                c[row * c_stride + j] += alpha * val * b[col * b_stride + j];
            }
        }
    }
}

/* omp/matrix/csr_kernels.cpp:76-99: `#pragma omp parallel for` over rows,
 * same loop nest -- the CPU baseline timed by
 * bench.py (kind "port"). */
ORACLE_API void omp_csr_spmv(i64 nrows, i64 nrhs, const i32* row_ptrs,
                             const i32* col_idxs, const double* vals,
                             const double* b, i64 b_stride, double* c,
                             i64 c_stride)
{
#pragma omp parallel for
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 j = 0; j < nrhs; ++j) c[row * c_stride + j] = 0.0;
        for (i64 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const double val = vals[k];
            const i64 col = col_idxs[k];
            for (i64 j = 0; j < nrhs; ++j) {
                c[row * c_stride + j] += val * b[col * b_stride + j];
            }
        }
    }
}

/* omp/matrix/csr_kernels.cpp:104-131 */
ORACLE_API void omp_csr_advanced_spmv(i64 nrows, i64 nrhs, double alpha,
                                      const i32* row_ptrs, const i32* col_idxs,
                                      const double* vals, const double* b,
                                      i64 b_stride, double beta, double* c,
                                      i64 c_stride)
{
#pragma omp parallel for
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 j = 0; j < nrhs; ++j) c[row * c_stride + j] *= beta;
        for (i64 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const double val = vals[k];
            const i64 col = col_idxs[k];
            for (i64 j = 0; j < nrhs; ++j) {
                c[row * c_stride + j] += alpha * val * b[col * b_stride + j];
            }
        }
    }
}
