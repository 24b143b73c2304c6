/* ORACLE (test infrastructure).  ELL / SELL-P / COO / Hybrid SpMV, the
 * CSR -> {COO, ELL, SELL-P, Hybrid} conversions and the index components. */
#include "oracle_common.h"

typedef uint64_t u64;
#define INVALID_INDEX (-1) /* invalid_index<IndexType>() */

/* reference/matrix/ell_kernels.cpp:57-99 (spmv) and :105-157 (advanced_spmv).
 * Column-major storage vals[row + i*stride]; padding has col == -1. */
ORACLE_API void ref_ell_spmv(i64 nrows, i64 nrhs, i64 num_stored_per_row,
                             i64 stride, const i32* col_idxs,
                             const double* vals, const double* b,
                             i64 b_stride, double* c, i64 c_stride)
{
    for (i64 j = 0; j < nrhs; ++j) {
        for (i64 row = 0; row < nrows; ++row) {
            double result = 0.0;
            for (i64 i = 0; i < num_stored_per_row; ++i) {
                const double val = vals[row + i * stride];
                const i32 col = col_idxs[row + i * stride];
                if (col != INVALID_INDEX) result += val * b[(i64)col * b_stride + j];
            }
            c[row * c_stride + j] = result;
        }
    }
}

ORACLE_API void ref_ell_advanced_spmv(i64 nrows, i64 nrhs, double alpha,
                                      i64 num_stored_per_row, i64 stride,
                                      const i32* col_idxs, const double* vals,
                                      const double* b, i64 b_stride,
                                      double beta, double* c, i64 c_stride)
{
    for (i64 j = 0; j < nrhs; ++j) {
        for (i64 row = 0; row < nrows; ++row) {
            double result = c[row * c_stride + j];
            result *= beta;
            for (i64 i = 0; i < num_stored_per_row; ++i) {
                const double val = vals[row + i * stride];
                const i32 col = col_idxs[row + i * stride];
                if (col != INVALID_INDEX)
                    result += alpha * val * b[(i64)col * b_stride + j];
            }
            c[row * c_stride + j] = result;
        }
    }
}

/* reference/matrix/sellp_kernels.cpp:57-89 / :95-131.  Element (row r of
 * slice s, i) lives at (slice_sets[s] + i) * slice_size + r (sellp.hpp:382). */
ORACLE_API void ref_sellp_spmv(i64 nrows, i64 nrhs, i64 slice_size,
                               const u64* slice_sets, const u64* slice_lengths,
                               const i32* col_idxs, const double* vals,
                               const double* b, i64 b_stride, double* c,
                               i64 c_stride)
{
    const i64 slice_num = (nrows + slice_size - 1 + slice_size - 1) / slice_size;
    for (i64 slice = 0; slice < slice_num; ++slice) {
        for (i64 row = 0; row < slice_size; ++row) {
            const i64 global_row = slice * slice_size + row;
            if (global_row >= nrows) break;
            for (i64 j = 0; j < nrhs; ++j) c[global_row * c_stride + j] = 0.0;
            for (u64 i = 0; i < slice_lengths[slice]; ++i) {
                const u64 idx = (slice_sets[slice] + i) * (u64)slice_size + (u64)row;
                const double val = vals[idx];
                const i32 col = col_idxs[idx];
                if (col != INVALID_INDEX)
                    for (i64 j = 0; j < nrhs; ++j)
                        c[global_row * c_stride + j] += val * b[(i64)col * b_stride + j];
            }
        }
    }
}

ORACLE_API void ref_sellp_advanced_spmv(i64 nrows, i64 nrhs, double alpha,
                                        i64 slice_size, const u64* slice_sets,
                                        const u64* slice_lengths,
                                        const i32* col_idxs, const double* vals,
                                        const double* b, i64 b_stride,
                                        double beta, double* c, i64 c_stride)
{
    const i64 slice_num = (nrows + slice_size - 1 + slice_size - 1) / slice_size;
    for (i64 slice = 0; slice < slice_num; ++slice) {
        for (i64 row = 0; row < slice_size; ++row) {
            const i64 global_row = slice * slice_size + row;
            if (global_row >= nrows) break;
            for (i64 j = 0; j < nrhs; ++j) c[global_row * c_stride + j] *= beta;
            for (u64 i = 0; i < slice_lengths[slice]; ++i) {
                const u64 idx = (slice_sets[slice] + i) * (u64)slice_size + (u64)row;
                const double val = vals[idx];
                const i32 col = col_idxs[idx];
                if (col != INVALID_INDEX)
                    for (i64 j = 0; j < nrhs; ++j)
                        c[global_row * c_stride + j] +=
                            alpha * val * b[(i64)col * b_stride + j];
            }
        }
    }
}

/* reference/matrix/coo_kernels.cpp:92-106 (spmv2), :112-131 (advanced_spmv2);
 * spmv = fill(0) + spmv2 (:63-71), advanced_spmv = scale(beta) + advanced_spmv2 (:77-88) */
ORACLE_API void ref_coo_spmv2(i64 nnz, i64 nrhs, const i32* row_idxs,
                              const i32* col_idxs, const double* vals,
                              const double* b, i64 b_stride, double* c,
                              i64 c_stride)
{
    for (i64 i = 0; i < nnz; ++i)
        for (i64 j = 0; j < nrhs; ++j)
            c[(i64)row_idxs[i] * c_stride + j] +=
                vals[i] * b[(i64)col_idxs[i] * b_stride + j];
}

ORACLE_API void ref_coo_advanced_spmv2(i64 nnz, i64 nrhs, double alpha,
                                       const i32* row_idxs, const i32* col_idxs,
                                       const double* vals, const double* b,
                                       i64 b_stride, double* c, i64 c_stride)
{
    for (i64 i = 0; i < nnz; ++i)
        for (i64 j = 0; j < nrhs; ++j)
            c[(i64)row_idxs[i] * c_stride + j] +=
                alpha * vals[i] * b[(i64)col_idxs[i] * b_stride + j];
}

ORACLE_API void ref_coo_spmv(i64 nrows, i64 nnz, i64 nrhs, const i32* row_idxs,
                             const i32* col_idxs, const double* vals,
                             const double* b, i64 b_stride, double* c,
                             i64 c_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j) c[i * c_stride + j] = 0.0;
    ref_coo_spmv2(nnz, nrhs, row_idxs, col_idxs, vals, b, b_stride, c, c_stride);
}

ORACLE_API void ref_coo_advanced_spmv(i64 nrows, i64 nnz, i64 nrhs,
                                      double alpha, const i32* row_idxs,
                                      const i32* col_idxs, const double* vals,
                                      const double* b, i64 b_stride,
                                      double beta, double* c, i64 c_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j) c[i * c_stride + j] *= beta;
    ref_coo_advanced_spmv2(nnz, nrhs, alpha, row_idxs, col_idxs, vals, b,
                           b_stride, c, c_stride);
}

/* reference/components/prefix_sum_kernels.cpp:43-53: exclusive, in place */
ORACLE_API void ref_prefix_sum_i32(i32* counts, i64 n)
{
    i32 partial = 0;
    for (i64 i = 0; i < n; ++i) {
        const i32 v = counts[i];
        counts[i] = partial;
        partial += v;
    }
}
ORACLE_API void ref_prefix_sum_i64(i64* counts, i64 n)
{
    i64 partial = 0;
    for (i64 i = 0; i < n; ++i) {
        const i64 v = counts[i];
        counts[i] = partial;
        partial += v;
    }
}

/* reference/components/format_conversion_kernels.cpp:50-62 */
ORACLE_API void ref_convert_ptrs_to_idxs(const i32* ptrs, i64 num_blocks,
                                         i32* idxs)
{
    for (i64 block = 0; block < num_blocks; ++block)
        for (i32 i = ptrs[block]; i < ptrs[block + 1]; ++i) idxs[i] = (i32)block;
}

/* :68-78 */
ORACLE_API void ref_convert_idxs_to_ptrs(const i32* idxs, i64 num_idxs,
                                         i64 num_blocks, i32* ptrs)
{
    for (i64 i = 0; i <= num_blocks; ++i) ptrs[i] = 0;
    for (i64 i = 0; i < num_idxs; ++i) ptrs[idxs[i]]++;
    ref_prefix_sum_i32(ptrs, num_blocks + 1);
}

/* :84-92 */
ORACLE_API void ref_convert_ptrs_to_sizes(const i32* ptrs, i64 num_blocks,
                                          u64* sizes)
{
    for (i64 b = 0; b < num_blocks; ++b) sizes[b] = (u64)(ptrs[b + 1] - ptrs[b]);
}

/* reference/matrix/ell_kernels.cpp:159-170 */
ORACLE_API i64 ref_compute_max_row_nnz(const i32* row_ptrs, i64 nrows)
{
    i64 m = 0;
    for (i64 i = 1; i <= nrows; ++i) {
        i64 d = row_ptrs[i] - row_ptrs[i - 1];
        if (d > m) m = d;
    }
    return m;
}

/* reference/matrix/csr_kernels.cpp:431-459 (convert_to_ell); ELL arrays are
 * stride * num_stored_per_row long */
ORACLE_API void ref_csr_convert_to_ell(i64 nrows, const i32* row_ptrs,
                                       const i32* col_idxs, const double* vals,
                                       i64 num_stored_per_row, i64 stride,
                                       i32* ell_cols, double* ell_vals)
{
    for (i64 row = 0; row < nrows; ++row) {
        for (i64 i = 0; i < num_stored_per_row; ++i) {
            ell_vals[row + i * stride] = 0.0;
            ell_cols[row + i * stride] = INVALID_INDEX;
        }
        for (i64 k = 0; k < row_ptrs[row + 1] - row_ptrs[row]; ++k) {
            ell_vals[row + k * stride] = vals[row_ptrs[row] + k];
            ell_cols[row + k * stride] = col_idxs[row_ptrs[row] + k];
        }
    }
}

/* reference/matrix/sellp_kernels.cpp:134-159 (compute_slice_sets):
 * slice_sets has num_slices + 1 entries */
ORACLE_API void ref_sellp_compute_slice_sets(const i32* row_ptrs, i64 nrows,
                                             i64 slice_size, i64 stride_factor,
                                             u64* slice_sets,
                                             u64* slice_lengths)
{
    const i64 num_slices = (nrows + slice_size - 1) / slice_size;
    for (i64 slice = 0; slice < num_slices; ++slice) {
        u64 len = 0;
        for (i64 lr = 0; lr < slice_size; ++lr) {
            const i64 row = slice * slice_size + lr;
            const i64 rl = row < nrows ? row_ptrs[row + 1] - row_ptrs[row] : 0;
            const u64 padded = (u64)((rl + stride_factor - 1) / stride_factor * stride_factor);
            if (padded > len) len = padded;
        }
        slice_lengths[slice] = len;
    }
    u64 partial = 0;
    for (i64 s = 0; s < num_slices; ++s) {
        slice_sets[s] = partial;
        partial += slice_lengths[s];
    }
    slice_sets[num_slices] = partial;
}

/* reference/matrix/csr_kernels.cpp:385-425 (convert_to_sellp) */
ORACLE_API void ref_csr_convert_to_sellp(i64 nrows, const i32* row_ptrs,
                                         const i32* col_idxs, const double* vals,
                                         i64 slice_size, const u64* slice_sets,
                                         const u64* slice_lengths,
                                         i32* out_cols, double* out_vals)
{
    const i64 slice_num = (nrows + slice_size - 1) / slice_size;
    for (i64 slice = 0; slice < slice_num; ++slice) {
        for (i64 row = 0; row < slice_size; ++row) {
            const i64 global_row = slice * slice_size + row;
            if (global_row >= nrows) break;
            u64 ind = slice_sets[slice] * (u64)slice_size + (u64)row;
            for (i32 k = row_ptrs[global_row]; k < row_ptrs[global_row + 1]; ++k) {
                out_vals[ind] = vals[k];
                out_cols[ind] = col_idxs[k];
                ind += (u64)slice_size;
            }
            const u64 end = (slice_sets[slice] + slice_lengths[slice]) * (u64)slice_size + (u64)row;
            for (u64 i = ind; i < end; i += (u64)slice_size) {
                out_cols[i] = INVALID_INDEX;
                out_vals[i] = 0.0;
            }
        }
    }
}

/* reference/matrix/hybrid_kernels.cpp:60-71 (compute_coo_row_ptrs): n+1 entries */
ORACLE_API void ref_hybrid_compute_coo_row_ptrs(const i32* row_ptrs, i64 nrows,
                                                i64 ell_lim, i64* coo_row_ptrs)
{
    for (i64 row = 0; row < nrows; ++row) {
        const i64 nnz = row_ptrs[row + 1] - row_ptrs[row];
        coo_row_ptrs[row] = nnz <= ell_lim ? 0 : nnz - ell_lim;
    }
    coo_row_ptrs[nrows] = 0;
    ref_prefix_sum_i64(coo_row_ptrs, nrows + 1);
}

/* reference/matrix/csr_kernels.cpp:768-812 (convert_to_hybrid) */
ORACLE_API void ref_csr_convert_to_hybrid(i64 nrows, const i32* row_ptrs,
                                          const i32* col_idxs, const double* vals,
                                          i64 ell_lim, i64 ell_stride,
                                          i32* ell_cols, double* ell_vals,
                                          i32* coo_rows, i32* coo_cols,
                                          double* coo_vals)
{
    for (i64 i = 0; i < ell_lim; ++i) {
        for (i64 j = 0; j < ell_stride; ++j) {
            ell_vals[j + i * ell_stride] = 0.0;
            ell_cols[j + i * ell_stride] = INVALID_INDEX;
        }
    }
    i64 csr_idx = 0, coo_idx = 0;
    for (i64 row = 0; row < nrows; ++row) {
        i64 ell_idx = 0;
        while (csr_idx < row_ptrs[row + 1]) {
            const double val = vals[csr_idx];
            if (ell_idx < ell_lim) {
                ell_vals[row + ell_idx * ell_stride] = val;
                ell_cols[row + ell_idx * ell_stride] = col_idxs[csr_idx];
                ell_idx++;
            } else {
                coo_vals[coo_idx] = val;
                coo_cols[coo_idx] = col_idxs[csr_idx];
                coo_rows[coo_idx] = (i32)row;
                coo_idx++;
            }
            csr_idx++;
        }
    }
}

/* Hybrid strategies, include/ginkgo/core/matrix/hybrid.hpp:206-370.
 * kind: 0 column_limit(num_columns), 1 imbalance_limit(percent),
 * 2 imbalance_bounded_limit(percent, ratio), 3 minimal_storage_limit,
 * 4 automatic = imbalance_bounded_limit(1/3, 0.001). */
static int cmp_u64(const void* a, const void* b)
{
    const u64 x = *(const u64*)a, y = *(const u64*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static u64 imbalance_limit(u64* row_nnz, i64 nrows, double percent)
{
    if (percent > 1.0) percent = 1.0;
    if (percent < 0.0) percent = 0.0;
    if (nrows == 0) return 0;
    qsort(row_nnz, (size_t)nrows, sizeof(u64), cmp_u64);
    if (percent < 1) return row_nnz[(u64)(nrows * percent)];
    return row_nnz[nrows - 1];
}
ORACLE_API i64 ref_hybrid_ell_width(const i32* row_ptrs, i64 nrows, int kind,
                                    double percent, double ratio,
                                    i64 num_columns)
{
    u64* row_nnz = (u64*)malloc(sizeof(u64) * (size_t)(nrows > 0 ? nrows : 1));
    ref_convert_ptrs_to_sizes(row_ptrs, nrows, row_nnz);
    u64 res = 0;
    if (kind == 0) {
        res = (u64)num_columns;
    } else if (kind == 1) {
        res = imbalance_limit(row_nnz, nrows, percent);
    } else if (kind == 2 || kind == 4) {
        if (kind == 4) {
            percent = 1.0 / 3.0;
            ratio = 0.001;
        }
        const u64 ell = imbalance_limit(row_nnz, nrows, percent);
        const u64 bound = (u64)(nrows * ratio);
        res = ell < bound ? ell : bound;
    } else {
        /* sizeof(IndexType) / (sizeof(ValueType) + 2 sizeof(IndexType)) for f64/i32 */
        res = imbalance_limit(row_nnz, nrows, 4.0 / (8.0 + 2 * 4.0));
    }
    free(row_nnz);
    return (i64)res;
}
