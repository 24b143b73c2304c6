/* ORACLE (test infrastructure).  Block-Jacobi: find_blocks, generate, apply;
 * scalar Jacobi.  reference/preconditioner/jacobi_kernels.cpp.
 * Full-precision block storage only (precision_reduction(0,0): the adaptive
 * precision path is outside the fp64 scope). */
#include "oracle_common.h"

/* block_interleaved_storage_scheme (include/ginkgo/core/preconditioner/jacobi.hpp:62-167)
 * and Jacobi::compute_storage_scheme (:578-609).  scheme[3] = {block_offset,
 * group_offset, group_power}. max_block_stride = 0 -> default (32 on CPU, the
 * wavefront size 64 on HIP). */
static i64 superior_power2(i64 v)
{
    i64 p = 1;
    while (p < v) p *= 2;
    return p;
}
ORACLE_API void ref_jacobi_storage_scheme(i64 max_block_size,
                                          i64 max_block_stride, i64* scheme)
{
    const i64 group_size = max_block_stride / superior_power2(max_block_size);
    const i64 block_offset = max_block_size;
    const i64 block_stride = group_size * block_offset;
    const i64 group_offset = max_block_size * block_stride;
    i64 gp = 0;
    while (((i64)1 << (gp + 1)) <= group_size) ++gp;
    scheme[0] = block_offset;
    scheme[1] = group_offset;
    scheme[2] = gp;
}
static i64 sch_group_size(const i64* s) { return (i64)1 << s[2]; }
static i64 sch_stride(const i64* s) { return s[0] << s[2]; }
static i64 sch_global_offset(const i64* s, i64 b)
{
    return s[1] * (b >> s[2]) + s[0] * (b & (sch_group_size(s) - 1));
}
ORACLE_API i64 ref_jacobi_storage_space(const i64* scheme, i64 num_blocks)
{
    const i64 gs = sch_group_size(scheme);
    return (num_blocks + gs - 1) / gs * scheme[1];
}

/* :66-137 find_natural_blocks + agglomerate_supervariables; returns num_blocks,
 * block_ptrs needs nrows + 1 entries */
ORACLE_API i64 ref_jacobi_find_blocks(i64 nrows, const i32* row_ptrs,
                                      const i32* col_idxs, i64 max_block_size,
                                      i32* block_ptrs)
{
    block_ptrs[0] = 0;
    if (nrows == 0) return 0;
    i64 num_blocks = 1;
    i32 current = 1;
    for (i64 i = 1; i < nrows; ++i) {
        const i32* prev = col_idxs + row_ptrs[i - 1];
        const i32* curr = col_idxs + row_ptrs[i];
        const i32* next = col_idxs + row_ptrs[i + 1];
        int same = (next - curr) == (curr - prev);
        if (same) same = memcmp(curr, prev, (size_t)(next - curr) * sizeof(i32)) == 0;
        if (current < max_block_size && same) {
            ++current;
        } else {
            block_ptrs[num_blocks] = block_ptrs[num_blocks - 1] + current;
            ++num_blocks;
            current = 1;
        }
    }
    block_ptrs[num_blocks] = block_ptrs[num_blocks - 1] + current;
    /* agglomerate */
    const i64 natural = num_blocks;
    num_blocks = 1;
    i32 cur = block_ptrs[1] - block_ptrs[0];
    for (i64 i = 1; i < natural; ++i) {
        const i32 bs = block_ptrs[i + 1] - block_ptrs[i];
        if (cur + bs <= max_block_size) {
            cur += bs;
        } else {
            block_ptrs[num_blocks] = block_ptrs[i];
            ++num_blocks;
            cur = bs;
        }
    }
    block_ptrs[num_blocks] = block_ptrs[natural];
    return num_blocks;
}

/* matrix_operations.hpp:51-66 (column-major argument, as called on row-major
 * blocks it yields max column sum -- reproduced as is) */
static double inf_norm(i64 nr, i64 nc, const double* m, i64 stride)
{
    double result = 0.0;
    for (i64 i = 0; i < nr; ++i) {
        double tmp = 0.0;
        for (i64 j = 0; j < nc; ++j) tmp += fabs(m[i + j * stride]);
        if (tmp > result) result = tmp;
    }
    return result;
}

/* :163-183 */
static void extract_block(const i32* rp, const i32* ci, const double* v, i32 bs,
                          i32 start, double* block, i64 stride)
{
    for (i32 i = 0; i < bs; ++i)
        for (i32 j = 0; j < bs; ++j) block[i * stride + j] = 0.0;
    for (i32 row = 0; row < bs; ++row) {
        for (i32 k = rp[start + row]; k < rp[start + row + 1]; ++k) {
            const i32 col = ci[k] - start;
            if (0 <= col && col < bs) block[row * stride + col] = v[k];
        }
    }
}

/* :186-240, :295-312: Gauss-Jordan with implicit (row) pivoting */
static int invert_block(i32 bs, i32* perm, double* block, i64 stride)
{
    for (i32 k = 0; k < bs; ++k) {
        i32 cp = 0;
        const double* col = block + k * stride + k;
        for (i32 i = 1; i < bs - k; ++i)
            if (fabs(col[cp * stride]) < fabs(col[i * stride])) cp = i;
        cp += k;
        for (i32 i = 0; i < bs; ++i) {
            double t = block[k * stride + i];
            block[k * stride + i] = block[cp * stride + i];
            block[cp * stride + i] = t;
        }
        i32 tp = perm[k];
        perm[k] = perm[cp];
        perm[cp] = tp;
        const double d = block[k * stride + k];
        if (d == 0.0) return 0;
        for (i32 i = 0; i < bs; ++i) block[i * stride + k] /= -d;
        block[k * stride + k] = 0.0;
        for (i32 i = 0; i < bs; ++i)
            for (i32 j = 0; j < bs; ++j)
                block[i * stride + j] += block[i * stride + k] * block[k * stride + j];
        for (i32 j = 0; j < bs; ++j) block[k * stride + j] /= d;
        block[k * stride + k] = 1.0 / d;
    }
    return 1;
}

/* :339-441 generate (prec == full precision for every block); conditioning
 * may be NULL; blocks must hold ref_jacobi_storage_space entries */
ORACLE_API void ref_jacobi_generate(i64 nrows, const i32* row_ptrs,
                                    const i32* col_idxs, const double* vals,
                                    i64 num_blocks, const i64* scheme,
                                    const i32* block_ptrs, double* conditioning,
                                    double* blocks)
{
    (void)nrows;
    const i64 stride = sch_stride(scheme);
    for (i64 b = 0; b < num_blocks; ++b) {
        const i32 bs = block_ptrs[b + 1] - block_ptrs[b];
        double* block = (double*)malloc(sizeof(double) * (size_t)(bs * bs + 1));
        i32* perm = (i32*)malloc(sizeof(i32) * (size_t)(bs + 1));
        for (i32 i = 0; i < bs; ++i) perm[i] = i;
        extract_block(row_ptrs, col_idxs, vals, bs, block_ptrs[b], block, bs);
        if (conditioning) conditioning[b] = inf_norm(bs, bs, block, bs);
        invert_block(bs, perm, block, bs);
        if (conditioning) conditioning[b] *= inf_norm(bs, bs, block, bs);
        /* permute_and_transpose_block :277-292 */
        double* out = blocks + sch_global_offset(scheme, b);
        for (i32 i = 0; i < bs; ++i)
            for (i32 j = 0; j < bs; ++j)
                out[i + perm[j] * stride] = block[i * bs + j];
        free(block);
        free(perm);
    }
}

/* :447-479 apply_block; :505-561 apply / simple_apply */
static void apply_block(i64 bs, i64 nrhs, const double* block, i64 stride,
                        double alpha, const double* b, i64 stride_b,
                        double beta, double* x, i64 stride_x)
{
    if (beta != 0.0) {
        for (i64 row = 0; row < bs; ++row)
            for (i64 col = 0; col < nrhs; ++col) x[row * stride_x + col] *= beta;
    } else {
        for (i64 row = 0; row < bs; ++row)
            for (i64 col = 0; col < nrhs; ++col) x[row * stride_x + col] = 0.0;
    }
    for (i64 inner = 0; inner < bs; ++inner)
        for (i64 row = 0; row < bs; ++row)
            for (i64 col = 0; col < nrhs; ++col)
                x[row * stride_x + col] +=
                    alpha * block[row + inner * stride] * b[inner * stride_b + col];
}

ORACLE_API void ref_jacobi_apply(i64 num_blocks, const i64* scheme,
                                 const i32* block_ptrs, const double* blocks,
                                 i64 nrhs, double alpha, const double* b,
                                 i64 b_stride, double beta, double* x,
                                 i64 x_stride)
{
    for (i64 i = 0; i < num_blocks; ++i) {
        const i64 bs = block_ptrs[i + 1] - block_ptrs[i];
        apply_block(bs, nrhs, blocks + sch_global_offset(scheme, i),
                    sch_stride(scheme), alpha, b + b_stride * block_ptrs[i],
                    b_stride, beta, x + x_stride * block_ptrs[i], x_stride);
    }
}

ORACLE_API void ref_jacobi_simple_apply(i64 num_blocks, const i64* scheme,
                                        const i32* block_ptrs,
                                        const double* blocks, i64 nrhs,
                                        const double* b, i64 b_stride,
                                        double* x, i64 x_stride)
{
    ref_jacobi_apply(num_blocks, scheme, block_ptrs, blocks, nrhs, 1.0, b,
                     b_stride, 0.0, x, x_stride);
}

/* csr::extract_diagonal (reference/matrix/csr_kernels.cpp) + :608-620 invert_diagonal */
ORACLE_API void ref_csr_extract_diagonal(i64 nrows, const i32* row_ptrs,
                                         const i32* col_idxs, const double* vals,
                                         double* diag)
{
    for (i64 row = 0; row < nrows; ++row) {
        diag[row] = 0.0;
        for (i32 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            if (col_idxs[k] == row) {
                diag[row] = vals[k];
                break;
            }
        }
    }
}
ORACLE_API void ref_jacobi_invert_diagonal(i64 n, const double* diag,
                                           double* inv_diag)
{
    for (i64 i = 0; i < n; ++i) inv_diag[i] = 1.0 / (diag[i] == 0.0 ? 1.0 : diag[i]);
}
/* :565-578 scalar_apply, :583-594 simple_scalar_apply */
ORACLE_API void ref_jacobi_scalar_apply(i64 nrows, i64 nrhs, const double* diag,
                                        double alpha, const double* b,
                                        i64 b_stride, double beta, double* x,
                                        i64 x_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j)
            x[i * x_stride + j] =
                beta * x[i * x_stride + j] + alpha * b[i * b_stride + j] * diag[i];
}
ORACLE_API void ref_jacobi_simple_scalar_apply(i64 nrows, i64 nrhs,
                                               const double* diag,
                                               const double* b, i64 b_stride,
                                               double* x, i64 x_stride)
{
    for (i64 i = 0; i < nrows; ++i)
        for (i64 j = 0; j < nrhs; ++j)
            x[i * x_stride + j] = b[i * b_stride + j] * diag[i];
}
