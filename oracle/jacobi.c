/* ORACLE (test infrastructure).  Block-Jacobi: find_blocks, generate, apply;
 * scalar Jacobi.  reference/preconditioner/jacobi_kernels.cpp.
 * Block storage in full precision (ref_jacobi_generate / _apply) and with the
 * adaptive per-group storage precision (ref_jacobi_generate_adaptive /
 * _apply_adaptive, precision_reduction encoded as in
 * include/ginkgo/core/base/types.hpp:257-368: (preserving << 4) | nonpreserving,
 * autodetect = 0xff). */
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

/* ---- adaptive precision block storage ------------------------------------
 * Reduced storage types for ValueType = double
 * (core/preconditioner/jacobi_utils.hpp:46-69, core/base/extended_float.hpp):
 *   (0,0) double                    (0,1) float
 *   (0,2) half                      (1,0) truncated<double,2> = upper 32 bits
 *   (1,1) truncated<float,2> = upper 16 bits of the float
 *   (2,0) truncated<double,4> = upper 16 bits of the double
 * half follows the HOST conversion of the reference (extended_float.hpp:357-376):
 * the significand is truncated, results below the half normal range are
 * flushed to signed zero, overflow gives infinity. */
enum { PR_P0N0 = 0x00, PR_P0N1 = 0x01, PR_P0N2 = 0x02, PR_P1N0 = 0x10, PR_P1N1 = 0x11, PR_P2N0 = 0x20 };
/* precision_reduction_descriptor (jacobi_utils.hpp:84-112) */
enum { D_P0N0 = 0x00, D_P0N2 = 0x01, D_P1N1 = 0x02, D_P2N0 = 0x04, D_P0N1 = 0x08, D_P1N0 = 0x10 };

static uint16_t float2half(uint32_t f)
{
    const uint16_t sign = (uint16_t)((f >> 16) & 0x8000u);
    const uint32_t exp_bits = f & 0x7f800000u;
    const uint32_t sig = f & 0x007fffffu;
    if (exp_bits == 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (sig ? 0x03ffu : 0u));
    /* shift_exponent: (exp >> 13) - bias_change, clamped to [0, 0x7c00] */
    const uint32_t e = exp_bits >> 13;
    const uint32_t bias_change = (0x3f800000u >> 13) - 0x3c00u;
    uint32_t he = e <= bias_change ? 0u : e - bias_change;
    if (he >= 0x7c00u) he = 0x7c00u;
    if (he == 0x7c00u) return (uint16_t)(sign | 0x7c00u); /* is_inf(exp) */
    if (he == 0u) return sign;                            /* is_denom: flushed */
    return (uint16_t)(sign | he | (sig >> 13));
}

static uint32_t half2float(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t exp_bits = h & 0x7c00u;
    const uint32_t sig = h & 0x03ffu;
    if (exp_bits == 0x7c00u) return sign | 0x7f800000u | (sig ? 0x007fffffu : 0u);
    if (exp_bits == 0u) return sign; /* denormals flushed */
    const uint32_t bias_change = 0x3f800000u - (0x3c00u << 13);
    return sign | ((exp_bits << 13) + bias_change) | (sig << 13);
}

/* static_cast<resolved_precision>(v) stored at element idx of base */
static void pr_store(u8 p, void* base, i64 idx, double v)
{
    uint64_t b64;
    uint32_t b32;
    float f = (float)v;
    memcpy(&b64, &v, 8);
    memcpy(&b32, &f, 4);
    switch (p) {
    case PR_P0N1: ((float*)base)[idx] = f; break;
    case PR_P0N2: ((uint16_t*)base)[idx] = float2half(b32); break;
    case PR_P1N0: ((uint32_t*)base)[idx] = (uint32_t)(b64 >> 32); break;
    case PR_P1N1: ((uint16_t*)base)[idx] = (uint16_t)(b32 >> 16); break;
    case PR_P2N0: ((uint16_t*)base)[idx] = (uint16_t)(b64 >> 48); break;
    default: ((double*)base)[idx] = v; break;
    }
}

/* default_converter<resolved_precision, double> */
static double pr_load(u8 p, const void* base, i64 idx)
{
    uint64_t b64;
    uint32_t b32;
    float f;
    double d;
    switch (p) {
    case PR_P0N1: return (double)((const float*)base)[idx];
    case PR_P0N2: b32 = half2float(((const uint16_t*)base)[idx]); memcpy(&f, &b32, 4); return (double)f;
    case PR_P1N0: b64 = (uint64_t)((const uint32_t*)base)[idx] << 32; memcpy(&d, &b64, 8); return d;
    case PR_P1N1: b32 = (uint32_t)((const uint16_t*)base)[idx] << 16; memcpy(&f, &b32, 4); return (double)f;
    case PR_P2N0: b64 = (uint64_t)((const uint16_t*)base)[idx] << 48; memcpy(&d, &b64, 8); return d;
    default: return ((const double*)base)[idx];
    }
}

/* round trip double -> reduced -> double (tests) */
ORACLE_API double ref_jacobi_round_to_precision(int precision, double v)
{
    double slot[1];
    pr_store((u8)precision, slot, 0, v);
    return pr_load((u8)precision, slot, 0);
}

/* :311-336 validate_precision_reduction_feasibility<reduced> on the inverted
 * block (row-major, stride bs) */
static int feasible(u8 reduced, i32 bs, const double* block)
{
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(bs * bs + 1));
    i32* perm = (i32*)malloc(sizeof(i32) * (size_t)(bs + 1));
    for (i32 i = 0; i < bs; ++i) perm[i] = i;
    for (i32 i = 0; i < bs * bs; ++i) tmp[i] = ref_jacobi_round_to_precision(reduced, block[i]);
    double cond = inf_norm(bs, bs, tmp, bs);
    const int ok = invert_block(bs, perm, tmp, bs);
    int result = 0;
    if (ok) {
        cond *= inf_norm(bs, bs, tmp, bs);
        result = cond >= 1.0 && cond * 0x1p-53 < 1e-3;
    }
    free(tmp);
    free(perm);
    return result;
}

static uint32_t singleton(u8 pr)
{
    return pr == PR_P0N1 ? D_P0N1 : pr == PR_P0N2 ? D_P0N2 : pr == PR_P1N0 ? D_P1N0
           : pr == PR_P1N1 ? D_P1N1 : pr == PR_P2N0 ? D_P2N0 : D_P0N0;
}

/* jacobi_utils.hpp:129-167; eps = 2^-(significand_bits + rounds_to_nearest) */
static uint32_t supported_reductions(double accuracy, double cond, i32 bs, const double* block)
{
#define ACCURATE(eps) (cond * (eps) < accuracy)
    int verified1 = 2;
    uint32_t supported = D_P0N0;
    if (ACCURATE(0x1p-4)) supported |= D_P2N0;                                          /* truncated<double,4> */
    if (ACCURATE(0x1p-7) && (verified1 = feasible(PR_P0N1, bs, block))) supported |= D_P1N1; /* truncated<float,2> */
    if (ACCURATE(0x1p-11) && verified1 != 0 && feasible(PR_P0N2, bs, block)) supported |= D_P0N2; /* half */
    if (ACCURATE(0x1p-20)) supported |= D_P1N0;                                         /* truncated<double,2> */
    if (ACCURATE(0x1p-24) &&
        (verified1 == 1 || (verified1 == 2 && (verified1 = feasible(PR_P0N1, bs, block))))) {
        supported |= D_P0N1; /* float */
    }
#undef ACCURATE
    return supported;
}

/* jacobi_utils.hpp:184-201 */
static u8 optimal_reduction(uint32_t supported)
{
    if (supported & D_P0N2) return PR_P0N2;
    if (supported & D_P1N1) return PR_P1N1;
    if (supported & D_P2N0) return PR_P2N0;
    if (supported & D_P0N1) return PR_P0N1;
    if (supported & D_P1N0) return PR_P1N0;
    return PR_P0N0;
}

/* :339-441 generate with block_precisions (in: requested per block, 0xff =
 * autodetect; out: the precision used, common to a group); conditioning is
 * required for autodetection */
ORACLE_API void ref_jacobi_generate_adaptive(i64 nrows, const i32* row_ptrs, const i32* col_idxs,
                                             const double* vals, i64 num_blocks,
                                             const i64* scheme, const i32* block_ptrs,
                                             double accuracy, double* conditioning,
                                             u8* block_precisions, double* blocks)
{
    (void)nrows;
    const i64 stride = sch_stride(scheme);
    const i64 gs = sch_group_size(scheme);
    double** block = (double**)malloc(sizeof(double*) * (size_t)gs);
    i32** perm = (i32**)malloc(sizeof(i32*) * (size_t)gs);
    for (i64 g = 0; g < num_blocks; g += gs) {
        uint32_t descriptors = ~(uint32_t)0;
        for (i64 b = 0; b < gs && g + b < num_blocks; ++b) {
            const i32 bs = block_ptrs[g + b + 1] - block_ptrs[g + b];
            block[b] = (double*)malloc(sizeof(double) * (size_t)(bs * bs + 1));
            perm[b] = (i32*)malloc(sizeof(i32) * (size_t)(bs + 1));
            for (i32 i = 0; i < bs; ++i) perm[b][i] = i;
            extract_block(row_ptrs, col_idxs, vals, bs, block_ptrs[g + b], block[b], bs);
            if (conditioning) conditioning[g + b] = inf_norm(bs, bs, block[b], bs);
            invert_block(bs, perm[b], block[b], bs);
            if (conditioning) conditioning[g + b] *= inf_norm(bs, bs, block[b], bs);
            const u8 local = block_precisions ? block_precisions[g + b] : PR_P0N0;
            if (local == 0xff && conditioning) {
                descriptors &= supported_reductions(accuracy, conditioning[g + b], bs, block[b]);
            } else {
                descriptors &= singleton(local);
            }
        }
        const u8 p = optimal_reduction(descriptors);
        for (i64 b = 0; b < gs && g + b < num_blocks; ++b) {
            if (block_precisions) block_precisions[g + b] = p;
            const i32 bs = block_ptrs[g + b + 1] - block_ptrs[g + b];
            void* group = blocks + scheme[1] * ((g + b) >> scheme[2]);
            const i64 block_ofs = scheme[0] * ((g + b) & (gs - 1));
            for (i32 i = 0; i < bs; ++i)
                for (i32 j = 0; j < bs; ++j)
                    pr_store(p, group, block_ofs + i + perm[b][j] * stride, block[b][i * bs + j]);
            free(block[b]);
            free(perm[b]);
        }
    }
    free(block);
    free(perm);
}

/* :497-561 apply / simple_apply with block_precisions: apply_block with the
 * converter, x += (alpha * double(block)) * b */
ORACLE_API void ref_jacobi_apply_adaptive(i64 num_blocks, const i64* scheme,
                                          const i32* block_ptrs, const u8* block_precisions,
                                          const double* blocks, i64 nrhs, double alpha,
                                          const double* b, i64 b_stride, double beta, double* x,
                                          i64 x_stride)
{
    const i64 stride = sch_stride(scheme);
    const i64 gs = sch_group_size(scheme);
    for (i64 i = 0; i < num_blocks; ++i) {
        const i64 bs = block_ptrs[i + 1] - block_ptrs[i];
        const u8 p = block_precisions ? block_precisions[i] : PR_P0N0;
        const void* group = blocks + scheme[1] * (i >> scheme[2]);
        const i64 block_ofs = scheme[0] * (i & (gs - 1));
        const double* bb = b + b_stride * block_ptrs[i];
        double* xx = x + x_stride * block_ptrs[i];
        for (i64 row = 0; row < bs; ++row)
            for (i64 col = 0; col < nrhs; ++col)
                xx[row * x_stride + col] = beta != 0.0 ? xx[row * x_stride + col] * beta : 0.0;
        for (i64 inner = 0; inner < bs; ++inner)
            for (i64 row = 0; row < bs; ++row)
                for (i64 col = 0; col < nrhs; ++col)
                    xx[row * x_stride + col] +=
                        alpha * pr_load(p, group, block_ofs + row + inner * stride) *
                        bb[inner * b_stride + col];
    }
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

/* jacobi::transpose_jacobi (reference/preconditioner/jacobi_kernels.cpp:629-658;
 * transpose_block = out[j + i*stride] = in[i + j*stride] in the block's
 * resolved precision, matrix_operations / jacobi_kernels.cpp:597-608). */
ORACLE_API void ref_jacobi_transpose(i64 num_blocks, const i64* scheme, const i32* block_ptrs,
                                     const u8* block_precisions, const double* blocks,
                                     double* out_blocks)
{
    const i64 stride = sch_stride(scheme);
    const i64 gs = sch_group_size(scheme);
    for (i64 i = 0; i < num_blocks; ++i) {
        const i64 bs = block_ptrs[i + 1] - block_ptrs[i];
        const u8 p = block_precisions ? block_precisions[i] : PR_P0N0;
        const char* group = (const char*)(blocks + scheme[1] * (i >> scheme[2]));
        char* out_group = (char*)(out_blocks + scheme[1] * (i >> scheme[2]));
        const i64 block_ofs = scheme[0] * (i & (gs - 1));
        const i64 esize = p == PR_P0N0 ? 8 : ((p == PR_P0N1 || p == PR_P1N0) ? 4 : 2);
        for (i64 r = 0; r < bs; ++r) {
            for (i64 c = 0; c < bs; ++c) {
                memcpy(out_group + esize * (block_ofs + c + r * stride),
                       group + esize * (block_ofs + r + c * stride), (size_t)esize);
            }
        }
    }
}
