// Block-Jacobi preconditioner for gfx950.  Replaces
// gko::kernels::hip::jacobi::{find_blocks, generate, apply, simple_apply,
// scalar_apply, simple_scalar_apply, invert_diagonal}
// (core/preconditioner/jacobi_kernels.hpp:50-190) and csr::extract_diagonal;
// semantics = reference/preconditioner/jacobi_kernels.cpp:66-625.
// Block storage in fp64 or, with the *_adaptive entry points, in the
// reference's adaptive precision: every storage group (= the blocks one wave
// owns) picks the smallest of {double, float, half, truncated<double,2>,
// truncated<float,2>, truncated<double,4>} that keeps cond * eps below the
// requested accuracy (core/preconditioner/jacobi_utils.hpp:46-201) -- the apply
// is HBM-bound on the block storage, so half the bytes is half the time.
//
// Storage = the reference's block_interleaved_storage_scheme with the HIP
// choice max_block_stride = wavefront size = 64 (jacobi.hpp:578-609): with
// S = pow2ceil(max_block_size), a group holds 64/S blocks, element (r, c) of
// block b sits at group_offset*(b / gs) + block_offset*(b % gs) + r + c*stride.
// One wave owns one group, lane = (block in group)*S + row: for a fixed column
// the 64 lanes of a wave read 64*8 consecutive bytes.
//
// apply: HBM-bound on the block storage (8*bs^2 B per block vs 16*bs B of
// vector): every lane streams its row of the inverse (one coalesced 8-B load
// per column, all issued up front), b is broadcast by shuffle, and the row sum
// is formed in the reference's `inner` order -> bit-identical results.
// generate: Gauss-Jordan with the reference's implicit row pivoting, each
// block in LDS, one lane per row; every element sees the same operations in
// the same order as reference invert_block -> bit-identical inverse.
#include <cstring>

#include "internal.hpp"

#include "sort_scan.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;

struct scheme_t {
    int64_t block_offset, group_offset;
    int group_power;
    __host__ __device__ int64_t stride() const { return block_offset << group_power; }
    __host__ __device__ int64_t global_offset(int64_t b) const
    {
        return group_offset * (b >> group_power) +
               block_offset * (b & ((int64_t{1} << group_power) - 1));
    }
};

// ---- find_blocks ----------------------------------------------------------

// same[i] = row i has the sparsity pattern of row i-1 (same[0] = 0)
__global__ __launch_bounds__(block) void compare_rows_kernel(
    int64_t nrows, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, uint8_t* __restrict__ same)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < nrows;
         i += static_cast<int64_t>(gridDim.x) * block) {
        uint8_t s = 0;
        if (i > 0) {
            const int32_t p0 = row_ptrs[i - 1], p1 = row_ptrs[i], p2 = row_ptrs[i + 1];
            if (p2 - p1 == p1 - p0) {
                s = 1;
                for (int32_t k = 0; k < p2 - p1; ++k) {
                    if (col_idxs[p0 + k] != col_idxs[p1 + k]) {
                        s = 0;
                        break;
                    }
                }
            }
        }
        same[i] = s;
    }
}

// The reference's two greedy passes (find_natural_blocks, then
// agglomerate_supervariables, jacobi_kernels.cpp:66-137 -- single-thread kernels
// in its GPU backends too) restated as data-parallel steps:
//  1. run starts: rows whose pattern differs from the previous row; a max-scan
//     gives every row the start rs(i) of its run of identical rows.  A natural
//     block starts at i  <=>  (i - rs(i)) % max_block_size == 0.
//  2. a second max-scan gives L(x) = last natural start <= x.  The agglomerated
//     block that starts at natural start r ends at the largest natural
//     boundary within r + max_block_size:  next(r) = L(r + max_bs), or n.
//  3. the block starts are the chain 0, next(0), next(next(0)), ...  A hop
//     covers at most max_bs rows, so the chain enters every chunk of 2048 rows
//     within its first max_bs rows: every chunk walks (in LDS) from each of
//     those <= 32 candidate entries to its exit, one thread composes the
//     exits of consecutive chunks, and every chunk re-walks from its real
//     entry marking the chain; flags -> scan -> block_ptrs.
// 120 ms -> well under 1 ms for 1.2M rows (profiles/r01_jacobi.log).
constexpr int fb_chunk = 2048;

__global__ __launch_bounds__(block) void fb_run_start_values_kernel(int n,
                                                                    const uint8_t* __restrict__ same,
                                                                    int* __restrict__ v)
{
    const int i = blockIdx.x * block + threadIdx.x;
    if (i < n) v[i] = same[i] ? -1 : i;
}

// in place: v holds rs(i) on entry, the natural-start value on exit
__global__ __launch_bounds__(block) void fb_natural_values_kernel(int n, int max_bs,
                                                                  int* __restrict__ v)
{
    const int i = blockIdx.x * block + threadIdx.x;
    if (i < n) v[i] = (i - v[i]) % max_bs == 0 ? i : -1;
}

// Mark = false: exits[chunk * 32 + e] = first chain row >= chunk end when the
// chain enters the chunk at its row e.  Mark = true: walk from the real entry
// and flag the chain rows.
template <bool Mark>
__global__ __launch_bounds__(block) void fb_walk_kernel(int n, int max_bs,
                                                        const int* __restrict__ last_natural,
                                                        int* __restrict__ exits,
                                                        const int* __restrict__ entry,
                                                        int* __restrict__ flags)
{
    __shared__ int s_next[fb_chunk];
    const int cs = blockIdx.x * fb_chunk;
    const int ce = min(cs + fb_chunk, n);
    for (int t = threadIdx.x; t < ce - cs; t += block) {
        const int r = cs + t;
        s_next[t] = r + max_bs >= n ? n : last_natural[r + max_bs];
    }
    __syncthreads();
    if (!Mark) {
        if (threadIdx.x < 32) {
            int r = cs + threadIdx.x;
            if (static_cast<int>(threadIdx.x) < max_bs && r < ce) {
                while (r < ce) r = s_next[r - cs];
            } else {
                r = n;
            }
            exits[blockIdx.x * 32 + threadIdx.x] = r;
        }
    } else if (threadIdx.x == 0) {
        int r = entry[blockIdx.x];
        while (r < ce) {
            flags[r] = 1;
            r = s_next[r - cs];
        }
    }
}

// entry[c] = first chain row >= c * fb_chunk (n once the chain has ended)
__global__ void fb_compose_kernel(int n, int nchunks, const int* __restrict__ exits,
                                  int* __restrict__ entry)
{
    int e = 0;
    for (int c = 0; c < nchunks; ++c) {
        entry[c] = e;
        const int ce = min((c + 1) * fb_chunk, n);
        if (e < ce) e = exits[c * 32 + (e - c * fb_chunk)];
    }
}

// offsets = exclusive scan of flags over n + 1 entries
__global__ __launch_bounds__(block) void fb_scatter_kernel(int n, const int* __restrict__ offsets,
                                                           int32_t* __restrict__ block_ptrs,
                                                           int64_t* __restrict__ num_blocks_out)
{
    const int i = blockIdx.x * block + threadIdx.x;
    if (i < n && offsets[i + 1] != offsets[i]) block_ptrs[offsets[i]] = i;
    if (i == n) {
        block_ptrs[offsets[n]] = n;
        *num_blocks_out = offsets[n];
    }
}

struct fb_layout {
    size_t same, a, flags, exits, entry, scan_tmp, psum_ws, total;
    size_t scan_tmp_bytes, psum_bytes;
};

fb_layout make_fb_layout(int64_t nrows)
{
    fb_layout l{};
    const size_t n = static_cast<size_t>(nrows > 0 ? nrows : 1);
    const size_t nchunks = (n + fb_chunk - 1) / fb_chunk;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    l.same = take(n);
    l.a = take(4 * (n + 1));
    l.flags = take(4 * (n + 1));
    l.exits = take(4 * 32 * nchunks);
    l.entry = take(4 * (nchunks + 1));
    const size_t tmp = scan_workspace_bytes(n);
    l.scan_tmp_bytes = tmp;
    l.scan_tmp = take(tmp);
    l.psum_bytes = gkomi_prefix_sum_workspace_bytes(static_cast<int64_t>(n) + 1);
    l.psum_ws = take(l.psum_bytes);
    l.total = off;
    return l;
}

// ---- generate ---------------------------------------------------------------

// precision_reduction as stored by the reference (types.hpp:257-368):
// (preserving << 4) | nonpreserving, autodetect = 0xff
enum : int { pr_p0n0 = 0x00, pr_p0n1 = 0x01, pr_p0n2 = 0x02, pr_p1n0 = 0x10, pr_p1n1 = 0x11, pr_p2n0 = 0x20,
             pr_autodetect = 0xff };
// precision_reduction_descriptor (jacobi_utils.hpp:84-112)
enum : unsigned { d_p0n0 = 0x00, d_p0n2 = 0x01, d_p1n1 = 0x02, d_p2n0 = 0x04, d_p0n1 = 0x08, d_p1n0 = 0x10 };

// gko::half's HOST conversion (core/base/extended_float.hpp:357-399), which the
// reference executor -- our oracle -- uses: significand truncated, results
// below the half normal range flushed to signed zero, overflow to infinity.
// (The reference's device code would call __float2half_rn instead.)
__device__ __forceinline__ unsigned short float2half_trunc(float f)
{
    const unsigned bits = __float_as_uint(f);
    const unsigned short sign = static_cast<unsigned short>((bits >> 16) & 0x8000u);
    const unsigned exp_bits = bits & 0x7f800000u;
    const unsigned sig = bits & 0x007fffffu;
    if (exp_bits == 0x7f800000u) return sign | 0x7c00u | (sig ? 0x03ffu : 0u);
    const unsigned e = exp_bits >> 13;
    const unsigned bias_change = (0x3f800000u >> 13) - 0x3c00u;
    unsigned he = e <= bias_change ? 0u : e - bias_change;
    if (he >= 0x7c00u) return sign | 0x7c00u;
    if (he == 0u) return sign;
    return sign | static_cast<unsigned short>(he) | static_cast<unsigned short>(sig >> 13);
}

__device__ __forceinline__ float half2float_flush(unsigned short h)
{
    // selects, not branches: this sits in the load loop of the apply kernel
    const unsigned sign = (static_cast<unsigned>(h) & 0x8000u) << 16;
    const unsigned exp_bits = h & 0x7c00u;
    const unsigned sig = h & 0x03ffu;
    const unsigned bias_change = 0x3f800000u - (0x3c00u << 13);
    const unsigned normal = sign | ((exp_bits << 13) + bias_change) | (sig << 13);
    const unsigned special = sign | 0x7f800000u | (sig ? 0x007fffffu : 0u);
    unsigned bits = exp_bits == 0u ? sign : normal;
    bits = exp_bits == 0x7c00u ? special : bits;
    return __uint_as_float(bits);
}

// static_cast<resolved_precision>(v) into element idx of the group's memory
template <int P>
__device__ __forceinline__ void store_reduced(void* group, int64_t idx, double v)
{
    if (P == pr_p0n1) {
        static_cast<float*>(group)[idx] = static_cast<float>(v);
    } else if (P == pr_p0n2) {
        static_cast<unsigned short*>(group)[idx] = float2half_trunc(static_cast<float>(v));
    } else if (P == pr_p1n0) {
        static_cast<unsigned*>(group)[idx] =
            static_cast<unsigned>(static_cast<unsigned long long>(__double_as_longlong(v)) >> 32);
    } else if (P == pr_p1n1) {
        static_cast<unsigned short*>(group)[idx] =
            static_cast<unsigned short>(__float_as_uint(static_cast<float>(v)) >> 16);
    } else if (P == pr_p2n0) {
        static_cast<unsigned short*>(group)[idx] = static_cast<unsigned short>(
            static_cast<unsigned long long>(__double_as_longlong(v)) >> 48);
    } else {
        static_cast<double*>(group)[idx] = v;
    }
}

// default_converter<resolved_precision, double>
template <int P>
__device__ __forceinline__ double load_reduced(const void* group, int64_t idx)
{
    if (P == pr_p0n1) return static_cast<double>(static_cast<const float*>(group)[idx]);
    if (P == pr_p0n2)
        return static_cast<double>(half2float_flush(static_cast<const unsigned short*>(group)[idx]));
    if (P == pr_p1n0)
        return __longlong_as_double(static_cast<long long>(
            static_cast<unsigned long long>(static_cast<const unsigned*>(group)[idx]) << 32));
    if (P == pr_p1n1)
        return static_cast<double>(__uint_as_float(
            static_cast<unsigned>(static_cast<const unsigned short*>(group)[idx]) << 16));
    if (P == pr_p2n0)
        return __longlong_as_double(static_cast<long long>(
            static_cast<unsigned long long>(static_cast<const unsigned short*>(group)[idx]) << 48));
    return static_cast<const double*>(group)[idx];
}

template <int P>
__device__ __forceinline__ double round_to(double v)
{
    double slot;
    store_reduced<P>(&slot, 0, v);
    return load_reduced<P>(&slot, 0);
}

// compute_inf_norm as the reference calls it on the row-major block:
// max_i sum_j |m[i + j*bs]| (matrix_operations.hpp:51-66).  One wave = one
// storage group, lane = (block in group) * S + r; col is a per-block scratch
// row; every lane of the block returns the norm.
template <int S>
__device__ double block_inf_norm(const double* blk, double* col, int bs, int r, bool row_active)
{
    constexpr int ld = S + 1;
    double t = 0.0;
    if (row_active)
        for (int j = 0; j < bs; ++j) t += fabs(blk[j * ld + r]);
    col[r] = row_active ? t : 0.0;
    __syncthreads();
    double norm = 0.0;
    for (int i = 0; i < bs; ++i) norm = fmax(norm, col[i]);
    __syncthreads();
    return norm;
}

// invert_block (:295-312) in LDS: Gauss-Jordan with the reference's implicit
// row pivoting, lane r owns row r / column r of its block; every element sees
// the same operations in the same order as the reference.  Returns false at a
// zero pivot like the reference's early return (the block then keeps its
// partial state).  max_bs = largest block size in the wave (uniform).
template <int S>
__device__ bool invert_block_lds(double* blk, int* perm, int bs, int r, bool row_active, int max_bs)
{
    constexpr int ld = S + 1;
    bool ok = true;
    for (int k = 0; k < max_bs; ++k) {
        const bool step = ok && k < bs;
        // choose_pivot: first row i >= k with the largest |blk[i][k]|
        int cp = k;
        if (step) {
            double best = fabs(blk[k * ld + k]);
            for (int i = k + 1; i < bs; ++i) {
                const double cand = fabs(blk[i * ld + k]);
                if (best < cand) {
                    best = cand;
                    cp = i;
                }
            }
        }
        // swap_rows(k, cp) and the permutation: lane r handles column r
        if (step && row_active && cp != k) {
            const double t = blk[k * ld + r];
            blk[k * ld + r] = blk[cp * ld + r];
            blk[cp * ld + r] = t;
            if (r == 0) {
                const int tp = perm[k];
                perm[k] = perm[cp];
                perm[cp] = tp;
            }
        }
        __syncthreads();
        const double d = step ? blk[k * ld + k] : 1.0;
        if (step && d == 0.0) ok = false;
        const bool go = step && ok;
        __syncthreads();
        // apply_gauss_jordan_transform(k, k) (:217-240)
        if (go && row_active) blk[r * ld + k] /= -d;
        __syncthreads();
        if (go && r == k) blk[k * ld + k] = 0.0;
        __syncthreads();
        if (go && row_active && r != k) {
            // row k adds 0 * row k to itself in the reference: left untouched here
            const double f = blk[r * ld + k];
            for (int j = 0; j < bs; ++j) blk[r * ld + j] += f * blk[k * ld + j];
        }
        __syncthreads();
        if (go && row_active) blk[k * ld + r] /= d;
        __syncthreads();
        if (go && r == k) blk[k * ld + k] = 1.0 / d;
        __syncthreads();
    }
    return ok;
}

// validate_precision_reduction_feasibility<reduced> (:311-336): would the
// inverse still be invertible and sanely conditioned after rounding?
template <int S, int P>
__device__ bool reduction_feasible(const double* blk, double* tmp, int* perm, double* col, int bs,
                                   int r, bool row_active, int max_bs)
{
    constexpr int ld = S + 1;
    for (int j = 0; j < S; ++j) tmp[r * ld + j] = row_active && j < bs ? round_to<P>(blk[r * ld + j]) : 0.0;
    perm[r] = r;
    __syncthreads();
    double cond = block_inf_norm<S>(tmp, col, bs, r, row_active);
    const bool ok = invert_block_lds<S>(tmp, perm, bs, r, row_active, max_bs);
    cond *= block_inf_norm<S>(tmp, col, bs, r, row_active);
    return ok && cond >= 1.0 && cond * 0x1p-53 < 1e-3;
}

__device__ __forceinline__ unsigned descriptor_singleton(int pr)
{
    return pr == pr_p0n1 ? d_p0n1 : pr == pr_p0n2 ? d_p0n2 : pr == pr_p1n0 ? d_p1n0
           : pr == pr_p1n1 ? d_p1n1 : pr == pr_p2n0 ? d_p2n0 : d_p0n0;
}

// get_supported_storage_reductions<double> (jacobi_utils.hpp:129-167) with the
// two verificators already evaluated (they are pure): same truth table
__device__ __forceinline__ unsigned supported_reductions(double accuracy, double cond, bool v1, bool v2)
{
    unsigned supported = d_p0n0;
    int verified1 = 2;
    if (cond * 0x1p-4 < accuracy) supported |= d_p2n0;
    if (cond * 0x1p-7 < accuracy) {
        verified1 = v1 ? 1 : 0;
        if (v1) supported |= d_p1n1;
    }
    if (cond * 0x1p-11 < accuracy && verified1 != 0 && v2) supported |= d_p0n2;
    if (cond * 0x1p-20 < accuracy) supported |= d_p1n0;
    if (cond * 0x1p-24 < accuracy) {
        if (verified1 == 2) verified1 = v1 ? 1 : 0;
        if (verified1 == 1) supported |= d_p0n1;
    }
    return supported;
}

// get_optimal_storage_reduction (jacobi_utils.hpp:184-201)
__device__ __forceinline__ int optimal_reduction(unsigned supported)
{
    if (supported & d_p0n2) return pr_p0n2;
    if (supported & d_p1n1) return pr_p1n1;
    if (supported & d_p2n0) return pr_p2n0;
    if (supported & d_p0n1) return pr_p0n1;
    if (supported & d_p1n0) return pr_p1n0;
    return pr_p0n0;
}

// permute_and_transpose_block (:277-292): out[i + perm[j]*stride] = blk[i][j]
template <int S, int P>
__device__ void store_block(const double* blk, const int* perm, int bs, int r, bool row_active,
                            void* group, int64_t block_ofs, int64_t stride)
{
    constexpr int ld = S + 1;
    if (row_active)
        for (int j = 0; j < bs; ++j) store_reduced<P>(group, block_ofs + r + perm[j] * stride, blk[r * ld + j]);
}

// Adaptive = false: every block in fp64 (block_precisions unused).
template <int S, bool Adaptive>
__global__ __launch_bounds__(64) void jacobi_generate_kernel(
    const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, int64_t num_blocks, scheme_t scheme,
    const int32_t* __restrict__ block_ptrs, double accuracy, double* __restrict__ conditioning,
    uint8_t* __restrict__ block_precisions, double* __restrict__ blocks)
{
    constexpr int gs = 64 / S;        // blocks per group = per wave
    constexpr int ld = S + 1;         // padded leading dimension: conflict-free column walks
    __shared__ double sblk[gs * S * ld];
    __shared__ double stmp[Adaptive ? gs * S * ld : 1];
    __shared__ int sperm[gs * S];
    __shared__ int sperm2[Adaptive ? gs * S : 1];
    __shared__ double scol[gs * S];   // |pivot candidates| / column sums
    const int lane = threadIdx.x;
    const int g = lane / S;           // block within the group
    const int r = lane % S;           // row (or column) handled by this lane
    const int64_t b = static_cast<int64_t>(blockIdx.x) * gs + g;
    const bool have = b < num_blocks;
    const int start = have ? block_ptrs[b] : 0;
    const int bs = have ? block_ptrs[b + 1] - start : 0;
    double* blk = sblk + g * S * ld;
    int* perm = sperm + g * S;
    double* col = scol + g * S;
    const bool row_active = r < bs;

    // extract_block (:163-183)
    for (int j = 0; j < S; ++j) blk[r * ld + j] = 0.0;
    perm[r] = r;
    if (row_active) {
        const int end = row_ptrs[start + r + 1];
        for (int k = row_ptrs[start + r]; k < end; ++k) {
            const int c = col_idxs[k] - start;
            if (0 <= c && c < bs) blk[r * ld + c] = vals[k];
        }
    }
    __syncthreads();
    double cond = 0.0;
    if (conditioning != nullptr) cond = block_inf_norm<S>(blk, col, bs, r, row_active);
    int max_bs = bs;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_bs = max(max_bs, __shfl_xor(max_bs, off, 64));
    invert_block_lds<S>(blk, perm, bs, r, row_active, max_bs);
    if (conditioning != nullptr) {
        cond *= block_inf_norm<S>(blk, col, bs, r, row_active);
        if (have && r == 0) conditioning[b] = cond;
    }

    int p = pr_p0n0;
    if (Adaptive) {
        // the group's precision: the best reduction every block supports (:387-416)
        const int local = have ? block_precisions[b] : pr_p0n0;
        const bool autodetect = have && local == pr_autodetect;
        unsigned desc = have ? descriptor_singleton(local) : ~0u;
        if (__any(autodetect)) {  // (wave-uniform: the checks below are cooperative)
            double* tmp = stmp + g * S * ld;
            int* perm2 = sperm2 + g * S;
            const bool v1 = reduction_feasible<S, pr_p0n1>(blk, tmp, perm2, col, bs, r, row_active, max_bs);
            const bool v2 = reduction_feasible<S, pr_p0n2>(blk, tmp, perm2, col, bs, r, row_active, max_bs);
            if (autodetect) desc = supported_reductions(accuracy, cond, v1, v2);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) desc &= __shfl_xor(desc, off, 64);
        p = optimal_reduction(desc);
        if (have && r == 0) block_precisions[b] = static_cast<uint8_t>(p);
    }
    // group memory is addressed in units of the reduced type (:425-436)
    void* group = blocks + scheme.group_offset * blockIdx.x;
    const int64_t block_ofs = scheme.block_offset * g;
    const int64_t stride = scheme.stride();
    switch (p) {
    case pr_p0n1: store_block<S, pr_p0n1>(blk, perm, bs, r, row_active, group, block_ofs, stride); break;
    case pr_p0n2: store_block<S, pr_p0n2>(blk, perm, bs, r, row_active, group, block_ofs, stride); break;
    case pr_p1n0: store_block<S, pr_p1n0>(blk, perm, bs, r, row_active, group, block_ofs, stride); break;
    case pr_p1n1: store_block<S, pr_p1n1>(blk, perm, bs, r, row_active, group, block_ofs, stride); break;
    case pr_p2n0: store_block<S, pr_p2n0>(blk, perm, bs, r, row_active, group, block_ofs, stride); break;
    default: store_block<S, pr_p0n0>(blk, perm, bs, r, row_active, group, block_ofs, stride); break;
    }
}

// ---- apply --------------------------------------------------------------------

// this lane's row of the inverse: one coalesced load per column.  Branch-free
// (clamped addresses, select afterwards) so that all loads are in flight
// together; with a branch per element the compiler waits for each load before
// the next (68 us instead of 31 us for 4-byte storage on 38k blocks of 32).
template <int S, int P>
__device__ __forceinline__ void load_row(const void* group, int idx0, int safe_idx, int stride,
                                         bool active, int bs, double (&rowv)[S])
{
    // (indices inside a group fit 32 bits: at most 32 columns x stride 64)
    const int base = active ? idx0 : safe_idx;
    const int step = active ? stride : 0;
    const int last = active ? bs - 1 : 0;
#pragma unroll
    for (int inner = 0; inner < S; ++inner) {
        rowv[inner] = load_reduced<P>(group, base + min(inner, last) * step);
    }
#pragma unroll
    for (int inner = 0; inner < S; ++inner) {
        rowv[inner] = (active && inner < bs) ? rowv[inner] : 0.0;
    }
}

// block_precisions == nullptr: fp64 storage
// Dot (one column, the fused CG of cg_solver.hip): the launch also leaves, per workgroup, the partial sums of
// b . x (= r . z) and b . b (= r . r) -- every lane holds both factors of its row anyway -- instead of a kernel of
// its own re-reading both vectors; a solve that has stopped skips the launch.
template <int S, bool Advanced, bool Dot = false>
__global__ __launch_bounds__(block) void jacobi_apply_kernel(
    int64_t num_blocks, scheme_t scheme, const int32_t* __restrict__ block_ptrs,
    const uint8_t* __restrict__ block_precisions, const double* __restrict__ blocks, int64_t nrhs,
    const double* __restrict__ alpha_p, const double* __restrict__ b, int64_t b_stride,
    const double* __restrict__ beta_p, double* __restrict__ x, int64_t x_stride,
    double* __restrict__ part_bx = nullptr, double* __restrict__ part_bb = nullptr,
    const uint8_t* __restrict__ stop_status = nullptr)
{
    if (Dot && status_has_stopped_uniform(stop_status)) return;
    constexpr int gs = 64 / S;
    const int lane = threadIdx.x & 63;
    const int g = lane / S, r = lane % S;
    const int64_t group_id = blockIdx.x * static_cast<int64_t>(block / 64) + (threadIdx.x >> 6);
    const int64_t blk_id = group_id * gs + g;
    const bool have = blk_id < num_blocks;
    const int start = have ? block_ptrs[blk_id] : 0;
    const int bs = have ? block_ptrs[blk_id + 1] - start : 0;
    const bool active = r < bs;
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // the group (= this wave) shares one storage precision; its memory is
    // addressed in units of that type (:509-527)
    const int64_t first = group_id * gs;
    const int p = (block_precisions != nullptr && first < num_blocks) ? block_precisions[first] : pr_p0n0;
    const void* group = blocks + (first < num_blocks ? scheme.group_offset * group_id : 0);
    const int idx0 = static_cast<int>(scheme.block_offset) * g + r;
    const int safe = first < num_blocks ? static_cast<int>(scheme.block_offset) * g : 0;  // always inside the storage
    const int stride = static_cast<int>(scheme.stride());
    double rowv[S];
    switch (p) {
    case pr_p0n1: load_row<S, pr_p0n1>(group, idx0, safe, stride, active, bs, rowv); break;
    case pr_p0n2: load_row<S, pr_p0n2>(group, idx0, safe, stride, active, bs, rowv); break;
    case pr_p1n0: load_row<S, pr_p1n0>(group, idx0, safe, stride, active, bs, rowv); break;
    case pr_p1n1: load_row<S, pr_p1n1>(group, idx0, safe, stride, active, bs, rowv); break;
    case pr_p2n0: load_row<S, pr_p2n0>(group, idx0, safe, stride, active, bs, rowv); break;
    default: load_row<S, pr_p0n0>(group, idx0, safe, stride, active, bs, rowv); break;
    }
    double bx = 0.0, bb = 0.0;
    for (int64_t j = 0; j < nrhs; ++j) {
        const double bv = active ? b[(start + r) * b_stride + j] : 0.0;
        double acc = 0.0;
        if (Advanced && active && beta != 0.0) acc = x[(start + r) * x_stride + j] * beta;
#pragma unroll
        for (int inner = 0; inner < S; ++inner) {
            const double bi = __shfl(bv, g * S + inner, 64);
            if (inner < bs) {
                acc += Advanced ? (alpha * rowv[inner]) * bi : rowv[inner] * bi;
            }
        }
        if (active) x[(start + r) * x_stride + j] = acc;
        if (Dot && active) {
            bx += bv * acc;
            bb += bv * bv;
        }
    }
    if (Dot) {
        __shared__ double red[2 * (block / 64)];
        bx = wave_reduce_sum(bx);
        bb = wave_reduce_sum(bb);
        if (lane == 0) {
            red[threadIdx.x >> 6] = bx;
            red[block / 64 + (threadIdx.x >> 6)] = bb;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double tx = 0.0, tb = 0.0;
#pragma unroll
            for (int w = 0; w < block / 64; ++w) {
                tx += red[w];
                tb += red[block / 64 + w];
            }
            part_bx[blockIdx.x] = tx;
            part_bb[blockIdx.x] = tb;
        }
    }
}

// ---- scalar Jacobi --------------------------------------------------------------

__global__ __launch_bounds__(block) void extract_diagonal_kernel(
    int64_t nrows, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    double* __restrict__ diag)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        double d = 0.0;
        const int32_t end = row_ptrs[row + 1];
        for (int32_t k = row_ptrs[row]; k < end; ++k) {
            if (col_idxs[k] == row) {
                d = vals[k];
                break;
            }
        }
        diag[row] = d;
    }
}

__global__ __launch_bounds__(block) void invert_diagonal_kernel(int64_t n,
                                                               const double* __restrict__ diag,
                                                               double* __restrict__ inv)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * block) {
        const double d = diag[i];
        inv[i] = 1.0 / (d == 0.0 ? 1.0 : d);
    }
}

template <bool Advanced>
__global__ __launch_bounds__(block) void scalar_apply_kernel(
    int64_t nrows, int64_t nrhs, const double* __restrict__ diag,
    const double* __restrict__ alpha_p, const double* __restrict__ b, int64_t b_stride,
    const double* __restrict__ beta_p, double* __restrict__ x, int64_t x_stride)
{
    const double alpha = Advanced ? alpha_p[0] : 1.0;
    const double beta = Advanced ? beta_p[0] : 0.0;
    const int64_t total = nrows * nrhs;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t row = i / nrhs, col = i % nrhs;
        const double bv = b[row * b_stride + col];
        double* xp = x + row * x_stride + col;
        // reference :565-594
        *xp = Advanced ? beta * (*xp) + (alpha * bv) * diag[row] : bv * diag[row];
    }
}

int pow2ceil(int v)
{
    int p = 1;
    while (p < v) p *= 2;
    return p;
}

// jacobi::transpose_jacobi (reference/preconditioner/jacobi_kernels.cpp:629-658):
// every stored block transposed in its own storage precision -- a move of
// 8-, 4- or 2-byte elements, so the result is bit-identical by construction.
// One thread per (block, r, c) of the max_block_size^2 slots of a block.
template <typename Elem>
__device__ __forceinline__ void transpose_elem(const double* in_group, double* out_group,
                                               int64_t block_ofs, int64_t stride, int r, int c)
{
    reinterpret_cast<Elem*>(out_group)[block_ofs + c + r * stride] =
        reinterpret_cast<const Elem*>(in_group)[block_ofs + r + c * stride];
}

__global__ __launch_bounds__(block) void jacobi_transpose_kernel(
    int64_t num_blocks, int max_bs, scheme_t scheme, const int32_t* __restrict__ block_ptrs,
    const uint8_t* __restrict__ precisions, const double* __restrict__ in, double* __restrict__ out)
{
    const int64_t idx = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    const int64_t per = static_cast<int64_t>(max_bs) * max_bs;
    const int64_t b = idx / per;
    if (b >= num_blocks) return;
    const int rc = static_cast<int>(idx - b * per);
    const int r = rc % max_bs, c = rc / max_bs;
    const int bs = block_ptrs[b + 1] - block_ptrs[b];
    if (r >= bs || c >= bs) return;
    const int64_t group_ofs = scheme.group_offset * (b >> scheme.group_power);
    const int64_t block_ofs = scheme.block_offset * (b & ((int64_t{1} << scheme.group_power) - 1));
    const int64_t stride = scheme.stride();
    const int p = precisions != nullptr ? precisions[b] : pr_p0n0;
    if (p == pr_p0n0) {
        transpose_elem<double>(in + group_ofs, out + group_ofs, block_ofs, stride, r, c);
    } else if (p == pr_p0n1 || p == pr_p1n0) {
        transpose_elem<unsigned>(in + group_ofs, out + group_ofs, block_ofs, stride, r, c);
    } else {
        transpose_elem<unsigned short>(in + group_ofs, out + group_ofs, block_ofs, stride, r, c);
    }
}

scheme_t make_scheme(int max_block_size)
{
    const int s = pow2ceil(max_block_size);
    const int group_size = 64 / s;
    scheme_t sc;
    sc.block_offset = max_block_size;
    sc.group_offset = static_cast<int64_t>(max_block_size) * group_size * max_block_size;
    sc.group_power = 0;
    while ((1 << (sc.group_power + 1)) <= group_size) ++sc.group_power;
    return sc;
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int gkomi_jacobi_storage_scheme(int max_block_size, int64_t out[4])
{
    if (out == nullptr || max_block_size < 1 || max_block_size > 32) return GKOMI_EINVAL;
    const scheme_t sc = make_scheme(max_block_size);
    out[0] = sc.block_offset;
    out[1] = sc.group_offset;
    out[2] = sc.group_power;
    out[3] = sc.stride();
    return GKOMI_SUCCESS;
}

extern "C" size_t gkomi_jacobi_storage_elements(int max_block_size, int64_t num_blocks)
{
    if (max_block_size < 1 || max_block_size > 32 || num_blocks < 0) return 0;
    const scheme_t sc = make_scheme(max_block_size);
    const int64_t gs = int64_t{1} << sc.group_power;
    return static_cast<size_t>(ceildiv(num_blocks, gs) * sc.group_offset);
}

extern "C" size_t gkomi_jacobi_find_blocks_workspace_bytes(int64_t nrows)
{
    return make_fb_layout(nrows).total;
}

extern "C" int gkomi_jacobi_find_blocks_i32(gkomi_stream_t s, int64_t nrows,
                                            const int32_t* row_ptrs, const int32_t* col_idxs,
                                            int max_block_size, int32_t* block_ptrs,
                                            int64_t* num_blocks_device, void* workspace,
                                            size_t workspace_bytes, int64_t* host_num_blocks)
{
    if (nrows < 0 || max_block_size < 1 || max_block_size > 32) return GKOMI_EINVAL;
    if (nrows > INT32_MAX - fb_chunk - 64) return GKOMI_ENOTSUPPORTED;
    const fb_layout l = make_fb_layout(nrows);
    if (workspace_bytes < l.total || workspace == nullptr) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    const int n = static_cast<int>(nrows);
    int err = 0;
    if (n == 0) {
        err = static_cast<int>(hipMemsetAsync(block_ptrs, 0, sizeof(int32_t), stream));
        if (!err) err = static_cast<int>(hipMemsetAsync(num_blocks_device, 0, sizeof(int64_t), stream));
    } else {
        char* ws = static_cast<char*>(workspace);
        uint8_t* same = reinterpret_cast<uint8_t*>(ws + l.same);
        int* a = reinterpret_cast<int*>(ws + l.a);
        int* flags = reinterpret_cast<int*>(ws + l.flags);
        int* exits = reinterpret_cast<int*>(ws + l.exits);
        int* entry = reinterpret_cast<int*>(ws + l.entry);
        const int nchunks = static_cast<int>(ceildiv(nrows, fb_chunk));
        const dim3 grid_n(static_cast<unsigned>(ceildiv(nrows, block)));
        const dim3 grid_n1(static_cast<unsigned>(ceildiv(nrows + 1, block)));
        hipLaunchKernelGGL(compare_rows_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block),
                           0, stream, nrows, row_ptrs, col_idxs, same);
        hipLaunchKernelGGL(fb_run_start_values_kernel, grid_n, dim3(block), 0, stream, n, same, a);
        size_t tmp_bytes = l.scan_tmp_bytes;
        err = inclusive_max_i32(stream, a, a, n, ws + l.scan_tmp, tmp_bytes);
        if (err) return err;
        hipLaunchKernelGGL(fb_natural_values_kernel, grid_n, dim3(block), 0, stream, n,
                           max_block_size, a);
        tmp_bytes = l.scan_tmp_bytes;
        err = inclusive_max_i32(stream, a, a, n, ws + l.scan_tmp, tmp_bytes);
        if (err) return err;
        err = static_cast<int>(hipMemsetAsync(flags, 0, sizeof(int) * (static_cast<size_t>(n) + 1), stream));
        if (err) return err;
        hipLaunchKernelGGL(fb_walk_kernel<false>, dim3(nchunks), dim3(block), 0, stream, n,
                           max_block_size, a, exits, entry, flags);
        hipLaunchKernelGGL(fb_compose_kernel, dim3(1), dim3(1), 0, stream, n, nchunks, exits, entry);
        hipLaunchKernelGGL(fb_walk_kernel<true>, dim3(nchunks), dim3(block), 0, stream, n,
                           max_block_size, a, exits, entry, flags);
        err = gkomi_prefix_sum_i32(s, flags, nrows + 1, ws + l.psum_ws, l.psum_bytes);
        if (err) return err;
        hipLaunchKernelGGL(fb_scatter_kernel, grid_n1, dim3(block), 0, stream, n, flags, block_ptrs,
                           num_blocks_device);
        err = check_launch();
    }
    if (err) return err;
    if (host_num_blocks != nullptr) {
        err = static_cast<int>(hipMemcpyAsync(host_num_blocks, num_blocks_device, sizeof(int64_t),
                                              hipMemcpyDeviceToHost, stream));
        if (err) return err;
        err = static_cast<int>(hipStreamSynchronize(stream));
    }
    return err;
}

namespace {

int jacobi_generate(gkomi_stream_t s, int64_t nrows, const int32_t* row_ptrs,
                    const int32_t* col_idxs, const double* vals, int64_t num_blocks,
                    int max_block_size, const int32_t* block_ptrs, double accuracy,
                    double* conditioning, uint8_t* block_precisions, double* blocks)
{
    if (nrows < 0 || num_blocks < 0 || max_block_size < 1 || max_block_size > 32) return GKOMI_EINVAL;
    if (num_blocks == 0) return GKOMI_SUCCESS;
    const scheme_t sc = make_scheme(max_block_size);
    const int sw = pow2ceil(max_block_size);
    const int64_t groups = ceildiv(num_blocks, 64 / sw);
    if (groups > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    hipStream_t stream = to_stream(s);
    // padding entries of the storage are never read by apply, but keep them defined
    int err = static_cast<int>(hipMemsetAsync(
        blocks, 0, sizeof(double) * gkomi_jacobi_storage_elements(max_block_size, num_blocks), stream));
    if (err) return err;
#define GKOMI_GEN(S)                                                                            \
    do {                                                                                        \
        if (block_precisions != nullptr) {                                                      \
            hipLaunchKernelGGL((jacobi_generate_kernel<S, true>),                               \
                               dim3(static_cast<unsigned>(groups)), dim3(64), 0, stream,        \
                               row_ptrs, col_idxs, vals, num_blocks, sc, block_ptrs, accuracy,  \
                               conditioning, block_precisions, blocks);                         \
        } else {                                                                                \
            hipLaunchKernelGGL((jacobi_generate_kernel<S, false>),                              \
                               dim3(static_cast<unsigned>(groups)), dim3(64), 0, stream,        \
                               row_ptrs, col_idxs, vals, num_blocks, sc, block_ptrs, accuracy,  \
                               conditioning, block_precisions, blocks);                         \
        }                                                                                       \
    } while (0)
    switch (sw) {
    case 1: GKOMI_GEN(1); break;
    case 2: GKOMI_GEN(2); break;
    case 4: GKOMI_GEN(4); break;
    case 8: GKOMI_GEN(8); break;
    case 16: GKOMI_GEN(16); break;
    default: GKOMI_GEN(32); break;
    }
#undef GKOMI_GEN
    return check_launch();
}

int jacobi_apply(gkomi_stream_t s, int64_t num_blocks, int max_block_size,
                 const int32_t* block_ptrs, const uint8_t* block_precisions, const double* blocks,
                 int64_t nrhs, const double* alpha, const double* b, int64_t b_stride,
                 const double* beta, double* x, int64_t x_stride)
{
    if (num_blocks < 0 || nrhs < 0 || max_block_size < 1 || max_block_size > 32) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (num_blocks == 0 || nrhs == 0) return GKOMI_SUCCESS;
    const scheme_t sc = make_scheme(max_block_size);
    const int sw = pow2ceil(max_block_size);
    const int64_t groups = ceildiv(num_blocks, 64 / sw);
    const int64_t grid = ceildiv(groups, block / 64);
    if (grid > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    hipStream_t stream = to_stream(s);
#define GKOMI_APPLY(S)                                                                        \
    do {                                                                                      \
        if (alpha != nullptr) {                                                               \
            hipLaunchKernelGGL((jacobi_apply_kernel<S, true>), dim3(static_cast<unsigned>(grid)), \
                               dim3(block), 0, stream, num_blocks, sc, block_ptrs,            \
                               block_precisions, blocks, nrhs, alpha, b, b_stride, beta, x,   \
                               x_stride);                                                     \
        } else {                                                                              \
            hipLaunchKernelGGL((jacobi_apply_kernel<S, false>), dim3(static_cast<unsigned>(grid)), \
                               dim3(block), 0, stream, num_blocks, sc, block_ptrs,            \
                               block_precisions, blocks, nrhs, alpha, b, b_stride, beta, x,   \
                               x_stride);                                                     \
        }                                                                                     \
    } while (0)
    switch (sw) {
    case 1: GKOMI_APPLY(1); break;
    case 2: GKOMI_APPLY(2); break;
    case 4: GKOMI_APPLY(4); break;
    case 8: GKOMI_APPLY(8); break;
    case 16: GKOMI_APPLY(16); break;
    default: GKOMI_APPLY(32); break;
    }
#undef GKOMI_APPLY
    return check_launch();
}

}  // namespace

// z = M^-1 r together with the partials of r . z and r . r (cg_solver.hip, internal.hpp): fp64 or adaptive block
// storage, one column.  Returns the number of partials (= workgroups) or a negative error code; 0 = not for this
// preconditioner (scalar Jacobi, several columns): the caller applies it and adds up the dots itself.
int gkomi::jacobi_apply_dot_launch(gkomi_stream_t s, const gkomi_jacobi_ctx* c, const double* in, double* out,
                                   double* part_rz, double* part_rr, size_t room, const uint8_t* stop_status)
{
    if (c == nullptr || c->nrhs != 1 || c->max_block_size <= 1 || c->max_block_size > 32 || c->num_blocks <= 0) return 0;
    const scheme_t sc = make_scheme(c->max_block_size);
    const int sw = pow2ceil(c->max_block_size);
    const int64_t groups = ceildiv(c->num_blocks, 64 / sw);
    const int64_t grid = ceildiv(groups, block / 64);
    // One partial per workgroup, about n / 256 of them, and every workgroup of the consumer (cg_fused_step1_kernel,
    // up to 1024 of them) re-adds them all, every iteration: beyond spmv_dot_uncompressed_partials the launch writes
    // them behind the first spmv_dot_max_partials slots and a second, small launch compresses them into those --
    // the rule of spmv_dot_plan::launch (at 256^3 rows 65 k partials would be ~1 GB of L2 reads per iteration).
    const bool squeeze = grid > spmv_dot_uncompressed_partials;
    if (grid + (squeeze ? spmv_dot_max_partials : 0) > static_cast<int64_t>(room)) return 0;
    double* raw_rz = squeeze ? part_rz + spmv_dot_max_partials : part_rz;
    double* raw_rr = squeeze ? part_rr + spmv_dot_max_partials : part_rr;
    hipStream_t stream = to_stream(s);
#define GKOMI_APPLY_DOT(S)                                                                                        \
    hipLaunchKernelGGL((jacobi_apply_kernel<S, false, true>), dim3(static_cast<unsigned>(grid)), dim3(block), 0,   \
                       stream, c->num_blocks, sc, c->block_ptrs, c->block_precisions, c->blocks, int64_t{1},      \
                       static_cast<const double*>(nullptr), in, int64_t{1}, static_cast<const double*>(nullptr),  \
                       out, int64_t{1}, raw_rz, raw_rr, stop_status)
    switch (sw) {
    case 2: GKOMI_APPLY_DOT(2); break;
    case 4: GKOMI_APPLY_DOT(4); break;
    case 8: GKOMI_APPLY_DOT(8); break;
    case 16: GKOMI_APPLY_DOT(16); break;
    default: GKOMI_APPLY_DOT(32); break;
    }
#undef GKOMI_APPLY_DOT
    int err = check_launch();
    if (err == 0 && squeeze) {
        err = compress_partials_launch(stream, raw_rz, static_cast<int>(grid), part_rz, spmv_dot_max_partials, raw_rr,
                                       part_rr, stop_status);
    }
    if (err) return -(err < 0 ? -err : err) - 1000;
    return squeeze ? spmv_dot_max_partials : static_cast<int>(grid);
}

extern "C" int gkomi_jacobi_generate_f64_i32(gkomi_stream_t s, int64_t nrows,
                                             const int32_t* row_ptrs, const int32_t* col_idxs,
                                             const double* vals, int64_t num_blocks,
                                             int max_block_size, const int32_t* block_ptrs,
                                             double* conditioning, double* blocks)
{
    return jacobi_generate(s, nrows, row_ptrs, col_idxs, vals, num_blocks, max_block_size,
                           block_ptrs, 0.0, conditioning, nullptr, blocks);
}

extern "C" int gkomi_jacobi_generate_adaptive_f64_i32(
    gkomi_stream_t s, int64_t nrows, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, int64_t num_blocks, int max_block_size, const int32_t* block_ptrs,
    double accuracy, double* conditioning, uint8_t* block_precisions, double* blocks)
{
    if (conditioning == nullptr || block_precisions == nullptr) return GKOMI_EINVAL;
    return jacobi_generate(s, nrows, row_ptrs, col_idxs, vals, num_blocks, max_block_size,
                           block_ptrs, accuracy, conditioning, block_precisions, blocks);
}

extern "C" int gkomi_jacobi_apply_f64_i32(gkomi_stream_t s, int64_t num_blocks,
                                          int max_block_size, const int32_t* block_ptrs,
                                          const double* blocks, int64_t nrhs, const double* alpha,
                                          const double* b, int64_t b_stride, const double* beta,
                                          double* x, int64_t x_stride)
{
    return jacobi_apply(s, num_blocks, max_block_size, block_ptrs, nullptr, blocks, nrhs, alpha, b,
                        b_stride, beta, x, x_stride);
}

extern "C" int gkomi_jacobi_apply_adaptive_f64_i32(
    gkomi_stream_t s, int64_t num_blocks, int max_block_size, const int32_t* block_ptrs,
    const uint8_t* block_precisions, const double* blocks, int64_t nrhs, const double* alpha,
    const double* b, int64_t b_stride, const double* beta, double* x, int64_t x_stride)
{
    if (block_precisions == nullptr) return GKOMI_EINVAL;
    return jacobi_apply(s, num_blocks, max_block_size, block_ptrs, block_precisions, blocks, nrhs,
                        alpha, b, b_stride, beta, x, x_stride);
}

extern "C" int gkomi_csr_extract_diagonal_f64_i32(gkomi_stream_t s, int64_t nrows,
                                                  const int32_t* row_ptrs,
                                                  const int32_t* col_idxs, const double* vals,
                                                  double* diag)
{
    if (nrows < 0) return GKOMI_EINVAL;
    if (nrows == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(extract_diagonal_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block),
                       0, to_stream(s), nrows, row_ptrs, col_idxs, vals, diag);
    return check_launch();
}

extern "C" int gkomi_jacobi_invert_diagonal_f64(gkomi_stream_t s, int64_t n, const double* diag,
                                                double* inv_diag)
{
    if (n < 0) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(invert_diagonal_kernel, dim3(grid_for(n, block)), dim3(block), 0,
                       to_stream(s), n, diag, inv_diag);
    return check_launch();
}

extern "C" int gkomi_jacobi_scalar_apply_f64(gkomi_stream_t s, int64_t nrows, int64_t nrhs,
                                             const double* inv_diag, const double* alpha,
                                             const double* b, int64_t b_stride,
                                             const double* beta, double* x, int64_t x_stride)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    dim3 grid(grid_for(nrows * nrhs, block));
    if (alpha != nullptr) {
        hipLaunchKernelGGL(scalar_apply_kernel<true>, grid, dim3(block), 0, to_stream(s), nrows,
                           nrhs, inv_diag, alpha, b, b_stride, beta, x, x_stride);
    } else {
        hipLaunchKernelGGL(scalar_apply_kernel<false>, grid, dim3(block), 0, to_stream(s), nrows,
                           nrhs, inv_diag, alpha, b, b_stride, beta, x, x_stride);
    }
    return check_launch();
}

extern "C" int gkomi_jacobi_transpose_f64_i32(gkomi_stream_t s, int64_t num_blocks, int max_block_size,
                                              const int32_t* block_ptrs, const uint8_t* block_precisions,
                                              const double* blocks, double* out_blocks)
{
    if (num_blocks < 0 || max_block_size < 1 || max_block_size > 32) return GKOMI_EINVAL;
    if (num_blocks == 0) return GKOMI_SUCCESS;
    if (blocks == out_blocks) return GKOMI_EINVAL;
    const int64_t total = num_blocks * max_block_size * max_block_size;
    const int64_t grid = ceildiv(total, block);
    if (grid > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    hipLaunchKernelGGL(jacobi_transpose_kernel, dim3(static_cast<unsigned>(grid)), dim3(block), 0, to_stream(s),
                       num_blocks, max_block_size, make_scheme(max_block_size), block_ptrs, block_precisions,
                       blocks, out_blocks);
    return check_launch();
}
