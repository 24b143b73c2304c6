// Index / format-conversion kernels for gfx950 (bit-exact integer work).
// Replaces components::{prefix_sum, convert_ptrs_to_idxs, convert_idxs_to_ptrs,
// convert_ptrs_to_sizes} (core/components/*_kernels.hpp),
// csr::{convert_to_ell, convert_to_sellp, convert_to_hybrid},
// sellp::compute_slice_sets, hybrid::compute_coo_row_ptrs
// (core/matrix/{csr,sellp,hybrid}_kernels.hpp); semantics =
// reference/components/{prefix_sum,format_conversion}_kernels.cpp,
// reference/matrix/csr_kernels.cpp:385-459, 768-812,
// reference/matrix/sellp_kernels.cpp:134-159,
// reference/matrix/hybrid_kernels.cpp:60-71.
//
// All HBM-bound scatters/scans; the exclusive scan is the classic three
// phases (1024-element workgroup scans, one workgroup scanning the workgroup
// totals, offset add) with caller-provided scratch.
#include "common.hpp"

#include <vector>
#include <algorithm>
#include <type_traits>

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr int scan_items = 4;
constexpr int scan_tile = block * scan_items;  // 1024 elements per workgroup

// exclusive scan of one tile held as scan_items consecutive values per thread;
// returns the tile total in every thread
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T (&v)[scan_items], T* smem)
{
    T local = 0;
#pragma unroll
    for (int i = 0; i < scan_items; ++i) {
        const T t = v[i];
        v[i] = local;
        local += t;
    }
    // inclusive scan of the per-thread totals across the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    T wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < block / 64; ++w) {
        if (w < wave) wave_off += smem[w];
        total += smem[w];
    }
    const T thread_off = wave_off + incl - local;
#pragma unroll
    for (int i = 0; i < scan_items; ++i) v[i] += thread_off;
    __syncthreads();
    return total;
}

template <typename T>
__global__ __launch_bounds__(block) void scan_tiles_kernel(T* __restrict__ data, int64_t n,
                                                          T* __restrict__ tile_totals)
{
    __shared__ T smem[block / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * scan_tile + threadIdx.x * scan_items;
    T v[scan_items];
#pragma unroll
    for (int i = 0; i < scan_items; ++i) v[i] = base + i < n ? data[base + i] : T{0};
    const T total = block_exclusive_scan(v, smem);
#pragma unroll
    for (int i = 0; i < scan_items; ++i)
        if (base + i < n) data[base + i] = v[i];
    if (threadIdx.x == 0) tile_totals[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the tile totals, in chunks with a carry
template <typename T>
__global__ __launch_bounds__(block) void scan_totals_kernel(T* __restrict__ totals, int64_t ntiles)
{
    __shared__ T smem[block / 64];
    T carry = 0;
    for (int64_t chunk = 0; chunk < ntiles; chunk += scan_tile) {
        const int64_t base = chunk + threadIdx.x * scan_items;
        T v[scan_items];
#pragma unroll
        for (int i = 0; i < scan_items; ++i) v[i] = base + i < ntiles ? totals[base + i] : T{0};
        const T total = block_exclusive_scan(v, smem);
#pragma unroll
        for (int i = 0; i < scan_items; ++i)
            if (base + i < ntiles) totals[base + i] = v[i] + carry;
        carry += total;
    }
}

template <typename T>
__global__ __launch_bounds__(block) void scan_add_offsets_kernel(T* __restrict__ data, int64_t n,
                                                                const T* __restrict__ tile_offsets)
{
    const T off = tile_offsets[blockIdx.x];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * scan_tile + threadIdx.x * scan_items;
#pragma unroll
    for (int i = 0; i < scan_items; ++i)
        if (base + i < n) data[base + i] += off;
}

template <typename T>
int prefix_sum_impl(hipStream_t s, T* data, int64_t n, void* ws, size_t ws_bytes)
{
    if (n < 0) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    const int64_t ntiles = ceildiv(n, scan_tile);
    if (ntiles > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    if (ws == nullptr || ws_bytes < sizeof(T) * static_cast<size_t>(ntiles)) return GKOMI_EWORKSPACE;
    T* totals = static_cast<T*>(ws);
    hipLaunchKernelGGL(scan_tiles_kernel<T>, dim3(static_cast<unsigned>(ntiles)), dim3(block), 0, s,
                       data, n, totals);
    if (ntiles > 1) {
        hipLaunchKernelGGL(scan_totals_kernel<T>, dim3(1), dim3(block), 0, s, totals, ntiles);
        hipLaunchKernelGGL(scan_add_offsets_kernel<T>, dim3(static_cast<unsigned>(ntiles)),
                           dim3(block), 0, s, data, n, totals);
    }
    return check_launch();
}

template <typename I>
__global__ __launch_bounds__(block) void ptrs_to_idxs_kernel(const I* __restrict__ ptrs,
                                                            int64_t num_blocks,
                                                            I* __restrict__ idxs)
{
    // one sub-wave of 8 lanes per block entry keeps short rows coalesced
    constexpr int sub = 8;
    const int64_t gid = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block / sub;
    const int lane = threadIdx.x % sub;
    for (int64_t blk = gid / sub; blk < num_blocks; blk += step) {
        const I end = ptrs[blk + 1];
        for (I i = ptrs[blk] + lane; i < end; i += sub) idxs[i] = static_cast<I>(blk);
    }
}

template <typename I>
__global__ __launch_bounds__(block) void count_idxs_kernel(const I* __restrict__ idxs,
                                                          int64_t num_idxs,
                                                          I* __restrict__ ptrs)
{
    using counter = typename std::conditional<sizeof(I) == 8, unsigned long long, int>::type;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < num_idxs;
         i += static_cast<int64_t>(gridDim.x) * block) {
        atomicAdd(reinterpret_cast<counter*>(ptrs + idxs[i]), counter{1});  // integer: exact in any order
    }
}

template <typename I>
__global__ __launch_bounds__(block) void ptrs_to_sizes_kernel(const I* __restrict__ ptrs,
                                                             int64_t num_blocks,
                                                             uint64_t* __restrict__ sizes)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < num_blocks;
         i += static_cast<int64_t>(gridDim.x) * block) {
        sizes[i] = static_cast<uint64_t>(ptrs[i + 1] - ptrs[i]);
    }
}

__global__ __launch_bounds__(block) void csr_to_ell_kernel(
    int64_t nrows, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, int64_t num_stored, int64_t stride,
    int32_t* __restrict__ ell_cols, double* __restrict__ ell_vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        const int32_t begin = row_ptrs[row];
        const int64_t len = row_ptrs[row + 1] - begin;
        for (int64_t i = 0; i < num_stored; ++i) {
            const bool in = i < len;
            ell_vals[row + i * stride] = in ? vals[begin + i] : 0.0;
            ell_cols[row + i * stride] = in ? col_idxs[begin + i] : -1;
        }
    }
}

__global__ __launch_bounds__(block) void slice_lengths_kernel(
    const int32_t* __restrict__ row_ptrs, int64_t nrows, int64_t slice_size, int64_t stride_factor,
    int64_t num_slices, uint64_t* __restrict__ slice_sets, uint64_t* __restrict__ slice_lengths)
{
    for (int64_t slice = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; slice <= num_slices;
         slice += static_cast<int64_t>(gridDim.x) * block) {
        uint64_t len = 0;
        if (slice < num_slices) {
            for (int64_t lr = 0; lr < slice_size; ++lr) {
                const int64_t row = slice * slice_size + lr;
                const int64_t rl = row < nrows ? row_ptrs[row + 1] - row_ptrs[row] : 0;
                const uint64_t padded =
                    static_cast<uint64_t>((rl + stride_factor - 1) / stride_factor * stride_factor);
                len = max(len, padded);
            }
            slice_lengths[slice] = len;
        }
        slice_sets[slice] = len;  // scanned afterwards; entry num_slices starts as 0
    }
}

__global__ __launch_bounds__(block) void csr_to_sellp_kernel(
    int64_t nrows, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, int64_t slice_size, const uint64_t* __restrict__ slice_sets,
    const uint64_t* __restrict__ slice_lengths, int32_t* __restrict__ out_cols,
    double* __restrict__ out_vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t slice = row / slice_size, local = row % slice_size;
        const int64_t len = static_cast<int64_t>(slice_lengths[slice]);
        const int64_t base = static_cast<int64_t>(slice_sets[slice]) * slice_size + local;
        const int32_t begin = row_ptrs[row];
        const int64_t rl = row_ptrs[row + 1] - begin;
        for (int64_t i = 0; i < len; ++i) {
            const bool in = i < rl;
            out_vals[base + i * slice_size] = in ? vals[begin + i] : 0.0;
            out_cols[base + i * slice_size] = in ? col_idxs[begin + i] : -1;
        }
    }
}

__global__ __launch_bounds__(block) void coo_overflow_counts_kernel(
    const int32_t* __restrict__ row_ptrs, int64_t nrows, int64_t ell_lim,
    int64_t* __restrict__ coo_row_ptrs)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row <= nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        int64_t v = 0;
        if (row < nrows) {
            const int64_t nnz = row_ptrs[row + 1] - row_ptrs[row];
            v = nnz <= ell_lim ? 0 : nnz - ell_lim;
        }
        coo_row_ptrs[row] = v;
    }
}

__global__ __launch_bounds__(block) void csr_to_hybrid_kernel(
    int64_t nrows, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, const int64_t* __restrict__ coo_row_ptrs, int64_t ell_lim,
    int64_t ell_stride, int32_t* __restrict__ ell_cols, double* __restrict__ ell_vals,
    int32_t* __restrict__ coo_rows, int32_t* __restrict__ coo_cols, double* __restrict__ coo_vals)
{
    // rows in [nrows, ell_stride) are padding rows of the ELL part
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < ell_stride;
         row += static_cast<int64_t>(gridDim.x) * block) {
        const bool real = row < nrows;
        const int32_t begin = real ? row_ptrs[row] : 0;
        const int64_t len = real ? row_ptrs[row + 1] - begin : 0;
        for (int64_t i = 0; i < ell_lim; ++i) {
            const bool in = i < len;
            ell_vals[row + i * ell_stride] = in ? vals[begin + i] : 0.0;
            ell_cols[row + i * ell_stride] = in ? col_idxs[begin + i] : -1;
        }
        if (real) {
            int64_t out = coo_row_ptrs[row];
            for (int64_t i = ell_lim; i < len; ++i, ++out) {
                coo_vals[out] = vals[begin + i];
                coo_cols[out] = col_idxs[begin + i];
                coo_rows[out] = static_cast<int32_t>(row);
            }
        }
    }
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_prefix_sum_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return sizeof(int64_t) * static_cast<size_t>(ceildiv(n, scan_tile));
}

extern "C" int gkomi_prefix_sum_i32(gkomi_stream_t s, int32_t* counts, int64_t n, void* workspace,
                                    size_t workspace_bytes)
{
    return prefix_sum_impl<int32_t>(to_stream(s), counts, n, workspace, workspace_bytes);
}

extern "C" int gkomi_prefix_sum_i64(gkomi_stream_t s, int64_t* counts, int64_t n, void* workspace,
                                    size_t workspace_bytes)
{
    return prefix_sum_impl<int64_t>(to_stream(s), counts, n, workspace, workspace_bytes);
}

extern "C" int gkomi_convert_ptrs_to_idxs_i32(gkomi_stream_t s, const int32_t* ptrs,
                                              int64_t num_blocks, int32_t* idxs)
{
    if (num_blocks < 0) return GKOMI_EINVAL;
    if (num_blocks == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(ptrs_to_idxs_kernel<int32_t>, dim3(grid_for(num_blocks * 8, block, 1 << 16)),
                       dim3(block), 0, to_stream(s), ptrs, num_blocks, idxs);
    return check_launch();
}

extern "C" int gkomi_convert_idxs_to_ptrs_i32(gkomi_stream_t s, const int32_t* idxs,
                                              int64_t num_idxs, int64_t num_blocks, int32_t* ptrs,
                                              void* workspace, size_t workspace_bytes)
{
    if (num_idxs < 0 || num_blocks < 0) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(s);
    int err = static_cast<int>(
        hipMemsetAsync(ptrs, 0, sizeof(int32_t) * static_cast<size_t>(num_blocks + 1), stream));
    if (err) return err;
    if (num_idxs > 0) {
        hipLaunchKernelGGL(count_idxs_kernel<int32_t>, dim3(grid_for(num_idxs, block)), dim3(block), 0,
                           stream, idxs, num_idxs, ptrs);
        err = check_launch();
        if (err) return err;
    }
    return gkomi_prefix_sum_i32(s, ptrs, num_blocks + 1, workspace, workspace_bytes);
}

extern "C" int gkomi_convert_ptrs_to_sizes_i32(gkomi_stream_t s, const int32_t* ptrs,
                                               int64_t num_blocks, uint64_t* sizes)
{
    if (num_blocks < 0) return GKOMI_EINVAL;
    if (num_blocks == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(ptrs_to_sizes_kernel<int32_t>, dim3(grid_for(num_blocks, block)), dim3(block), 0,
                       to_stream(s), ptrs, num_blocks, sizes);
    return check_launch();
}

// <int64>: the components:: kernels of the index type a > 2^31-nonzero matrix needs
extern "C" int gkomi_convert_ptrs_to_idxs_i64(gkomi_stream_t s, const int64_t* ptrs, int64_t num_blocks, int64_t* idxs)
{
    if (num_blocks < 0) return GKOMI_EINVAL;
    if (num_blocks == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(ptrs_to_idxs_kernel<int64_t>, dim3(grid_for(num_blocks * 8, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), ptrs, num_blocks, idxs);
    return check_launch();
}

extern "C" int gkomi_convert_idxs_to_ptrs_i64(gkomi_stream_t s, const int64_t* idxs, int64_t num_idxs, int64_t num_blocks,
                                              int64_t* ptrs, void* workspace, size_t workspace_bytes)
{
    if (num_idxs < 0 || num_blocks < 0) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(s);
    int err = static_cast<int>(hipMemsetAsync(ptrs, 0, sizeof(int64_t) * static_cast<size_t>(num_blocks + 1), stream));
    if (err) return err;
    if (num_idxs > 0) {
        hipLaunchKernelGGL(count_idxs_kernel<int64_t>, dim3(grid_for(num_idxs, block)), dim3(block), 0, stream, idxs,
                           num_idxs, ptrs);
        err = check_launch();
        if (err) return err;
    }
    return gkomi_prefix_sum_i64(s, ptrs, num_blocks + 1, workspace, workspace_bytes);
}

extern "C" int gkomi_convert_ptrs_to_sizes_i64(gkomi_stream_t s, const int64_t* ptrs, int64_t num_blocks, uint64_t* sizes)
{
    if (num_blocks < 0) return GKOMI_EINVAL;
    if (num_blocks == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(ptrs_to_sizes_kernel<int64_t>, dim3(grid_for(num_blocks, block)), dim3(block), 0, to_stream(s), ptrs,
                       num_blocks, sizes);
    return check_launch();
}

extern "C" int gkomi_csr_convert_to_ell_f64_i32(gkomi_stream_t s, int64_t nrows,
                                                const int32_t* row_ptrs, const int32_t* col_idxs,
                                                const double* vals, int64_t num_stored_per_row,
                                                int64_t stride, int32_t* ell_col_idxs,
                                                double* ell_vals)
{
    if (nrows < 0 || num_stored_per_row < 0 || stride < nrows) return GKOMI_EINVAL;
    if (nrows == 0 || num_stored_per_row == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(csr_to_ell_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), nrows, row_ptrs, col_idxs, vals, num_stored_per_row, stride,
                       ell_col_idxs, ell_vals);
    return check_launch();
}

extern "C" int gkomi_sellp_compute_slice_sets_i32(gkomi_stream_t s, const int32_t* row_ptrs,
                                                  int64_t nrows, int64_t slice_size,
                                                  int64_t stride_factor, uint64_t* slice_sets,
                                                  uint64_t* slice_lengths, void* workspace,
                                                  size_t workspace_bytes)
{
    if (nrows < 0 || slice_size <= 0 || stride_factor <= 0) return GKOMI_EINVAL;
    const int64_t num_slices = ceildiv(nrows, slice_size);
    hipLaunchKernelGGL(slice_lengths_kernel, dim3(grid_for(num_slices + 1, block)), dim3(block), 0,
                       to_stream(s), row_ptrs, nrows, slice_size, stride_factor, num_slices,
                       slice_sets, slice_lengths);
    int err = check_launch();
    if (err) return err;
    return gkomi_prefix_sum_i64(s, reinterpret_cast<int64_t*>(slice_sets), num_slices + 1,
                                workspace, workspace_bytes);
}

extern "C" int gkomi_csr_convert_to_sellp_f64_i32(
    gkomi_stream_t s, int64_t nrows, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, int64_t slice_size, const uint64_t* slice_sets,
    const uint64_t* slice_lengths, int32_t* out_col_idxs, double* out_vals)
{
    if (nrows < 0 || slice_size <= 0) return GKOMI_EINVAL;
    if (nrows == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(csr_to_sellp_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), nrows, row_ptrs, col_idxs, vals, slice_size, slice_sets,
                       slice_lengths, out_col_idxs, out_vals);
    return check_launch();
}

extern "C" int gkomi_hybrid_compute_coo_row_ptrs_i32(gkomi_stream_t s, const int32_t* row_ptrs,
                                                     int64_t nrows, int64_t ell_lim,
                                                     int64_t* coo_row_ptrs, void* workspace,
                                                     size_t workspace_bytes)
{
    if (nrows < 0 || ell_lim < 0) return GKOMI_EINVAL;
    hipLaunchKernelGGL(coo_overflow_counts_kernel, dim3(grid_for(nrows + 1, block)), dim3(block),
                       0, to_stream(s), row_ptrs, nrows, ell_lim, coo_row_ptrs);
    int err = check_launch();
    if (err) return err;
    return gkomi_prefix_sum_i64(s, coo_row_ptrs, nrows + 1, workspace, workspace_bytes);
}

extern "C" int gkomi_csr_convert_to_hybrid_f64_i32(
    gkomi_stream_t s, int64_t nrows, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, const int64_t* coo_row_ptrs, int64_t ell_lim, int64_t ell_stride,
    int32_t* ell_col_idxs, double* ell_vals, int32_t* coo_row_idxs, int32_t* coo_col_idxs,
    double* coo_vals)
{
    if (nrows < 0 || ell_lim < 0 || ell_stride < nrows) return GKOMI_EINVAL;
    if (ell_stride == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(csr_to_hybrid_kernel, dim3(grid_for(ell_stride, block, 1 << 16)),
                       dim3(block), 0, to_stream(s), nrows, row_ptrs, col_idxs, vals, coo_row_ptrs,
                       ell_lim, ell_stride, ell_col_idxs, ell_vals, coo_row_idxs, coo_col_idxs,
                       coo_vals);
    return check_launch();
}

// Hybrid strategies (include/ginkgo/core/matrix/hybrid.hpp:206-370).  Like the
// reference (strategy_type::compute_hybrid_config sorts a host copy of
// row_nnz), this runs on the host: blocking D2H copy of row_ptrs.
extern "C" int gkomi_hybrid_ell_width_i32(gkomi_stream_t s, const int32_t* row_ptrs,
                                          int64_t nrows, int kind, double percent, double ratio,
                                          int64_t num_columns, int64_t* host_result)
{
    if (nrows < 0 || host_result == nullptr || kind < 0 || kind > 4) return GKOMI_EINVAL;
    if (kind == 0) {
        *host_result = num_columns;
        return GKOMI_SUCCESS;
    }
    if (nrows == 0) {
        *host_result = 0;
        return GKOMI_SUCCESS;
    }
    std::vector<int32_t> ptrs(static_cast<size_t>(nrows) + 1);
    hipStream_t stream = to_stream(s);
    int err = static_cast<int>(hipMemcpyAsync(ptrs.data(), row_ptrs, sizeof(int32_t) * ptrs.size(),
                                              hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    if (err) return err;
    std::vector<uint64_t> row_nnz(static_cast<size_t>(nrows));
    for (int64_t i = 0; i < nrows; ++i) row_nnz[i] = static_cast<uint64_t>(ptrs[i + 1] - ptrs[i]);
    auto imbalance = [&](double p) -> uint64_t {
        p = std::min(p, 1.0);
        p = std::max(p, 0.0);
        std::sort(row_nnz.begin(), row_nnz.end());
        if (p < 1) return row_nnz[static_cast<size_t>(nrows * p)];
        return row_nnz[nrows - 1];
    };
    uint64_t res = 0;
    if (kind == 1) {
        res = imbalance(percent);
    } else if (kind == 2 || kind == 4) {
        if (kind == 4) {
            percent = 1.0 / 3.0;
            ratio = 0.001;
        }
        res = std::min(imbalance(percent), static_cast<uint64_t>(nrows * ratio));
    } else {
        res = imbalance(static_cast<double>(sizeof(int32_t)) /
                        (sizeof(double) + 2 * sizeof(int32_t)));
    }
    *host_result = static_cast<int64_t>(res);
    return GKOMI_SUCCESS;
}
