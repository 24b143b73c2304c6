// Row-partitioned distributed matrix setup for gfx950.  Replaces
// gko::kernels::hip::partition::{build_ranges_from_global_size,
// build_from_contiguous, build_from_mapping, build_starting_indices} and
// distributed_matrix::build_local_nonlocal (core/distributed/*_kernels.hpp;
// the reference's GPU version is Thrust sort/unique/scan,
// common/cuda_hip/distributed/matrix_kernels.hpp.inc); semantics =
// reference/distributed/partition_kernels.cpp:42-160,
// reference/distributed/matrix_kernels.cpp:49-190.
//
// Partition metadata (range bounds, part ids, starting indices) is O(#ranges)
// and handled on the host, where core/distributed/matrix.cpp consumes it for
// the communication plan.  build_local_nonlocal works on the device-resident
// COO input: classify -> exclusive scans -> stable compaction; the non-local
// columns are renumbered by sorting the 64-bit keys (owning part << 40 | global
// column) with the library's radix sort (sort_scan.hip), unique-ing with a flag scan and binary
// searching each entry's key.  All integer work: bit-exact.
#include <cstring>

#include "common.hpp"

#include "sort_scan.hpp"

#include <algorithm>

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr int col_bits = 40;  // global columns < 2^40, parts < 2^23

struct partition_view {
    const int64_t* bounds;
    const int32_t* part_ids;
    const int32_t* starts;
    int64_t num_ranges;
};

__device__ __forceinline__ int64_t find_range(int64_t idx, const partition_view& p)
{
    // upper_bound(bounds + 1, bounds + num_ranges + 1, idx) - (bounds + 1)
    int64_t lo = 0, hi = p.num_ranges;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (p.bounds[mid + 1] <= idx) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// kind[i]: 0 not mine, 1 local, 2 non-local; flags for the two scans
__global__ __launch_bounds__(block) void classify_kernel(
    int64_t nnz, const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
    partition_view rp, partition_view cp, int32_t local_part, int64_t* __restrict__ local_flag,
    int64_t* __restrict__ nonlocal_flag, uint64_t* __restrict__ keys)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i <= nnz;
         i += static_cast<int64_t>(gridDim.x) * block) {
        int64_t lf = 0, nf = 0;
        uint64_t key = ~0ull;
        if (i < nnz) {
            const int64_t rr = find_range(rows[i], rp);
            if (rp.part_ids[rr] == local_part) {
                const int64_t cr = find_range(cols[i], cp);
                const int32_t part = cp.part_ids[cr];
                if (part == local_part) {
                    lf = 1;
                } else {
                    nf = 1;
                    key = (static_cast<uint64_t>(part) << col_bits) | static_cast<uint64_t>(cols[i]);
                }
            }
        }
        local_flag[i] = lf;
        nonlocal_flag[i] = nf;
        if (i < nnz) keys[i] = key;
    }
}

__global__ __launch_bounds__(block) void compact_kernel(
    int64_t nnz, const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
    const double* __restrict__ vals, partition_view rp, partition_view cp,
    const int64_t* __restrict__ local_pos, const int64_t* __restrict__ nonlocal_pos,
    const uint64_t* __restrict__ keys, const uint64_t* __restrict__ unique_keys, int64_t num_unique,
    int32_t* __restrict__ l_rows, int32_t* __restrict__ l_cols, double* __restrict__ l_vals,
    int32_t* __restrict__ nl_rows, int32_t* __restrict__ nl_cols, double* __restrict__ nl_vals)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < nnz;
         i += static_cast<int64_t>(gridDim.x) * block) {
        const bool is_local = local_pos[i + 1] != local_pos[i];
        const bool is_nonlocal = nonlocal_pos[i + 1] != nonlocal_pos[i];
        if (!is_local && !is_nonlocal) continue;
        const int64_t rr = find_range(rows[i], rp);
        const int32_t lrow = static_cast<int32_t>(rows[i] - rp.bounds[rr]) + rp.starts[rr];
        if (is_local) {
            const int64_t cr = find_range(cols[i], cp);
            const int64_t o = local_pos[i];
            l_rows[o] = lrow;
            l_cols[o] = static_cast<int32_t>(cols[i] - cp.bounds[cr]) + cp.starts[cr];
            l_vals[o] = vals[i];
        } else {
            const uint64_t key = keys[i];
            int64_t lo = 0, hi = num_unique;
            while (lo < hi) {  // lower_bound
                const int64_t mid = (lo + hi) / 2;
                if (unique_keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            const int64_t o = nonlocal_pos[i];
            nl_rows[o] = lrow;
            nl_cols[o] = static_cast<int32_t>(lo);
            nl_vals[o] = vals[i];
        }
    }
}

// flag[i] = 1 for the first occurrence of a key in the sorted array (sentinel ~0 excluded)
__global__ __launch_bounds__(block) void unique_flag_kernel(int64_t n,
                                                           const uint64_t* __restrict__ sorted,
                                                           int64_t* __restrict__ flag)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i <= n;
         i += static_cast<int64_t>(gridDim.x) * block) {
        int64_t f = 0;
        if (i < n) {
            const uint64_t k = sorted[i];
            f = (k != ~0ull && (i == 0 || sorted[i - 1] != k)) ? 1 : 0;
        }
        flag[i] = f;
    }
}

__global__ __launch_bounds__(block) void unique_scatter_kernel(
    int64_t n, const uint64_t* __restrict__ sorted, const int64_t* __restrict__ pos,
    uint64_t* __restrict__ unique_keys)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * block) {
        if (pos[i + 1] != pos[i]) unique_keys[pos[i]] = sorted[i];
    }
}

__global__ __launch_bounds__(block) void gather_info_kernel(
    int64_t num_unique, const uint64_t* __restrict__ unique_keys, partition_view cp,
    int32_t* __restrict__ gather_idxs, int32_t* __restrict__ recv_sizes,
    int64_t* __restrict__ non_local_to_global)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < num_unique;
         i += static_cast<int64_t>(gridDim.x) * block) {
        const uint64_t key = unique_keys[i];
        const int64_t col = static_cast<int64_t>(key & ((1ull << col_bits) - 1));
        const int32_t part = static_cast<int32_t>(key >> col_bits);
        const int64_t r = find_range(col, cp);
        gather_idxs[i] = static_cast<int32_t>(col - cp.bounds[r]) + cp.starts[r];
        non_local_to_global[i] = col;
        atomicAdd(recv_sizes + part, 1);  // integer histogram: exact
    }
}

// distributed_vector::build_local: the entries whose row this part owns, scattered into the dense local block
// (common/cuda_hip/distributed/vector_kernels.hpp.inc:33-98 does it with thrust::upper_bound + scatter_if)
__global__ __launch_bounds__(block) void vector_build_local_kernel(
    int64_t nnz, const int64_t* __restrict__ rows, const int64_t* __restrict__ cols, const double* __restrict__ vals,
    partition_view rp, int32_t local_part, double* __restrict__ local, int64_t stride)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < nnz;
         i += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t rr = find_range(rows[i], rp);
        if (rp.part_ids[rr] != local_part) continue;
        const int64_t lrow = (rows[i] - rp.bounds[rr]) + rp.starts[rr];
        local[lrow * stride + cols[i]] = vals[i];
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct dist_layout {
    size_t local_pos, nonlocal_pos, keys, sorted, unique_pos, unique_keys, scan, sort_tmp, total;
    size_t sort_tmp_bytes;
};

dist_layout make_layout(int64_t nnz)
{
    dist_layout l{};
    const size_t n1 = static_cast<size_t>(nnz) + 1;
    size_t off = 0;
    l.local_pos = off; off += align_up(8 * n1, 256);
    l.nonlocal_pos = off; off += align_up(8 * n1, 256);
    l.keys = off; off += align_up(8 * n1, 256);
    l.sorted = off; off += align_up(8 * n1, 256);
    l.unique_pos = off; off += align_up(8 * n1, 256);
    l.unique_keys = off; off += align_up(8 * n1, 256);
    l.scan = off; off += align_up(gkomi_prefix_sum_workspace_bytes(nnz + 1) + 8, 256);
    const size_t tmp = nnz > 0 ? radix_sort_workspace_bytes(nnz, sizeof(uint64_t), false) : 0;
    l.sort_tmp_bytes = tmp;
    l.sort_tmp = off; off += align_up(tmp + 8, 256);
    l.total = off;
    return l;
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

// ---- Partition metadata: host arrays, O(number of ranges) -------------------------

extern "C" int gkomi_partition_build_ranges_from_global_size(int64_t num_parts, int64_t global_size,
                                                             int64_t* host_ranges)
{
    if (num_parts <= 0 || global_size < 0 || host_ranges == nullptr) return GKOMI_EINVAL;
    const int64_t per = global_size / num_parts;
    const int64_t rest = global_size - num_parts * per;
    host_ranges[0] = 0;
    for (int64_t i = 1; i < num_parts + 1; ++i) {
        host_ranges[i] = host_ranges[i - 1] + per + ((i - 1) < rest ? 1 : 0);
    }
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_partition_build_from_contiguous(int64_t num_parts, const int64_t* host_ranges,
                                                     int64_t* host_range_bounds,
                                                     int32_t* host_part_ids)
{
    if (num_parts < 0 || host_ranges == nullptr) return GKOMI_EINVAL;
    host_range_bounds[0] = 0;
    for (int64_t i = 0; i < num_parts; ++i) {
        host_range_bounds[i + 1] = host_ranges[i + 1];
        host_part_ids[i] = static_cast<int32_t>(i);
    }
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_partition_build_from_mapping(int64_t n, const int32_t* host_mapping,
                                                  int64_t* host_range_bounds,
                                                  int32_t* host_part_ids, int64_t* host_num_ranges)
{
    if (n < 0 || host_num_ranges == nullptr) return GKOMI_EINVAL;
    int64_t range_idx = 0;
    int32_t range_part = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (host_mapping[i] != range_part) {
            host_range_bounds[range_idx] = i;
            host_part_ids[range_idx] = host_mapping[i];
            ++range_idx;
            range_part = host_mapping[i];
        }
    }
    host_range_bounds[range_idx] = n;
    *host_num_ranges = range_idx;
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_partition_build_starting_indices(const int64_t* host_range_bounds,
                                                      const int32_t* host_part_ids,
                                                      int64_t num_ranges, int64_t num_parts,
                                                      int32_t* host_starting_indices,
                                                      int32_t* host_part_sizes,
                                                      int64_t* host_num_empty_parts)
{
    if (num_ranges < 0 || num_parts < 0) return GKOMI_EINVAL;
    std::fill_n(host_part_sizes, num_parts, 0);
    for (int64_t r = 0; r < num_ranges; ++r) {
        const int32_t part = host_part_ids[r];
        if (part < 0 || part >= num_parts) return GKOMI_EINVAL;
        host_starting_indices[r] = host_part_sizes[part];
        host_part_sizes[part] += static_cast<int32_t>(host_range_bounds[r + 1] - host_range_bounds[r]);
    }
    if (host_num_empty_parts != nullptr) {
        *host_num_empty_parts = std::count(host_part_sizes, host_part_sizes + num_parts, 0);
    }
    return GKOMI_SUCCESS;
}

// partition::has_ordered_parts (reference/distributed/partition_kernels.cpp:139-155): the part ids of consecutive
// ranges never decrease.  (Partition::has_ordered_parts asks it only of partitions whose parts are connected --
// num_parts - num_empty_parts == num_ranges, core/distributed/partition.cpp:120-138.)
extern "C" int gkomi_partition_has_ordered_parts(const int32_t* host_part_ids, int64_t num_ranges, int64_t* host_result)
{
    if (num_ranges < 0 || host_result == nullptr || (num_ranges > 0 && host_part_ids == nullptr)) return GKOMI_EINVAL;
    *host_result = 1;
    for (int64_t i = 1; i < num_ranges; ++i) {
        if (host_part_ids[i] < host_part_ids[i - 1]) {
            *host_result = 0;
            break;
        }
    }
    return GKOMI_SUCCESS;
}

// distributed_vector::build_local (reference/distributed/vector_kernels.cpp:47-96; Vector::read_distributed,
// core/distributed/vector.cpp:120-170): local(local row of rows[i], cols[i]) = vals[i] for the entries of
// local_part; everything else of `local` is left as the caller set it (the reference fills it with zeros first).
// Device arrays; the partition arrays are device copies of the host metadata as for build_local_nonlocal.
// Entries with the same (row, column): one of them wins (the reference's sequential loop keeps the last, its
// GPU scatter_if any).
extern "C" int gkomi_dist_vector_build_local_f64(gkomi_stream_t s, int64_t nnz, const int64_t* rows, const int64_t* cols,
                                                 const double* vals, const int64_t* range_bounds, const int32_t* part_ids,
                                                 const int32_t* starts, int64_t num_ranges, int32_t local_part,
                                                 double* local, int64_t local_stride)
{
    if (nnz < 0 || num_ranges < 0 || local_stride < 0) return GKOMI_EINVAL;
    if (nnz == 0) return GKOMI_SUCCESS;
    if (rows == nullptr || cols == nullptr || vals == nullptr || local == nullptr || range_bounds == nullptr ||
        part_ids == nullptr || starts == nullptr) {
        return GKOMI_EINVAL;
    }
    const partition_view rp{range_bounds, part_ids, starts, num_ranges};
    hipLaunchKernelGGL(vector_build_local_kernel, dim3(grid_for(nnz, block)), dim3(block), 0, to_stream(s), nnz, rows, cols,
                       vals, rp, local_part, local, local_stride);
    return check_launch();
}

// ---- build_local_nonlocal: two phases, because the caller owns the outputs ---------

extern "C" size_t gkomi_dist_build_workspace_bytes(int64_t nnz)
{
    if (nnz < 0) return 0;
    return make_layout(nnz).total;
}

// Phase 1: classify, scan, sort and unique.  host_sizes = {num_local,
// num_non_local, num_unique_non_local_cols} (blocking copy).
extern "C" int gkomi_dist_build_local_nonlocal_sizes(
    gkomi_stream_t s, int64_t nnz, const int64_t* rows, const int64_t* cols,
    const int64_t* row_range_bounds, const int32_t* row_part_ids, const int32_t* row_starts,
    int64_t row_num_ranges, const int64_t* col_range_bounds, const int32_t* col_part_ids,
    const int32_t* col_starts, int64_t col_num_ranges, int32_t local_part, void* workspace,
    size_t workspace_bytes, int64_t host_sizes[3])
{
    if (nnz < 0 || host_sizes == nullptr) return GKOMI_EINVAL;
    const dist_layout l = make_layout(nnz);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    int64_t* local_pos = reinterpret_cast<int64_t*>(ws + l.local_pos);
    int64_t* nonlocal_pos = reinterpret_cast<int64_t*>(ws + l.nonlocal_pos);
    uint64_t* keys = reinterpret_cast<uint64_t*>(ws + l.keys);
    uint64_t* sorted = reinterpret_cast<uint64_t*>(ws + l.sorted);
    int64_t* unique_pos = reinterpret_cast<int64_t*>(ws + l.unique_pos);
    uint64_t* unique_keys = reinterpret_cast<uint64_t*>(ws + l.unique_keys);
    void* scan_ws = ws + l.scan;
    const size_t scan_bytes = gkomi_prefix_sum_workspace_bytes(nnz + 1) + 8;
    partition_view rp{row_range_bounds, row_part_ids, row_starts, row_num_ranges};
    partition_view cp{col_range_bounds, col_part_ids, col_starts, col_num_ranges};
    hipLaunchKernelGGL(classify_kernel, dim3(grid_for(nnz + 1, block)), dim3(block), 0, stream, nnz,
                       rows, cols, rp, cp, local_part, local_pos, nonlocal_pos, keys);
    int err = check_launch();
    if (err) return err;
    err = gkomi_prefix_sum_i64(s, local_pos, nnz + 1, scan_ws, scan_bytes);
    if (err) return err;
    err = gkomi_prefix_sum_i64(s, nonlocal_pos, nnz + 1, scan_ws, scan_bytes);
    if (err) return err;
    if (nnz > 0) {
        size_t tmp_bytes = l.sort_tmp_bytes;
        // keys = part << 40 | global column: 64-bit stable radix sort (sort_scan.hip)
        err = radix_sort_u64(stream, nnz, keys, sorted, nullptr, nullptr, 64, ws + l.sort_tmp, tmp_bytes);
        if (err) return err;
    }
    hipLaunchKernelGGL(unique_flag_kernel, dim3(grid_for(nnz + 1, block)), dim3(block), 0, stream,
                       nnz, sorted, unique_pos);
    err = gkomi_prefix_sum_i64(s, unique_pos, nnz + 1, scan_ws, scan_bytes);
    if (err) return err;
    if (nnz > 0) {
        hipLaunchKernelGGL(unique_scatter_kernel, dim3(grid_for(nnz, block)), dim3(block), 0, stream,
                           nnz, sorted, unique_pos, unique_keys);
    }
    err = check_launch();
    if (err) return err;
    int64_t tails[3];
    err = static_cast<int>(hipMemcpyAsync(&tails[0], local_pos + nnz, 8, hipMemcpyDeviceToHost, stream));
    if (!err) err = static_cast<int>(hipMemcpyAsync(&tails[1], nonlocal_pos + nnz, 8, hipMemcpyDeviceToHost, stream));
    if (!err) err = static_cast<int>(hipMemcpyAsync(&tails[2], unique_pos + nnz, 8, hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    host_sizes[0] = tails[0];
    host_sizes[1] = tails[1];
    host_sizes[2] = tails[2];
    return err;
}

// Phase 2: fill the caller-allocated outputs from the same workspace.
// recv_sizes: device int32[num_parts].
extern "C" int gkomi_dist_build_local_nonlocal_fill(
    gkomi_stream_t s, int64_t nnz, const int64_t* rows, const int64_t* cols, const double* vals,
    const int64_t* row_range_bounds, const int32_t* row_part_ids, const int32_t* row_starts,
    int64_t row_num_ranges, const int64_t* col_range_bounds, const int32_t* col_part_ids,
    const int32_t* col_starts, int64_t col_num_ranges, int64_t num_parts, const void* workspace,
    int64_t num_unique, int32_t* local_row_idxs, int32_t* local_col_idxs, double* local_vals,
    int32_t* non_local_row_idxs, int32_t* non_local_col_idxs, double* non_local_vals,
    int32_t* gather_idxs, int32_t* recv_sizes, int64_t* non_local_to_global)
{
    if (nnz < 0 || workspace == nullptr || num_parts < 0) return GKOMI_EINVAL;
    const dist_layout l = make_layout(nnz);
    hipStream_t stream = to_stream(s);
    const char* ws = static_cast<const char*>(workspace);
    const int64_t* local_pos = reinterpret_cast<const int64_t*>(ws + l.local_pos);
    const int64_t* nonlocal_pos = reinterpret_cast<const int64_t*>(ws + l.nonlocal_pos);
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(ws + l.keys);
    const uint64_t* unique_keys = reinterpret_cast<const uint64_t*>(ws + l.unique_keys);
    partition_view rp{row_range_bounds, row_part_ids, row_starts, row_num_ranges};
    partition_view cp{col_range_bounds, col_part_ids, col_starts, col_num_ranges};
    int err = static_cast<int>(
        hipMemsetAsync(recv_sizes, 0, sizeof(int32_t) * static_cast<size_t>(num_parts), stream));
    if (err) return err;
    if (nnz > 0) {
        hipLaunchKernelGGL(compact_kernel, dim3(grid_for(nnz, block)), dim3(block), 0, stream, nnz,
                           rows, cols, vals, rp, cp, local_pos, nonlocal_pos, keys, unique_keys,
                           num_unique, local_row_idxs, local_col_idxs, local_vals,
                           non_local_row_idxs, non_local_col_idxs, non_local_vals);
    }
    if (num_unique > 0) {
        hipLaunchKernelGGL(gather_info_kernel, dim3(grid_for(num_unique, block)), dim3(block), 0,
                           stream, num_unique, unique_keys, cp, gather_idxs, recv_sizes,
                           non_local_to_global);
    }
    return check_launch();
}
