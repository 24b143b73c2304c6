// Matrix assembly on the device for gfx950: the device_matrix_data kernels the
// format readers run before any SpMV.  Replaces
// gko::kernels::hip::components::{sort_row_major, sum_duplicates,
// remove_zeros, soa_to_aos, aos_to_soa}
// (core/base/device_matrix_data_kernels.hpp; the reference's GPU version is
// Thrust, common/cuda_hip/base/device_matrix_data_kernels.hpp.inc); semantics =
// reference/base/device_matrix_data_kernels.cpp:52-190.  `Csr::read` itself is
// then convert_idxs_to_ptrs on the sorted row indices (core/matrix/csr.cpp:
// 453-470, conversions.hip).
//
//  * sort_row_major: stable LSD radix sort (sort_scan.hip) of the 64-bit keys
//    row << 32 | col carrying the entry's position, then one gather of the
//    values.  The reference uses std::sort, which leaves the order of duplicate
//    (row, col) entries unspecified; this sort keeps their input order.
//  * sum_duplicates: head flags -> exclusive scan -> one thread per output
//    entry adds its run left to right starting from 0 (the reference's
//    `new = 0; new += v...`) -> bit-identical sums (including -0 + 0 = +0).
//  * remove_zeros: flag -> scan -> stable scatter.
// The compacting kernels write into caller-provided arrays of capacity nnz and
// return the new count to the host (blocking, like the reference's
// array::resize_and_reset decision).
#include <cstring>

#include "common.hpp"

#include "sort_scan.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

struct layout {
    size_t keys_in, keys_out, idx_in, idx_out, vals, flags, scan_ws, sort_tmp, total;
};

size_t sort_tmp_bytes(int64_t nnz) { return radix_sort_workspace_bytes(nnz, sizeof(uint64_t), true); }

layout make_layout(int64_t nnz)
{
    layout l{};
    const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align256(bytes);
        return at;
    };
    l.keys_in = take(8 * n);
    l.keys_out = take(8 * n);
    l.idx_in = take(4 * n);
    l.idx_out = take(4 * n);
    l.vals = take(8 * n);
    l.flags = take(4 * (n + 1));
    l.scan_ws = take(gkomi_prefix_sum_workspace_bytes(static_cast<int64_t>(n) + 1));
    l.sort_tmp = take(sort_tmp_bytes(static_cast<int64_t>(n)));
    l.total = off;
    return l;
}

__global__ __launch_bounds__(block) void make_keys_kernel(int64_t nnz,
                                                          const int32_t* __restrict__ rows,
                                                          const int32_t* __restrict__ cols,
                                                          uint64_t* __restrict__ keys,
                                                          uint32_t* __restrict__ idx)
{
    const int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (i < nnz) {
        // matrix_data_entry::operator< compares (row, column) as signed values;
        // flipping the sign bits makes the unsigned key order the same
        const uint32_t r = static_cast<uint32_t>(rows[i]) ^ 0x80000000u;
        const uint32_t c = static_cast<uint32_t>(cols[i]) ^ 0x80000000u;
        keys[i] = static_cast<uint64_t>(r) << 32 | c;
        idx[i] = static_cast<uint32_t>(i);
    }
}

__global__ __launch_bounds__(block) void apply_sort_kernel(int64_t nnz,
                                                           const uint64_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ idx,
                                                           const double* __restrict__ vals_in,
                                                           int32_t* __restrict__ rows,
                                                           int32_t* __restrict__ cols,
                                                           double* __restrict__ vals)
{
    const int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (i < nnz) {
        const uint64_t k = keys[i];
        rows[i] = static_cast<int32_t>(static_cast<uint32_t>(k >> 32) ^ 0x80000000u);
        cols[i] = static_cast<int32_t>(static_cast<uint32_t>(k) ^ 0x80000000u);
        vals[i] = vals_in[idx[i]];
    }
}

__global__ __launch_bounds__(block) void nonzero_flags_kernel(int64_t nnz,
                                                              const double* __restrict__ vals,
                                                              int32_t* __restrict__ flags)
{
    const int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (i < nnz) flags[i] = vals[i] != 0.0 ? 1 : 0;  // is_nonzero: NaN counts as nonzero
    if (i == nnz) flags[i] = 0;
}

__global__ __launch_bounds__(block) void head_flags_kernel(int64_t nnz,
                                                           const int32_t* __restrict__ rows,
                                                           const int32_t* __restrict__ cols,
                                                           int32_t* __restrict__ flags)
{
    const int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (i < nnz) {
        flags[i] = (i == 0 || rows[i] != rows[i - 1] || cols[i] != cols[i - 1]) ? 1 : 0;
    }
    if (i == nnz) flags[i] = 0;
}

// offsets = exclusive scan of the flags (offsets[nnz] = number kept)
__global__ __launch_bounds__(block) void compact_kernel(
    int64_t nnz, const int32_t* __restrict__ offsets, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ cols, const double* __restrict__ vals,
    int32_t* __restrict__ out_rows, int32_t* __restrict__ out_cols, double* __restrict__ out_vals)
{
    const int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (i < nnz && offsets[i + 1] != offsets[i]) {
        const int32_t o = offsets[i];
        out_rows[o] = rows[i];
        out_cols[o] = cols[i];
        out_vals[o] = vals[i];
    }
}

__global__ __launch_bounds__(block) void sum_runs_kernel(
    int64_t nnz, const int32_t* __restrict__ offsets, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ cols, const double* __restrict__ vals,
    int32_t* __restrict__ out_rows, int32_t* __restrict__ out_cols, double* __restrict__ out_vals)
{
    const int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (i < nnz && offsets[i + 1] != offsets[i]) {  // head of a run of equal (row, col)
        const int32_t o = offsets[i];
        double acc = 0.0;
        int64_t k = i;
        do {
            acc += vals[k];
            ++k;
        } while (k < nnz && offsets[k + 1] == offsets[k]);
        out_rows[o] = rows[i];
        out_cols[o] = cols[i];
        out_vals[o] = acc;
    }
}

int read_count(hipStream_t stream, const int32_t* dev_count, int64_t* host_nnz)
{
    int32_t h = 0;
    int err = static_cast<int>(
        hipMemcpyAsync(&h, dev_count, sizeof(h), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    *host_nnz = h;
    return err;
}

template <typename FlagKernel, typename MoveKernel, typename... FlagArgs>
int compact(hipStream_t stream, int64_t nnz, const int32_t* rows, const int32_t* cols,
            const double* vals, int32_t* out_rows, int32_t* out_cols, double* out_vals,
            void* workspace, size_t workspace_bytes, int64_t* host_nnz, FlagKernel flag_kernel,
            MoveKernel move_kernel, FlagArgs... flag_args)
{
    if (nnz < 0 || host_nnz == nullptr) return GKOMI_EINVAL;
    *host_nnz = 0;
    if (nnz == 0) return GKOMI_SUCCESS;
    if (nnz >= INT32_MAX) return GKOMI_ENOTSUPPORTED;
    const layout l = make_layout(nnz);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    int32_t* flags = reinterpret_cast<int32_t*>(ws + l.flags);
    const dim3 grid(static_cast<unsigned>(ceildiv(nnz + 1, block)));
    hipLaunchKernelGGL(flag_kernel, grid, dim3(block), 0, stream, nnz, flag_args..., flags);
    int err = gkomi_prefix_sum_i32(stream, flags, nnz + 1, ws + l.scan_ws,
                                   l.sort_tmp - l.scan_ws);
    if (err) return err;
    hipLaunchKernelGGL(move_kernel, grid, dim3(block), 0, stream, nnz, flags, rows, cols, vals,
                       out_rows, out_cols, out_vals);
    err = check_launch();
    if (err) return err;
    return read_count(stream, flags + nnz, host_nnz);
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_matrix_data_workspace_bytes(int64_t nnz)
{
    return make_layout(nnz < 0 ? 0 : nnz).total;
}

extern "C" int gkomi_matrix_data_sort_row_major_f64_i32(gkomi_stream_t s, int64_t nnz,
                                                        int32_t* row_idxs, int32_t* col_idxs,
                                                        double* values, void* workspace,
                                                        size_t workspace_bytes)
{
    if (nnz < 0) return GKOMI_EINVAL;
    if (nnz == 0) return GKOMI_SUCCESS;
    if (nnz >= INT32_MAX) return GKOMI_ENOTSUPPORTED;
    const layout l = make_layout(nnz);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    uint64_t* keys_in = reinterpret_cast<uint64_t*>(ws + l.keys_in);
    uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + l.keys_out);
    uint32_t* idx_in = reinterpret_cast<uint32_t*>(ws + l.idx_in);
    uint32_t* idx_out = reinterpret_cast<uint32_t*>(ws + l.idx_out);
    double* vals_copy = reinterpret_cast<double*>(ws + l.vals);
    const dim3 grid(static_cast<unsigned>(ceildiv(nnz, block)));
    hipLaunchKernelGGL(make_keys_kernel, grid, dim3(block), 0, stream, nnz, row_idxs, col_idxs,
                       keys_in, idx_in);
    int err = static_cast<int>(hipMemcpyAsync(vals_copy, values, 8 * static_cast<size_t>(nnz),
                                              hipMemcpyDeviceToDevice, stream));
    if (err) return err;
    // stable LSD radix sort of the 64-bit (row, column) keys carrying the entry's position (sort_scan.hip)
    err = radix_sort_u64(stream, nnz, keys_in, keys_out, idx_in, idx_out, 64, ws + l.sort_tmp, l.total - l.sort_tmp);
    if (err) return err;
    hipLaunchKernelGGL(apply_sort_kernel, grid, dim3(block), 0, stream, nnz, keys_out, idx_out,
                       vals_copy, row_idxs, col_idxs, values);
    return check_launch();
}

extern "C" int gkomi_matrix_data_remove_zeros_f64_i32(
    gkomi_stream_t s, int64_t nnz, const int32_t* row_idxs, const int32_t* col_idxs,
    const double* values, int32_t* out_row_idxs, int32_t* out_col_idxs, double* out_values,
    void* workspace, size_t workspace_bytes, int64_t* host_nnz)
{
    return compact(to_stream(s), nnz, row_idxs, col_idxs, values, out_row_idxs, out_col_idxs,
                   out_values, workspace, workspace_bytes, host_nnz, nonzero_flags_kernel,
                   compact_kernel, values);
}

extern "C" int gkomi_matrix_data_sum_duplicates_f64_i32(
    gkomi_stream_t s, int64_t nnz, const int32_t* row_idxs, const int32_t* col_idxs,
    const double* values, int32_t* out_row_idxs, int32_t* out_col_idxs, double* out_values,
    void* workspace, size_t workspace_bytes, int64_t* host_nnz)
{
    return compact(to_stream(s), nnz, row_idxs, col_idxs, values, out_row_idxs, out_col_idxs,
                   out_values, workspace, workspace_bytes, host_nnz, head_flags_kernel,
                   sum_runs_kernel, row_idxs, col_idxs);
}
