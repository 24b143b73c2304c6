// The <float, int32> instantiation of the core of the path (GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE,
// include/ginkgo/core/base/types.hpp:544-560): csr::spmv / advanced_spmv (core/matrix/csr_kernels.hpp:58-75), the dense
// BLAS-1 kernels (core/matrix/dense_kernels.hpp), the CG kernels (core/solver/cg_kernels.hpp:54-80), stop::residual_norm
// (core/stop/residual_norm_kernels.hpp) and a Cg driver on them (core/solver/cg.cpp:107-193).
//
// Kernels of their own, not the double ones re-typed: 8 B per nonzero instead of 12, 4-B vector elements.  The SpMV is the
// row-cut LDS design of csr_spmv.hip's stream kernel: a workgroup owns 256 consecutive rows, streams their nonzeros in tiles
// of 2048 (coalesced 4-B loads, the gathers of b behind them), leaves the products in LDS, and the thread that owns a row
// adds its products left to right -- the reference's order, so csr::spmv is bit-identical to the reference executor's
// float instantiation for any row lengths (tests/test_f32_gpu.py).  Every intermediate is a float; -ffp-contract=off.
// Elementwise kernels are bit-exact, reductions are two-stage and reproducible (tolerance parity), as in dense.hip.
// Not tuned beyond that: the measured path of this library is double (BASELINE.json: fp64).
#include "common.hpp"

#include <algorithm>
#include <utility>

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr int spmv_tile = 2048;

template <bool Advanced>
__global__ __launch_bounds__(block) void f32_csr_spmv_kernel(int nrows, const int32_t* __restrict__ row_ptrs,
                                                            const int32_t* __restrict__ col_idxs, const float* __restrict__ vals,
                                                            const float* __restrict__ b, int64_t b_stride, float* __restrict__ c,
                                                            int64_t c_stride, const float* __restrict__ alpha_p,
                                                            const float* __restrict__ beta_p)
{
    __shared__ float prod[spmv_tile];
    b += blockIdx.y;
    c += blockIdx.y;
    const int r0 = blockIdx.x * block;
    const int r1 = min(r0 + block, nrows);
    const int row = r0 + threadIdx.x;
    const bool mine = row < r1;
    const int ra = mine ? row_ptrs[row] : 0;
    const int rb = mine ? row_ptrs[row + 1] : 0;
    const int p0 = row_ptrs[r0], p1 = row_ptrs[r1];
    const float alpha = Advanced ? alpha_p[0] : 1.0f;
    float sum = 0.0f;
    if (Advanced && mine) sum = c[row * c_stride] * beta_p[0];
    for (int base = p0; base < p1; base += spmv_tile) {
        const int count = min(spmv_tile, p1 - base);
        for (int i = threadIdx.x; i < count; i += block) {
            const float v = vals[base + i];
            const float x = b[col_idxs[base + i] * b_stride];
            prod[i] = Advanced ? (alpha * v) * x : v * x;
        }
        __syncthreads();
        const int lo = max(ra, base) - base, hi = min(rb, base + count) - base;
        for (int k = lo; k < hi; ++k) sum = sum + prod[k];
        __syncthreads();
    }
    if (mine) c[row * c_stride] = sum;
}

enum class ew { fill, copy, scale, inv_scale, add_scaled, sub_scaled };

// y(row, col) = op(alpha(col), x(row, col), y(row, col)); one thread per entry, entries of a row next to each other
template <ew Op>
__global__ __launch_bounds__(block) void f32_ew_kernel(int64_t nrows, int64_t ncols, const float* __restrict__ alpha_p,
                                                      int64_t alpha_ncols, float value, const float* __restrict__ x, int64_t x_stride,
                                                      float* __restrict__ y, int64_t y_stride)
{
    const int64_t total = nrows * ncols;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t row = i / ncols, col = i - row * ncols;
        float* dst = y + row * y_stride + col;
        const float a = (Op == ew::fill || Op == ew::copy) ? 0.0f : alpha_p[alpha_ncols == 1 ? 0 : col];
        if (Op == ew::fill) {
            *dst = value;
        } else if (Op == ew::copy) {
            *dst = x[row * x_stride + col];
        } else if (Op == ew::scale) {
            *dst = *dst * a;
        } else if (Op == ew::inv_scale) {
            *dst = *dst / a;
        } else if (Op == ew::add_scaled) {
            *dst = *dst + a * x[row * x_stride + col];
        } else {
            *dst = *dst - a * x[row * x_stride + col];
        }
    }
}

constexpr int red_rows = 4096;  // rows per partial sum

// partial[col * nparts + part] = sum over the part's rows of x * y (y == x: squares), added in row order by ... one
// workgroup per (part, column): lane-strided partial sums, then a fixed tree
__global__ __launch_bounds__(block) void f32_dot_partials_kernel(int64_t nrows, const float* __restrict__ x, int64_t x_stride,
                                                                const float* __restrict__ y, int64_t y_stride, int nparts,
                                                                float* __restrict__ partial)
{
    __shared__ float red[block];
    const int64_t col = blockIdx.y;
    const int64_t lo = static_cast<int64_t>(blockIdx.x) * red_rows, hi = min(lo + red_rows, nrows);
    float acc = 0.0f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += block) acc = acc + x[i * x_stride + col] * y[i * y_stride + col];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = block / 2; d > 0; d /= 2) {
        if (threadIdx.x < d) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[col * nparts + blockIdx.x] = red[0];
}

// result[col] = [sqrt of] the partials of the column, in index order
__global__ void f32_sum_partials_kernel(int64_t ncols, int nparts, const float* __restrict__ partial, float* __restrict__ result,
                                        int take_sqrt)
{
    const int64_t col = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (col >= ncols) return;
    float total = 0.0f;
    for (int k = 0; k < nparts; ++k) total = total + partial[col * nparts + k];
    result[col] = take_sqrt ? sqrtf(total) : total;
}

__device__ __forceinline__ bool has_stopped(uint8_t s) { return (s & GKOMI_STATUS_ID_MASK) != 0; }

__global__ __launch_bounds__(block) void f32_cg_initialize_kernel(int64_t nrows, int64_t nrhs, const float* __restrict__ b,
                                                                 int64_t b_stride, float* __restrict__ r, int64_t r_stride,
                                                                 float* __restrict__ z, int64_t z_stride, float* __restrict__ p,
                                                                 int64_t p_stride, float* __restrict__ q, int64_t q_stride,
                                                                 float* __restrict__ prev_rho, float* __restrict__ rho,
                                                                 uint8_t* __restrict__ stop_status)
{
    const int64_t g = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (g < nrhs) {
        rho[g] = 0.0f;
        prev_rho[g] = 1.0f;
        stop_status[g] = 0;
    }
    const int64_t total = nrows * nrhs;
    for (int64_t i = g; i < total; i += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t row = i / nrhs, col = i - row * nrhs;
        r[row * r_stride + col] = b[row * b_stride + col];
        z[row * z_stride + col] = 0.0f;
        p[row * p_stride + col] = 0.0f;
        q[row * q_stride + col] = 0.0f;
    }
}

__global__ __launch_bounds__(block) void f32_cg_step_1_kernel(int64_t nrows, int64_t nrhs, float* __restrict__ p, int64_t p_stride,
                                                             const float* __restrict__ z, int64_t z_stride,
                                                             const float* __restrict__ rho, const float* __restrict__ prev_rho,
                                                             const uint8_t* __restrict__ stop_status)
{
    const int64_t total = nrows * nrhs;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t row = i / nrhs, col = i - row * nrhs;
        if (has_stopped(stop_status[col])) continue;
        const float zv = z[row * z_stride + col];
        if (prev_rho[col] == 0.0f) {
            p[row * p_stride + col] = zv;
        } else {
            const float tmp = rho[col] / prev_rho[col];
            p[row * p_stride + col] = zv + tmp * p[row * p_stride + col];
        }
    }
}

__global__ __launch_bounds__(block) void f32_cg_step_2_kernel(int64_t nrows, int64_t nrhs, float* __restrict__ x, int64_t x_stride,
                                                             float* __restrict__ r, int64_t r_stride, const float* __restrict__ p,
                                                             int64_t p_stride, const float* __restrict__ q, int64_t q_stride,
                                                             const float* __restrict__ beta, const float* __restrict__ rho,
                                                             const uint8_t* __restrict__ stop_status)
{
    const int64_t total = nrows * nrhs;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t row = i / nrhs, col = i - row * nrhs;
        if (has_stopped(stop_status[col]) || beta[col] == 0.0f) continue;
        const float tmp = rho[col] / beta[col];
        x[row * x_stride + col] = x[row * x_stride + col] + tmp * p[row * p_stride + col];
        r[row * r_stride + col] = r[row * r_stride + col] - tmp * q[row * q_stride + col];
    }
}

// one thread: the loops of reference/stop/residual_norm_kernels.cpp:57-83; flags = {all_converged, one_changed}
__global__ void f32_residual_norm_kernel(int64_t nrhs, const float* __restrict__ tau, const float* __restrict__ orig_tau, float goal,
                                         uint8_t id, int set_finalized, uint8_t* __restrict__ stop_status, uint8_t* __restrict__ flags)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint8_t all = 1, changed = 0;
    for (int64_t i = 0; i < nrhs; ++i) {
        if (tau[i] < goal * orig_tau[i]) {
            if (!has_stopped(stop_status[i])) {
                stop_status[i] = static_cast<uint8_t>(GKOMI_STATUS_CONVERGED | (set_finalized ? GKOMI_STATUS_FINALIZED : 0) | id);
            }
            changed = 1;
        }
    }
    for (int64_t i = 0; i < nrhs; ++i) {
        if (!has_stopped(stop_status[i])) {
            all = 0;
            break;
        }
    }
    flags[0] = all;
    flags[1] = changed;
}

template <ew Op>
int launch_ew(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* alpha, int64_t alpha_ncols, float value, const float* x,
              int64_t x_stride, float* y, int64_t y_stride)
{
    if (nrows < 0 || ncols < 0 || y_stride < ncols) return GKOMI_EINVAL;
    if (nrows == 0 || ncols == 0) return GKOMI_SUCCESS;
    if ((Op != ew::fill && Op != ew::copy) && (alpha == nullptr || (alpha_ncols != 1 && alpha_ncols != ncols))) return GKOMI_EINVAL;
    hipLaunchKernelGGL(f32_ew_kernel<Op>, dim3(grid_for(nrows * ncols, block)), dim3(block), 0, to_stream(s), nrows, ncols, alpha,
                       alpha_ncols, value, x, x_stride, y, y_stride);
    return check_launch();
}

int reduce(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* x, int64_t x_stride, const float* y, int64_t y_stride,
           float* result, void* workspace, size_t workspace_bytes, int take_sqrt)
{
    if (nrows < 0 || ncols < 0 || ncols > 65535) return GKOMI_EINVAL;
    if (ncols == 0) return GKOMI_SUCCESS;
    const int nparts = static_cast<int>(std::max<int64_t>(1, ceildiv(nrows, red_rows)));
    if (workspace == nullptr || workspace_bytes < sizeof(float) * static_cast<size_t>(nparts) * ncols) return GKOMI_EWORKSPACE;
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(f32_dot_partials_kernel, dim3(nparts, static_cast<unsigned>(ncols)), dim3(block), 0, to_stream(s), nrows, x,
                       x_stride, y, y_stride, nparts, partial);
    hipLaunchKernelGGL(f32_sum_partials_kernel, dim3(static_cast<unsigned>(ceildiv(ncols, 64))), dim3(64), 0, to_stream(s), ncols,
                       nparts, partial, result, take_sqrt);
    return check_launch();
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int gkomi_csr_spmv_f32_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
                                      const int32_t* col_idxs, const float* vals, const float* b, int64_t b_stride, float* c,
                                      int64_t c_stride, const float* alpha, const float* beta)
{
    if (nrows < 0 || ncols < 0 || nrhs < 0 || nnz < 0) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrows > INT32_MAX - 1024 || nrhs > 65535 || nnz > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    const dim3 grid(static_cast<unsigned>(ceildiv(nrows, block)), static_cast<unsigned>(nrhs));
    if (alpha != nullptr) {
        hipLaunchKernelGGL(f32_csr_spmv_kernel<true>, grid, dim3(block), 0, to_stream(s), static_cast<int>(nrows), row_ptrs, col_idxs,
                           vals, b, b_stride, c, c_stride, alpha, beta);
    } else {
        hipLaunchKernelGGL(f32_csr_spmv_kernel<false>, grid, dim3(block), 0, to_stream(s), static_cast<int>(nrows), row_ptrs, col_idxs,
                           vals, b, b_stride, c, c_stride, alpha, beta);
    }
    return check_launch();
}

extern "C" int gkomi_dense_fill_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, float* x, int64_t stride, float value)
{
    return launch_ew<ew::fill>(s, nrows, ncols, nullptr, 1, value, nullptr, 0, x, stride);
}

extern "C" int gkomi_dense_copy_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* in, int64_t in_stride, float* out,
                                    int64_t out_stride)
{
    if (in_stride < ncols) return GKOMI_EINVAL;
    return launch_ew<ew::copy>(s, nrows, ncols, nullptr, 1, 0.0f, in, in_stride, out, out_stride);
}

extern "C" int gkomi_dense_scale_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* alpha, int64_t alpha_ncols, float* x,
                                     int64_t stride)
{
    return launch_ew<ew::scale>(s, nrows, ncols, alpha, alpha_ncols, 0.0f, nullptr, 0, x, stride);
}

extern "C" int gkomi_dense_inv_scale_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* alpha, int64_t alpha_ncols,
                                         float* x, int64_t stride)
{
    return launch_ew<ew::inv_scale>(s, nrows, ncols, alpha, alpha_ncols, 0.0f, nullptr, 0, x, stride);
}

extern "C" int gkomi_dense_add_scaled_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* alpha, int64_t alpha_ncols,
                                          const float* x, int64_t x_stride, float* y, int64_t y_stride)
{
    if (x_stride < ncols) return GKOMI_EINVAL;
    return launch_ew<ew::add_scaled>(s, nrows, ncols, alpha, alpha_ncols, 0.0f, x, x_stride, y, y_stride);
}

extern "C" int gkomi_dense_sub_scaled_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* alpha, int64_t alpha_ncols,
                                          const float* x, int64_t x_stride, float* y, int64_t y_stride)
{
    if (x_stride < ncols) return GKOMI_EINVAL;
    return launch_ew<ew::sub_scaled>(s, nrows, ncols, alpha, alpha_ncols, 0.0f, x, x_stride, y, y_stride);
}

extern "C" size_t gkomi_dense_reduction_workspace_bytes_f32(int64_t nrows, int64_t ncols)
{
    if (nrows < 0 || ncols < 0) return 0;
    return sizeof(float) * static_cast<size_t>(std::max<int64_t>(1, ceildiv(nrows, red_rows))) * static_cast<size_t>(std::max<int64_t>(ncols, 1)) + 16;
}

extern "C" int gkomi_dense_compute_dot_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* x, int64_t x_stride,
                                           const float* y, int64_t y_stride, float* result, void* workspace, size_t workspace_bytes)
{
    return reduce(s, nrows, ncols, x, x_stride, y, y_stride, result, workspace, workspace_bytes, 0);
}

extern "C" int gkomi_dense_compute_norm2_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* x, int64_t x_stride,
                                             float* result, void* workspace, size_t workspace_bytes)
{
    return reduce(s, nrows, ncols, x, x_stride, x, x_stride, result, workspace, workspace_bytes, 1);
}

extern "C" int gkomi_cg_initialize_f32(gkomi_stream_t s, int64_t nrows, int64_t nrhs, const float* b, int64_t b_stride, float* r,
                                       int64_t r_stride, float* z, int64_t z_stride, float* p, int64_t p_stride, float* q,
                                       int64_t q_stride, float* prev_rho, float* rho, uint8_t* stop_status)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(f32_cg_initialize_kernel, dim3(grid_for(std::max<int64_t>(nrows * nrhs, nrhs), block)), dim3(block), 0,
                       to_stream(s), nrows, nrhs, b, b_stride, r, r_stride, z, z_stride, p, p_stride, q, q_stride, prev_rho, rho,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_cg_step_1_f32(gkomi_stream_t s, int64_t nrows, int64_t nrhs, float* p, int64_t p_stride, const float* z,
                                   int64_t z_stride, const float* rho, const float* prev_rho, const uint8_t* stop_status)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(f32_cg_step_1_kernel, dim3(grid_for(nrows * nrhs, block)), dim3(block), 0, to_stream(s), nrows, nrhs, p, p_stride,
                       z, z_stride, rho, prev_rho, stop_status);
    return check_launch();
}

extern "C" int gkomi_cg_step_2_f32(gkomi_stream_t s, int64_t nrows, int64_t nrhs, float* x, int64_t x_stride, float* r, int64_t r_stride,
                                   const float* p, int64_t p_stride, const float* q, int64_t q_stride, const float* beta,
                                   const float* rho, const uint8_t* stop_status)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(f32_cg_step_2_kernel, dim3(grid_for(nrows * nrhs, block)), dim3(block), 0, to_stream(s), nrows, nrhs, x, x_stride,
                       r, r_stride, p, p_stride, q, q_stride, beta, rho, stop_status);
    return check_launch();
}

extern "C" int gkomi_residual_norm_f32(gkomi_stream_t s, int64_t nrhs, const float* tau, const float* orig_tau, float rel_residual_goal,
                                       uint8_t stopping_id, int set_finalized, uint8_t* stop_status, uint8_t* device_flags,
                                       uint8_t* host_flags)
{
    if (nrhs < 0 || device_flags == nullptr) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(s);
    hipLaunchKernelGGL(f32_residual_norm_kernel, dim3(1), dim3(64), 0, stream, nrhs, tau, orig_tau, rel_residual_goal, stopping_id,
                       set_finalized, stop_status, device_flags);
    int err = check_launch();
    if (err || host_flags == nullptr) return err;
    err = static_cast<int>(hipMemcpyAsync(host_flags, device_flags, 2, hipMemcpyDeviceToHost, stream));
    if (err) return err;
    return static_cast<int>(hipStreamSynchronize(stream));
}

// ---- Cg<float>::apply_dense_impl (core/solver/cg.cpp:107-193), Identity preconditioner, Combined(Iteration(max_iters),
// ResidualNorm(reduction, baseline)): the reference's kernel sequence on the kernels above, one right-hand side, the
// criterion looked at on the host every iteration like the reference does (two blocking 1-byte copies there, one 2-byte
// copy here).  workspace: gkomi_cg_workspace_bytes_f32(n).  host_info = {iterations, converged, ||r||, baseline norm}.
extern "C" size_t gkomi_cg_workspace_bytes_f32(int64_t n)
{
    if (n < 0) return 0;
    const size_t vec = (sizeof(float) * static_cast<size_t>(n > 0 ? n : 1) + 255) / 256 * 256;
    return 4 * vec + 256 + gkomi_dense_reduction_workspace_bytes_f32(n, 1) + 256;
}

extern "C" int gkomi_cg_solve_f32(gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
                                  const float* vals, const float* b, float* x, int64_t max_iters, float reduction, int baseline,
                                  void* workspace, size_t workspace_bytes, double* host_info)
{
    if (n < 0 || nnz < 0 || max_iters < 0 || baseline < 0 || baseline > 2) return GKOMI_EINVAL;
    if (workspace == nullptr || workspace_bytes < gkomi_cg_workspace_bytes_f32(n)) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    const size_t vec = (sizeof(float) * static_cast<size_t>(n > 0 ? n : 1) + 255) / 256 * 256;
    float* r = reinterpret_cast<float*>(ws);
    float* z = reinterpret_cast<float*>(ws + vec);
    float* p = reinterpret_cast<float*>(ws + 2 * vec);
    float* q = reinterpret_cast<float*>(ws + 3 * vec);
    float* small = reinterpret_cast<float*>(ws + 4 * vec);  // prev_rho, rho, beta, tau, orig_tau, one, minus one
    uint8_t* status = reinterpret_cast<uint8_t*>(small + 16);
    uint8_t* flags = status + 8;
    void* red = ws + 4 * vec + 256;
    const size_t red_bytes = gkomi_dense_reduction_workspace_bytes_f32(n, 1);
    float *prev_rho = small, *rho = small + 1, *beta = small + 2, *tau = small + 3, *orig_tau = small + 4, *one = small + 5,
          *neg = small + 6;
#define GKOMI_TRY(expr)        \
    do {                       \
        const int e_ = (expr); \
        if (e_) return e_;     \
    } while (0)
    GKOMI_TRY(gkomi_cg_initialize_f32(s, n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, prev_rho, rho, status));
    GKOMI_TRY(gkomi_dense_fill_f32(s, 1, 1, one, 1, 1.0f));
    GKOMI_TRY(gkomi_dense_fill_f32(s, 1, 1, neg, 1, -1.0f));
    GKOMI_TRY(gkomi_csr_spmv_f32_i32(s, n, n, 1, nnz, row_ptrs, col_idxs, vals, x, 1, r, 1, neg, one));  // r = b - A x
    if (baseline == 2) {
        GKOMI_TRY(gkomi_dense_fill_f32(s, 1, 1, orig_tau, 1, 1.0f));
    } else {
        GKOMI_TRY(gkomi_dense_compute_norm2_f32(s, n, 1, baseline == 0 ? b : r, 1, orig_tau, red, red_bytes));
    }
    long long iter = -1;
    int converged = 0;
    uint8_t host_flags[2] = {0, 0};
    while (true) {
        GKOMI_TRY(gkomi_dense_copy_f32(s, n, 1, r, 1, z, 1));  // Identity::apply
        GKOMI_TRY(gkomi_dense_compute_dot_f32(s, n, 1, r, 1, z, 1, rho, red, red_bytes));
        ++iter;
        GKOMI_TRY(gkomi_dense_compute_norm2_f32(s, n, 1, r, 1, tau, red, red_bytes));
        if (iter >= max_iters) break;  // Iteration criterion first (Combined)
        GKOMI_TRY(gkomi_residual_norm_f32(s, 1, tau, orig_tau, reduction, 1, 1, status, flags, host_flags));
        if (host_flags[0]) {
            converged = 1;
            break;
        }
        GKOMI_TRY(gkomi_cg_step_1_f32(s, n, 1, p, 1, z, 1, rho, prev_rho, status));
        GKOMI_TRY(gkomi_csr_spmv_f32_i32(s, n, n, 1, nnz, row_ptrs, col_idxs, vals, p, 1, q, 1, nullptr, nullptr));
        GKOMI_TRY(gkomi_dense_compute_dot_f32(s, n, 1, p, 1, q, 1, beta, red, red_bytes));
        GKOMI_TRY(gkomi_cg_step_2_f32(s, n, 1, x, 1, r, 1, p, 1, q, 1, beta, rho, status));
        std::swap(prev_rho, rho);
    }
    if (host_info != nullptr) {
        float h[2] = {0.0f, 0.0f};
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h[0], tau, sizeof(float), hipMemcpyDeviceToHost, stream)));
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h[1], orig_tau, sizeof(float), hipMemcpyDeviceToHost, stream)));
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
        host_info[0] = static_cast<double>(iter);
        host_info[1] = static_cast<double>(converged);
        host_info[2] = h[0];
        host_info[3] = h[1];
    }
#undef GKOMI_TRY
    return GKOMI_SUCCESS;
}
