// Dense BLAS-1 solver steps for gfx950.  Replaces the dense:: entries of
// core/matrix/dense_kernels.hpp that the Krylov solvers use (scale,
// inv_scale, add_scaled, sub_scaled, fill, copy, compute_dot,
// compute_conj_dot, compute_norm2, compute_norm1, compute_squared_norm2,
// compute_sqrt, row_gather); semantics =
// reference/matrix/dense_kernels.cpp:127-447.
//
// All of these are pure HBM streaming (16-24 B per element).  Vectors
// (ncols == 1, stride == 1) and contiguous blocks (stride == ncols) take a
// flat 16-B-per-lane grid-stride path; strided matrices take a 2-D path with
// one grid row per column.  Reductions are two-stage (per-block partials,
// then one block adds the partials in index order): no float atomics, so
// results are bitwise reproducible from run to run.
#include "common.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr int red_max_blocks = 2048;

enum class ew_op { scale, inv_scale, add_scaled, sub_scaled };

template <ew_op Op>
__device__ __forceinline__ double ew_apply(double a, double x, double y)
{
    switch (Op) {
    case ew_op::scale: return y * a;
    case ew_op::inv_scale: return y / a;
    case ew_op::add_scaled: return y + a * x;
    case ew_op::sub_scaled: return y - a * x;
    }
    return y;
}

template <ew_op Op>
constexpr bool needs_x = (Op == ew_op::add_scaled || Op == ew_op::sub_scaled);

// flat path: n contiguous values, one scalar alpha
template <ew_op Op>
__global__ __launch_bounds__(block) void ew_flat_kernel(
    int64_t n, const double* __restrict__ alpha_p, const double* __restrict__ x,
    double* __restrict__ y)
{
    const double a = alpha_p[0];
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    double2* y2 = reinterpret_cast<double2*>(y);
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < n2; i += step) {
        double2 yv = y2[i];
        double2 xv = make_double2(0.0, 0.0);
        if (needs_x<Op>) xv = x2[i];
        yv.x = ew_apply<Op>(a, xv.x, yv.x);
        yv.y = ew_apply<Op>(a, xv.y, yv.y);
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        y[n - 1] = ew_apply<Op>(a, needs_x<Op> ? x[n - 1] : 0.0, y[n - 1]);
    }
}

// general path: nrows x ncols with strides, alpha per column or shared
template <ew_op Op>
__global__ __launch_bounds__(block) void ew_2d_kernel(
    int64_t nrows, int64_t ncols, const double* __restrict__ alpha_p,
    int64_t alpha_ncols, const double* __restrict__ x, int64_t x_stride,
    double* __restrict__ y, int64_t y_stride)
{
    const int64_t total = nrows * ncols;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        const int64_t row = i / ncols;
        const int64_t col = i % ncols;
        const double a = alpha_p[alpha_ncols == 1 ? 0 : col];
        const double xv = needs_x<Op> ? x[row * x_stride + col] : 0.0;
        y[row * y_stride + col] = ew_apply<Op>(a, xv, y[row * y_stride + col]);
    }
}

template <ew_op Op>
int launch_ew(hipStream_t s, int64_t nrows, int64_t ncols, const double* alpha,
              int64_t alpha_ncols, const double* x, int64_t x_stride,
              double* y, int64_t y_stride)
{
    if (nrows < 0 || ncols < 0) return GKOMI_EINVAL;
    if (alpha_ncols != 1 && alpha_ncols != ncols) return GKOMI_EINVAL;
    if (nrows == 0 || ncols == 0) return GKOMI_SUCCESS;
    if (y_stride < ncols || (needs_x<Op> && x_stride < ncols)) return GKOMI_EINVAL;
    const bool contiguous = y_stride == ncols && (!needs_x<Op> || x_stride == ncols);
    const bool aligned = reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                         (!needs_x<Op> || reinterpret_cast<uintptr_t>(x) % 16 == 0);
    if (contiguous && aligned && (alpha_ncols == 1)) {
        const int64_t n = nrows * ncols;
        hipLaunchKernelGGL(ew_flat_kernel<Op>, dim3(grid_for(n / 2 + 1, block)),
                           dim3(block), 0, s, n, alpha, x, y);
    } else {
        hipLaunchKernelGGL(ew_2d_kernel<Op>,
                           dim3(grid_for(nrows * ncols, block)), dim3(block),
                           0, s, nrows, ncols, alpha, alpha_ncols, x, x_stride,
                           y, y_stride);
    }
    return check_launch();
}

__global__ __launch_bounds__(block) void fill_kernel(int64_t nrows,
                                                    int64_t ncols,
                                                    double* __restrict__ x,
                                                    int64_t stride, double v)
{
    const int64_t total = nrows * ncols;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        x[(i / ncols) * stride + (i % ncols)] = v;
    }
}

__global__ __launch_bounds__(block) void fill_flat_kernel(int64_t n,
                                                         double* __restrict__ x,
                                                         double v)
{
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    double2* x2 = reinterpret_cast<double2*>(x);
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < n2; i += step) {
        x2[i] = make_double2(v, v);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = v;
}

__global__ __launch_bounds__(block) void copy_kernel(
    int64_t nrows, int64_t ncols, const double* __restrict__ in,
    int64_t in_stride, double* __restrict__ out, int64_t out_stride)
{
    const int64_t total = nrows * ncols;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        const int64_t row = i / ncols, col = i % ncols;
        out[row * out_stride + col] = in[row * in_stride + col];
    }
}

__global__ __launch_bounds__(block) void copy_flat_kernel(
    int64_t n, const double* __restrict__ in, double* __restrict__ out)
{
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    const double2* in2 = reinterpret_cast<const double2*>(in);
    double2* out2 = reinterpret_cast<double2*>(out);
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < n2; i += step) {
        out2[i] = in2[i];
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = in[n - 1];
}

__global__ __launch_bounds__(block) void sqrt_kernel(int64_t nrows,
                                                    int64_t ncols,
                                                    double* __restrict__ x,
                                                    int64_t stride)
{
    const int64_t total = nrows * ncols;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        double* p = x + (i / ncols) * stride + (i % ncols);
        *p = sqrt(*p);
    }
}

__global__ __launch_bounds__(block) void row_gather_kernel(
    int64_t nout, int64_t ncols, const int32_t* __restrict__ rows,
    const double* __restrict__ in, int64_t in_stride, double* __restrict__ out,
    int64_t out_stride)
{
    const int64_t total = nout * ncols;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        const int64_t row = i / ncols, col = i % ncols;
        out[row * out_stride + col] =
            in[static_cast<int64_t>(rows[row]) * in_stride + col];
    }
}

// ---- reductions ---------------------------------------------------------

enum class red_op { dot, sqnorm, norm1 };

template <red_op Op>
__device__ __forceinline__ double red_term(double x, double y)
{
    switch (Op) {
    case red_op::dot: return x * y;
    case red_op::sqnorm: return x * x;
    case red_op::norm1: return fabs(x);
    }
    return 0.0;
}

// stage 1, vector fast path: partial[blockIdx.x] = sum over this block's share
template <red_op Op>
__global__ __launch_bounds__(block) void reduce_flat_kernel(
    int64_t n, const double* __restrict__ x, const double* __restrict__ y,
    double* __restrict__ partial)
{
    __shared__ double smem[block / wave_size];
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < n2; i += step) {
        const double2 xv = x2[i];
        double2 yv = xv;
        if (Op == red_op::dot) yv = y2[i];
        acc0 += red_term<Op>(xv.x, yv.x);
        acc1 += red_term<Op>(xv.y, yv.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        acc0 += red_term<Op>(x[n - 1], Op == red_op::dot ? y[n - 1] : x[n - 1]);
    }
    const double total = block_reduce_sum<block>(acc0 + acc1, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// stage 1, general path: grid.y = column
template <red_op Op>
__global__ __launch_bounds__(block) void reduce_2d_kernel(
    int64_t nrows, const double* __restrict__ x, int64_t x_stride,
    const double* __restrict__ y, int64_t y_stride,
    double* __restrict__ partial)
{
    __shared__ double smem[block / wave_size];
    const int64_t col = blockIdx.y;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < nrows; i += step) {
        const double xv = x[i * x_stride + col];
        const double yv = Op == red_op::dot ? y[i * y_stride + col] : xv;
        acc += red_term<Op>(xv, yv);
    }
    const double total = block_reduce_sum<block>(acc, smem);
    if (threadIdx.x == 0) partial[col * gridDim.x + blockIdx.x] = total;
}

// stage 2: one block per column adds the partials in index order
template <bool Sqrt>
__global__ __launch_bounds__(block) void reduce_final_kernel(
    int nparts, const double* __restrict__ partial, double* __restrict__ result)
{
    __shared__ double smem[block / wave_size];
    const double* p = partial + static_cast<int64_t>(blockIdx.x) * nparts;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += block) acc += p[i];
    const double total = block_reduce_sum<block>(acc, smem);
    if (threadIdx.x == 0) result[blockIdx.x] = Sqrt ? sqrt(total) : total;
}

int reduction_blocks(int64_t nrows)
{
    // two 16-B loads per thread before it pays to add a workgroup (short
    // vectors are latency-bound: more workgroups = more loads in flight)
    int64_t g = ceildiv(nrows, static_cast<int64_t>(block) * 2 * 2);
    if (g < 1) g = 1;
    if (g > red_max_blocks) g = red_max_blocks;
    return static_cast<int>(g);
}

template <red_op Op, bool Sqrt>
int launch_reduce(hipStream_t s, int64_t nrows, int64_t ncols, const double* x,
                  int64_t x_stride, const double* y, int64_t y_stride,
                  double* result, void* workspace, size_t workspace_bytes)
{
    if (nrows < 0 || ncols < 0) return GKOMI_EINVAL;
    if (ncols == 0) return GKOMI_SUCCESS;
    if (ncols > 65535) return GKOMI_ENOTSUPPORTED;
    if (nrows == 0) {
        return static_cast<int>(
            hipMemsetAsync(result, 0, sizeof(double) * ncols, s));
    }
    if (workspace_bytes < gkomi_dense_reduction_workspace_bytes(nrows, ncols) ||
        workspace == nullptr) {
        return GKOMI_EWORKSPACE;
    }
    double* partial = static_cast<double*>(workspace);
    const int g = reduction_blocks(nrows);
    const bool flat = ncols == 1 && x_stride == 1 &&
                      (Op != red_op::dot || y_stride == 1) &&
                      reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                      (Op != red_op::dot ||
                       reinterpret_cast<uintptr_t>(y) % 16 == 0);
    if (flat) {
        hipLaunchKernelGGL(reduce_flat_kernel<Op>, dim3(g), dim3(block), 0, s,
                           nrows, x, y, partial);
    } else {
        hipLaunchKernelGGL(reduce_2d_kernel<Op>, dim3(g, ncols), dim3(block),
                           0, s, nrows, x, x_stride, y, y_stride, partial);
    }
    int err = check_launch();
    if (err) return err;
    hipLaunchKernelGGL(reduce_final_kernel<Sqrt>, dim3(ncols), dim3(block), 0,
                       s, g, partial, result);
    return check_launch();
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_dense_reduction_workspace_bytes(int64_t nrows,
                                                        int64_t ncols)
{
    if (nrows <= 0 || ncols <= 0) return 0;
    return sizeof(double) * static_cast<size_t>(reduction_blocks(nrows)) *
           static_cast<size_t>(ncols);
}

extern "C" int gkomi_dense_fill_f64(gkomi_stream_t s, int64_t nrows,
                                    int64_t ncols, double* x, int64_t stride,
                                    double value)
{
    if (nrows < 0 || ncols < 0) return GKOMI_EINVAL;
    if (nrows == 0 || ncols == 0) return GKOMI_SUCCESS;
    if (stride < ncols) return GKOMI_EINVAL;
    if (stride == ncols && reinterpret_cast<uintptr_t>(x) % 16 == 0) {
        const int64_t n = nrows * ncols;
        hipLaunchKernelGGL(fill_flat_kernel, dim3(grid_for(n / 2 + 1, block)),
                           dim3(block), 0, to_stream(s), n, x, value);
    } else {
        hipLaunchKernelGGL(fill_kernel, dim3(grid_for(nrows * ncols, block)),
                           dim3(block), 0, to_stream(s), nrows, ncols, x,
                           stride, value);
    }
    return check_launch();
}

extern "C" int gkomi_dense_copy_f64(gkomi_stream_t s, int64_t nrows,
                                    int64_t ncols, const double* in,
                                    int64_t in_stride, double* out,
                                    int64_t out_stride)
{
    if (nrows < 0 || ncols < 0) return GKOMI_EINVAL;
    if (nrows == 0 || ncols == 0) return GKOMI_SUCCESS;
    if (in_stride < ncols || out_stride < ncols) return GKOMI_EINVAL;
    if (in_stride == ncols && out_stride == ncols &&
        reinterpret_cast<uintptr_t>(in) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(out) % 16 == 0) {
        const int64_t n = nrows * ncols;
        hipLaunchKernelGGL(copy_flat_kernel, dim3(grid_for(n / 2 + 1, block)),
                           dim3(block), 0, to_stream(s), n, in, out);
    } else {
        hipLaunchKernelGGL(copy_kernel, dim3(grid_for(nrows * ncols, block)),
                           dim3(block), 0, to_stream(s), nrows, ncols, in,
                           in_stride, out, out_stride);
    }
    return check_launch();
}

extern "C" int gkomi_dense_scale_f64(gkomi_stream_t s, int64_t nrows,
                                     int64_t ncols, const double* alpha,
                                     int64_t alpha_ncols, double* x,
                                     int64_t stride)
{
    return launch_ew<ew_op::scale>(to_stream(s), nrows, ncols, alpha,
                                   alpha_ncols, nullptr, 0, x, stride);
}

extern "C" int gkomi_dense_inv_scale_f64(gkomi_stream_t s, int64_t nrows,
                                         int64_t ncols, const double* alpha,
                                         int64_t alpha_ncols, double* x,
                                         int64_t stride)
{
    return launch_ew<ew_op::inv_scale>(to_stream(s), nrows, ncols, alpha,
                                       alpha_ncols, nullptr, 0, x, stride);
}

extern "C" int gkomi_dense_add_scaled_f64(gkomi_stream_t s, int64_t nrows,
                                          int64_t ncols, const double* alpha,
                                          int64_t alpha_ncols, const double* x,
                                          int64_t x_stride, double* y,
                                          int64_t y_stride)
{
    return launch_ew<ew_op::add_scaled>(to_stream(s), nrows, ncols, alpha,
                                        alpha_ncols, x, x_stride, y, y_stride);
}

extern "C" int gkomi_dense_sub_scaled_f64(gkomi_stream_t s, int64_t nrows,
                                          int64_t ncols, const double* alpha,
                                          int64_t alpha_ncols, const double* x,
                                          int64_t x_stride, double* y,
                                          int64_t y_stride)
{
    return launch_ew<ew_op::sub_scaled>(to_stream(s), nrows, ncols, alpha,
                                        alpha_ncols, x, x_stride, y, y_stride);
}

extern "C" int gkomi_dense_compute_dot_f64(gkomi_stream_t s, int64_t nrows,
                                           int64_t ncols, const double* x,
                                           int64_t x_stride, const double* y,
                                           int64_t y_stride, double* result,
                                           void* workspace,
                                           size_t workspace_bytes)
{
    return launch_reduce<red_op::dot, false>(to_stream(s), nrows, ncols, x,
                                             x_stride, y, y_stride, result,
                                             workspace, workspace_bytes);
}

extern "C" int gkomi_dense_compute_norm2_f64(gkomi_stream_t s, int64_t nrows,
                                             int64_t ncols, const double* x,
                                             int64_t x_stride, double* result,
                                             void* workspace,
                                             size_t workspace_bytes)
{
    return launch_reduce<red_op::sqnorm, true>(to_stream(s), nrows, ncols, x,
                                               x_stride, x, x_stride, result,
                                               workspace, workspace_bytes);
}

extern "C" int gkomi_dense_compute_squared_norm2_f64(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, const double* x,
    int64_t x_stride, double* result, void* workspace, size_t workspace_bytes)
{
    return launch_reduce<red_op::sqnorm, false>(to_stream(s), nrows, ncols, x,
                                                x_stride, x, x_stride, result,
                                                workspace, workspace_bytes);
}

extern "C" int gkomi_dense_compute_norm1_f64(gkomi_stream_t s, int64_t nrows,
                                             int64_t ncols, const double* x,
                                             int64_t x_stride, double* result,
                                             void* workspace,
                                             size_t workspace_bytes)
{
    return launch_reduce<red_op::norm1, false>(to_stream(s), nrows, ncols, x,
                                               x_stride, x, x_stride, result,
                                               workspace, workspace_bytes);
}

extern "C" int gkomi_dense_compute_sqrt_f64(gkomi_stream_t s, int64_t nrows,
                                            int64_t ncols, double* x,
                                            int64_t stride)
{
    if (nrows < 0 || ncols < 0) return GKOMI_EINVAL;
    if (nrows == 0 || ncols == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(sqrt_kernel, dim3(grid_for(nrows * ncols, block)),
                       dim3(block), 0, to_stream(s), nrows, ncols, x, stride);
    return check_launch();
}

extern "C" int gkomi_dense_row_gather_f64_i32(gkomi_stream_t s, int64_t nout,
                                              int64_t ncols,
                                              const int32_t* rows,
                                              const double* in,
                                              int64_t in_stride, double* out,
                                              int64_t out_stride)
{
    if (nout < 0 || ncols < 0) return GKOMI_EINVAL;
    if (nout == 0 || ncols == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(row_gather_kernel, dim3(grid_for(nout * ncols, block)),
                       dim3(block), 0, to_stream(s), nout, ncols, rows, in,
                       in_stride, out, out_stride);
    return check_launch();
}
