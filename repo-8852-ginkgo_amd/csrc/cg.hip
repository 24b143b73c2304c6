// CG step kernels for gfx950.  Replaces gko::kernels::hip::cg::{initialize,
// step_1, step_2} (core/solver/cg_kernels.hpp:54-80); semantics =
// reference/solver/cg_kernels.cpp:53-123 (per-column gating by
// stopping_status, safe division by zero prev_rho / beta).
#include "common.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;

__global__ __launch_bounds__(block) void cg_initialize_kernel(
    int64_t nrows, int64_t nrhs, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ r, int64_t r_stride, double* __restrict__ z,
    int64_t z_stride, double* __restrict__ p, int64_t p_stride,
    double* __restrict__ q, int64_t q_stride, double* __restrict__ prev_rho,
    double* __restrict__ rho, uint8_t* __restrict__ stop_status)
{
    const int64_t gid = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    if (gid < nrhs) {
        rho[gid] = 0.0;
        prev_rho[gid] = 1.0;
        stop_status[gid] = 0;
    }
    const int64_t total = nrows * nrhs;
    for (int64_t i = gid; i < total; i += step) {
        const int64_t row = i / nrhs, col = i % nrhs;
        r[row * r_stride + col] = b[row * b_stride + col];
        z[row * z_stride + col] = 0.0;
        p[row * p_stride + col] = 0.0;
        q[row * q_stride + col] = 0.0;
    }
}

__global__ __launch_bounds__(block) void cg_step_1_kernel(
    int64_t nrows, int64_t nrhs, double* __restrict__ p, int64_t p_stride,
    const double* __restrict__ z, int64_t z_stride,
    const double* __restrict__ rho, const double* __restrict__ prev_rho,
    const uint8_t* __restrict__ stop_status)
{
    const int64_t total = nrows * nrhs;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        const int64_t row = i / nrhs, col = i % nrhs;
        if (status_has_stopped(stop_status[col])) continue;
        const double pr = prev_rho[col];
        const double zv = z[row * z_stride + col];
        double* pp = p + row * p_stride + col;
        if (pr == 0.0) {
            *pp = zv;
        } else {
            const double tmp = rho[col] / pr;
            *pp = zv + tmp * (*pp);
        }
    }
}

// single-rhs contiguous fast path, 16 B per lane
__global__ __launch_bounds__(block) void cg_step_1_vec_kernel(
    int64_t n, double* __restrict__ p, const double* __restrict__ z,
    const double* __restrict__ rho, const double* __restrict__ prev_rho,
    const uint8_t* __restrict__ stop_status)
{
    if (status_has_stopped(stop_status[0])) return;
    const double pr = prev_rho[0];
    const bool restart = pr == 0.0;
    const double tmp = restart ? 0.0 : rho[0] / pr;
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    double2* p2 = reinterpret_cast<double2*>(p);
    const double2* z2 = reinterpret_cast<const double2*>(z);
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < n2; i += step) {
        double2 zv = z2[i];
        if (!restart) {
            const double2 pv = p2[i];
            zv.x = zv.x + tmp * pv.x;
            zv.y = zv.y + tmp * pv.y;
        }
        p2[i] = zv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        p[n - 1] = restart ? z[n - 1] : z[n - 1] + tmp * p[n - 1];
    }
}

__global__ __launch_bounds__(block) void cg_step_2_kernel(
    int64_t nrows, int64_t nrhs, double* __restrict__ x, int64_t x_stride,
    double* __restrict__ r, int64_t r_stride, const double* __restrict__ p,
    int64_t p_stride, const double* __restrict__ q, int64_t q_stride,
    const double* __restrict__ beta, const double* __restrict__ rho,
    const uint8_t* __restrict__ stop_status)
{
    const int64_t total = nrows * nrhs;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < total; i += step) {
        const int64_t row = i / nrhs, col = i % nrhs;
        if (status_has_stopped(stop_status[col])) continue;
        const double bt = beta[col];
        if (bt != 0.0) {
            const double tmp = rho[col] / bt;
            x[row * x_stride + col] += tmp * p[row * p_stride + col];
            r[row * r_stride + col] -= tmp * q[row * q_stride + col];
        }
    }
}

__global__ __launch_bounds__(block) void cg_step_2_vec_kernel(
    int64_t n, double* __restrict__ x, double* __restrict__ r,
    const double* __restrict__ p, const double* __restrict__ q,
    const double* __restrict__ beta, const double* __restrict__ rho,
    const uint8_t* __restrict__ stop_status)
{
    if (status_has_stopped(stop_status[0])) return;
    const double bt = beta[0];
    if (bt == 0.0) return;
    const double tmp = rho[0] / bt;
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    double2* x2 = reinterpret_cast<double2*>(x);
    double2* r2 = reinterpret_cast<double2*>(r);
    const double2* p2 = reinterpret_cast<const double2*>(p);
    const double2* q2 = reinterpret_cast<const double2*>(q);
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < n2; i += step) {
        double2 xv = x2[i], rv = r2[i];
        const double2 pv = p2[i], qv = q2[i];
        xv.x += tmp * pv.x;
        xv.y += tmp * pv.y;
        rv.x -= tmp * qv.x;
        rv.y -= tmp * qv.y;
        x2[i] = xv;
        r2[i] = rv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        x[n - 1] += tmp * p[n - 1];
        r[n - 1] -= tmp * q[n - 1];
    }
}

inline bool aligned16(const void* p)
{
    return reinterpret_cast<uintptr_t>(p) % 16 == 0;
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int gkomi_cg_initialize_f64(
    gkomi_stream_t s, int64_t nrows, int64_t nrhs, const double* b,
    int64_t b_stride, double* r, int64_t r_stride, double* z, int64_t z_stride,
    double* p, int64_t p_stride, double* q, int64_t q_stride, double* prev_rho,
    double* rho, uint8_t* stop_status)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    const int64_t work = nrows * nrhs > nrhs ? nrows * nrhs : nrhs;
    // the first nrhs threads also reset the scalars: the grid always covers them
    int g = grid_for(work, block);
    if (static_cast<int64_t>(g) * block < nrhs) g = static_cast<int>(ceildiv(nrhs, block));
    hipLaunchKernelGGL(cg_initialize_kernel, dim3(g), dim3(block), 0,
                       to_stream(s), nrows, nrhs, b, b_stride, r, r_stride, z,
                       z_stride, p, p_stride, q, q_stride, prev_rho, rho,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_cg_step_1_f64(gkomi_stream_t s, int64_t nrows,
                                   int64_t nrhs, double* p, int64_t p_stride,
                                   const double* z, int64_t z_stride,
                                   const double* rho, const double* prev_rho,
                                   const uint8_t* stop_status)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (nrhs == 1 && p_stride == 1 && z_stride == 1 && aligned16(p) &&
        aligned16(z)) {
        hipLaunchKernelGGL(cg_step_1_vec_kernel,
                           dim3(grid_for(nrows / 2 + 1, block)), dim3(block), 0,
                           to_stream(s), nrows, p, z, rho, prev_rho,
                           stop_status);
    } else {
        hipLaunchKernelGGL(cg_step_1_kernel,
                           dim3(grid_for(nrows * nrhs, block)), dim3(block), 0,
                           to_stream(s), nrows, nrhs, p, p_stride, z, z_stride,
                           rho, prev_rho, stop_status);
    }
    return check_launch();
}

extern "C" int gkomi_cg_step_2_f64(gkomi_stream_t s, int64_t nrows,
                                   int64_t nrhs, double* x, int64_t x_stride,
                                   double* r, int64_t r_stride, const double* p,
                                   int64_t p_stride, const double* q,
                                   int64_t q_stride, const double* beta,
                                   const double* rho,
                                   const uint8_t* stop_status)
{
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (nrhs == 1 && x_stride == 1 && r_stride == 1 && p_stride == 1 &&
        q_stride == 1 && aligned16(x) && aligned16(r) && aligned16(p) &&
        aligned16(q)) {
        hipLaunchKernelGGL(cg_step_2_vec_kernel,
                           dim3(grid_for(nrows / 2 + 1, block)), dim3(block), 0,
                           to_stream(s), nrows, x, r, p, q, beta, rho,
                           stop_status);
    } else {
        hipLaunchKernelGGL(cg_step_2_kernel,
                           dim3(grid_for(nrows * nrhs, block)), dim3(block), 0,
                           to_stream(s), nrows, nrhs, x, x_stride, r, r_stride,
                           p, p_stride, q, q_stride, beta, rho, stop_status);
    }
    return check_launch();
}
