// Kernels of the fused CG drivers (one GPU: cg_solver.hip; row-partitioned:
// dist_cg.hip).  Included into each of them inside its own anonymous
// namespace: K1 = criterion + p update, K3 = x, r update + r.r partials, the
// two-output dot partials, the device-resident scalars.  See cg_solver.hip for
// the iteration they form.
#pragma once
#include "internal.hpp"

#include <cmath>

namespace gkomi {
namespace {

// the fused kernels use 1024-thread workgroups: same thread count on the chip,
// 4x fewer partials for every consumer workgroup to re-add (K3 re-reads the
// ~3900 p.q partials of K2: 15 MB of L2 traffic instead of 61 MB)
constexpr int fblock = 1024;
constexpr int max_parts = 1024;
constexpr uint8_t id_iteration = 1;  // Combined: ids count from 1 in criteria order
constexpr uint8_t id_residual = 2;

// device-resident solver scalars (the reference's 1x1 Dense workspace scalars)
struct cg_scalars {
    double rho[2];      // rho of iteration it lives in rho[it & 1]
    double tau;         // ||r|| at the last evaluated check
    double orig_tau;    // baseline norm
    double beta;
    long long stop_iter;  // iteration index at which the criterion fired
    unsigned char status;
    unsigned char pad[7];
};

__device__ __forceinline__ double sum_partials(const double* __restrict__ part,
                                               int nparts, double* smem)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += fblock) acc += part[i];
    acc = wave_reduce_sum(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = acc;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < fblock / wave_size; ++w) total += smem[w];
    return total;  // identical in every thread of every workgroup
}

// K1.  rho_part/tau_part may alias (Identity preconditioner: z == r).
__global__ __launch_bounds__(fblock) void cg_fused_step1_kernel(
    int64_t n, double* __restrict__ p, const double* __restrict__ z,
    const double* __restrict__ rho_part, int n_rho,
    const double* __restrict__ tau_part, int n_tau, cg_scalars* scal,
    long long it, long long max_iters, double goal, host_watch_line* watch = nullptr)
{
    __shared__ double smem[fblock / wave_size];
    const bool stopped_before = status_has_stopped(scal->status);
    if (stopped_before) {
        // (the host's view of the solve, internal.hpp: iteration reached, iteration stopped)
        if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, scal->stop_iter);
        return;
    }
    // the first sweep's loads do not depend on the scalars: issue them before
    // the partial sums so their latency hides behind the reduction
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * fblock;
    const int64_t i0 = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x;
    double2 z0 = make_double2(0.0, 0.0), p0 = make_double2(0.0, 0.0);
    if (i0 < n2) {
        z0 = reinterpret_cast<const double2*>(z)[i0];
        p0 = reinterpret_cast<const double2*>(p)[i0];
    }
    const double rho = sum_partials(rho_part, n_rho, smem);
    const double tau2 = rho_part == tau_part ? rho : sum_partials(tau_part, n_tau, smem);
    const double tau = sqrt(tau2);
    const double orig = scal->orig_tau;
    uint8_t st = 0;
    // Combined: Iteration is asked first, then ResidualNorm
    if (it >= max_iters) {
        st = id_iteration | GKOMI_STATUS_FINALIZED;
    } else if (tau < goal * orig) {
        st = GKOMI_STATUS_CONVERGED | id_residual | GKOMI_STATUS_FINALIZED;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rho[it & 1] = rho;
        scal->tau = tau;
        if (st) {
            scal->stop_iter = it;
            scal->status = st;
        }
        host_watch_publish(watch, it, st ? it : -1ll);
    }
    if (st) return;
    const double prev = scal->rho[(it + 1) & 1];
    const bool restart = prev == 0.0;
    const double tmp = restart ? 0.0 : rho / prev;
    double2* p2 = reinterpret_cast<double2*>(p);
    const double2* z2 = reinterpret_cast<const double2*>(z);
    if (i0 < n2) {
        if (!restart) {
            z0.x = z0.x + tmp * p0.x;
            z0.y = z0.y + tmp * p0.y;
        }
        p2[i0] = z0;
    }
    for (int64_t i = i0 + step; i < n2; i += step) {
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

// K3.  Leaves partial[blockIdx.x] = sum of r_new^2 over this workgroup's share.
__global__ __launch_bounds__(fblock) void cg_fused_step2_kernel(
    int64_t n, double* __restrict__ x, double* __restrict__ r,
    const double* __restrict__ p, const double* __restrict__ q,
    const double* __restrict__ beta_part, int n_beta, cg_scalars* scal,
    long long it, double* __restrict__ rr_part)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) return;
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * fblock;
    const int64_t i0 = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x;
    double2* x2 = reinterpret_cast<double2*>(x);
    double2* r2 = reinterpret_cast<double2*>(r);
    const double2* p2 = reinterpret_cast<const double2*>(p);
    const double2* q2 = reinterpret_cast<const double2*>(q);
    // first sweep's loads before the partial sums (independent of beta)
    double2 x0 = make_double2(0.0, 0.0), r0 = x0, p0 = x0, q0 = x0;
    if (i0 < n2) {
        x0 = x2[i0];
        r0 = r2[i0];
        p0 = p2[i0];
        q0 = q2[i0];
    }
    const double beta = sum_partials(beta_part, n_beta, smem);
    const double rho = scal->rho[it & 1];
    const bool update = beta != 0.0;
    const double tmp = update ? rho / beta : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal->beta = beta;
    double acc0 = 0.0, acc1 = 0.0;
    if (i0 < n2) {
        if (update) {
            x0.x += tmp * p0.x;
            x0.y += tmp * p0.y;
            r0.x -= tmp * q0.x;
            r0.y -= tmp * q0.y;
            x2[i0] = x0;
            r2[i0] = r0;
        }
        acc0 += r0.x * r0.x;
        acc1 += r0.y * r0.y;
    }
    for (int64_t i = i0 + step; i < n2; i += step) {
        double2 rv = r2[i];
        if (update) {
            double2 xv = x2[i];
            const double2 pv = p2[i], qv = q2[i];
            xv.x += tmp * pv.x;
            xv.y += tmp * pv.y;
            rv.x -= tmp * qv.x;
            rv.y -= tmp * qv.y;
            x2[i] = xv;
            r2[i] = rv;
        }
        acc0 += rv.x * rv.x;
        acc1 += rv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        if (update) {
            x[n - 1] += tmp * p[n - 1];
            r[n - 1] -= tmp * q[n - 1];
        }
        acc0 += r[n - 1] * r[n - 1];
    }
    __syncthreads();
    const double total = block_reduce_sum<fblock>(acc0 + acc1, smem);
    if (threadIdx.x == 0) rr_part[blockIdx.x] = total;
}

// partial[blockIdx.x] = sum x*y over the workgroup's share; two outputs so that
// r.z and r.r come from one pass when a preconditioner is present
__global__ __launch_bounds__(fblock) void cg_dot2_partials_kernel(
    int64_t n, const double* __restrict__ r, const double* __restrict__ z,
    const cg_scalars* scal, double* __restrict__ rz_part,
    double* __restrict__ rr_part)
{
    __shared__ double smem[fblock / wave_size];
    if (scal != nullptr && status_has_stopped(scal->status)) return;
    // r and z are workspace vectors (256-B aligned): 16 B per lane
    const int64_t step = static_cast<int64_t>(gridDim.x) * fblock;
    const int64_t n2 = n / 2;
    const double2* r2 = reinterpret_cast<const double2*>(r);
    const double2* z2 = reinterpret_cast<const double2*>(z);
    double a = 0.0, bb = 0.0, a1 = 0.0, b1 = 0.0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x;
         i < n2; i += step) {
        const double2 rv = r2[i], zv = z2[i];
        a += rv.x * zv.x;
        a1 += rv.y * zv.y;
        bb += rv.x * rv.x;
        b1 += rv.y * rv.y;
    }
    a += a1;
    bb += b1;
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        a += r[n - 1] * z[n - 1];
        bb += r[n - 1] * r[n - 1];
    }
    const double ta = block_reduce_sum<fblock>(a, smem);
    __syncthreads();
    const double tb = block_reduce_sum<fblock>(bb, smem);
    if (threadIdx.x == 0) {
        rz_part[blockIdx.x] = ta;
        if (rr_part != nullptr) rr_part[blockIdx.x] = tb;
    }
}

__global__ void cg_init_scalars_kernel(cg_scalars* scal, const double* orig_tau,
                                       int baseline_absolute)
{
    scal->rho[0] = 0.0;
    scal->rho[1] = 1.0;  // prev_rho = 1 (reference cg::initialize)
    scal->tau = 0.0;
    scal->orig_tau = baseline_absolute ? 1.0 : orig_tau[0];
    scal->beta = 0.0;
    scal->stop_iter = -1;
    scal->status = 0;
}

int vec_grid(int64_t n) { return fused_vec_grid(n); }  // internal.hpp

}  // namespace
}  // namespace gkomi
