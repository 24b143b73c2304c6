// Native CG driver for gfx950: Cg::apply_dense_impl (core/solver/cg.cpp:107-193)
// with the criteria Combined(Iteration, ResidualNorm)
// (core/stop/combined.cpp:40, core/stop/residual_norm.cpp:119-228).
//
// mode 0 replays the reference's kernel sequence one launch per kernel and
// checks the criterion on the host every iteration (a blocking 2-byte D2H copy
// per iteration, as hip/stop/residual_norm_kernels.hip.cpp:119-120 does).
//
// mode 1 is the MI355X design: three launches per iteration, every scalar on
// the device, no per-iteration host round trip.
//   K1 step1 : every workgroup re-adds the <=1024 partials of rho = r.z and
//              tau^2 = r.r left by K3 (same order in every workgroup, so all
//              agree bit for bit), evaluates the criterion, and -- unless
//              stopped -- p = z + (rho/prev_rho) p.
//   K2 spmv  : q = A p (stream kernel) + partials of beta = p.q.
//   K3 step2 : re-adds the beta partials, x += (rho/beta) p, r -= (rho/beta) q,
//              and leaves the partials of r.r for the next K1.
// Once the criterion fires, K1 records the iteration and sets the
// stopping_status; all later launches return immediately, so x, r and the
// iteration count are exactly those of the iteration that stopped.  The host
// polls the status every `check_every` iterations.
// HBM traffic per iteration (n rows, Identity): K1 3n, K2 matrix + 2n (+n
// for p in the epilogue, L2-resident), K3 6n values -- vs 18n + matrix in the
// reference's accounting (core/solver/cg.cpp:148-156).
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
    long long it, long long max_iters, double goal)
{
    __shared__ double smem[fblock / wave_size];
    const bool stopped_before = status_has_stopped(scal->status);
    if (stopped_before) return;
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

int vec_grid(int64_t n)
{
    int64_t g = ceildiv(n / 2 + 1, fblock);
    if (g > max_parts) g = max_parts;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct workspace_layout {
    size_t r, z, p, q, scalars, part_a, part_b, part_c, red, small, total;
};

workspace_layout make_layout(int64_t n, int64_t nrhs)
{
    workspace_layout l{};
    const size_t vec = align_up(sizeof(double) * static_cast<size_t>(n) * nrhs, 256);
    size_t off = 0;
    l.r = off; off += vec;
    l.z = off; off += vec;
    l.p = off; off += vec;
    l.q = off; off += vec;
    l.scalars = off; off += 256;
    const size_t nb = static_cast<size_t>(csr_spmv_dot_num_partials(static_cast<int>(n)));
    l.part_a = off; off += align_up(sizeof(double) * max_parts, 256);   // r.z / r.r
    l.part_b = off; off += align_up(sizeof(double) * max_parts, 256);   // r.r with a preconditioner
    l.part_c = off; off += align_up(sizeof(double) * (nb + 1), 256);    // p.q
    l.red = off; off += align_up(gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8, 256);
    // mode 0 scalars: alpha-free set {prev_rho, rho, beta, tau, orig_tau, one, neg_one} x nrhs,
    // then stop_status[nrhs] and 2 flag bytes
    l.small = off; off += align_up(sizeof(double) * 8 * static_cast<size_t>(nrhs) + nrhs + 16, 256);
    l.total = off;
    return l;
}

#define GKOMI_TRY(expr)          \
    do {                         \
        int err_ = (expr);       \
        if (err_) return err_;   \
    } while (0)

int identity_or_precond(gkomi_apply_fn precond, void* ctx, gkomi_stream_t s,
                        int64_t n, int64_t nrhs, const double* r, double* z)
{
    if (precond == nullptr) {
        // matrix::Identity::apply copies (core/matrix/identity.cpp)
        return gkomi_dense_copy_f64(s, n, nrhs, r, nrhs, z, nrhs);
    }
    return precond(ctx, s, r, z);
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_cg_workspace_bytes(int64_t n, int64_t nrhs)
{
    if (n < 0 || nrhs <= 0) return 0;
    return make_layout(n, nrhs).total;
}

namespace {
int cg_solve_impl(gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A, gkomi_apply_fn precond,
                  void* precond_ctx, const double* b, double* x, int64_t max_iters,
                  double reduction_factor, int baseline, int mode, int check_every, void* workspace,
                  size_t workspace_bytes, double* host_info)
{
    const int64_t nnz = A.nnz;
    const int32_t* row_ptrs = A.row_ptrs;
    const int32_t* col_idxs = A.col_idxs;
    const double* vals = A.vals;
    if (mode == 1 && (!A.is_csr() || n == 0)) mode = 0;  // the fused path needs the CSR arrays (and rows)
    if (n < 0 || nrhs <= 0 || max_iters < 0) return GKOMI_EINVAL;
    if (baseline < 0 || baseline > 2 || (mode != 0 && mode != 1)) return GKOMI_EINVAL;
    if (mode == 1 && nrhs != 1) return GKOMI_ENOTSUPPORTED;
    if (n > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    const workspace_layout l = make_layout(n, nrhs);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    double* r = reinterpret_cast<double*>(ws + l.r);
    double* z = reinterpret_cast<double*>(ws + l.z);
    double* p = reinterpret_cast<double*>(ws + l.p);
    double* q = reinterpret_cast<double*>(ws + l.q);
    void* red = ws + l.red;
    const size_t red_bytes = gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8;
    double* small = reinterpret_cast<double*>(ws + l.small);
    double* prev_rho = small;
    double* rho = small + nrhs;
    double* beta = small + 2 * nrhs;
    double* tau = small + 3 * nrhs;
    double* orig_tau = small + 4 * nrhs;
    double* one = small + 5 * nrhs;
    double* neg_one = small + 6 * nrhs;
    uint8_t* stop_status = reinterpret_cast<uint8_t*>(small + 8 * nrhs);
    uint8_t* dev_flags = stop_status + nrhs + (8 - nrhs % 8) % 8;

    // cg::initialize, then r = b - A x (advanced apply), cg.cpp:137-142
    GKOMI_TRY(gkomi_cg_initialize_f64(s, n, nrhs, b, nrhs, r, nrhs, z, nrhs, p,
                                      nrhs, q, nrhs, prev_rho, rho, stop_status));
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, one, nrhs, 1.0));
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, neg_one, nrhs, -1.0));
    GKOMI_TRY(A.apply(s, nrhs, neg_one, x, one, r));
    // criterion generate: baseline norm (residual_norm.cpp:119-189)
    if (baseline == 0) {
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, b, nrhs, orig_tau, red, red_bytes));
    } else if (baseline == 1) {
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, orig_tau, red, red_bytes));
    } else {
        GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, orig_tau, nrhs, 1.0));
    }

    long long iterations = -1;
    int converged = 0;

    if (mode == 0) {
        uint8_t host_flags[2] = {0, 0};
        long long iter = -1;
        while (true) {
            GKOMI_TRY(identity_or_precond(precond, precond_ctx, s, n, nrhs, r, z));
            GKOMI_TRY(gkomi_dense_compute_dot_f64(s, n, nrhs, r, nrhs, z, nrhs, rho, red, red_bytes));
            ++iter;
            bool stop = false;
            if (iter >= max_iters) {
                GKOMI_TRY(gkomi_set_all_statuses(s, nrhs, id_iteration, 1, stop_status));
                stop = true;
            } else {
                GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, tau, red, red_bytes));
                GKOMI_TRY(gkomi_residual_norm_f64(s, nrhs, tau, orig_tau, reduction_factor,
                                                  id_residual, 1, stop_status, dev_flags,
                                                  host_flags));
                stop = host_flags[0] != 0;
                converged = stop ? 1 : 0;
            }
            if (stop) break;
            GKOMI_TRY(gkomi_cg_step_1_f64(s, n, nrhs, p, nrhs, z, nrhs, rho, prev_rho, stop_status));
            GKOMI_TRY(A.apply(s, nrhs, nullptr, p, nullptr, q));
            GKOMI_TRY(gkomi_dense_compute_dot_f64(s, n, nrhs, p, nrhs, q, nrhs, beta, red, red_bytes));
            GKOMI_TRY(gkomi_cg_step_2_f64(s, n, nrhs, x, nrhs, r, nrhs, p, nrhs, q, nrhs, beta, rho,
                                          stop_status));
            std::swap(prev_rho, rho);
        }
        iterations = iter;
        if (host_info != nullptr) {
            // final recurrence residual norms for the report
            GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, tau, red, red_bytes));
            for (int64_t j = 0; j < nrhs; ++j) {
                GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 2 + 2 * j, tau + j, sizeof(double),
                                                          hipMemcpyDeviceToHost, stream)));
                GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 3 + 2 * j, orig_tau + j,
                                                          sizeof(double), hipMemcpyDeviceToHost, stream)));
            }
        }
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    } else {
        cg_scalars* scal = reinterpret_cast<cg_scalars*>(ws + l.scalars);
        double* part_a = reinterpret_cast<double*>(ws + l.part_a);
        double* part_b = reinterpret_cast<double*>(ws + l.part_b);
        double* part_c = reinterpret_cast<double*>(ws + l.part_c);
        const int g = vec_grid(n);
        const int nb = csr_spmv_dot_num_partials(static_cast<int>(n));
        const bool swizzle = csr_auto_swizzle(n, nnz);
        const bool aligned = reinterpret_cast<uintptr_t>(vals) % 16 == 0 &&
                             reinterpret_cast<uintptr_t>(col_idxs) % 8 == 0 &&
                             reinterpret_cast<uintptr_t>(x) % 16 == 0;
        if (!aligned) return GKOMI_ENOTSUPPORTED;
        cg_scalars polled{};  // per call: concurrent solves on other streams / threads do not share it
        if (check_every < 1) check_every = 1;
        hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(1), 0, stream, scal, orig_tau,
                           baseline == 2 ? 1 : 0);
        GKOMI_TRY(check_launch());
        // partials of r.z (and r.r) for the first check
        const double* zz = precond == nullptr ? r : z;
        if (precond != nullptr) GKOMI_TRY(precond(precond_ctx, s, r, z));
        hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, zz,
                           static_cast<const cg_scalars*>(nullptr), part_a,
                           precond == nullptr ? nullptr : part_b);
        GKOMI_TRY(check_launch());
        const double* tau_part = precond == nullptr ? part_a : part_b;
        long long it = 0;
        bool done = false;
        while (!done) {
            for (int c = 0; c < check_every; ++c, ++it) {
                hipLaunchKernelGGL(cg_fused_step1_kernel, dim3(g), dim3(fblock), 0, stream, n, p, zz,
                                   part_a, g, tau_part, g, scal, it,
                                   static_cast<long long>(max_iters), reduction_factor);
                GKOMI_TRY(csr_spmv_dot_launch(stream, static_cast<int>(n), nnz, row_ptrs, col_idxs,
                                              vals, p, q, part_c, &scal->status, swizzle));
                hipLaunchKernelGGL(cg_fused_step2_kernel, dim3(g), dim3(fblock), 0, stream, n, x, r, p,
                                   q, part_c, nb, scal, it, precond == nullptr ? part_a : part_b);
                if (precond != nullptr) {
                    GKOMI_TRY(precond(precond_ctx, s, r, z));
                    hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r,
                                       z, static_cast<const cg_scalars*>(scal), part_a,
                                       static_cast<double*>(nullptr));
                }
                if (it >= max_iters) {  // the launch with it == max_iters stops for sure
                    ++it;
                    break;
                }
            }
            GKOMI_TRY(check_launch());
            GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&polled, scal, sizeof(cg_scalars),
                                                      hipMemcpyDeviceToHost, stream)));
            GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
            const cg_scalars* h = &polled;
            if (h->status & GKOMI_STATUS_ID_MASK) {
                done = true;
                iterations = h->stop_iter;
                converged = (h->status & GKOMI_STATUS_CONVERGED) ? 1 : 0;
                if (host_info != nullptr) {
                    host_info[2] = h->tau;
                    host_info[3] = h->orig_tau;
                }
            }
        }
    }
    if (host_info != nullptr) {
        host_info[0] = static_cast<double>(iterations);
        host_info[1] = static_cast<double>(converged);
    }
    return GKOMI_SUCCESS;
}
}  // namespace

extern "C" int gkomi_cg_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz,
    const int32_t* row_ptrs, const int32_t* col_idxs, const double* vals,
    int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int mode, int check_every,
    void* workspace, size_t workspace_bytes, double* host_info)
{
    return cg_solve_impl(s, n, nrhs,
                         make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint),
                         precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, mode,
                         check_every, workspace, workspace_bytes, host_info);
}

// the system matrix behind a callback: the reference kernel sequence (mode 0)
extern "C" int gkomi_cg_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                     gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                     gkomi_apply_fn precond, void* precond_ctx, const double* b,
                                     double* x, int64_t max_iters, double reduction_factor,
                                     int baseline, void* workspace, size_t workspace_bytes,
                                     double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return cg_solve_impl(s, n, nrhs, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                         max_iters, reduction_factor, baseline, 0, 1, workspace, workspace_bytes,
                         host_info);
}
