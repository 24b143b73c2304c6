// BiCGSTAB, FCG and CGS for gfx950 (the Krylov solvers that
// share CG's BLAS-1).  Replaces gko::kernels::hip::{bicgstab, fcg, cgs}::
// {initialize, step_1, step_2, step_3, finalize}
// (core/solver/{bicgstab,fcg,cgs}_kernels.hpp; the reference's GPU versions
// are the unified kernels common/unified/solver/*_kernels.cpp) and provides
// native drivers for {Bicgstab,Fcg,Cgs}::apply_dense_impl
// (core/solver/bicgstab.cpp:107-234, fcg.cpp:104-196, cgs.cpp:107-205).
// Semantics = reference/solver/{bicgstab,fcg,cgs}_kernels.cpp.
//
// The step kernels are pure streaming (3n..7n values, HBM-bound): one thread
// per element, the per-column scalars read from device memory, every
// expression written exactly as the reference's (-ffp-contract=off) ->
// bit-identical.  Scalars a step defines (alpha, omega, beta) are recomputed by
// every thread from their inputs and stored once by row 0.
#include <algorithm>
#include <utility>

#include "common.hpp"
#include "internal.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr uint8_t finalized_mask = 0x40;  // stopping_status::is_finalized (stopping_status.hpp)

#define GKOMI_ELEMENTWISE(i, j)                                                         \
    const int64_t idx_ = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;        \
    if (idx_ >= n * nrhs) return;                                                       \
    const int64_t i = idx_ / nrhs;                                                      \
    const int64_t j = idx_ - i * nrhs

// ---- BiCGSTAB ---------------------------------------------------------------
__global__ __launch_bounds__(block) void bicgstab_initialize_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ r, int64_t r_stride, double* __restrict__ rr, int64_t rr_stride,
    double* __restrict__ y, int64_t y_stride, double* __restrict__ s, int64_t s_stride,
    double* __restrict__ t, int64_t t_stride, double* __restrict__ z, int64_t z_stride,
    double* __restrict__ v, int64_t v_stride, double* __restrict__ p, int64_t p_stride,
    double* __restrict__ prev_rho, double* __restrict__ rho, double* __restrict__ alpha,
    double* __restrict__ beta, double* __restrict__ gamma, double* __restrict__ omega,
    uint8_t* __restrict__ stop_status)
{
    const int64_t idx = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (idx < nrhs) {
        rho[idx] = prev_rho[idx] = alpha[idx] = beta[idx] = gamma[idx] = omega[idx] = 1.0;
        stop_status[idx] = 0;
    }
    if (idx >= n * nrhs) return;
    const int64_t i = idx / nrhs, j = idx - i * nrhs;
    r[i * r_stride + j] = b[i * b_stride + j];
    rr[i * rr_stride + j] = z[i * z_stride + j] = v[i * v_stride + j] = s[i * s_stride + j] =
        t[i * t_stride + j] = y[i * y_stride + j] = p[i * p_stride + j] = 0.0;
}

__global__ __launch_bounds__(block) void bicgstab_step_1_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ r, int64_t r_stride,
    double* __restrict__ p, int64_t p_stride, const double* __restrict__ v, int64_t v_stride,
    const double* __restrict__ rho, const double* __restrict__ prev_rho,
    const double* __restrict__ alpha, const double* __restrict__ omega,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    if (prev_rho[j] * omega[j] != 0.0) {
        const double tmp = rho[j] / prev_rho[j] * alpha[j] / omega[j];
        p[i * p_stride + j] =
            r[i * r_stride + j] + tmp * (p[i * p_stride + j] - omega[j] * v[i * v_stride + j]);
    } else {
        p[i * p_stride + j] = r[i * r_stride + j];
    }
}

__global__ __launch_bounds__(block) void bicgstab_step_2_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ r, int64_t r_stride,
    double* __restrict__ s, int64_t s_stride, const double* __restrict__ v, int64_t v_stride,
    const double* __restrict__ rho, double* __restrict__ alpha, const double* __restrict__ beta,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    double a = 0.0;
    if (beta[j] != 0.0) {
        a = rho[j] / beta[j];
        s[i * s_stride + j] = r[i * r_stride + j] - a * v[i * v_stride + j];
    } else {
        s[i * s_stride + j] = r[i * r_stride + j];
    }
    if (i == 0) alpha[j] = a;  // nobody reads alpha in this kernel
}

__global__ __launch_bounds__(block) void bicgstab_step_3_kernel(
    int64_t n, int64_t nrhs, double* __restrict__ x, int64_t x_stride, double* __restrict__ r,
    int64_t r_stride, const double* __restrict__ s, int64_t s_stride,
    const double* __restrict__ t, int64_t t_stride, const double* __restrict__ y,
    int64_t y_stride, const double* __restrict__ z, int64_t z_stride,
    const double* __restrict__ alpha, const double* __restrict__ beta,
    const double* __restrict__ gamma, double* __restrict__ omega,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    const double om = beta[j] != 0.0 ? gamma[j] / beta[j] : 0.0;
    x[i * x_stride + j] += alpha[j] * y[i * y_stride + j] + om * z[i * z_stride + j];
    r[i * r_stride + j] = s[i * s_stride + j] - om * t[i * t_stride + j];
    if (i == 0) omega[j] = om;
}

// x += alpha * y for the columns that stopped but are not finalized yet ...
__global__ __launch_bounds__(block) void bicgstab_finalize_kernel(
    int64_t n, int64_t nrhs, double* __restrict__ x, int64_t x_stride,
    const double* __restrict__ y, int64_t y_stride, const double* __restrict__ alpha,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    const uint8_t st = stop_status[j];
    if (status_has_stopped(st) && !(st & finalized_mask)) {
        x[i * x_stride + j] += alpha[j] * y[i * y_stride + j];
    }
}

// ... which then become finalized (separate launch: every row reads the status)
__global__ __launch_bounds__(block) void finalize_status_kernel(int64_t nrhs, bool have_rows,
                                                                uint8_t* __restrict__ stop_status)
{
    const int64_t j = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (j < nrhs && have_rows && status_has_stopped(stop_status[j])) stop_status[j] |= finalized_mask;
}

// ---- FCG ----------------------------------------------------------------------
__global__ __launch_bounds__(block) void fcg_initialize_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ r, int64_t r_stride, double* __restrict__ z, int64_t z_stride,
    double* __restrict__ p, int64_t p_stride, double* __restrict__ q, int64_t q_stride,
    double* __restrict__ t, int64_t t_stride, double* __restrict__ prev_rho,
    double* __restrict__ rho, double* __restrict__ rho_t, uint8_t* __restrict__ stop_status)
{
    const int64_t idx = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (idx < nrhs) {
        rho[idx] = 0.0;
        prev_rho[idx] = rho_t[idx] = 1.0;
        stop_status[idx] = 0;
    }
    if (idx >= n * nrhs) return;
    const int64_t i = idx / nrhs, j = idx - i * nrhs;
    t[i * t_stride + j] = r[i * r_stride + j] = b[i * b_stride + j];
    z[i * z_stride + j] = p[i * p_stride + j] = q[i * q_stride + j] = 0.0;
}

__global__ __launch_bounds__(block) void fcg_step_1_kernel(
    int64_t n, int64_t nrhs, double* __restrict__ p, int64_t p_stride,
    const double* __restrict__ z, int64_t z_stride, const double* __restrict__ rho_t,
    const double* __restrict__ prev_rho, const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    if (prev_rho[j] == 0.0) {
        p[i * p_stride + j] = z[i * z_stride + j];
    } else {
        const double tmp = rho_t[j] / prev_rho[j];
        p[i * p_stride + j] = z[i * z_stride + j] + tmp * p[i * p_stride + j];
    }
}

__global__ __launch_bounds__(block) void fcg_step_2_kernel(
    int64_t n, int64_t nrhs, double* __restrict__ x, int64_t x_stride, double* __restrict__ r,
    int64_t r_stride, double* __restrict__ t, int64_t t_stride, const double* __restrict__ p,
    int64_t p_stride, const double* __restrict__ q, int64_t q_stride,
    const double* __restrict__ beta, const double* __restrict__ rho,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    if (beta[j] != 0.0) {
        const double tmp = rho[j] / beta[j];
        const double prev_r = r[i * r_stride + j];
        x[i * x_stride + j] += tmp * p[i * p_stride + j];
        const double new_r = prev_r - tmp * q[i * q_stride + j];
        r[i * r_stride + j] = new_r;
        t[i * t_stride + j] = new_r - prev_r;
    }
}

// ---- CGS ----------------------------------------------------------------------
__global__ __launch_bounds__(block) void cgs_initialize_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ r, int64_t r_stride, double* __restrict__ r_tld, int64_t r_tld_stride,
    double* __restrict__ p, int64_t p_stride, double* __restrict__ q, int64_t q_stride,
    double* __restrict__ u, int64_t u_stride, double* __restrict__ u_hat, int64_t u_hat_stride,
    double* __restrict__ v_hat, int64_t v_hat_stride, double* __restrict__ t, int64_t t_stride,
    double* __restrict__ alpha, double* __restrict__ beta, double* __restrict__ gamma,
    double* __restrict__ prev_rho, double* __restrict__ rho, uint8_t* __restrict__ stop_status)
{
    const int64_t idx = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (idx < nrhs) {
        rho[idx] = 0.0;
        prev_rho[idx] = alpha[idx] = beta[idx] = gamma[idx] = 1.0;
        stop_status[idx] = 0;
    }
    if (idx >= n * nrhs) return;
    const int64_t i = idx / nrhs, j = idx - i * nrhs;
    r[i * r_stride + j] = r_tld[i * r_tld_stride + j] = b[i * b_stride + j];
    u[i * u_stride + j] = u_hat[i * u_hat_stride + j] = p[i * p_stride + j] = q[i * q_stride + j] =
        v_hat[i * v_hat_stride + j] = t[i * t_stride + j] = 0.0;
}

// beta is rewritten only when prev_rho != 0, and then nobody reads the old value
__global__ __launch_bounds__(block) void cgs_step_1_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ r, int64_t r_stride,
    double* __restrict__ u, int64_t u_stride, double* __restrict__ p, int64_t p_stride,
    const double* __restrict__ q, int64_t q_stride, double* beta, const double* __restrict__ rho,
    const double* __restrict__ prev_rho, const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    const bool update = prev_rho[j] != 0.0;
    const double bt = update ? rho[j] / prev_rho[j] : beta[j];
    const double uu = r[i * r_stride + j] + bt * q[i * q_stride + j];
    u[i * u_stride + j] = uu;
    p[i * p_stride + j] = uu + bt * (q[i * q_stride + j] + bt * p[i * p_stride + j]);
    if (i == 0 && update) beta[j] = bt;
}

__global__ __launch_bounds__(block) void cgs_step_2_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ u, int64_t u_stride,
    const double* __restrict__ v_hat, int64_t v_hat_stride, double* __restrict__ q,
    int64_t q_stride, double* __restrict__ t, int64_t t_stride, double* alpha,
    const double* __restrict__ rho, const double* __restrict__ gamma,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    const bool update = gamma[j] != 0.0;
    const double a = update ? rho[j] / gamma[j] : alpha[j];
    const double qq = u[i * u_stride + j] - a * v_hat[i * v_hat_stride + j];
    q[i * q_stride + j] = qq;
    t[i * t_stride + j] = u[i * u_stride + j] + qq;
    if (i == 0 && update) alpha[j] = a;
}

__global__ __launch_bounds__(block) void cgs_step_3_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ t, int64_t t_stride,
    const double* __restrict__ u_hat, int64_t u_hat_stride, double* __restrict__ r,
    int64_t r_stride, double* __restrict__ x, int64_t x_stride, const double* __restrict__ alpha,
    const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    x[i * x_stride + j] += alpha[j] * u_hat[i * u_hat_stride + j];
    r[i * r_stride + j] -= alpha[j] * t[i * t_stride + j];
}

// ---- BiCG ---------------------------------------------------------------------
__global__ __launch_bounds__(block) void bicg_initialize_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ r, int64_t r_stride, double* __restrict__ z, int64_t z_stride,
    double* __restrict__ p, int64_t p_stride, double* __restrict__ q, int64_t q_stride,
    double* __restrict__ prev_rho, double* __restrict__ rho, double* __restrict__ r2,
    int64_t r2_stride, double* __restrict__ z2, int64_t z2_stride, double* __restrict__ p2,
    int64_t p2_stride, double* __restrict__ q2, int64_t q2_stride,
    uint8_t* __restrict__ stop_status)
{
    const int64_t idx = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (idx < nrhs) {
        rho[idx] = 0.0;
        prev_rho[idx] = 1.0;
        stop_status[idx] = 0;
    }
    if (idx >= n * nrhs) return;
    const int64_t i = idx / nrhs, j = idx - i * nrhs;
    r[i * r_stride + j] = r2[i * r2_stride + j] = b[i * b_stride + j];
    z[i * z_stride + j] = p[i * p_stride + j] = q[i * q_stride + j] = 0.0;
    z2[i * z2_stride + j] = p2[i * p2_stride + j] = q2[i * q2_stride + j] = 0.0;
}

__global__ __launch_bounds__(block) void bicg_step_1_kernel(
    int64_t n, int64_t nrhs, double* __restrict__ p, int64_t p_stride,
    const double* __restrict__ z, int64_t z_stride, double* __restrict__ p2, int64_t p2_stride,
    const double* __restrict__ z2, int64_t z2_stride, const double* __restrict__ rho,
    const double* __restrict__ prev_rho, const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    if (prev_rho[j] == 0.0) {
        p[i * p_stride + j] = z[i * z_stride + j];
        p2[i * p2_stride + j] = z2[i * z2_stride + j];
    } else {
        const double tmp = rho[j] / prev_rho[j];
        p[i * p_stride + j] = z[i * z_stride + j] + tmp * p[i * p_stride + j];
        p2[i * p2_stride + j] = z2[i * z2_stride + j] + tmp * p2[i * p2_stride + j];
    }
}

__global__ __launch_bounds__(block) void bicg_step_2_kernel(
    int64_t n, int64_t nrhs, double* __restrict__ x, int64_t x_stride, double* __restrict__ r,
    int64_t r_stride, double* __restrict__ r2, int64_t r2_stride, const double* __restrict__ p,
    int64_t p_stride, const double* __restrict__ q, int64_t q_stride,
    const double* __restrict__ q2, int64_t q2_stride, const double* __restrict__ beta,
    const double* __restrict__ rho, const uint8_t* __restrict__ stop_status)
{
    GKOMI_ELEMENTWISE(i, j);
    if (status_has_stopped(stop_status[j])) return;
    if (beta[j] != 0.0) {
        const double tmp = rho[j] / beta[j];
        x[i * x_stride + j] += tmp * p[i * p_stride + j];
        r[i * r_stride + j] -= tmp * q[i * q_stride + j];
        r2[i * r2_stride + j] -= tmp * q2[i * q2_stride + j];
    }
}

__global__ __launch_bounds__(block) void ir_initialize_kernel(int64_t nrhs,
                                                              uint8_t* __restrict__ stop_status)
{
    const int64_t j = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (j < nrhs) stop_status[j] = 0;
}

// the scalar updates of steps that define a scalar must also happen for n == 0
// (the reference loops over the columns first): tiny single-block kernels
__global__ void cgs_scalar_kernel(int64_t nrhs, double* out, const double* num, const double* den,
                                  const uint8_t* stop_status)
{
    for (int64_t j = threadIdx.x; j < nrhs; j += blockDim.x) {
        if (!status_has_stopped(stop_status[j]) && den[j] != 0.0) out[j] = num[j] / den[j];
    }
}
__global__ void bicgstab_omega_kernel(int64_t nrhs, double* omega, const double* gamma,
                                      const double* beta, const uint8_t* stop_status)
{
    for (int64_t j = threadIdx.x; j < nrhs; j += blockDim.x) {
        if (!status_has_stopped(stop_status[j])) omega[j] = beta[j] != 0.0 ? gamma[j] / beta[j] : 0.0;
    }
}

// deferred criterion: the device remembers where every column had stopped; the
// host looks only every few iterations (the step kernels skip stopped columns,
// so the extra iterations launched meanwhile change nothing)
struct stop_record {
    long long iter;
    int phase;
    int pad_;
};

__global__ void record_stop_kernel(const uint8_t* __restrict__ flags, long long iter, int phase,
                                   stop_record* __restrict__ rec)
{
    if (flags[0] != 0 && rec->iter < 0) {
        rec->iter = iter;
        rec->phase = phase;
    }
}

bool bad_dims(int64_t n, int64_t nrhs) { return n < 0 || nrhs < 0; }
dim3 grid_of(int64_t n, int64_t nrhs) { return dim3(static_cast<unsigned>(ceildiv(std::max<int64_t>(n * nrhs, nrhs), block))); }

#define GKOMI_TRY(expr)        \
    do {                       \
        const int e_ = (expr); \
        if (e_) return e_;     \
    } while (0)

// ---- drivers ------------------------------------------------------------------
struct solver_layout {
    size_t vec[8], small, red, parts, total;
};

solver_layout make_solver_layout(int64_t n, int64_t nrhs, int nvec)
{
    solver_layout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    for (int k = 0; k < nvec; ++k) l.vec[k] = take(sizeof(double) * static_cast<size_t>(n) * nrhs + 8);
    // 10 scalar rows + statuses + flags
    l.small = take(sizeof(double) * 10 * nrhs + 2 * static_cast<size_t>(nrhs) + 128);
    l.red = take(gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8);
    // fused single-rhs drivers: 3 x <=1024 partials of the vector kernels and
    // 3 x one partial per SpMV row block, plus the device scalars
    l.parts = take(sizeof(double) * (3 * 1024 + 3 * (spmv_dot_partials_room(n))) + 256);
    l.total = off;
    return l;
}

// what the three drivers share: r = b - A x, the baseline norm, the criterion
struct driver_common {
    gkomi_stream_t s;
    hipStream_t stream;
    int64_t n, nrhs;
    sysmat A;
    gkomi_apply_fn precond;
    void* precond_ctx;
    int64_t max_iters;
    double reduction;
    double *tau, *orig_tau, *one, *neg_one;
    uint8_t *stop_status, *dev_flags;
    void* red;
    size_t red_bytes;
    int converged = 0;
    uint8_t host_flags[2] = {0, 0};

    int spmv(const double* in, double* out) const
    {
        return A.apply(s, nrhs, nullptr, in, nullptr, out);
    }
    int apply_precond(const double* in, double* out) const
    {
        if (precond == nullptr) return gkomi_dense_copy_f64(s, n, nrhs, in, nrhs, out, nrhs);
        return precond(precond_ctx, s, in, out);
    }
    int dot(const double* a, const double* b2, double* result) const
    {
        return gkomi_dense_compute_dot_f64(s, n, nrhs, a, nrhs, b2, nrhs, result, red, red_bytes);
    }
    int start(const double* b, const double* x, double* r, int baseline)
    {
        GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, one, nrhs, 1.0));
        GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, neg_one, nrhs, -1.0));
        // r = b - A x (r already holds b)
        GKOMI_TRY(A.apply(s, nrhs, neg_one, x, one, r));
        if (baseline == 0) {
            return gkomi_dense_compute_norm2_f64(s, n, nrhs, b, nrhs, orig_tau, red, red_bytes);
        }
        if (baseline == 1) {
            return gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, orig_tau, red, red_bytes);
        }
        return gkomi_dense_fill_f64(s, 1, nrhs, orig_tau, nrhs, 1.0);
    }
    stop_record* record = nullptr;  // device
    int64_t check_every = 1;
    int64_t unpolled = 0;
    stop_record host_record{-1, 0, 0};

    int poll()
    {
        unpolled = 0;
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&host_record, record, sizeof(stop_record),
                                                  hipMemcpyDeviceToHost, stream)));
        return static_cast<int>(hipStreamSynchronize(stream));
    }
    // Combined(Iteration [id 1], ResidualNorm [id 1]) on `residual`; *stop = every
    // column has stopped.  The criterion itself is evaluated on the device at
    // every call, exactly where the reference evaluates it; the host learns the
    // outcome every `check_every` calls (stop_iter() is then the iteration the
    // device recorded, not the current one).
    int check(int64_t iter, const double* residual, bool set_finalized, int phase, bool* stop)
    {
        *stop = false;
        if (iter >= max_iters) {
            GKOMI_TRY(poll());  // converged during the iterations not looked at yet?
            if (host_record.iter < 0) {
                GKOMI_TRY(gkomi_set_all_statuses(s, nrhs, 1, set_finalized ? 1 : 0, stop_status));
                converged = 0;
                host_record.iter = iter;
                host_record.phase = phase;
            } else {
                converged = 1;
            }
            *stop = true;
            return 0;
        }
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, residual, nrhs, tau, red, red_bytes));
        GKOMI_TRY(gkomi_residual_norm_f64(s, nrhs, tau, orig_tau, reduction, 1,
                                          set_finalized ? 1 : 0, stop_status, dev_flags, nullptr));
        hipLaunchKernelGGL(record_stop_kernel, dim3(1), dim3(1), 0, stream, dev_flags,
                           static_cast<long long>(iter), phase, record);
        if (++unpolled >= check_every) {
            GKOMI_TRY(poll());
            if (host_record.iter >= 0) {
                converged = 1;
                *stop = true;
            }
        }
        return 0;
    }
    int64_t stop_iter() const { return static_cast<int64_t>(host_record.iter); }
    int finish(int64_t iter, const double* residual, double* host_info)
    {
        if (host_info != nullptr) {
            // report the norm of the final residual
            GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, residual, nrhs, tau, red, red_bytes));
            for (int64_t j = 0; j < nrhs; ++j) {
                GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 2 + 2 * j, tau + j,
                                                          sizeof(double), hipMemcpyDeviceToHost, stream)));
                GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 3 + 2 * j, orig_tau + j,
                                                          sizeof(double), hipMemcpyDeviceToHost, stream)));
            }
            host_info[0] = static_cast<double>(iter);
            host_info[1] = static_cast<double>(converged);
        }
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
        return precond_status(precond, precond_ctx, s);
    }
};

int make_common(driver_common& c, gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A,
                gkomi_apply_fn precond, void* precond_ctx,
                int64_t max_iters, double reduction, int baseline, int64_t check_every, char* ws,
                const solver_layout& l, double** scalars)
{
    if (n < 0 || nrhs <= 0 || max_iters < 0 || baseline < 0 || baseline > 2) return GKOMI_EINVAL;
    c.s = s;
    c.stream = to_stream(s);
    c.n = n; c.nrhs = nrhs;
    c.A = A;
    c.precond = precond; c.precond_ctx = precond_ctx;
    c.max_iters = max_iters; c.reduction = reduction;
    double* small = reinterpret_cast<double*>(ws + l.small);
    c.tau = small;
    c.orig_tau = small + nrhs;
    c.one = small + 2 * nrhs;
    c.neg_one = small + 3 * nrhs;
    *scalars = small + 4 * nrhs;  // 6 rows for the solver's own scalars
    c.stop_status = reinterpret_cast<uint8_t*>(small + 10 * nrhs);
    c.dev_flags = c.stop_status + nrhs + (8 - nrhs % 8) % 8;
    c.record = reinterpret_cast<stop_record*>(c.dev_flags + 16);
    c.check_every = check_every < 1 ? 1 : check_every;
    const stop_record init{-1, 0, 0};
    if (int err = static_cast<int>(hipMemcpyAsync(c.record, &init, sizeof(init), hipMemcpyHostToDevice, c.stream))) return err;
    if (int err = static_cast<int>(hipStreamSynchronize(c.stream))) return err;  // `init` is a stack object
    c.red = ws + l.red;
    c.red_bytes = gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8;
    return 0;
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

// ---- kernel entry points ---------------------------------------------------------
extern "C" int gkomi_bicgstab_initialize_f64(
    gkomi_stream_t s, int64_t n, int64_t nrhs, const double* b, int64_t b_stride, double* r,
    int64_t r_stride, double* rr, int64_t rr_stride, double* y, int64_t y_stride, double* sv,
    int64_t s_stride, double* t, int64_t t_stride, double* z, int64_t z_stride, double* v,
    int64_t v_stride, double* p, int64_t p_stride, double* prev_rho, double* rho, double* alpha,
    double* beta, double* gamma, double* omega, uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(bicgstab_initialize_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n,
                       nrhs, b, b_stride, r, r_stride, rr, rr_stride, y, y_stride, sv, s_stride, t,
                       t_stride, z, z_stride, v, v_stride, p, p_stride, prev_rho, rho, alpha, beta,
                       gamma, omega, stop_status);
    return check_launch();
}

extern "C" int gkomi_bicgstab_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* r,
                                         int64_t r_stride, double* p, int64_t p_stride,
                                         const double* v, int64_t v_stride, const double* rho,
                                         const double* prev_rho, const double* alpha,
                                         const double* omega, const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(bicgstab_step_1_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n,
                       nrhs, r, r_stride, p, p_stride, v, v_stride, rho, prev_rho, alpha, omega,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_bicgstab_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* r,
                                         int64_t r_stride, double* sv, int64_t s_stride,
                                         const double* v, int64_t v_stride, const double* rho,
                                         double* alpha, const double* beta,
                                         const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;  // the reference sets alpha inside the row loop
    hipLaunchKernelGGL(bicgstab_step_2_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n,
                       nrhs, r, r_stride, sv, s_stride, v, v_stride, rho, alpha, beta, stop_status);
    return check_launch();
}

extern "C" int gkomi_bicgstab_step_3_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* x,
                                         int64_t x_stride, double* r, int64_t r_stride,
                                         const double* sv, int64_t s_stride, const double* t,
                                         int64_t t_stride, const double* y, int64_t y_stride,
                                         const double* z, int64_t z_stride, const double* alpha,
                                         const double* beta, const double* gamma, double* omega,
                                         const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    if (n == 0) {
        hipLaunchKernelGGL(bicgstab_omega_kernel, dim3(1), dim3(block), 0, to_stream(s), nrhs, omega,
                           gamma, beta, stop_status);
        return check_launch();
    }
    hipLaunchKernelGGL(bicgstab_step_3_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n,
                       nrhs, x, x_stride, r, r_stride, sv, s_stride, t, t_stride, y, y_stride, z,
                       z_stride, alpha, beta, gamma, omega, stop_status);
    return check_launch();
}

extern "C" int gkomi_bicgstab_finalize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* x,
                                           int64_t x_stride, const double* y, int64_t y_stride,
                                           const double* alpha, uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipStream_t stream = to_stream(s);
    if (n > 0) {
        hipLaunchKernelGGL(bicgstab_finalize_kernel, grid_of(n, nrhs), dim3(block), 0, stream, n,
                           nrhs, x, x_stride, y, y_stride, alpha, stop_status);
    }
    // the reference finalizes inside the row loop: nothing happens for n == 0
    hipLaunchKernelGGL(finalize_status_kernel, dim3(static_cast<unsigned>(ceildiv(nrhs, block))),
                       dim3(block), 0, stream, nrhs, n > 0, stop_status);
    return check_launch();
}

extern "C" int gkomi_fcg_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* b,
                                        int64_t b_stride, double* r, int64_t r_stride, double* z,
                                        int64_t z_stride, double* p, int64_t p_stride, double* q,
                                        int64_t q_stride, double* t, int64_t t_stride,
                                        double* prev_rho, double* rho, double* rho_t,
                                        uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(fcg_initialize_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs,
                       b, b_stride, r, r_stride, z, z_stride, p, p_stride, q, q_stride, t, t_stride,
                       prev_rho, rho, rho_t, stop_status);
    return check_launch();
}

extern "C" int gkomi_fcg_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* p,
                                    int64_t p_stride, const double* z, int64_t z_stride,
                                    const double* rho_t, const double* prev_rho,
                                    const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(fcg_step_1_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, p,
                       p_stride, z, z_stride, rho_t, prev_rho, stop_status);
    return check_launch();
}

extern "C" int gkomi_fcg_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* x,
                                    int64_t x_stride, double* r, int64_t r_stride, double* t,
                                    int64_t t_stride, const double* p, int64_t p_stride,
                                    const double* q, int64_t q_stride, const double* beta,
                                    const double* rho, const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(fcg_step_2_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, x,
                       x_stride, r, r_stride, t, t_stride, p, p_stride, q, q_stride, beta, rho,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_cgs_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* b,
                                        int64_t b_stride, double* r, int64_t r_stride,
                                        double* r_tld, int64_t r_tld_stride, double* p,
                                        int64_t p_stride, double* q, int64_t q_stride, double* u,
                                        int64_t u_stride, double* u_hat, int64_t u_hat_stride,
                                        double* v_hat, int64_t v_hat_stride, double* t,
                                        int64_t t_stride, double* alpha, double* beta,
                                        double* gamma, double* prev_rho, double* rho,
                                        uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(cgs_initialize_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs,
                       b, b_stride, r, r_stride, r_tld, r_tld_stride, p, p_stride, q, q_stride, u,
                       u_stride, u_hat, u_hat_stride, v_hat, v_hat_stride, t, t_stride, alpha, beta,
                       gamma, prev_rho, rho, stop_status);
    return check_launch();
}

extern "C" int gkomi_cgs_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* r,
                                    int64_t r_stride, double* u, int64_t u_stride, double* p,
                                    int64_t p_stride, const double* q, int64_t q_stride,
                                    double* beta, const double* rho, const double* prev_rho,
                                    const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    if (n == 0) {
        hipLaunchKernelGGL(cgs_scalar_kernel, dim3(1), dim3(block), 0, to_stream(s), nrhs, beta, rho,
                           prev_rho, stop_status);
        return check_launch();
    }
    hipLaunchKernelGGL(cgs_step_1_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, r,
                       r_stride, u, u_stride, p, p_stride, q, q_stride, beta, rho, prev_rho,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_cgs_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* u,
                                    int64_t u_stride, const double* v_hat, int64_t v_hat_stride,
                                    double* q, int64_t q_stride, double* t, int64_t t_stride,
                                    double* alpha, const double* rho, const double* gamma,
                                    const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    if (n == 0) {
        hipLaunchKernelGGL(cgs_scalar_kernel, dim3(1), dim3(block), 0, to_stream(s), nrhs, alpha, rho,
                           gamma, stop_status);
        return check_launch();
    }
    hipLaunchKernelGGL(cgs_step_2_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, u,
                       u_stride, v_hat, v_hat_stride, q, q_stride, t, t_stride, alpha, rho, gamma,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_cgs_step_3_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* t,
                                    int64_t t_stride, const double* u_hat, int64_t u_hat_stride,
                                    double* r, int64_t r_stride, double* x, int64_t x_stride,
                                    const double* alpha, const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(cgs_step_3_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, t,
                       t_stride, u_hat, u_hat_stride, r, r_stride, x, x_stride, alpha, stop_status);
    return check_launch();
}

// ---- drivers ------------------------------------------------------------------------
extern "C" size_t gkomi_krylov_workspace_bytes(int64_t n, int64_t nrhs)
{
    if (n < 0 || nrhs <= 0) return 0;
    return make_solver_layout(n, nrhs, 8).total;
}

#define GKOMI_DRIVER_PROLOGUE(NVEC)                                                              \
    if (n < 0 || nrhs <= 0) return GKOMI_EINVAL;                                                 \
    const solver_layout l = make_solver_layout(n, nrhs, 8);                                      \
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;              \
    char* ws = static_cast<char*>(workspace);                                                    \
    driver_common c;                                                                             \
    double* sc = nullptr;                                                                        \
    GKOMI_TRY(make_common(c, s, n, nrhs, A, precond, precond_ctx, max_iters, reduction_factor,   \
                          baseline, check_every, ws, l, &sc));                                   \
    auto V = [&](int k) { return reinterpret_cast<double*>(ws + l.vec[k]); }

namespace {
int bicgstab_solve_impl(gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A_, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * nrhs * 10);
    GKOMI_DRIVER_PROLOGUE(8);
    double *r = V(0), *z = V(1), *y = V(2), *v = V(3), *sv = V(4), *t = V(5), *p = V(6), *rr = V(7);
    double *alpha = sc, *beta = sc + nrhs, *gamma = sc + 2 * nrhs, *prev_rho = sc + 3 * nrhs,
           *rho = sc + 4 * nrhs, *omega = sc + 5 * nrhs;
    GKOMI_TRY(gkomi_bicgstab_initialize_f64(s, n, nrhs, b, nrhs, r, nrhs, rr, nrhs, y, nrhs, sv, nrhs,
                                            t, nrhs, z, nrhs, v, nrhs, p, nrhs, prev_rho, rho, alpha,
                                            beta, gamma, omega, c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, r, nrhs, rr, nrhs));
    int64_t iter = -1;
    while (true) {
        ++iter;
        GKOMI_TRY(c.dot(rr, r, rho));
        bool stop = false;
        GKOMI_TRY(c.check(iter, r, true, 1, &stop));
        if (stop) break;
        GKOMI_TRY(gkomi_bicgstab_step_1_f64(s, n, nrhs, r, nrhs, p, nrhs, v, nrhs, rho, prev_rho,
                                            alpha, omega, c.stop_status));
        GKOMI_TRY(c.apply_precond(p, y));
        GKOMI_TRY(c.spmv(y, v));
        GKOMI_TRY(c.dot(rr, v, beta));
        GKOMI_TRY(gkomi_bicgstab_step_2_f64(s, n, nrhs, r, nrhs, sv, nrhs, v, nrhs, rho, alpha, beta,
                                            c.stop_status));
        GKOMI_TRY(c.check(iter, sv, false, 2, &stop));
        // the reference finalizes "if (one_changed)"; the kernel only touches
        // columns that stopped without being finalized, so calling it always
        // is the same thing without asking the host
        GKOMI_TRY(gkomi_bicgstab_finalize_f64(s, n, nrhs, x, nrhs, y, nrhs, alpha, c.stop_status));
        if (stop) break;
        GKOMI_TRY(c.apply_precond(sv, z));
        GKOMI_TRY(c.spmv(z, t));
        GKOMI_TRY(c.dot(sv, t, gamma));
        GKOMI_TRY(c.dot(t, t, beta));
        GKOMI_TRY(gkomi_bicgstab_step_3_f64(s, n, nrhs, x, nrhs, r, nrhs, sv, nrhs, t, nrhs, y, nrhs,
                                            z, nrhs, alpha, beta, gamma, omega, c.stop_status));
        std::swap(prev_rho, rho);
    }
    return c.finish(c.stop_iter(), c.host_record.phase == 2 ? sv : r, host_info);
}


// ---- fused BiCGSTAB, one right-hand side --------------------------------------------
//
// The reference sequence above costs 25 launches per iteration (two-stage dots,
// the criterion, the step kernels).  The fused driver keeps the same recurrences
// and check points in 5 launches (+ the preconditioner's):
//   KA  re-adds the partials of rho = rr.r and |r|^2 left by KE, evaluates the
//       criterion on r (phase 1) and -- unless stopped -- updates p (step_1)
//   KB  v = A y with the partials of rr.v in the SpMV epilogue
//   KC  re-adds them: alpha, s = r - alpha v (step_2), partials of |s|^2
//   KD  t = A z with the partials of s.t and t.t in the epilogue
//   KE  first re-adds the partials of |s|^2 and evaluates the criterion on s (phase 2; a launch of its own until
//       round 3: 4.8 us per iteration): if it fires, x += alpha y (finalize) and nothing else -- KD ran once for
//       nothing; otherwise re-adds KD's: omega, x += alpha y + omega z, r = s - omega t (step_3), partials of rr.r
//       and |r|^2 for the next KA
// Every workgroup re-adds the partials in the same order, so all agree on the
// scalars bit for bit; workgroup 0 stores them for the kernels that follow.
// Once a criterion fires the remaining launches return at once, so x and the
// iteration count are those of the stopping iteration whatever `check_every`.
constexpr int fblock = 1024;
constexpr int fused_max_parts = 1024;

struct bicgstab_scalars {
    double rho[2];  // rho of iteration `it` lives in rho[it & 1]
    double alpha, omega, tau, orig_tau;
    long long stop_iter;
    long long stop2_iter;   // iteration whose half step converged (written by KE only), -1 before
    int phase;
    unsigned char status;   // written by KA only
    unsigned char status2;  // written by KE only (the half step's criterion)
    unsigned char pad[2];
};

__device__ __forceinline__ double sum_partials_f(const double* __restrict__ part, int nparts,
                                                 double* smem)
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

__device__ __forceinline__ bool fused_stopped(const bicgstab_scalars* scal)
{
    return status_has_stopped(scal->status) || status_has_stopped(scal->status2);
}

__global__ void bicgstab_fused_init_kernel(bicgstab_scalars* scal, const double* orig_tau)
{
    scal->rho[0] = 1.0;
    scal->rho[1] = 1.0;  // prev_rho = rho = alpha = omega = 1 (bicgstab::initialize)
    scal->alpha = 1.0;
    scal->omega = 1.0;
    scal->tau = 0.0;
    scal->orig_tau = orig_tau[0];
    scal->stop_iter = -1;
    scal->stop2_iter = -1;
    scal->phase = 0;
    scal->status = 0;
    scal->status2 = 0;
}

// pa[block] = sum a*b, pb[block] = sum c*c over the workgroup's share (pb optional)
__global__ __launch_bounds__(fblock) void fused_dot2_partials_kernel(
    int64_t n, const double* __restrict__ a, const double* __restrict__ b,
    const double* __restrict__ c, const bicgstab_scalars* scal, double* __restrict__ pa,
    double* __restrict__ pb)
{
    __shared__ double smem[fblock / wave_size];
    if (scal != nullptr && fused_stopped(scal)) return;
    const int64_t step = static_cast<int64_t>(gridDim.x) * fblock;
    double u = 0.0, w = 0.0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x; i < n; i += step) {
        u += a[i] * b[i];
        if (pb != nullptr) {
            const double cv = c[i];
            w += cv * cv;
        }
    }
    const double tu = block_reduce_sum<fblock>(u, smem);
    __syncthreads();
    const double tw = block_reduce_sum<fblock>(w, smem);
    if (threadIdx.x == 0) {
        pa[blockIdx.x] = tu;
        if (pb != nullptr) pb[blockIdx.x] = tw;
    }
}

// The vector kernels move 16 B per lane and issue the loads of their first
// sweep (which do not depend on the scalars) before re-adding the partials, so
// the reduction's latency hides behind them (as in cg_solver.hip).
struct pair_sweep {
    int64_t n2, step, i0;
    __device__ pair_sweep(int64_t n)
        : n2(n / 2), step(static_cast<int64_t>(gridDim.x) * fblock),
          i0(blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x)
    {}
    __device__ bool first() const { return i0 < n2; }
    __device__ bool tail(int64_t n) const { return (n & 1) && blockIdx.x == 0 && threadIdx.x == 0; }
};

__device__ __forceinline__ double2 ld2(const double* p, int64_t i)
{
    return reinterpret_cast<const double2*>(p)[i];
}
__device__ __forceinline__ void st2(double* p, int64_t i, double2 v)
{
    reinterpret_cast<double2*>(p)[i] = v;
}

// KA
__global__ __launch_bounds__(fblock) void bicgstab_fused_step1_kernel(
    int64_t n, const double* __restrict__ r, double* __restrict__ p, const double* __restrict__ v,
    const double* __restrict__ rho_part, const double* __restrict__ tau_part, int nparts,
    bicgstab_scalars* scal, long long it, long long max_iters, double goal, host_watch_line* watch = nullptr)
{
    __shared__ double smem[fblock / wave_size];
    if (fused_stopped(scal)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, scal->stop_iter);  // internal.hpp
        return;
    }
    const pair_sweep sw(n);
    double2 r0 = make_double2(0.0, 0.0), p0 = r0, v0 = r0;
    if (sw.first()) {
        r0 = ld2(r, sw.i0);
        p0 = ld2(p, sw.i0);
        v0 = ld2(v, sw.i0);
    }
    const double rho = sum_partials_f(rho_part, nparts, smem);
    const double tau = sqrt(sum_partials_f(tau_part, nparts, smem));
    uint8_t st = 0;
    if (it >= max_iters) {
        st = 1 | GKOMI_STATUS_FINALIZED;
    } else if (tau < goal * scal->orig_tau) {
        st = GKOMI_STATUS_CONVERGED | 1 | GKOMI_STATUS_FINALIZED;
    }
    const double prev = scal->rho[(it + 1) & 1];
    const double alpha = scal->alpha, omega = scal->omega;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rho[it & 1] = rho;
        scal->tau = tau;  // also at the iteration limit: the norm of the residual that is returned
        if (st) {
            scal->stop_iter = it;
            scal->phase = 1;
            scal->status = st;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, st ? it : -1ll);
    if (st) return;
    // bicgstab::step_1 (reference/solver/bicgstab_kernels.cpp)
    const bool update = prev * omega != 0.0;
    const double tmp = update ? rho / prev * alpha / omega : 0.0;
    auto step1 = [&](double rv, double pv, double vv) { return update ? rv + tmp * (pv - omega * vv) : rv; };
    if (sw.first()) st2(p, sw.i0, make_double2(step1(r0.x, p0.x, v0.x), step1(r0.y, p0.y, v0.y)));
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        const double2 rv = ld2(r, i), pv = ld2(p, i), vv = ld2(v, i);
        st2(p, i, make_double2(step1(rv.x, pv.x, vv.x), step1(rv.y, pv.y, vv.y)));
    }
    if (sw.tail(n)) p[n - 1] = step1(r[n - 1], p[n - 1], v[n - 1]);
}

// KC
__global__ __launch_bounds__(fblock) void bicgstab_fused_step2_kernel(
    int64_t n, const double* __restrict__ r, double* __restrict__ sv, const double* __restrict__ v,
    const double* __restrict__ beta_part, int nparts, bicgstab_scalars* scal, long long it,
    double* __restrict__ ss_part)
{
    __shared__ double smem[fblock / wave_size];
    if (fused_stopped(scal)) return;
    const pair_sweep sw(n);
    double2 r0 = make_double2(0.0, 0.0), v0 = r0;
    if (sw.first()) {
        r0 = ld2(r, sw.i0);
        v0 = ld2(v, sw.i0);
    }
    const double beta = sum_partials_f(beta_part, nparts, smem);
    const double rho = scal->rho[it & 1];
    const bool update = beta != 0.0;
    const double alpha = update ? rho / beta : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal->alpha = alpha;
    // bicgstab::step_2
    auto step2 = [&](double rv, double vv) { return update ? rv - alpha * vv : rv; };
    double a0 = 0.0, a1 = 0.0;
    if (sw.first()) {
        const double2 o = make_double2(step2(r0.x, v0.x), step2(r0.y, v0.y));
        st2(sv, sw.i0, o);
        a0 += o.x * o.x;
        a1 += o.y * o.y;
    }
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        const double2 rv = ld2(r, i), vv = ld2(v, i);
        const double2 o = make_double2(step2(rv.x, vv.x), step2(rv.y, vv.y));
        st2(sv, i, o);
        a0 += o.x * o.x;
        a1 += o.y * o.y;
    }
    if (sw.tail(n)) {
        const double o = step2(r[n - 1], v[n - 1]);
        sv[n - 1] = o;
        a0 += o * o;
    }
    __syncthreads();
    const double total = block_reduce_sum<fblock>(a0 + a1, smem);
    if (threadIdx.x == 0) ss_part[blockIdx.x] = total;
}

// KE
__global__ __launch_bounds__(fblock) void bicgstab_fused_step3_kernel(
    int64_t n, double* __restrict__ x, double* __restrict__ r, const double* __restrict__ sv,
    const double* __restrict__ t, const double* __restrict__ y, const double* __restrict__ z,
    const double* __restrict__ rr, const double* __restrict__ gamma_part,
    const double* __restrict__ tt_part, int nparts, bicgstab_scalars* scal,
    double* __restrict__ rho_part, double* __restrict__ tau_part,
    const double* __restrict__ ss_part, int nss, long long it, double goal)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) return;
    // the half step's criterion (KD0 of round 1 was a launch of its own: 4.8 us per iteration for a kernel that moves
    // data once per solve).  status2 / stop2_iter are written here: workgroup 0 may be writing them while a late
    // workgroup of the same launch starts, so "stopped in an earlier launch" is read from the one 8-byte word
    const long long stopped_at = scal->stop2_iter;
    if (stopped_at >= 0 && stopped_at != it) return;
    const pair_sweep sw(n);
    double2 x0 = make_double2(0.0, 0.0), y0 = x0, z0 = x0, s0 = x0, t0 = x0, q0 = x0;
    if (sw.first()) {
        x0 = ld2(x, sw.i0);
        y0 = ld2(y, sw.i0);
        z0 = ld2(z, sw.i0);
        s0 = ld2(sv, sw.i0);
        t0 = ld2(t, sw.i0);
        q0 = ld2(rr, sw.i0);
    }
    const double tau_s = sqrt(sum_partials_f(ss_part, nss, smem));
    if (tau_s < goal * scal->orig_tau) {
        // bicgstab::finalize (core/solver/bicgstab.cpp:196-203): s has converged, x += alpha y and nothing else
        // (the preconditioner apply and the SpMV between step 2 and here ran on that s for nothing, once per solve)
        const double alpha = scal->alpha;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal->tau = tau_s;
            scal->stop_iter = it;
            scal->stop2_iter = it;
            scal->phase = 2;
            scal->status2 = GKOMI_STATUS_CONVERGED | 1 | GKOMI_STATUS_FINALIZED;
        }
        if (sw.first()) st2(x, sw.i0, make_double2(x0.x + alpha * y0.x, x0.y + alpha * y0.y));
        for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
            const double2 xv = ld2(x, i), yv = ld2(y, i);
            st2(x, i, make_double2(xv.x + alpha * yv.x, xv.y + alpha * yv.y));
        }
        if (sw.tail(n)) x[n - 1] += alpha * y[n - 1];
        return;
    }
    const double gamma = sum_partials_f(gamma_part, nparts, smem);
    const double beta = sum_partials_f(tt_part, nparts, smem);
    const double omega = beta != 0.0 ? gamma / beta : 0.0;
    const double alpha = scal->alpha;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal->omega = omega;
    // bicgstab::step_3
    double a0 = 0.0, a1 = 0.0;
    auto step3 = [&](double xv, double yv, double zv, double svv, double tv, double qv, double* xo,
                     double* ro) {
        *xo = xv + (alpha * yv + omega * zv);
        *ro = svv - omega * tv;
        a0 += qv * *ro;
        a1 += *ro * *ro;
    };
    if (sw.first()) {
        double2 xo, ro;
        step3(x0.x, y0.x, z0.x, s0.x, t0.x, q0.x, &xo.x, &ro.x);
        step3(x0.y, y0.y, z0.y, s0.y, t0.y, q0.y, &xo.y, &ro.y);
        st2(x, sw.i0, xo);
        st2(r, sw.i0, ro);
    }
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        const double2 xv = ld2(x, i), yv = ld2(y, i), zv = ld2(z, i), svv = ld2(sv, i), tv = ld2(t, i),
                      qv = ld2(rr, i);
        double2 xo, ro;
        step3(xv.x, yv.x, zv.x, svv.x, tv.x, qv.x, &xo.x, &ro.x);
        step3(xv.y, yv.y, zv.y, svv.y, tv.y, qv.y, &xo.y, &ro.y);
        st2(x, i, xo);
        st2(r, i, ro);
    }
    if (sw.tail(n)) {
        const int64_t i = n - 1;
        double xo, ro;
        step3(x[i], y[i], z[i], sv[i], t[i], rr[i], &xo, &ro);
        x[i] = xo;
        r[i] = ro;
    }
    __syncthreads();
    const double t0s = block_reduce_sum<fblock>(a0, smem);
    __syncthreads();
    const double t1s = block_reduce_sum<fblock>(a1, smem);
    if (threadIdx.x == 0) {
        rho_part[blockIdx.x] = t0s;
        tau_part[blockIdx.x] = t1s;
    }
}

int bicgstab_fused_impl(gkomi_stream_t s, int64_t n, const sysmat& A_, gkomi_apply_fn precond,
                        void* precond_ctx, const double* b, double* x, int64_t max_iters,
                        double reduction_factor, int baseline, int64_t check_every, void* workspace,
                        size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * 1 * 10);
    const int64_t nrhs = 1;
    if (n > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0) {  // the vector kernels move 16 B per lane
        return bicgstab_solve_impl(s, n, 1, A, precond, precond_ctx, b, x, max_iters, reduction_factor,
                                   baseline, check_every, workspace, workspace_bytes, host_info);
    }
    GKOMI_DRIVER_PROLOGUE(8);
    double *r = V(0), *z = V(1), *y = V(2), *v = V(3), *sv = V(4), *t = V(5), *p = V(6), *rr = V(7);
    GKOMI_TRY(gkomi_bicgstab_initialize_f64(s, n, 1, b, 1, r, 1, rr, 1, y, 1, sv, 1, t, 1, z, 1, v, 1, p,
                                            1, sc, sc + 1, sc + 2, sc + 3, sc + 4, sc + 5,
                                            c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    GKOMI_TRY(gkomi_dense_copy_f64(s, n, 1, r, 1, rr, 1));
    hipStream_t stream = c.stream;
    double* parts = reinterpret_cast<double*>(ws + l.parts);
    bicgstab_scalars* scal = reinterpret_cast<bicgstab_scalars*>(parts);
    double* part_rho = parts + 32;
    double* part_tau = part_rho + fused_max_parts;
    double* part_ss = part_tau + fused_max_parts;
    const size_t per_spmv = spmv_dot_partials_room(n);
    double* part_beta = part_ss + fused_max_parts;
    double* part_gamma = part_beta + per_spmv;
    double* part_tt = part_gamma + per_spmv;
    const int g = fused_vec_grid(n);  // 16 B per lane (internal.hpp)
    // the dots in the SpMV's epilogue for CSR / ELL / SELL-P (internal.hpp)
    const spmv_dot_plan spmv(A);
    const bool csr_epilogue = spmv.fused();
    const int nb = csr_epilogue ? spmv.num_partials : g;
    if (precond == nullptr) {  // Identity: y = p, z = s without the copies
        y = p;
        z = sv;
    }
    hipLaunchKernelGGL(bicgstab_fused_init_kernel, dim3(1), dim3(1), 0, stream, scal, c.orig_tau);
    hipLaunchKernelGGL(fused_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, rr, r, r,
                       static_cast<const bicgstab_scalars*>(nullptr), part_rho, part_tau);
    GKOMI_TRY(check_launch());
    // q = A in, partials of w.q (and q.q) -- in the SpMV epilogue when A is CSR
    auto spmv_dots = [&](const double* in, double* out, const double* w, double* pw, double* pq) {
        if (csr_epilogue) {
            return spmv.launch(stream, in, out, pw, &scal->status, w, pq);
        }
        GKOMI_TRY(A.apply(s, 1, nullptr, in, nullptr, out));
        hipLaunchKernelGGL(fused_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, w, out,
                           out, static_cast<const bicgstab_scalars*>(scal), pw, pq);
        return check_launch();
    };
    bicgstab_scalars h{};
    long long it = 0;
    bool done = false;
    // the host's view of the solve (host_watch, internal.hpp): the first kernel of every iteration reports the
    // iteration it evaluated; the host stays a few iterations ahead and looks at device memory only once the
    // criterion has fired -- no blocking look every check_every iterations
    host_watch watch;
    const long long lag = std::min<long long>(c.check_every, precond == nullptr ? 4 * host_watch_lag : host_watch_lag);
    while (!done) {
        bool last = false;
        for (int64_t k = 0; k < (watch.dev != nullptr ? 1 : c.check_every) && !done; ++k, ++it) {
            hipLaunchKernelGGL(bicgstab_fused_step1_kernel, dim3(g), dim3(fblock), 0, stream, n, r, p, v,
                               part_rho, part_tau, g, scal, it, static_cast<long long>(max_iters),
                               reduction_factor, watch.dev);
            if (it >= max_iters) {  // this launch stops for sure
                ++it;
                last = true;
                break;
            }
            if (precond != nullptr) GKOMI_TRY(precond(precond_ctx, s, p, y));
            GKOMI_TRY(spmv_dots(y, v, rr, part_beta, nullptr));
            hipLaunchKernelGGL(bicgstab_fused_step2_kernel, dim3(g), dim3(fblock), 0, stream, n, r, sv,
                               v, part_beta, nb, scal, it, part_ss);
            if (precond != nullptr) GKOMI_TRY(precond(precond_ctx, s, sv, z));
            GKOMI_TRY(spmv_dots(z, t, sv, part_gamma, part_tt));
            hipLaunchKernelGGL(bicgstab_fused_step3_kernel, dim3(g), dim3(fblock), 0, stream, n, x, r,
                               sv, t, y, z, rr, part_gamma, part_tt, nb, scal, part_rho, part_tau, part_ss, g, it,
                               reduction_factor);
        }
        GKOMI_TRY(check_launch());
        if (watch.dev != nullptr && !last) {
            if (it - 1 < lag) continue;
            if (watch.wait(stream, it - 1 - lag)) {
                if (watch.stop_iter() < 0) continue;  // still running: no look at device memory
            } else {
                watch.dev = nullptr;  // its stores do not reach this host: the blocking look from here on
            }
        }
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h, scal, sizeof(h), hipMemcpyDeviceToHost, stream)));
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
        done = h.stop_iter >= 0;
    }
    const unsigned char st = h.status | h.status2;
    if (host_info != nullptr) {
        host_info[0] = static_cast<double>(h.stop_iter);
        host_info[1] = (st & GKOMI_STATUS_CONVERGED) ? 1.0 : 0.0;
        host_info[2] = h.tau;
        host_info[3] = h.orig_tau;
    }
    return precond_status(precond, precond_ctx, s);
}

}  // namespace

extern "C" int gkomi_bicgstab_solve_fused_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    return bicgstab_fused_impl(s, n, make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy,
                                                     max_row_nnz_hint),
                               precond, precond_ctx, b, x, max_iters, reduction_factor, baseline,
                               check_every, workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_bicgstab_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return bicgstab_fused_impl(s, n, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                               max_iters, reduction_factor, baseline, check_every, workspace,
                               workspace_bytes, host_info);
}

extern "C" int gkomi_bicgstab_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    return bicgstab_solve_impl(s, n, nrhs,
                            make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint),
                            precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, check_every,
                            workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_bicgstab_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                        gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                        gkomi_apply_fn precond, void* precond_ctx, const double* b,
                                        double* x, int64_t max_iters, double reduction_factor,
                                        int baseline, int64_t check_every, void* workspace,
                                        size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return bicgstab_solve_impl(s, n, nrhs, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                            max_iters, reduction_factor, baseline, check_every, workspace, workspace_bytes,
                            host_info);
}

namespace {
int fcg_solve_impl(gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A_, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * nrhs * 7);
    GKOMI_DRIVER_PROLOGUE(5);
    double *r = V(0), *z = V(1), *p = V(2), *q = V(3), *t = V(4);
    double *beta = sc, *prev_rho = sc + nrhs, *rho = sc + 2 * nrhs, *rho_t = sc + 3 * nrhs;
    GKOMI_TRY(gkomi_fcg_initialize_f64(s, n, nrhs, b, nrhs, r, nrhs, z, nrhs, p, nrhs, q, nrhs, t,
                                       nrhs, prev_rho, rho, rho_t, c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    int64_t iter = -1;
    while (true) {
        GKOMI_TRY(c.apply_precond(r, z));
        GKOMI_TRY(c.dot(r, z, rho));
        GKOMI_TRY(c.dot(t, z, rho_t));
        ++iter;
        bool stop = false;
        GKOMI_TRY(c.check(iter, r, true, 1, &stop));
        if (stop) break;
        GKOMI_TRY(gkomi_fcg_step_1_f64(s, n, nrhs, p, nrhs, z, nrhs, rho_t, prev_rho, c.stop_status));
        GKOMI_TRY(c.spmv(p, q));
        GKOMI_TRY(c.dot(p, q, beta));
        GKOMI_TRY(gkomi_fcg_step_2_f64(s, n, nrhs, x, nrhs, r, nrhs, t, nrhs, p, nrhs, q, nrhs, beta,
                                       rho, c.stop_status));
        std::swap(prev_rho, rho);
    }
    return c.finish(c.stop_iter(), r, host_info);
}

}  // namespace

// ---- fused FCG, one right-hand side: 3 launches per iteration -----------------------
// (the structure of the fused CG of cg_solver.hip plus the vector t = r_new - r_old)
//   FA  re-adds the partials of rho = r.z, rho_t = t.z and |r|^2, evaluates the
//       criterion, p = z + (rho_t / prev_rho) p (fcg::step_1)
//   FB  q = A p with the p.q partials in the SpMV epilogue
//   FC  beta -> x += (rho/beta) p, r -= (rho/beta) q, t = r_new - r_old
//       (fcg::step_2) and, without a preconditioner (z = r), the partials for
//       the next FA; with one: z = M r, then a three-dot partials kernel
namespace {

struct fcg_scalars {
    double rho[2];  // rho of iteration `it` lives in rho[it & 1]
    double tau, orig_tau;
    long long stop_iter;
    unsigned char status;
    unsigned char pad[7];
};

__global__ void fcg_fused_init_kernel(fcg_scalars* scal, const double* orig_tau)
{
    scal->rho[0] = 0.0;
    scal->rho[1] = 1.0;  // prev_rho = 1 (fcg::initialize)
    scal->tau = 0.0;
    scal->orig_tau = orig_tau[0];
    scal->stop_iter = -1;
    scal->status = 0;
}

// p0[b] = sum r*z, p1[b] = sum t*z, p2[b] = sum r*r
__global__ __launch_bounds__(fblock) void fused_dot3_partials_kernel(
    int64_t n, const double* __restrict__ r, const double* __restrict__ z, const double* __restrict__ t,
    const unsigned char* status, double* __restrict__ p0, double* __restrict__ p1,
    double* __restrict__ p2)
{
    __shared__ double smem[fblock / wave_size];
    if (status != nullptr && status_has_stopped(status[0])) return;
    const int64_t step = static_cast<int64_t>(gridDim.x) * fblock;
    double a = 0.0, b = 0.0, c = 0.0;
    const bool vec = ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(z) |
                       reinterpret_cast<uintptr_t>(t)) & 15) == 0;
    if (vec) {  // 16 B per lane
        const int64_t n2 = n / 2;
        double a1 = 0.0, b1 = 0.0, c1 = 0.0;
        for (int64_t i = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x; i < n2; i += step) {
            const double2 rv = ld2(r, i), zv = ld2(z, i), tv = ld2(t, i);
            a += rv.x * zv.x;
            a1 += rv.y * zv.y;
            b += tv.x * zv.x;
            b1 += tv.y * zv.y;
            c += rv.x * rv.x;
            c1 += rv.y * rv.y;
        }
        a += a1;
        b += b1;
        c += c1;
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            a += r[n - 1] * z[n - 1];
            b += t[n - 1] * z[n - 1];
            c += r[n - 1] * r[n - 1];
        }
    } else {
        for (int64_t i = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x; i < n; i += step) {
            const double rv = r[i], zv = z[i];
            a += rv * zv;
            b += t[i] * zv;
            c += rv * rv;
        }
    }
    const double ta = block_reduce_sum<fblock>(a, smem);
    __syncthreads();
    const double tb = block_reduce_sum<fblock>(b, smem);
    __syncthreads();
    const double tc = block_reduce_sum<fblock>(c, smem);
    if (threadIdx.x == 0) {
        p0[blockIdx.x] = ta;
        p1[blockIdx.x] = tb;
        p2[blockIdx.x] = tc;
    }
}

// FA.  z may alias r (Identity).
__global__ __launch_bounds__(fblock) void fcg_fused_step1_kernel(
    int64_t n, double* __restrict__ p, const double* __restrict__ z,
    const double* __restrict__ rho_part, const double* __restrict__ rhot_part,
    const double* __restrict__ tau_part, int nparts, fcg_scalars* scal, long long it,
    long long max_iters, double goal, host_watch_line* watch = nullptr)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, scal->stop_iter);  // internal.hpp
        return;
    }
    const pair_sweep sw(n);
    double2 z0 = make_double2(0.0, 0.0), p0 = z0;
    if (sw.first()) {
        z0 = ld2(z, sw.i0);
        p0 = ld2(p, sw.i0);
    }
    const double rho = sum_partials_f(rho_part, nparts, smem);
    const double rho_t = sum_partials_f(rhot_part, nparts, smem);
    const double tau = sqrt(rho_part == tau_part ? rho : sum_partials_f(tau_part, nparts, smem));
    uint8_t st = 0;
    if (it >= max_iters) {
        st = 1 | GKOMI_STATUS_FINALIZED;
    } else if (tau < goal * scal->orig_tau) {
        st = GKOMI_STATUS_CONVERGED | 1 | GKOMI_STATUS_FINALIZED;
    }
    const double prev = scal->rho[(it + 1) & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rho[it & 1] = rho;
        scal->tau = tau;  // also at the iteration limit: the norm of the residual that is returned
        if (st) {
            scal->stop_iter = it;
            scal->status = st;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, st ? it : -1ll);
    if (st) return;
    const bool restart = prev == 0.0;
    const double tmp = restart ? 0.0 : rho_t / prev;
    auto step1 = [&](double zv, double pv) { return restart ? zv : zv + tmp * pv; };
    if (sw.first()) st2(p, sw.i0, make_double2(step1(z0.x, p0.x), step1(z0.y, p0.y)));
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        const double2 zv = ld2(z, i), pv = ld2(p, i);
        st2(p, i, make_double2(step1(zv.x, pv.x), step1(zv.y, pv.y)));
    }
    if (sw.tail(n)) p[n - 1] = step1(z[n - 1], p[n - 1]);
}

// FC.  With partials != nullptr (Identity): leaves r.r and t.r of the new r, t.
__global__ __launch_bounds__(fblock) void fcg_fused_step2_kernel(
    int64_t n, double* __restrict__ x, double* __restrict__ r, double* __restrict__ t,
    const double* __restrict__ p, const double* __restrict__ q,
    const double* __restrict__ beta_part, int nparts, fcg_scalars* scal, long long it,
    double* __restrict__ rr_part, double* __restrict__ tr_part)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) return;
    const pair_sweep sw(n);
    double2 x0 = make_double2(0.0, 0.0), r0 = x0, p0 = x0, q0 = x0, t0 = x0;
    if (sw.first()) {
        x0 = ld2(x, sw.i0);
        r0 = ld2(r, sw.i0);
        p0 = ld2(p, sw.i0);
        q0 = ld2(q, sw.i0);
        t0 = ld2(t, sw.i0);
    }
    const double beta = sum_partials_f(beta_part, nparts, smem);
    const double rho = scal->rho[it & 1];
    const bool update = beta != 0.0;
    const double tmp = update ? rho / beta : 0.0;
    double a0 = 0.0, a1 = 0.0;
    // fcg::step_2; returns through the references, accumulates r.r and t.r
    auto step2 = [&](double& xv, double& rv, double& tv, double pv, double qv) {
        if (update) {
            const double prev_r = rv;
            xv += tmp * pv;
            rv = prev_r - tmp * qv;
            tv = rv - prev_r;
        }
        a0 += rv * rv;
        a1 += tv * rv;
    };
    if (sw.first()) {
        step2(x0.x, r0.x, t0.x, p0.x, q0.x);
        step2(x0.y, r0.y, t0.y, p0.y, q0.y);
        if (update) {
            st2(x, sw.i0, x0);
            st2(r, sw.i0, r0);
            st2(t, sw.i0, t0);
        }
    }
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        double2 xv = ld2(x, i), rv = ld2(r, i), tv = ld2(t, i);
        const double2 pv = ld2(p, i), qv = ld2(q, i);
        step2(xv.x, rv.x, tv.x, pv.x, qv.x);
        step2(xv.y, rv.y, tv.y, pv.y, qv.y);
        if (update) {
            st2(x, i, xv);
            st2(r, i, rv);
            st2(t, i, tv);
        }
    }
    if (sw.tail(n)) {
        const int64_t i = n - 1;
        double xv = x[i], rv = r[i], tv = t[i];
        step2(xv, rv, tv, p[i], q[i]);
        if (update) {
            x[i] = xv;
            r[i] = rv;
            t[i] = tv;
        }
    }
    if (rr_part == nullptr) return;
    __syncthreads();
    const double s0 = block_reduce_sum<fblock>(a0, smem);
    __syncthreads();
    const double s1 = block_reduce_sum<fblock>(a1, smem);
    if (threadIdx.x == 0) {
        rr_part[blockIdx.x] = s0;
        tr_part[blockIdx.x] = s1;
    }
}

int fcg_fused_impl(gkomi_stream_t s, int64_t n, const sysmat& A_, gkomi_apply_fn precond,
                   void* precond_ctx, const double* b, double* x, int64_t max_iters,
                   double reduction_factor, int baseline, int64_t check_every, void* workspace,
                   size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * 1 * 7);
    const int64_t nrhs = 1;
    if (n > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0) {
        return fcg_solve_impl(s, n, 1, A, precond, precond_ctx, b, x, max_iters, reduction_factor, baseline,
                              check_every, workspace, workspace_bytes, host_info);
    }
    GKOMI_DRIVER_PROLOGUE(5);
    double *r = V(0), *z = V(1), *p = V(2), *q = V(3), *t = V(4);
    GKOMI_TRY(gkomi_fcg_initialize_f64(s, n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, t, 1, sc + 1, sc + 2, sc + 3,
                                       c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    // t = b from initialize, exactly like the reference (fcg.cpp:137 does not refresh it after r = b - A x)
    hipStream_t stream = c.stream;
    double* parts = reinterpret_cast<double*>(ws + l.parts);
    fcg_scalars* scal = reinterpret_cast<fcg_scalars*>(parts);
    double* part_rho = parts + 32;
    double* part_rhot = part_rho + fused_max_parts;
    double* part_tau = part_rhot + fused_max_parts;
    double* part_beta = part_tau + fused_max_parts;
    const size_t per_spmv = spmv_dot_partials_room(n);  // >= g: room for three arrays (layout)
    const int g = fused_vec_grid(n);
    // the dots in the SpMV's epilogue for CSR / ELL / SELL-P (internal.hpp)
    const spmv_dot_plan spmv(A);
    const bool csr_epilogue = spmv.fused();
    const int nb = csr_epilogue ? spmv.num_partials : g;
    const bool identity = precond == nullptr;
    if (identity) z = r;
    hipLaunchKernelGGL(fcg_fused_init_kernel, dim3(1), dim3(1), 0, stream, scal, c.orig_tau);
    if (!identity) GKOMI_TRY(precond(precond_ctx, s, r, z));
    hipLaunchKernelGGL(fused_dot3_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, z, t,
                       static_cast<const unsigned char*>(nullptr), part_rho, part_rhot, part_tau);
    GKOMI_TRY(check_launch());
    fcg_scalars h{};
    long long it = 0;
    bool done = false;
    // the host's view of the solve (host_watch, internal.hpp): the first kernel of every iteration reports the
    // iteration it evaluated; the host stays a few iterations ahead and looks at device memory only once the
    // criterion has fired -- no blocking look every check_every iterations
    host_watch watch;
    const long long lag = std::min<long long>(c.check_every, precond == nullptr ? 4 * host_watch_lag : host_watch_lag);
    while (!done) {
        bool last = false;
        for (int64_t k = 0; k < (watch.dev != nullptr ? 1 : c.check_every) && !done; ++k, ++it) {
            hipLaunchKernelGGL(fcg_fused_step1_kernel, dim3(g), dim3(fblock), 0, stream, n, p, z, part_rho,
                               part_rhot, identity ? part_rho : part_tau, g, scal, it,
                               static_cast<long long>(max_iters), reduction_factor, watch.dev);
            if (it >= max_iters) {
                ++it;
                last = true;
                break;
            }
            if (csr_epilogue) {
                GKOMI_TRY(spmv.launch(stream, p, q, part_beta, &scal->status));
            } else {
                GKOMI_TRY(A.apply(s, 1, nullptr, p, nullptr, q));
                // only p.q is wanted: the other two sums land in scratch
                hipLaunchKernelGGL(fused_dot3_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, p, q, p,
                                   &scal->status, part_beta, part_beta + per_spmv, part_beta + 2 * per_spmv);
            }
            hipLaunchKernelGGL(fcg_fused_step2_kernel, dim3(g), dim3(fblock), 0, stream, n, x, r, t, p, q,
                               part_beta, nb, scal, it, identity ? part_rho : static_cast<double*>(nullptr),
                               identity ? part_rhot : static_cast<double*>(nullptr));
            if (!identity) {
                GKOMI_TRY(precond(precond_ctx, s, r, z));
                hipLaunchKernelGGL(fused_dot3_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, z, t,
                                   &scal->status, part_rho, part_rhot, part_tau);
            }
        }
        GKOMI_TRY(check_launch());
        if (watch.dev != nullptr && !last) {
            if (it - 1 < lag) continue;
            if (watch.wait(stream, it - 1 - lag)) {
                if (watch.stop_iter() < 0) continue;  // still running: no look at device memory
            } else {
                watch.dev = nullptr;  // its stores do not reach this host: the blocking look from here on
            }
        }
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h, scal, sizeof(h), hipMemcpyDeviceToHost, stream)));
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
        done = h.stop_iter >= 0;
    }
    if (host_info != nullptr) {
        host_info[0] = static_cast<double>(h.stop_iter);
        host_info[1] = (h.status & GKOMI_STATUS_CONVERGED) ? 1.0 : 0.0;
        host_info[2] = h.tau;
        host_info[3] = h.orig_tau;
    }
    return precond_status(precond, precond_ctx, s);
}

}  // namespace

extern "C" int gkomi_fcg_solve_fused_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    return fcg_fused_impl(s, n, make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy,
                                                max_row_nnz_hint),
                          precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, check_every,
                          workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_fcg_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return fcg_fused_impl(s, n, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x, max_iters,
                          reduction_factor, baseline, check_every, workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_fcg_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    return fcg_solve_impl(s, n, nrhs,
                            make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint),
                            precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, check_every,
                            workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_fcg_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                        gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                        gkomi_apply_fn precond, void* precond_ctx, const double* b,
                                        double* x, int64_t max_iters, double reduction_factor,
                                        int baseline, int64_t check_every, void* workspace,
                                        size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return fcg_solve_impl(s, n, nrhs, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                            max_iters, reduction_factor, baseline, check_every, workspace, workspace_bytes,
                            host_info);
}

namespace {
int cgs_solve_impl(gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A_, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * nrhs * 11);
    GKOMI_DRIVER_PROLOGUE(8);
    double *r = V(0), *r_tld = V(1), *p = V(2), *q = V(3), *u = V(4), *u_hat = V(5), *v_hat = V(6),
           *t = V(7);
    double *alpha = sc, *beta = sc + nrhs, *gamma = sc + 2 * nrhs, *prev_rho = sc + 3 * nrhs,
           *rho = sc + 4 * nrhs;
    GKOMI_TRY(gkomi_cgs_initialize_f64(s, n, nrhs, b, nrhs, r, nrhs, r_tld, nrhs, p, nrhs, q, nrhs, u,
                                       nrhs, u_hat, nrhs, v_hat, nrhs, t, nrhs, alpha, beta, gamma,
                                       prev_rho, rho, c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, r, nrhs, r_tld, nrhs));
    int64_t iter = -1;
    while (true) {
        GKOMI_TRY(c.dot(r, r_tld, rho));
        ++iter;
        bool stop = false;
        GKOMI_TRY(c.check(iter, r, true, 1, &stop));
        if (stop) break;
        GKOMI_TRY(gkomi_cgs_step_1_f64(s, n, nrhs, r, nrhs, u, nrhs, p, nrhs, q, nrhs, beta, rho,
                                       prev_rho, c.stop_status));
        GKOMI_TRY(c.apply_precond(p, t));
        GKOMI_TRY(c.spmv(t, v_hat));
        GKOMI_TRY(c.dot(r_tld, v_hat, gamma));
        GKOMI_TRY(gkomi_cgs_step_2_f64(s, n, nrhs, u, nrhs, v_hat, nrhs, q, nrhs, t, nrhs, alpha, rho,
                                       gamma, c.stop_status));
        GKOMI_TRY(c.apply_precond(t, u_hat));
        GKOMI_TRY(c.spmv(u_hat, t));
        GKOMI_TRY(gkomi_cgs_step_3_f64(s, n, nrhs, t, nrhs, u_hat, nrhs, r, nrhs, x, nrhs, alpha,
                                       c.stop_status));
        std::swap(prev_rho, rho);
    }
    return c.finish(c.stop_iter(), r, host_info);
}

}  // namespace

// ---- fused CGS, one right-hand side: 5 launches per iteration ------------------------
//   GA  re-adds the partials of rho = r.r_tld and |r|^2, evaluates the criterion,
//       u = r + beta q, p = u + beta (q + beta p) (cgs::step_1)
//   GB  v_hat = A (M p) with the r_tld.v_hat partials in the SpMV epilogue
//   GC  alpha, q = u - alpha v_hat, t = u + q (cgs::step_2)
//   GD  A (M t)
//   GE  x += alpha u_hat, r -= alpha t (cgs::step_3), partials for the next GA
namespace {

struct cgs_scalars {
    double rho[2];
    double alpha, beta, tau, orig_tau;
    long long stop_iter;
    unsigned char status;
    unsigned char pad[7];
};

__global__ void cgs_fused_init_kernel(cgs_scalars* scal, const double* orig_tau)
{
    scal->rho[0] = 0.0;
    scal->rho[1] = 1.0;  // prev_rho = alpha = beta = gamma = 1 (cgs::initialize)
    scal->alpha = 1.0;
    scal->beta = 1.0;
    scal->tau = 0.0;
    scal->orig_tau = orig_tau[0];
    scal->stop_iter = -1;
    scal->status = 0;
}

// GA
__global__ __launch_bounds__(fblock) void cgs_fused_step1_kernel(
    int64_t n, const double* __restrict__ r, double* __restrict__ u, double* __restrict__ p,
    const double* __restrict__ q, const double* __restrict__ rho_part,
    const double* __restrict__ tau_part, int nparts, cgs_scalars* scal, long long it,
    long long max_iters, double goal, host_watch_line* watch = nullptr)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, scal->stop_iter);  // internal.hpp
        return;
    }
    const pair_sweep sw(n);
    double2 r0 = make_double2(0.0, 0.0), q0 = r0, p0 = r0;
    if (sw.first()) {
        r0 = ld2(r, sw.i0);
        q0 = ld2(q, sw.i0);
        p0 = ld2(p, sw.i0);
    }
    const double rho = sum_partials_f(rho_part, nparts, smem);
    const double tau = sqrt(sum_partials_f(tau_part, nparts, smem));
    uint8_t st = 0;
    if (it >= max_iters) {
        st = 1 | GKOMI_STATUS_FINALIZED;
    } else if (tau < goal * scal->orig_tau) {
        st = GKOMI_STATUS_CONVERGED | 1 | GKOMI_STATUS_FINALIZED;
    }
    const double prev = scal->rho[(it + 1) & 1];
    const bool update = prev != 0.0;
    // beta is rewritten only when prev_rho != 0, and then nobody reads the old value
    const double bt = update ? rho / prev : scal->beta;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->rho[it & 1] = rho;
        scal->tau = tau;  // also at the iteration limit: the norm of the residual that is returned
        if (st) {
            scal->stop_iter = it;
            scal->status = st;
        } else if (update) {
            scal->beta = bt;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) host_watch_publish(watch, it, st ? it : -1ll);
    if (st) return;
    auto step1 = [&](double rv, double qv, double pv, double* uo, double* po) {
        const double uu = rv + bt * qv;
        *uo = uu;
        *po = uu + bt * (qv + bt * pv);
    };
    if (sw.first()) {
        double2 uo, po;
        step1(r0.x, q0.x, p0.x, &uo.x, &po.x);
        step1(r0.y, q0.y, p0.y, &uo.y, &po.y);
        st2(u, sw.i0, uo);
        st2(p, sw.i0, po);
    }
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        const double2 rv = ld2(r, i), qv = ld2(q, i), pv = ld2(p, i);
        double2 uo, po;
        step1(rv.x, qv.x, pv.x, &uo.x, &po.x);
        step1(rv.y, qv.y, pv.y, &uo.y, &po.y);
        st2(u, i, uo);
        st2(p, i, po);
    }
    if (sw.tail(n)) {
        const int64_t i = n - 1;
        double uo, po;
        step1(r[i], q[i], p[i], &uo, &po);
        u[i] = uo;
        p[i] = po;
    }
}

// GC
__global__ __launch_bounds__(fblock) void cgs_fused_step2_kernel(
    int64_t n, const double* __restrict__ u, const double* __restrict__ v_hat,
    double* __restrict__ q, double* __restrict__ t, const double* __restrict__ gamma_part,
    int nparts, cgs_scalars* scal, long long it)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) return;
    const pair_sweep sw(n);
    double2 u0 = make_double2(0.0, 0.0), v0 = u0;
    if (sw.first()) {
        u0 = ld2(u, sw.i0);
        v0 = ld2(v_hat, sw.i0);
    }
    const double gamma = sum_partials_f(gamma_part, nparts, smem);
    const bool update = gamma != 0.0;
    const double a = update ? scal->rho[it & 1] / gamma : scal->alpha;
    if (update && blockIdx.x == 0 && threadIdx.x == 0) scal->alpha = a;
    auto step2 = [&](double uv, double vv, double* qo, double* to) {
        const double qq = uv - a * vv;
        *qo = qq;
        *to = uv + qq;
    };
    if (sw.first()) {
        double2 qo, to;
        step2(u0.x, v0.x, &qo.x, &to.x);
        step2(u0.y, v0.y, &qo.y, &to.y);
        st2(q, sw.i0, qo);
        st2(t, sw.i0, to);
    }
    for (int64_t i = sw.i0 + sw.step; i < sw.n2; i += sw.step) {
        const double2 uv = ld2(u, i), vv = ld2(v_hat, i);
        double2 qo, to;
        step2(uv.x, vv.x, &qo.x, &to.x);
        step2(uv.y, vv.y, &qo.y, &to.y);
        st2(q, i, qo);
        st2(t, i, to);
    }
    if (sw.tail(n)) {
        const int64_t i = n - 1;
        double qo, to;
        step2(u[i], v_hat[i], &qo, &to);
        q[i] = qo;
        t[i] = to;
    }
}

// GE: x += alpha xdir, r -= alpha rdir; partials of r.r_tld and r.r
__global__ __launch_bounds__(fblock) void cgs_fused_step3_kernel(
    int64_t n, double* __restrict__ x, double* __restrict__ r, const double* __restrict__ xdir,
    const double* __restrict__ rdir, const double* __restrict__ r_tld, cgs_scalars* scal,
    double* __restrict__ rho_part, double* __restrict__ tau_part)
{
    __shared__ double smem[fblock / wave_size];
    if (status_has_stopped(scal->status)) return;
    const double alpha = scal->alpha;
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * fblock;
    double a0 = 0.0, a1 = 0.0;
    auto step3 = [&](double& xv, double& rv, double xd, double rd, double rt) {
        xv += alpha * xd;
        rv -= alpha * rd;
        a0 += rv * rt;
        a1 += rv * rv;
    };
    for (int64_t i = blockIdx.x * static_cast<int64_t>(fblock) + threadIdx.x; i < n2; i += step) {
        double2 xv = ld2(x, i), rv = ld2(r, i);
        const double2 xd = ld2(xdir, i), rd = ld2(rdir, i), rt = ld2(r_tld, i);
        step3(xv.x, rv.x, xd.x, rd.x, rt.x);
        step3(xv.y, rv.y, xd.y, rd.y, rt.y);
        st2(x, i, xv);
        st2(r, i, rv);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        double xv = x[i], rv = r[i];
        step3(xv, rv, xdir[i], rdir[i], r_tld[i]);
        x[i] = xv;
        r[i] = rv;
    }
    __syncthreads();
    const double s0 = block_reduce_sum<fblock>(a0, smem);
    __syncthreads();
    const double s1 = block_reduce_sum<fblock>(a1, smem);
    if (threadIdx.x == 0) {
        rho_part[blockIdx.x] = s0;
        tau_part[blockIdx.x] = s1;
    }
}

int cgs_fused_impl(gkomi_stream_t s, int64_t n, const sysmat& A_, gkomi_apply_fn precond,
                   void* precond_ctx, const double* b, double* x, int64_t max_iters,
                   double reduction_factor, int baseline, int64_t check_every, void* workspace,
                   size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * 1 * 11);
    const int64_t nrhs = 1;
    if (n > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    if (reinterpret_cast<uintptr_t>(x) % 16 != 0) {
        return cgs_solve_impl(s, n, 1, A, precond, precond_ctx, b, x, max_iters, reduction_factor, baseline,
                              check_every, workspace, workspace_bytes, host_info);
    }
    GKOMI_DRIVER_PROLOGUE(8);
    double *r = V(0), *r_tld = V(1), *p = V(2), *q = V(3), *u = V(4), *u_hat = V(5), *v_hat = V(6),
           *t = V(7);
    GKOMI_TRY(gkomi_cgs_initialize_f64(s, n, 1, b, 1, r, 1, r_tld, 1, p, 1, q, 1, u, 1, u_hat, 1, v_hat, 1, t, 1,
                                       sc, sc + 1, sc + 2, sc + 3, sc + 4, c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    GKOMI_TRY(gkomi_dense_copy_f64(s, n, 1, r, 1, r_tld, 1));
    hipStream_t stream = c.stream;
    double* parts = reinterpret_cast<double*>(ws + l.parts);
    cgs_scalars* scal = reinterpret_cast<cgs_scalars*>(parts);
    double* part_rho = parts + 32;
    double* part_tau = part_rho + fused_max_parts;
    double* part_gamma = part_tau + 2 * fused_max_parts;
    const size_t per_spmv = spmv_dot_partials_room(n);
    const int g = fused_vec_grid(n);
    // the dots in the SpMV's epilogue for CSR / ELL / SELL-P (internal.hpp)
    const spmv_dot_plan spmv(A);
    const bool csr_epilogue = spmv.fused();
    const int nb = csr_epilogue ? spmv.num_partials : g;
    const bool identity = precond == nullptr;
    hipLaunchKernelGGL(cgs_fused_init_kernel, dim3(1), dim3(1), 0, stream, scal, c.orig_tau);
    // partials of r.r_tld and r.r (the third sum is scratch)
    hipLaunchKernelGGL(fused_dot3_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, r_tld, r,
                       static_cast<const unsigned char*>(nullptr), part_rho, part_gamma + per_spmv, part_tau);
    GKOMI_TRY(check_launch());
    cgs_scalars h{};
    long long it = 0;
    bool done = false;
    // the host's view of the solve (host_watch, internal.hpp): the first kernel of every iteration reports the
    // iteration it evaluated; the host stays a few iterations ahead and looks at device memory only once the
    // criterion has fired -- no blocking look every check_every iterations
    host_watch watch;
    const long long lag = std::min<long long>(c.check_every, precond == nullptr ? 4 * host_watch_lag : host_watch_lag);
    while (!done) {
        bool last = false;
        for (int64_t k = 0; k < (watch.dev != nullptr ? 1 : c.check_every) && !done; ++k, ++it) {
            hipLaunchKernelGGL(cgs_fused_step1_kernel, dim3(g), dim3(fblock), 0, stream, n, r, u, p, q, part_rho,
                               part_tau, g, scal, it, static_cast<long long>(max_iters), reduction_factor, watch.dev);
            if (it >= max_iters) {
                ++it;
                last = true;
                break;
            }
            // v_hat = A (M p), gamma = r_tld . v_hat
            const double* mp = p;
            if (!identity) {
                GKOMI_TRY(precond(precond_ctx, s, p, t));
                mp = t;
            }
            if (csr_epilogue) {
                GKOMI_TRY(spmv.launch(stream, mp, v_hat, part_gamma, &scal->status, r_tld));
            } else {
                GKOMI_TRY(A.apply(s, 1, nullptr, mp, nullptr, v_hat));
                hipLaunchKernelGGL(fused_dot3_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r_tld, v_hat,
                                   r_tld, &scal->status, part_gamma, part_gamma + per_spmv,
                                   part_gamma + 2 * per_spmv);
            }
            hipLaunchKernelGGL(cgs_fused_step2_kernel, dim3(g), dim3(fblock), 0, stream, n, u, v_hat, q, t,
                               part_gamma, nb, scal, it);
            // Identity: u_hat = t, and A t goes to the u_hat buffer; otherwise
            // u_hat = M t and A u_hat overwrites t, as in the reference
            const double *xdir, *rdir;
            if (identity) {
                GKOMI_TRY(A.apply(s, 1, nullptr, t, nullptr, u_hat));
                xdir = t;
                rdir = u_hat;
            } else {
                GKOMI_TRY(precond(precond_ctx, s, t, u_hat));
                GKOMI_TRY(A.apply(s, 1, nullptr, u_hat, nullptr, t));
                xdir = u_hat;
                rdir = t;
            }
            hipLaunchKernelGGL(cgs_fused_step3_kernel, dim3(g), dim3(fblock), 0, stream, n, x, r, xdir, rdir,
                               r_tld, scal, part_rho, part_tau);
        }
        GKOMI_TRY(check_launch());
        if (watch.dev != nullptr && !last) {
            if (it - 1 < lag) continue;
            if (watch.wait(stream, it - 1 - lag)) {
                if (watch.stop_iter() < 0) continue;  // still running: no look at device memory
            } else {
                watch.dev = nullptr;  // its stores do not reach this host: the blocking look from here on
            }
        }
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h, scal, sizeof(h), hipMemcpyDeviceToHost, stream)));
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
        done = h.stop_iter >= 0;
    }
    if (host_info != nullptr) {
        host_info[0] = static_cast<double>(h.stop_iter);
        host_info[1] = (h.status & GKOMI_STATUS_CONVERGED) ? 1.0 : 0.0;
        host_info[2] = h.tau;
        host_info[3] = h.orig_tau;
    }
    return precond_status(precond, precond_ctx, s);
}

}  // namespace

extern "C" int gkomi_cgs_solve_fused_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    return cgs_fused_impl(s, n, make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy,
                                                max_row_nnz_hint),
                          precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, check_every,
                          workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_cgs_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return cgs_fused_impl(s, n, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x, max_iters,
                          reduction_factor, baseline, check_every, workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_cgs_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    return cgs_solve_impl(s, n, nrhs,
                            make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint),
                            precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, check_every,
                            workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_cgs_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                        gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                        gkomi_apply_fn precond, void* precond_ctx, const double* b,
                                        double* x, int64_t max_iters, double reduction_factor,
                                        int baseline, int64_t check_every, void* workspace,
                                        size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return cgs_solve_impl(s, n, nrhs, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                            max_iters, reduction_factor, baseline, check_every, workspace, workspace_bytes,
                            host_info);
}

// ---- BiCG / IR ---------------------------------------------------------------------------
extern "C" int gkomi_bicg_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const double* b,
                                         int64_t b_stride, double* r, int64_t r_stride, double* z,
                                         int64_t z_stride, double* p, int64_t p_stride, double* q,
                                         int64_t q_stride, double* prev_rho, double* rho,
                                         double* r2, int64_t r2_stride, double* z2,
                                         int64_t z2_stride, double* p2, int64_t p2_stride,
                                         double* q2, int64_t q2_stride, uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(bicg_initialize_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs,
                       b, b_stride, r, r_stride, z, z_stride, p, p_stride, q, q_stride, prev_rho, rho,
                       r2, r2_stride, z2, z2_stride, p2, p2_stride, q2, q2_stride, stop_status);
    return check_launch();
}

extern "C" int gkomi_bicg_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* p,
                                     int64_t p_stride, const double* z, int64_t z_stride,
                                     double* p2, int64_t p2_stride, const double* z2,
                                     int64_t z2_stride, const double* rho, const double* prev_rho,
                                     const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(bicg_step_1_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, p,
                       p_stride, z, z_stride, p2, p2_stride, z2, z2_stride, rho, prev_rho,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_bicg_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* x,
                                     int64_t x_stride, double* r, int64_t r_stride, double* r2,
                                     int64_t r2_stride, const double* p, int64_t p_stride,
                                     const double* q, int64_t q_stride, const double* q2,
                                     int64_t q2_stride, const double* beta, const double* rho,
                                     const uint8_t* stop_status)
{
    if (bad_dims(n, nrhs)) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(bicg_step_2_kernel, grid_of(n, nrhs), dim3(block), 0, to_stream(s), n, nrhs, x,
                       x_stride, r, r_stride, r2, r2_stride, p, p_stride, q, q_stride, q2, q2_stride,
                       beta, rho, stop_status);
    return check_launch();
}

extern "C" int gkomi_ir_initialize(gkomi_stream_t s, int64_t nrhs, uint8_t* stop_status)
{
    if (nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(ir_initialize_kernel, dim3(static_cast<unsigned>(ceildiv(nrhs, block))),
                       dim3(block), 0, to_stream(s), nrhs, stop_status);
    return check_launch();
}

// Bicg::apply_dense_impl (core/solver/bicg.cpp:117-232).  The conjugate-transposed
// system matrix is passed in (t_*: csr::transpose of A, what the reference
// builds at the top of every apply); precond_t applies the transposed
// preconditioner (NULL with precond == NULL: Identity; a symmetric
// preconditioner passes the same callback twice).
extern "C" int gkomi_bicg_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, const int32_t* t_row_ptrs,
    const int32_t* t_col_idxs, const double* t_vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn precond, void* precond_ctx, gkomi_apply_fn precond_t, void* precond_t_ctx,
    const double* b, double* x, int64_t max_iters, double reduction_factor, int baseline,
    int64_t check_every, void* workspace, size_t workspace_bytes, double* host_info)
{
    if ((precond == nullptr) != (precond_t == nullptr)) return GKOMI_EINVAL;
    const sysmat A = make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint);
    GKOMI_DRIVER_PROLOGUE(8);
    double *r = V(0), *z = V(1), *p = V(2), *q = V(3), *r2 = V(4), *z2 = V(5), *p2 = V(6),
           *q2 = V(7);
    double *beta = sc, *prev_rho = sc + nrhs, *rho = sc + 2 * nrhs;
    GKOMI_TRY(gkomi_bicg_initialize_f64(s, n, nrhs, b, nrhs, r, nrhs, z, nrhs, p, nrhs, q, nrhs,
                                        prev_rho, rho, r2, nrhs, z2, nrhs, p2, nrhs, q2, nrhs,
                                        c.stop_status));
    GKOMI_TRY(c.start(b, x, r, baseline));
    GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, r, nrhs, r2, nrhs));
    int64_t iter = -1;
    while (true) {
        GKOMI_TRY(c.apply_precond(r, z));
        if (precond_t == nullptr) {
            GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, r2, nrhs, z2, nrhs));
        } else {
            GKOMI_TRY(precond_t(precond_t_ctx, s, r2, z2));
        }
        GKOMI_TRY(c.dot(z, r2, rho));
        ++iter;
        bool stop = false;
        GKOMI_TRY(c.check(iter, r, true, 1, &stop));
        if (stop) break;
        GKOMI_TRY(gkomi_bicg_step_1_f64(s, n, nrhs, p, nrhs, z, nrhs, p2, nrhs, z2, nrhs, rho, prev_rho,
                                        c.stop_status));
        GKOMI_TRY(c.spmv(p, q));
        GKOMI_TRY(gkomi_csr_spmv_f64_i32(s, n, n, nrhs, nnz, t_row_ptrs, t_col_idxs, t_vals, p2, nrhs,
                                         q2, nrhs, nullptr, nullptr, spmv_strategy, -1));
        GKOMI_TRY(c.dot(p2, q, beta));
        GKOMI_TRY(gkomi_bicg_step_2_f64(s, n, nrhs, x, nrhs, r, nrhs, r2, nrhs, p, nrhs, q, nrhs, q2,
                                        nrhs, beta, rho, c.stop_status));
        std::swap(prev_rho, rho);
    }
    return c.finish(c.stop_iter(), r, host_info);
}

// Ir::apply_dense_impl (core/solver/ir.cpp:186-277) with the caller's x as the
// initial guess: residual = b - A x, criterion, x += relaxation_factor *
// inner(residual); inner == NULL is the Identity (Richardson).  The criterion
// is looked at on the host every iteration (the update is not status-aware in
// the reference either: a stopped column keeps being relaxed until all stop).
extern "C" int gkomi_ir_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn inner, void* inner_ctx, double relaxation_factor, const double* b, double* x,
    int64_t max_iters, double reduction_factor, int baseline, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    gkomi_apply_fn precond = inner;
    void* precond_ctx = inner_ctx;
    const int64_t check_every = 1;
    const sysmat A = make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint);
    GKOMI_DRIVER_PROLOGUE(2);
    double *residual = V(0), *inner_solution = V(1);
    double* relax = sc;
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, relax, nrhs, relaxation_factor));
    GKOMI_TRY(gkomi_ir_initialize(s, nrhs, c.stop_status));
    GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, b, nrhs, residual, nrhs));
    GKOMI_TRY(c.start(b, x, residual, baseline));
    int64_t iter = -1;
    while (true) {
        ++iter;
        if (iter > 0) {
            GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, b, nrhs, residual, nrhs));
            GKOMI_TRY(A.apply(s, nrhs, c.neg_one, x, c.one, residual));
        }
        bool stop = false;
        GKOMI_TRY(c.check(iter, residual, true, 1, &stop));
        if (stop) break;
        GKOMI_TRY(c.apply_precond(residual, inner_solution));
        GKOMI_TRY(gkomi_dense_add_scaled_f64(s, n, nrhs, relax, nrhs, inner_solution, nrhs, x, nrhs));
    }
    return c.finish(c.stop_iter(), residual, host_info);
}
