// GMRES for gfx950: the step kernels of core/solver/{common_gmres,gmres}_kernels.hpp
// (initialize, restart, hessenberg_qr, solve_krylov, multi_axpy; semantics =
// reference/solver/common_gmres_kernels.cpp:60-217, gmres_kernels.cpp:55-100)
// and a native driver for Gmres::apply_dense_impl (core/solver/gmres.cpp:139-372).
//
// The Arnoldi step keeps the reference's modified Gram-Schmidt order (dot,
// axpy, dot, axpy ...), so the Hessenberg entries agree with the reference up
// to the reduction order of each dot.  The small-matrix kernels (Givens QR on
// one Hessenberg column, triangular solve) run one thread per right-hand
// side exactly like the reference loops -> bit-identical given equal inputs.
// HBM: step k moves (5k+8) n values (core/solver/gmres.cpp:217-222).
#include "cg_persistent.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <string>
#include <utility>

namespace gkomi {
namespace {

constexpr int block = 256;

__global__ __launch_bounds__(block) void gmres_initialize_kernel(
    int64_t n, int64_t nrhs, int64_t krylov_dim, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ residual, int64_t r_stride, double* __restrict__ gsin,
    double* __restrict__ gcos, uint8_t* __restrict__ stop_status)
{
    const int64_t gid = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = gid; i < krylov_dim * nrhs; i += step) {
        gsin[i] = 0.0;
        gcos[i] = 0.0;
    }
    for (int64_t i = gid; i < nrhs; i += step) stop_status[i] = 0;
    for (int64_t i = gid; i < n * nrhs; i += step) {
        const int64_t row = i / nrhs, col = i % nrhs;
        residual[row * r_stride + col] = b[row * b_stride + col];
    }
}

__global__ __launch_bounds__(block) void gmres_restart_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ residual, int64_t r_stride,
    const double* __restrict__ residual_norm, double* __restrict__ rnc,
    double* __restrict__ krylov_bases, int64_t kb_stride, uint64_t* __restrict__ final_iter_nums)
{
    const int64_t gid = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t i = gid; i < nrhs; i += step) {
        rnc[i] = residual_norm[i];
        final_iter_nums[i] = 0;
    }
    for (int64_t i = gid; i < n * nrhs; i += step) {
        const int64_t row = i / nrhs, col = i % nrhs;
        krylov_bases[row * kb_stride + col] = residual[row * r_stride + col] / residual_norm[col];
    }
}

// the reference's loop body for right-hand side i (common_gmres_kernels.cpp hessenberg_qr)
__device__ __forceinline__ void hessenberg_qr_column(
    int64_t i, int64_t nrhs, double* __restrict__ gsin, double* __restrict__ gcos,
    double* __restrict__ residual_norm, double* __restrict__ rnc, double* __restrict__ hess,
    int64_t h_stride, int64_t iter, uint64_t* __restrict__ final_iter_nums,
    const uint8_t* __restrict__ stop_status)
{
    if (status_has_stopped(stop_status[i])) return;
    final_iter_nums[i]++;
    for (int64_t j = 0; j < iter; ++j) {
        const double c = gcos[j * nrhs + i], s = gsin[j * nrhs + i];
        const double h0 = hess[j * h_stride + i], h1 = hess[(j + 1) * h_stride + i];
        const double temp = c * h0 + s * h1;
        hess[(j + 1) * h_stride + i] = -s * h0 + c * h1;
        hess[j * h_stride + i] = temp;
    }
    const double this_h = hess[iter * h_stride + i];
    const double next_h = hess[(iter + 1) * h_stride + i];
    double c, s;
    if (this_h == 0.0) {
        c = 0.0;
        s = 1.0;
    } else {
        const double scale = fabs(this_h) + fabs(next_h);
        const double hyp = scale * sqrt(fabs(this_h / scale) * fabs(this_h / scale) +
                                        fabs(next_h / scale) * fabs(next_h / scale));
        c = this_h / hyp;
        s = next_h / hyp;
    }
    gcos[iter * nrhs + i] = c;
    gsin[iter * nrhs + i] = s;
    hess[iter * h_stride + i] = c * this_h + s * next_h;
    hess[(iter + 1) * h_stride + i] = 0.0;
    const double r = rnc[iter * nrhs + i];
    const double next_r = -s * r;
    rnc[(iter + 1) * nrhs + i] = next_r;
    rnc[iter * nrhs + i] = c * r;
    residual_norm[i] = fabs(next_r);
}

// one thread per right-hand side
__global__ __launch_bounds__(block) void gmres_hessenberg_qr_kernel(
    int64_t nrhs, double* __restrict__ gsin, double* __restrict__ gcos,
    double* __restrict__ residual_norm, double* __restrict__ rnc, double* __restrict__ hess,
    int64_t h_stride, int64_t iter, uint64_t* __restrict__ final_iter_nums,
    const uint8_t* __restrict__ stop_status)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < nrhs;
         i += static_cast<int64_t>(gridDim.x) * block) {
        hessenberg_qr_column(i, nrhs, gsin, gcos, residual_norm, rnc, hess, h_stride, iter, final_iter_nums,
                             stop_status);
    }
}

__global__ __launch_bounds__(block) void gmres_solve_krylov_kernel(
    int64_t nrhs, const double* __restrict__ rnc, const double* __restrict__ hessenberg,
    int64_t h_stride, double* __restrict__ y, const uint64_t* __restrict__ final_iter_nums,
    const uint8_t* __restrict__ stop_status)
{
    for (int64_t k = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; k < nrhs;
         k += static_cast<int64_t>(gridDim.x) * block) {
        if (stop_status[k] & GKOMI_STATUS_FINALIZED) continue;
        const int64_t m = static_cast<int64_t>(final_iter_nums[k]);
        for (int64_t i = m - 1; i >= 0; --i) {
            double temp = rnc[i * nrhs + k];
            for (int64_t j = i + 1; j < m; ++j) {
                temp -= hessenberg[i * h_stride + j * nrhs + k] * y[j * nrhs + k];
            }
            y[i * nrhs + k] = temp / hessenberg[i * h_stride + i * nrhs + k];
        }
    }
}

// One right-hand side: the same back substitution, operand by operand in the same order, out of LDS -- the loop
// above waits for a dependent load from memory in every one of its m (m + 1) / 2 steps (45 us per restart of
// GMRES(30); this one 4 us, profiles/r03_arnoldi_blocked.md).
constexpr int solve_krylov_max = 64;
__global__ __launch_bounds__(256) void gmres_solve_krylov_single_kernel(
    const double* __restrict__ rnc, const double* __restrict__ hessenberg, int64_t h_stride,
    double* __restrict__ y, const uint64_t* __restrict__ final_iter_nums, const uint8_t* __restrict__ stop_status)
{
    __shared__ double lh[solve_krylov_max * solve_krylov_max];
    __shared__ double ly[solve_krylov_max];
    if (stop_status[0] & GKOMI_STATUS_FINALIZED) return;
    const int m = static_cast<int>(final_iter_nums[0]);
    for (int e = threadIdx.x; e < m * m; e += 256) {
        const int i = e / m, j = e % m;
        if (j >= i) lh[i * solve_krylov_max + j] = hessenberg[i * h_stride + j];
    }
    for (int e = threadIdx.x; e < m; e += 256) ly[e] = rnc[e];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = m - 1; i >= 0; --i) {
            double temp = ly[i];
            for (int j = i + 1; j < m; ++j) temp -= lh[i * solve_krylov_max + j] * ly[j];
            ly[i] = temp / lh[i * solve_krylov_max + i];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < m; e += 256) y[e] = ly[e];
}

// before_preconditioner = V y: each thread owns one (row, rhs) entry and adds
// the Krylov vectors in index order, like the reference
__global__ __launch_bounds__(block) void gmres_multi_axpy_kernel(
    int64_t n, int64_t nrhs, const double* __restrict__ krylov_bases, int64_t kb_stride,
    const double* __restrict__ y, double* __restrict__ before, int64_t bp_stride,
    const uint64_t* __restrict__ final_iter_nums, const uint8_t* __restrict__ stop_status)
{
    const int64_t total = n * nrhs;
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * block) {
        const int64_t i = idx / nrhs, k = idx % nrhs;
        if (stop_status[k] & GKOMI_STATUS_FINALIZED) continue;
        const int64_t m = static_cast<int64_t>(final_iter_nums[k]);
        double acc = 0.0;
        for (int64_t j = 0; j < m; ++j) {
            acc += krylov_bases[(i + j * n) * kb_stride + k] * y[j * nrhs + k];
        }
        before[i * bp_stride + k] = acc;
    }
}

__global__ __launch_bounds__(block) void gmres_finalize_status_kernel(
    int64_t nrhs, uint8_t* __restrict__ stop_status)
{
    for (int64_t k = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; k < nrhs;
         k += static_cast<int64_t>(gridDim.x) * block) {
        const uint8_t st = stop_status[k];
        if (!(st & GKOMI_STATUS_FINALIZED) && status_has_stopped(st)) {
            stop_status[k] = st | GKOMI_STATUS_FINALIZED;
        }
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- fused modified Gram-Schmidt (single right-hand side) ---------------------
// The reference's Arnoldi step (core/solver/gmres.cpp:300-319) is, per basis
// vector i, hessenberg(i) = dot(next, basis_i) followed by next -= hessenberg(i)
// * basis_i: 5n values of traffic and two launches + a reduction finish each.
// Fused: launch i subtracts h_{i-1} * basis_{i-1} from next AND accumulates the
// partial sums of dot(next, basis_i) in the same pass (4n values, one launch);
// every workgroup re-adds the previous launch's partials (fixed order) to get
// h_{i-1}, so no separate finishing kernel and no host round trip.  The last
// launch produces the partials of ||next||^2; the scaling kernel re-adds them,
// writes hessenberg(k+1) = ||next|| and scales next.  Same operations in the
// same order as the reference (MGS), only the summation order of the dots
// differs (two-stage, deterministic).
constexpr int arnoldi_block = 1024;
constexpr int arnoldi_max_blocks = 1024;

// deterministic sum of `count` partials by the whole workgroup; the total is
// identical in every thread of every workgroup
__device__ double arnoldi_sum_partials(const double* __restrict__ partials, int count, double* smem)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += arnoldi_block) acc += partials[i];
    acc = wave_reduce_sum(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = acc;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < arnoldi_block / wave_size; ++w) total += smem[w];
    return total;
}

__device__ __forceinline__ double2 ld2(const double* p, int64_t i)
{
    return reinterpret_cast<const double2*>(p)[i];
}

// next -= h_prev * prev (if prev), partial_out[block] = sum next * (with ? with : next).
// 16 B per lane; the first sweep's loads are issued before the partials are
// re-added (they do not depend on h_prev), so the reduction hides behind them.
__global__ __launch_bounds__(arnoldi_block) void gmres_arnoldi_step_kernel(
    int64_t n, double* __restrict__ next, const double* __restrict__ prev,
    const double* __restrict__ with, const double* __restrict__ partial_in, int count_in,
    double* __restrict__ h_prev_out, double* __restrict__ partial_out)
{
    __shared__ double smem[arnoldi_block / wave_size];
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * arnoldi_block;
    const int64_t i0 = blockIdx.x * static_cast<int64_t>(arnoldi_block) + threadIdx.x;
    double2 v0 = make_double2(0.0, 0.0), p0 = v0, w0 = v0;
    if (i0 < n2) {
        v0 = ld2(next, i0);
        if (prev != nullptr) p0 = ld2(prev, i0);
        if (with != nullptr) w0 = ld2(with, i0);
    }
    double h = 0.0;
    if (prev != nullptr) {
        h = arnoldi_sum_partials(partial_in, count_in, smem);
        if (blockIdx.x == 0 && threadIdx.x == 0) *h_prev_out = h;
    }
    double a0 = 0.0, a1 = 0.0;
    double2* next2 = reinterpret_cast<double2*>(next);
    if (i0 < n2) {
        if (prev != nullptr) {
            v0.x -= h * p0.x;
            v0.y -= h * p0.y;
            next2[i0] = v0;
        }
        a0 += v0.x * (with != nullptr ? w0.x : v0.x);
        a1 += v0.y * (with != nullptr ? w0.y : v0.y);
    }
    for (int64_t i = i0 + step; i < n2; i += step) {
        double2 v = ld2(next, i);
        if (prev != nullptr) {
            const double2 pv = ld2(prev, i);
            v.x -= h * pv.x;
            v.y -= h * pv.y;
            next2[i] = v;
        }
        if (with != nullptr) {
            const double2 wv = ld2(with, i);
            a0 += v.x * wv.x;
            a1 += v.y * wv.y;
        } else {
            a0 += v.x * v.x;
            a1 += v.y * v.y;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double v = next[n - 1];
        if (prev != nullptr) {
            v -= h * prev[n - 1];
            next[n - 1] = v;
        }
        a0 += v * (with != nullptr ? with[n - 1] : v);
    }
    __syncthreads();
    const double total = block_reduce_sum<arnoldi_block>(a0 + a1, smem);
    if (threadIdx.x == 0) partial_out[blockIdx.x] = total;
}

// hn = sqrt(sum partials) -> *hn_out; next /= hn
__global__ __launch_bounds__(arnoldi_block) void gmres_arnoldi_scale_kernel(
    int64_t n, double* __restrict__ next, const double* __restrict__ partial_in, int count_in,
    double* __restrict__ hn_out)
{
    __shared__ double smem[arnoldi_block / wave_size];
    const int64_t n2 = n / 2;
    const int64_t step = static_cast<int64_t>(gridDim.x) * arnoldi_block;
    const int64_t i0 = blockIdx.x * static_cast<int64_t>(arnoldi_block) + threadIdx.x;
    double2 v0 = make_double2(0.0, 0.0);
    if (i0 < n2) v0 = ld2(next, i0);
    const double hn = sqrt(arnoldi_sum_partials(partial_in, count_in, smem));
    if (blockIdx.x == 0 && threadIdx.x == 0) *hn_out = hn;
    double2* next2 = reinterpret_cast<double2*>(next);
    if (i0 < n2) next2[i0] = make_double2(v0.x / hn, v0.y / hn);
    for (int64_t i = i0 + step; i < n2; i += step) {
        const double2 v = ld2(next, i);
        next2[i] = make_double2(v.x / hn, v.y / hn);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) next[n - 1] /= hn;
}

// odd n: the basis vectors kb + n * i are only 8-byte aligned for odd i -> scalar
// kernels (same algorithm, 8 B per lane)
__device__ double sum_partials_block(const double* __restrict__ partials, int count, double* smem)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += arnoldi_block) acc += partials[i];
    return block_reduce_sum<arnoldi_block>(acc, smem);
}

// next -= h_prev * prev (if prev), partial_out[block] = sum next * (with ? with : next)
__global__ __launch_bounds__(arnoldi_block) void gmres_arnoldi_step_scalar_kernel(
    int64_t n, double* __restrict__ next, const double* __restrict__ prev,
    const double* __restrict__ with, const double* __restrict__ partial_in, int count_in,
    double* __restrict__ h_prev_out, double* __restrict__ partial_out)
{
    __shared__ double smem[arnoldi_block / wave_size];
    __shared__ double h_s;
    double h = 0.0;
    if (prev != nullptr) {
        const double total = sum_partials_block(partial_in, count_in, smem);
        if (threadIdx.x == 0) {
            h_s = total;
            if (blockIdx.x == 0) *h_prev_out = total;
        }
        __syncthreads();
        h = h_s;
    }
    double acc = 0.0;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(arnoldi_block) + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * arnoldi_block) {
        double v = next[e];
        if (prev != nullptr) {
            v -= h * prev[e];
            next[e] = v;
        }
        acc += v * (with != nullptr ? with[e] : v);
    }
    __syncthreads();
    const double total = block_reduce_sum<arnoldi_block>(acc, smem);
    if (threadIdx.x == 0) partial_out[blockIdx.x] = total;
}

// hn = sqrt(sum partials) -> *hn_out; next /= hn
__global__ __launch_bounds__(arnoldi_block) void gmres_arnoldi_scale_scalar_kernel(
    int64_t n, double* __restrict__ next, const double* __restrict__ partial_in, int count_in,
    double* __restrict__ hn_out)
{
    __shared__ double smem[arnoldi_block / wave_size];
    __shared__ double h_s;
    const double total = sum_partials_block(partial_in, count_in, smem);
    if (threadIdx.x == 0) {
        h_s = sqrt(total);
        if (blockIdx.x == 0) *hn_out = h_s;
    }
    __syncthreads();
    const double hn = h_s;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(arnoldi_block) + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * arnoldi_block) {
        next[e] /= hn;
    }
}

// ---- one launch per Arnoldi step instead of one per basis vector ----------------
// The modified Gram-Schmidt sweep of iteration k needs k + 2 device-wide sums, one
// after the other (h_i = v_i . w can only start when w has lost its v_{i-1}
// component).  With a launch per sum that is k + 2 kernel boundaries and w
// re-read and re-written every time (4 n values per basis vector).  Here one
// workgroup per CU keeps its part of w in registers for the whole sweep, streams
// each basis vector once (the next one is already in flight while the
// workgroups meet) and the sums travel through the meetings of
// cg_persistent.hpp: n values per basis vector and ~3 us per sum.
// Same operations in the same order as the reference's loop
// (core/solver/gmres.cpp:300-319); the partial sums are grouped per workgroup.
// A meeting that times out leaves next_k untouched (w is written at the very
// end) and raises ctl->overrun: the driver sees it at its next poll.
struct gmres_stop_record {
    long long iter;
};

// what follows the sweep in the reference's iteration, done by one thread of workgroup 0 instead
// of three more launches: the Givens update of the new Hessenberg column, then -- for the NEXT
// iteration, which the driver would open with it -- the ResidualNorm criterion and the record
// of the first iteration at which the solve had stopped
struct gmres_iteration_tail {
    double* gsin;
    double* gcos;
    double* residual_norm;
    double* rnc;
    uint64_t* final_iter_nums;
    uint8_t* stop_status;
    const double* orig_tau;
    double goal;
    uint8_t* flags;
    gmres_stop_record* record;
    long long next_iter;
    int restart_iter;
    host_watch_line* watch;  // the host's view of the solve (internal.hpp), or nullptr
};

// (one thread of workgroup 0, behind the sweep)
__device__ __forceinline__ void gmres_finish_iteration(double* hess_iter, int64_t h_stride, const gmres_iteration_tail& tail)
{
    hessenberg_qr_column(0, 1, tail.gsin, tail.gcos, tail.residual_norm, tail.rnc, hess_iter, h_stride,
                         tail.restart_iter, tail.final_iter_nums, tail.stop_status);
    // stop::ResidualNorm on the updated estimate (residual_norm_kernel<false> of stop.hip, one column,
    // not finalized) and gmres_record_stop_kernel
    uint8_t st = tail.stop_status[0];
    uint8_t one_changed = 0;
    if (tail.residual_norm[0] < tail.goal * tail.orig_tau[0]) {
        if (!status_has_stopped(st)) {
            st |= GKOMI_STATUS_CONVERGED | 2;  // id_residual
            tail.stop_status[0] = st;
        }
        one_changed = 1;
    }
    const uint8_t all = status_has_stopped(st) ? 1 : 0;
    tail.flags[0] = all;
    tail.flags[1] = one_changed;
    if (all && tail.record->iter < 0) tail.record->iter = tail.next_iter;
    host_watch_publish(tail.watch, tail.next_iter, tail.record->iter);
}

template <int R>
__global__ __launch_bounds__(pcg_block) void gmres_arnoldi_persistent_kernel(
    int n, int chunk, double* __restrict__ next_k, const double* __restrict__ kb, int steps,
    double* __restrict__ hess_iter, int64_t h_stride, pcg_slot* slots, int stride, int nap, pcg_control* ctl,
    long long meeting, long long max_polls, gmres_iteration_tail tail)
{
    __shared__ double smem[pcg_block / wave_size + 1];
    if (__hip_atomic_load(&ctl->overrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const int nwg = gridDim.x;
    const int tid = threadIdx.x;
    const int b0 = min(static_cast<int>(blockIdx.x) * chunk, n);
    const int b1 = min(b0 + chunk, n);
    double w[R], v[R], vn[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = b0 + k * pcg_block + tid;
        w[k] = row < b1 ? next_k[row] : 0.0;
        v[k] = row < b1 ? kb[row] : 0.0;
        vn[k] = 0.0;
    }
    for (int i = 0; i < steps; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) acc += w[k] * v[k];
        if (i + 1 < steps) {  // the next basis vector travels while the workgroups meet
            const double* nxt = kb + static_cast<int64_t>(n) * (i + 1);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int row = b0 + k * pcg_block + tid;
                vn[k] = row < b1 ? nxt[row] : 0.0;
            }
        }
        const double mine = pcg_block_sum<pcg_block>(acc, smem);
        double h = 0.0;
        if (!pcg_meet<pcg_block>(slots, stride, nap, nwg, ++meeting, mine, smem, ctl, max_polls, &h)) return;
        if (blockIdx.x == 0 && tid == 0) hess_iter[i * h_stride] = h;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            w[k] -= h * v[k];
            v[k] = vn[k];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) acc += w[k] * w[k];
    const double mine = pcg_block_sum<pcg_block>(acc, smem);
    double hn2 = 0.0;
    if (!pcg_meet<pcg_block>(slots, stride, nap, nwg, ++meeting, mine, smem, ctl, max_polls, &hn2)) return;
    const double hn = sqrt(hn2);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = b0 + k * pcg_block + tid;
        if (row < b1) next_k[row] = w[k] / hn;
    }
    if (blockIdx.x == 0 && tid == 0) {
        hess_iter[steps * h_stride] = hn;
        gmres_finish_iteration(hess_iter, h_stride, tail);
    }
}

// ---- the same sweep with one meeting per B basis vectors ---------------------------------------------
// What the sweep above costs is its meetings: 16 + 5.3 (k + 1) us on the 108^3 system, of which streaming a basis
// vector is 2 us (profiles/r03_arnoldi_blocked.md).  Modified Gram-Schmidt against v_0 .. v_{B-1} is
//     h_i = v_i . (w - sum_{m<i} h_m v_m) = v_i . w - sum_{m<i} (v_i . v_m) h_m ,
// a unit lower triangular system in the B sums v_i . w and the B (B - 1) / 2 sums v_i . v_m -- none of which needs an
// h.  So: B vectors travel together into registers, the B + B (B - 1) / 2 sums go through ONE meeting
// (pcg_meet_values), every thread solves the little system (same bits everywhere) and w loses the B components in
// the reference's order; the next B vectors are in flight meanwhile.  An identity, not an approximation: nothing
// assumes the basis orthogonal (the v_i . v_m are the measured ones); against the sweep above only the rounding of
// h_i differs (sum of products vs product of sums), by O(eps |v_i . v_m| |h_m|).  Each basis vector is still read
// exactly once.  512 threads per workgroup: 256 registers per lane hold w, the block and the block in flight.
constexpr int arn_block = 512;

// The workgroup's NV sums, each published (slot v of this workgroup, pcg_meet_values<..., Published = true>) by the
// wave that adds it up: through LDS transposed, wave v adds value v of all lanes in a fixed order.
template <int NV>
__device__ __forceinline__ void arn_sum_and_publish(const double (&acc)[NV], double* lp, double* flag, pcg_slot* slots,
                                                    int stride, int nwg, long long meeting)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int v = 0; v < NV; ++v) lp[v * arn_block + tid] = acc[v];
    if (tid == 0) *flag = 1.0;  // "nobody gave up"
    pcg_sync_lds();
    const int wave = tid >> 6, lane = tid & 63;
    pcg_slot* mine = slots + (meeting & 1) * static_cast<int64_t>(nwg + pcg_copies) * stride + blockIdx.x * stride;
    for (int v = wave; v < NV; v += arn_block / wave_size) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < arn_block / wave_size; ++k) s += lp[v * arn_block + lane + wave_size * k];
        s = wave_reduce_sum(s);
        if (lane == 0) pcg_publish(mine + v, s, meeting);
    }
}

// R2 = pairs of consecutive rows per lane (rows chunk start + 2 (r * 512 + lane) and the one behind it); Wide: n and
// the chunk are even, so a pair is one aligned 16-byte load.
template <int R2, int B, bool Wide>
__global__ __launch_bounds__(arn_block) void gmres_arnoldi_blocked_kernel(
    int n, int chunk, double* __restrict__ next_k, const double* __restrict__ kb, int steps,
    double* __restrict__ hess_iter, int64_t h_stride, pcg_slot* slots, int stride, int nap, pcg_control* ctl,
    long long meeting, long long max_polls, gmres_iteration_tail tail, bool nt_basis)
{
    constexpr int R = 2 * R2;
    constexpr int NV = B + B * (B - 1) / 2;
    constexpr int nwaves = arn_block / wave_size;
    __shared__ double lp[NV * arn_block];
    __shared__ double lsum[NV];
    __shared__ double lred[nwaves * NV + 1];
    __shared__ double smem[nwaves + 1];
    if (__hip_atomic_load(&ctl->overrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const int nwg = gridDim.x;
    const int tid = threadIdx.x;
    // Workgroup 0 owns no rows: it is the one that adds up every meeting's values, and its polls must not queue
    // behind a block of basis vectors in flight (vector-memory loads return in order) -- with rows of its own, a
    // meeting ended when ITS next block had landed, i.e. meetings and streaming took turns (8.4 us per block of
    // three vectors: 5.1 of bytes + 3.3 of meeting; profiles/r03_arnoldi_blocked.md).
    const int b0 = blockIdx.x == 0 ? n : min((static_cast<int>(blockIdx.x) - 1) * chunk, n);
    const int b1 = min(b0 + chunk, n);
    double w[R], v[B][R], vn[B][R];
    auto load_vector = [&](const double* src, bool wanted, double (&dst)[R]) {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const int row = b0 + 2 * (r * arn_block + tid);
            dst[2 * r] = dst[2 * r + 1] = 0.0;
            if (Wide) {
                if (wanted && row < b1) {
                    typedef double nt_pair __attribute__((ext_vector_type(2)));
                    double2 pair;
                    if (nt_basis) {
                        const nt_pair t = __builtin_nontemporal_load(reinterpret_cast<const nt_pair*>(src + row));
                        pair = make_double2(t.x, t.y);
                    } else {
                        pair = *reinterpret_cast<const double2*>(src + row);
                    }
                    dst[2 * r] = pair.x;
                    dst[2 * r + 1] = pair.y;
                }
            } else {
                if (wanted && row < b1) dst[2 * r] = src[row];
                if (wanted && row + 1 < b1) dst[2 * r + 1] = src[row + 1];
            }
        }
    };
    auto load_block = [&](int m, double (&dst)[B][R]) {
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int k = m * B + i;
            load_vector(kb + static_cast<int64_t>(n) * min(k, steps - 1), k < steps, dst[i]);
        }
    };
    load_vector(next_k, true, w);
    load_block(0, vn);
    const int nblocks = (steps + B - 1) / B;
#ifdef GKOMI_ARN_STAMPS
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_ = wall_clock64();
#define ARN_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wall_clock64(); ph[i] += n_ - t_; t_ = n_; } while (0)
#define ARN_STAMP_NOWAIT(i) do { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - t_; t_ = n_; } while (0)
#else
#define ARN_STAMP(i)
#define ARN_STAMP_NOWAIT(i)
#endif
    for (int m = 0; m < nblocks; ++m) {
        ARN_STAMP(0);  // waiting for the block in flight
#pragma unroll
        for (int i = 0; i < B; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[i][r] = vn[i][r];
        }

        double acc[NV];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            acc[i] = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[i] += w[r] * v[i][r];
        }
#pragma unroll
        for (int a = 1; a < B; ++a) {
#pragma unroll
            for (int c = 0; c < a; ++c) {
                double g = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) g += v[a][r] * v[c][r];
                acc[B + a * (a - 1) / 2 + c] = g;
            }
        }
        ARN_STAMP_NOWAIT(1);  // dots
        ++meeting;
        arn_sum_and_publish<NV>(acc, lp, lred + nwaves * NV, slots, stride, nwg, meeting);
        ARN_STAMP_NOWAIT(2);  // sums of the workgroup
        // the next block travels while the workgroups meet; asked for behind the publication of this block's sums
        // (in front of it the loads' way into a busy memory pipe delayed every workgroup's sums: 3.9 -> 2.5 us)
        auto next_block = [&]() {
            if (m + 1 < nblocks) load_block(m + 1, vn);
        };
        if (!pcg_meet_values<arn_block, NV, true>(slots, stride, nap, nwg, meeting, lsum, lred, lp, ctl, max_polls, next_block)) return;
        ARN_STAMP_NOWAIT(3);  // meeting
        double h[B];
#pragma unroll
        for (int a = 0; a < B; ++a) {
            double t = lsum[a];
#pragma unroll
            for (int c = 0; c < a; ++c) t -= lsum[B + a * (a - 1) / 2 + c] * h[c];
            h[a] = t;
        }
        if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
            for (int a = 0; a < B; ++a) {
                if (m * B + a < steps) hess_iter[(m * B + a) * h_stride] = h[a];
            }
        }
#pragma unroll
        for (int a = 0; a < B; ++a) {
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] -= h[a] * v[a][r];
        }
        ARN_STAMP_NOWAIT(4);  // solve + update
    }
#ifdef GKOMI_ARN_STAMPS
    if (steps == 30 && tid == 0 && (blockIdx.x == 1 || blockIdx.x == 0 || blockIdx.x == 200)) {
        printf("wg %d: %d blocks; wait for block %llu, dots %llu, sums %llu, meeting %llu, update %llu (10 ns ticks)\n",
               static_cast<int>(blockIdx.x), nblocks, ph[0], ph[1], ph[2], ph[3], ph[4]);
    }
#endif
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) acc += w[r] * w[r];
    const double mine = pcg_block_sum<arn_block>(acc, smem);
    double hn2 = 0.0;
    if (!pcg_meet<arn_block>(slots, stride, nap, nwg, ++meeting, mine, smem, ctl, max_polls, &hn2)) return;
    const double hn = sqrt(hn2);
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const int row = b0 + 2 * (r * arn_block + tid);
        if (Wide) {
            if (row < b1) *reinterpret_cast<double2*>(next_k + row) = make_double2(w[2 * r] / hn, w[2 * r + 1] / hn);
        } else {
            if (row < b1) next_k[row] = w[2 * r] / hn;
            if (row + 1 < b1) next_k[row + 1] = w[2 * r + 1] / hn;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        hess_iter[steps * h_stride] = hn;
        gmres_finish_iteration(hess_iter, h_stride, tail);
    }
}

// first iteration at which every column had stopped (-1 until then): the
// criterion runs on the device at every iteration, the host looks every
// `gmres_check_every` iterations and before every restart
constexpr int64_t gmres_check_every = 4;

__global__ void gmres_record_stop_kernel(const uint8_t* __restrict__ flags, long long iter,
                                         gmres_stop_record* record, host_watch_line* watch)
{
    if (flags[0] && record->iter < 0) record->iter = iter;
    host_watch_publish(watch, iter, record->iter);
}

struct gmres_layout {
    size_t residual, pv, before, after, kb, hess, gsin, gcos, rnc, y, small, fin, red, partials, pcg_ctl, pcg_slots,
        total;
};

gmres_layout make_layout(int64_t n, int64_t nrhs, int64_t d)
{
    gmres_layout l{};
    const size_t vec = align_up(sizeof(double) * static_cast<size_t>(n) * nrhs, 256);
    size_t off = 0;
    l.residual = off; off += vec;
    l.pv = off; off += vec;
    l.before = off; off += vec;
    l.after = off; off += vec;
    l.kb = off; off += align_up(sizeof(double) * static_cast<size_t>(n) * nrhs * (d + 1), 256);
    l.hess = off; off += align_up(sizeof(double) * static_cast<size_t>((d + 1) * d * nrhs), 256);
    l.gsin = off; off += align_up(sizeof(double) * static_cast<size_t>(d * nrhs), 256);
    l.gcos = off; off += align_up(sizeof(double) * static_cast<size_t>(d * nrhs), 256);
    l.rnc = off; off += align_up(sizeof(double) * static_cast<size_t>((d + 1) * nrhs), 256);
    l.y = off; off += align_up(sizeof(double) * static_cast<size_t>(d * nrhs), 256);
    // residual_norm, orig_tau, one, neg_one (nrhs each), then stop_status + flags
    l.small = off; off += align_up(sizeof(double) * 4 * static_cast<size_t>(nrhs) + nrhs + 48, 256);
    l.fin = off; off += align_up(sizeof(uint64_t) * static_cast<size_t>(nrhs), 256);
    l.red = off; off += align_up(gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8, 256);
    l.partials = off; off += align_up(sizeof(double) * 2 * arnoldi_max_blocks, 256);
    l.pcg_ctl = off; off += align_up(sizeof(pcg_control), 256);
    l.pcg_slots = off; off += align_up(sizeof(pcg_slot) * 2 * (max_parts + pcg_copies) * pcg_default_stride, 256);
    l.total = off;
    return l;
}

#define GKOMI_TRY(expr)        \
    do {                       \
        int err_ = (expr);     \
        if (err_) return err_; \
    } while (0)

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int gkomi_gmres_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                          int64_t krylov_dim, const double* b, int64_t b_stride,
                                          double* residual, int64_t r_stride, double* givens_sin,
                                          double* givens_cos, uint8_t* stop_status)
{
    if (n < 0 || nrhs < 0 || krylov_dim < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    const int64_t work = std::max<int64_t>(n * nrhs, krylov_dim * nrhs);
    hipLaunchKernelGGL(gmres_initialize_kernel, dim3(grid_for(work, block)), dim3(block), 0,
                       to_stream(s), n, nrhs, krylov_dim, b, b_stride, residual, r_stride,
                       givens_sin, givens_cos, stop_status);
    return check_launch();
}

extern "C" int gkomi_gmres_restart_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                       const double* residual, int64_t r_stride,
                                       const double* residual_norm,
                                       double* residual_norm_collection, double* krylov_bases,
                                       int64_t kb_stride, uint64_t* final_iter_nums)
{
    if (n < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(gmres_restart_kernel, dim3(grid_for(std::max<int64_t>(n * nrhs, nrhs), block)),
                       dim3(block), 0, to_stream(s), n, nrhs, residual, r_stride, residual_norm,
                       residual_norm_collection, krylov_bases, kb_stride, final_iter_nums);
    return check_launch();
}

extern "C" int gkomi_gmres_hessenberg_qr_f64(gkomi_stream_t s, int64_t nrhs, double* givens_sin,
                                             double* givens_cos, double* residual_norm,
                                             double* residual_norm_collection,
                                             double* hessenberg_iter, int64_t h_stride,
                                             int64_t iter, uint64_t* final_iter_nums,
                                             const uint8_t* stop_status)
{
    if (nrhs < 0 || iter < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(gmres_hessenberg_qr_kernel, dim3(grid_for(nrhs, block)), dim3(block), 0,
                       to_stream(s), nrhs, givens_sin, givens_cos, residual_norm,
                       residual_norm_collection, hessenberg_iter, h_stride, iter, final_iter_nums,
                       stop_status);
    return check_launch();
}

extern "C" int gkomi_gmres_solve_krylov_f64(gkomi_stream_t s, int64_t nrhs,
                                            const double* residual_norm_collection,
                                            const double* hessenberg, int64_t h_stride, double* y,
                                            const uint64_t* final_iter_nums,
                                            const uint8_t* stop_status)
{
    if (nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(gmres_solve_krylov_kernel, dim3(grid_for(nrhs, block)), dim3(block), 0,
                       to_stream(s), nrhs, residual_norm_collection, hessenberg, h_stride, y,
                       final_iter_nums, stop_status);
    return check_launch();
}

// the driver's call: it knows the restart length, so the one-column case goes through LDS
static int solve_krylov_launch(gkomi_stream_t s, int64_t nrhs, int64_t krylov_dim, const double* rnc,
                               const double* hessenberg, int64_t h_stride, double* y,
                               const uint64_t* final_iter_nums, const uint8_t* stop_status)
{
    const char* e = std::getenv("GKOMI_GMRES_SOLVE_KRYLOV");  // "generic": the test's switch (bit-identity of the two)
    if (nrhs == 1 && krylov_dim <= solve_krylov_max && (e == nullptr || std::string(e) != "generic")) {
        hipLaunchKernelGGL(gmres_solve_krylov_single_kernel, dim3(1), dim3(256), 0, to_stream(s), rnc, hessenberg,
                           h_stride, y, final_iter_nums, stop_status);
        return check_launch();
    }
    return gkomi_gmres_solve_krylov_f64(s, nrhs, rnc, hessenberg, h_stride, y, final_iter_nums, stop_status);
}

extern "C" int gkomi_gmres_multi_axpy_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                          const double* krylov_bases, int64_t kb_stride,
                                          const double* y, double* before_preconditioner,
                                          int64_t bp_stride, const uint64_t* final_iter_nums,
                                          uint8_t* stop_status)
{
    if (n < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipStream_t stream = to_stream(s);
    if (n > 0) {
        hipLaunchKernelGGL(gmres_multi_axpy_kernel, dim3(grid_for(n * nrhs, block)), dim3(block), 0,
                           stream, n, nrhs, krylov_bases, kb_stride, y, before_preconditioner,
                           bp_stride, final_iter_nums, stop_status);
    }
    // "if has_stopped: finalize" after the column is done (gmres_kernels.cpp:95-97)
    hipLaunchKernelGGL(gmres_finalize_status_kernel, dim3(grid_for(nrhs, block)), dim3(block), 0,
                       stream, nrhs, stop_status);
    return check_launch();
}

extern "C" size_t gkomi_gmres_workspace_bytes(int64_t n, int64_t nrhs, int64_t krylov_dim)
{
    if (n < 0 || nrhs <= 0 || krylov_dim <= 0) return 0;
    return make_layout(n, nrhs, krylov_dim).total;
}

namespace {
std::atomic<int64_t> gmres_meeting_fallbacks{0};
}
extern "C" int64_t gkomi_gmres_meeting_fallbacks(void) { return gmres_meeting_fallbacks.load(); }

namespace {
int gmres_solve_impl(gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A_, gkomi_apply_fn precond,
                     void* precond_ctx, const double* b, double* x, int64_t krylov_dim,
                     int64_t max_iters, double reduction_factor, int baseline, void* workspace,
                     size_t workspace_bytes, double* host_info, bool allow_persistent = true)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    // What stays in the Infinity Cache between two uses (measured on the 108^3 system, 310 MB of basis, 106 MB of
    // matrix; profiles/r03_arnoldi_blocked.md): when matrix + basis do not fit, the basis is read with nontemporal
    // loads (it is the larger stream and every vector of it is read once per iteration) and the matrix keeps the
    // cache -- 30.9 -> 28.7 ms without a preconditioner; behind a preconditioner (its factors or blocks are streams the
    // driver knows nothing about) the matrix streams as well -- 43.8 -> 42.0 ms with ParILU.
    const int64_t vec_bytes = static_cast<int64_t>(sizeof(double)) * n * nrhs;
    const bool nt_basis = A.is_csr() && A.storage_bytes() + vec_bytes * (krylov_dim + 6) > infinity_cache_bytes;
    A.note_working_set(nt_basis ? vec_bytes * 6 + (precond != nullptr ? infinity_cache_bytes : 0)
                                : vec_bytes * (krylov_dim + 6));
    if (n < 0 || nrhs <= 0 || krylov_dim <= 0 || max_iters < 0) return GKOMI_EINVAL;
    if (baseline < 0 || baseline > 2) return GKOMI_EINVAL;
    const gmres_layout l = make_layout(n, nrhs, krylov_dim);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    auto D = [&](size_t off) { return reinterpret_cast<double*>(ws + off); };
    double *residual = D(l.residual), *pv = D(l.pv), *before = D(l.before), *after = D(l.after);
    double *kb = D(l.kb), *hess = D(l.hess), *gsin = D(l.gsin), *gcos = D(l.gcos);
    double *rnc = D(l.rnc), *y = D(l.y);
    double* small = D(l.small);
    double *residual_norm = small, *orig_tau = small + nrhs, *one = small + 2 * nrhs,
           *neg_one = small + 3 * nrhs;
    uint8_t* stop_status = reinterpret_cast<uint8_t*>(small + 4 * nrhs);
    uint8_t* dev_flags = stop_status + nrhs + (8 - nrhs % 8) % 8;
    uint64_t* final_iter_nums = reinterpret_cast<uint64_t*>(ws + l.fin);
    void* red = ws + l.red;
    const size_t red_bytes = gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8;
    const int64_t h_stride = krylov_dim * nrhs;
    constexpr uint8_t id_iteration = 1, id_residual = 2;

    auto apply_precond = [&](const double* in, double* out) -> int {
        if (precond == nullptr) return gkomi_dense_copy_f64(s, n, nrhs, in, nrhs, out, nrhs);
        return precond(precond_ctx, s, in, out);
    };
    auto residual_and_restart = [&]() -> int {
        // residual = b - A x; residual_norm; restart (gmres.cpp:184-195 / 260-275)
        GKOMI_TRY(gkomi_dense_copy_f64(s, n, nrhs, b, nrhs, residual, nrhs));
        GKOMI_TRY(A.apply(s, nrhs, neg_one, x, one, residual));
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, residual, nrhs, residual_norm, red,
                                                red_bytes));
        return gkomi_gmres_restart_f64(s, n, nrhs, residual, nrhs, residual_norm, rnc, kb, nrhs,
                                       final_iter_nums);
    };
    auto update_solution = [&](int64_t) -> int {
        GKOMI_TRY(solve_krylov_launch(s, nrhs, krylov_dim, rnc, hess, h_stride, y, final_iter_nums,
                                               stop_status));
        GKOMI_TRY(gkomi_gmres_multi_axpy_f64(s, n, nrhs, kb, nrhs, y, before, nrhs,
                                             final_iter_nums, stop_status));
        GKOMI_TRY(apply_precond(before, after));
        return gkomi_dense_add_scaled_f64(s, n, nrhs, one, 1, after, nrhs, x, nrhs);
    };

    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, one, nrhs, 1.0));
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, neg_one, nrhs, -1.0));
    GKOMI_TRY(gkomi_dense_fill_f64(s, krylov_dim + 1, h_stride, hess, h_stride, 0.0));
    GKOMI_TRY(gkomi_gmres_initialize_f64(s, n, nrhs, krylov_dim, b, nrhs, residual, nrhs, gsin,
                                         gcos, stop_status));
    GKOMI_TRY(residual_and_restart());
    if (baseline == 0) {
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, b, nrhs, orig_tau, red, red_bytes));
    } else if (baseline == 1) {
        GKOMI_TRY(gkomi_dense_copy_f64(s, 1, nrhs, residual_norm, nrhs, orig_tau, nrhs));
    } else {
        GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, orig_tau, nrhs, 1.0));
    }

    long long total_iter = -1;
    int64_t restart_iter = 0;
    int converged = 0;
    // The criterion is evaluated on the device at every iteration, exactly where
    // the reference evaluates it; stopped columns are frozen by their statuses
    // (final_iter_nums, residual_norm, the Givens data no longer change), so the
    // iterations the host lets run before it looks change nothing in x.
    gmres_stop_record* record = reinterpret_cast<gmres_stop_record*>(dev_flags + 8);
    gmres_stop_record host_record{-1};
    GKOMI_TRY(static_cast<int>(hipMemcpyAsync(record, &host_record, sizeof(host_record), hipMemcpyHostToDevice, stream)));
    GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    int64_t unpolled = 0;
    // one launch per Arnoldi step (gmres_arnoldi_persistent_kernel) when w fits the register files
    static const bool persistent_on = [] {
        const char* e = std::getenv("GKOMI_GMRES_PERSISTENT");
        return e == nullptr || e[0] != '0';
    }();
    // GKOMI_GMRES_ARNOLDI=sweep: one meeting per basis vector (gmres_arnoldi_persistent_kernel) instead of one per
    // block of them -- the A/B switch of tools/, not a product setting
    const bool blocked_sweep = [] {
        const char* e = std::getenv("GKOMI_GMRES_ARNOLDI");
        return e == nullptr || std::string(e) != "sweep";
    }();
    static const int cus = device_cu_count();
    pcg_control* pctl = reinterpret_cast<pcg_control*>(ws + l.pcg_ctl);
    pcg_slot* pslots = reinterpret_cast<pcg_slot*>(ws + l.pcg_slots);
    // (an even chunk of an even n: the blocked sweep loads pairs of rows)
    const int pchunk = cus > 0 ? static_cast<int>(ceildiv(n, cus) + (n % 2 == 0 ? ceildiv(n, cus) % 2 : 0)) : 0;
    const int prows = static_cast<int>(ceildiv(pchunk, pcg_block));
    bool persistent = allow_persistent && persistent_on && nrhs == 1 && cus >= 8 && cus <= max_parts &&
                      n >= 64 * static_cast<int64_t>(cus) && n <= INT32_MAX && prows <= pcg_max_rows_per_thread &&
                      persistent_try_acquire();
    struct release_guard {
        bool held;
        ~release_guard() { if (held) persistent_release(); }
    } release{persistent};
    long long meeting = 0;
    bool criterion_done = false;
    host_watch watch;  // the device reports the iteration it has reached into pinned host memory
    pcg_control host_ctl{};
    if (persistent) {
        hipLaunchKernelGGL(pcg_clear_kernel, dim3(1), dim3(256), 0, stream, pslots, pcg_default_stride,
                           2 * (cus + pcg_copies), pctl);
    }
    auto poll = [&]() -> int {
        unpolled = 0;
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&host_record, record, sizeof(host_record), hipMemcpyDeviceToHost, stream)));
        if (persistent) {
            GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&host_ctl, pctl, sizeof(unsigned int), hipMemcpyDeviceToHost, stream)));
        }
        return static_cast<int>(hipStreamSynchronize(stream));
    };
    // A meeting of the single-launch Arnoldi step timed out (workgroups not resident together?): the
    // launches since then returned early, so the Hessenberg column, the Givens terms, the residual norms
    // and the stop record may be stale or unwritten.  Nothing of the current restart cycle is used: x
    // still holds the solution of the last COMPLETED restart (only update_solution writes x, and every
    // path to it passes a poll) -- solve again from there with one launch per sum.  Checked right after
    // every poll, before update_solution and before the stop / max_iters exits.
    auto solve_again_without_meetings = [&]() -> int {
        gmres_meeting_fallbacks.fetch_add(1);
        release.held = false;
        persistent_release();
        const long long done = std::max<long long>(total_iter - restart_iter, 0);
        const int err = gmres_solve_impl(s, n, nrhs, A, precond, precond_ctx, b, x, krylov_dim,
                                         std::max<int64_t>(max_iters - done, 0), reduction_factor, baseline,
                                         workspace, workspace_bytes, host_info, false);
        if (host_info != nullptr) host_info[0] += static_cast<double>(done);
        return err;
    };
    const long long meet_max_polls = [] {
        const char* e = std::getenv("GKOMI_MEET_MAX_POLLS");  // test hook: how long a meeting waits
        return e != nullptr && e[0] != 0 ? std::max(1ll, atoll(e)) : 1ll << 22;
    }();
    while (true) {
        ++total_iter;
        bool stop = false;
        if (total_iter >= max_iters) {
            GKOMI_TRY(poll());  // converged during the iterations not looked at yet?
            if (persistent && host_ctl.overrun != 0) return solve_again_without_meetings();
            if (host_record.iter >= 0) {
                converged = 1;
                total_iter = host_record.iter;
            } else {
                GKOMI_TRY(gkomi_set_all_statuses(s, nrhs, id_iteration, 0, stop_status));
            }
            stop = true;
        } else {
            // the criterion gets residual_norm directly (gmres.cpp:240-246), not finalized; the
            // single-launch Arnoldi step of the iteration before has already evaluated it
            if (!criterion_done) {
                GKOMI_TRY(gkomi_residual_norm_f64(s, nrhs, residual_norm, orig_tau, reduction_factor,
                                                  id_residual, 0, stop_status, dev_flags, nullptr));
                hipLaunchKernelGGL(gmres_record_stop_kernel, dim3(1), dim3(1), 0, stream, dev_flags, total_iter,
                                   record, watch.dev);
            }
            criterion_done = false;
            ++unpolled;
            bool look = false;  // a blocking look at device memory
            if (watch.dev != nullptr) {
                // the device tells the host how far it is (host_watch): before a restart the host needs the
                // criterion of THIS iteration, otherwise it only keeps within host_watch_lag iterations
                const long long target = restart_iter == krylov_dim ? total_iter : total_iter - host_watch_lag;
                if (target >= 0) {
                    if (watch.wait(stream, target)) {
                        host_record.iter = watch.stop_iter();
                    } else {
                        look = true;  // nothing to see and the stream has drained: a meeting timed out
                        watch.dev = nullptr;                 // (or the stores do not reach this host)
                    }
                }
            } else {
                look = unpolled >= gmres_check_every || restart_iter == krylov_dim;
            }
            if (look) {
                GKOMI_TRY(poll());
                if (persistent && host_ctl.overrun != 0) return solve_again_without_meetings();
            }
            if (host_record.iter >= 0) {
                stop = true;
                converged = 1;
                total_iter = host_record.iter;
            }
        }
        if (stop) break;
        if (restart_iter == krylov_dim) {
            GKOMI_TRY(update_solution(restart_iter));
            GKOMI_TRY(residual_and_restart());
            restart_iter = 0;
        }
        double* this_k = kb + n * nrhs * restart_iter;
        double* next_k = kb + n * nrhs * (restart_iter + 1);
        // Identity preconditioner: matrix::Identity::apply copies this_k -- the SpMV reads it in place
        if (precond != nullptr) GKOMI_TRY(apply_precond(this_k, pv));
        double* hess_iter = hess + nrhs * restart_iter;
        GKOMI_TRY(A.apply(s, nrhs, nullptr, precond != nullptr ? pv : this_k, nullptr, next_k));
        if (persistent) {
            const int steps = static_cast<int>(restart_iter + 1);
            const gmres_iteration_tail tail{gsin, gcos, residual_norm, rnc, final_iter_nums, stop_status, orig_tau,
                                            reduction_factor, dev_flags, record, total_iter + 1,
                                            static_cast<int>(restart_iter), watch.dev};
            // pairs of rows per lane of the blocked sweep's 512-thread workgroups (prows <= 8 above: at most 8)
            const int64_t per = ceildiv(n, cus - 1);  // workgroup 0 serves the meetings
            const int bchunk = static_cast<int>(per + (n % 2 == 0 ? per % 2 : 0));
            const int bpairs = static_cast<int>(ceildiv(bchunk, 2 * arn_block));
            const bool wide = n % 2 == 0 && bchunk % 2 == 0;
            if (blocked_sweep && cus <= arn_block && bpairs <= 8) {  // (workgroup 0 keeps cus x values of a meeting in LDS)
#define GKOMI_ARNB(R2, B)                                                                                     \
    do {                                                                                                      \
        if (wide) {                                                                                           \
            hipLaunchKernelGGL((gmres_arnoldi_blocked_kernel<R2, B, true>), dim3(cus), dim3(arn_block), 0,    \
                               stream, static_cast<int>(n), bchunk, next_k, kb, steps, hess_iter, h_stride,   \
                               pslots, pcg_default_stride, 1, pctl, meeting, meet_max_polls, tail, nt_basis);          \
        } else {                                                                                              \
            hipLaunchKernelGGL((gmres_arnoldi_blocked_kernel<R2, B, false>), dim3(cus), dim3(arn_block), 0,   \
                               stream, static_cast<int>(n), bchunk, next_k, kb, steps, hess_iter, h_stride,   \
                               pslots, pcg_default_stride, 1, pctl, meeting, meet_max_polls, tail, nt_basis);          \
        }                                                                                                     \
        meeting += (steps + (B) - 1) / (B) + 1;                                                               \
    } while (0)
                // block sizes measured on the 108^3 system (5 pairs per lane): 3 vectors 31.1 ms per solve, 4 (spills
                // seven registers) 32.2, 2 32.7 -- the sweep is bound by the basis' bytes, not by its meetings
                if (bpairs <= 1) {
                    GKOMI_ARNB(1, 4);
                } else if (bpairs <= 3) {
                    GKOMI_ARNB(3, 4);
                } else if (bpairs <= 5) {
                    GKOMI_ARNB(5, 3);
                } else {
                    GKOMI_ARNB(8, 2);
                }
#undef GKOMI_ARNB
            } else {
#define GKOMI_ARN(R)                                                                                  \
    hipLaunchKernelGGL(gmres_arnoldi_persistent_kernel<R>, dim3(cus), dim3(pcg_block), 0, stream,     \
                       static_cast<int>(n), pchunk, next_k, kb, steps, hess_iter, h_stride, pslots,   \
                       pcg_default_stride, 1, pctl, meeting, meet_max_polls, tail)
                if (prows <= 1) {
                    GKOMI_ARN(1);
                } else if (prows <= 2) {
                    GKOMI_ARN(2);
                } else if (prows <= 4) {
                    GKOMI_ARN(4);
                } else {
                    GKOMI_ARN(8);
                }
#undef GKOMI_ARN
                meeting += steps + 1;
            }
            criterion_done = true;
            GKOMI_TRY(check_launch());
            restart_iter++;
            continue;  // the Givens update is part of the launch
        } else if (nrhs == 1 && n > 0) {
            // fused modified Gram-Schmidt: one pass over next_k per basis vector
            const int blocks = static_cast<int>(
                std::min<int64_t>(arnoldi_max_blocks, std::max<int64_t>(1, ceildiv(n / 2 + 1, arnoldi_block))));
            double* partial[2] = {D(l.partials), D(l.partials) + arnoldi_max_blocks};
            for (int64_t i = 0; i <= restart_iter + 1; ++i) {
                const double* prev = i > 0 ? kb + n * (i - 1) : nullptr;
                const double* with = i <= restart_iter ? kb + n * i : nullptr;
                double* h_prev = i > 0 ? hess_iter + (i - 1) * h_stride : nullptr;
                if (n % 2 == 0) {
                    hipLaunchKernelGGL(gmres_arnoldi_step_kernel, dim3(blocks), dim3(arnoldi_block), 0,
                                       stream, n, next_k, prev, with, partial[(i + 1) & 1], blocks,
                                       h_prev, partial[i & 1]);
                } else {
                    hipLaunchKernelGGL(gmres_arnoldi_step_scalar_kernel, dim3(blocks), dim3(arnoldi_block), 0,
                                       stream, n, next_k, prev, with, partial[(i + 1) & 1], blocks,
                                       h_prev, partial[i & 1]);
                }
            }
            double* hn = hess_iter + (restart_iter + 1) * h_stride;
            if (n % 2 == 0) {
                hipLaunchKernelGGL(gmres_arnoldi_scale_kernel, dim3(blocks), dim3(arnoldi_block), 0,
                                   stream, n, next_k, partial[(restart_iter + 1) & 1], blocks, hn);
            } else {
                hipLaunchKernelGGL(gmres_arnoldi_scale_scalar_kernel, dim3(blocks), dim3(arnoldi_block), 0,
                                   stream, n, next_k, partial[(restart_iter + 1) & 1], blocks, hn);
            }
            GKOMI_TRY(check_launch());
        } else {
            for (int64_t i = 0; i <= restart_iter; ++i) {
                double* h = hess_iter + i * h_stride;
                const double* basis = kb + n * nrhs * i;
                GKOMI_TRY(gkomi_dense_compute_dot_f64(s, n, nrhs, next_k, nrhs, basis, nrhs, h, red,
                                                      red_bytes));
                GKOMI_TRY(gkomi_dense_sub_scaled_f64(s, n, nrhs, h, nrhs, basis, nrhs, next_k, nrhs));
            }
            double* hn = hess_iter + (restart_iter + 1) * h_stride;
            GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, next_k, nrhs, hn, red, red_bytes));
            GKOMI_TRY(gkomi_dense_inv_scale_f64(s, n, nrhs, hn, nrhs, next_k, nrhs));
        }
        GKOMI_TRY(gkomi_gmres_hessenberg_qr_f64(s, nrhs, gsin, gcos, residual_norm, rnc, hess_iter,
                                                h_stride, restart_iter, final_iter_nums,
                                                stop_status));
        restart_iter++;
    }
    GKOMI_TRY(update_solution(restart_iter));
    if (host_info != nullptr) {
        for (int64_t j = 0; j < nrhs; ++j) {
            GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 2 + 2 * j, residual_norm + j,
                                                      sizeof(double), hipMemcpyDeviceToHost, stream)));
            GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 3 + 2 * j, orig_tau + j,
                                                      sizeof(double), hipMemcpyDeviceToHost, stream)));
        }
        host_info[0] = static_cast<double>(total_iter);
        host_info[1] = static_cast<double>(converged);
    }
    GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    return precond_status(precond, precond_ctx, s);
}
}  // namespace

extern "C" int gkomi_gmres_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x, int64_t krylov_dim,
    int64_t max_iters, double reduction_factor, int baseline, void* workspace,
    size_t workspace_bytes, double* host_info)
{
    return gmres_solve_impl(s, n, nrhs,
                            make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint),
                            precond, precond_ctx, b, x, krylov_dim, max_iters, reduction_factor, baseline,
                            workspace, workspace_bytes, host_info);
}

extern "C" int gkomi_gmres_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                        gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                        gkomi_apply_fn precond, void* precond_ctx, const double* b,
                                        double* x, int64_t krylov_dim, int64_t max_iters,
                                        double reduction_factor, int baseline, void* workspace,
                                        size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return gmres_solve_impl(s, n, nrhs, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                            krylov_dim, max_iters, reduction_factor, baseline, workspace, workspace_bytes,
                            host_info);
}
