// Sparse triangular solves for gfx950 = the ILU preconditioner apply
// (include/ginkgo/core/preconditioner/ilu.hpp:265-305: L^-1 then U^-1).
// Replaces gko::kernels::hip::{lower_trs,upper_trs}::{generate, solve,
// should_perform_transpose} (core/solver/{lower,upper}_trs_kernels.hpp), which
// call hipSPARSE csrsv2 in the reference; semantics =
// reference/solver/lower_trs_kernels.cpp:90-120, upper_trs_kernels.cpp:90-123.
//
// Sync-free algorithm, one launch per right-hand side, one thread per row:
//  * x is pre-filled with a sentinel NaN payload; a row may use x[col] once it
//    no longer reads as the sentinel.  The 8-byte value is its own "ready"
//    flag: stored with ONE agent-scope (sc1, write-through) store and polled
//    with agent-scope (sc1) loads, the data-tagged granule hand-off of
//    MI355X_MICROARCH.md (per-XCD L2s are not coherent, per-CU L1s never
//    refreshed: plain loads would spin on stale lines).
//  * rows are handed out in chunks through an atomic ticket, so every row a
//    chunk depends on belongs to a workgroup that has already started: no
//    dependence on dispatch order or placement.
//  * no lane ever spins inside a loop another lane of its wave needs to leave:
//    each pass consumes the dependencies that are ready, a finished row stores
//    at once, and the wave leaves together (`__all`).  Spins are bounded; on
//    overrun the kernel raises a flag in the workspace instead of hanging.
// Per row the subtractions happen in storage order -> bit-identical to the
// reference.  Latency-bound (dependency chain x L2 round trip), as SURVEY 8(d)
// says; bytes 12*nnz + 4(n+1) + 16n.
#include "common.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr unsigned long long sentinel_bits = 0x7ff8dead0badbeefull;
constexpr long long max_passes = 1ll << 26;

struct trs_workspace {
    unsigned int ticket;
    unsigned int overrun;
};

__global__ __launch_bounds__(block) void trs_prepare_kernel(int64_t n, double* __restrict__ x,
                                                           int64_t x_stride,
                                                           trs_workspace* __restrict__ ws)
{
    const int64_t gid = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
    if (gid == 0) {
        ws->ticket = 0;
        ws->overrun = 0;
    }
    for (int64_t i = gid; i < n; i += static_cast<int64_t>(gridDim.x) * block) {
        reinterpret_cast<unsigned long long*>(x)[i * x_stride] = sentinel_bits;
    }
}

template <bool Lower>
__global__ __launch_bounds__(block) void trs_syncfree_kernel(
    int64_t n, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, bool unit_diag, const double* __restrict__ b,
    int64_t b_stride, double* x, int64_t x_stride, trs_workspace* ws)
{
    __shared__ unsigned int chunk_s;
    if (threadIdx.x == 0) chunk_s = atomicAdd(&ws->ticket, 1u);
    __syncthreads();
    const int64_t pos = static_cast<int64_t>(chunk_s) * block + threadIdx.x;
    bool done = pos >= n;
    const int64_t row = Lower ? pos : n - 1 - pos;
    int k = 0, end = 0;
    double sum = 0.0, diag = 1.0;
    if (!done) {
        k = row_ptrs[row];
        end = row_ptrs[row + 1];
        sum = b[row * b_stride];
    }
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(x);
    for (long long pass = 0; pass < max_passes; ++pass) {
        bool progressed = false;
        if (!done) {
            while (k < end) {
                const int col = col_idxs[k];
                if (col == row) {
                    diag = vals[k];
                    ++k;
                    continue;
                }
                if (Lower ? col > row : col < row) {  // other triangle: not part of the solve
                    ++k;
                    continue;
                }
                const unsigned long long bits = __hip_atomic_load(
                    xb + col * x_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (bits == sentinel_bits) break;  // not ready yet
                sum -= vals[k] * __longlong_as_double(static_cast<long long>(bits));
                ++k;
                progressed = true;
            }
            if (k == end) {
                double r = unit_diag ? sum : sum / diag;
                unsigned long long out = static_cast<unsigned long long>(__double_as_longlong(r));
                if (out == sentinel_bits) out ^= 1ull;  // a result must never read as "pending"
                __hip_atomic_store(xb + row * x_stride, out, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                done = true;
                progressed = true;
            }
        }
        if (__all(done)) return;
        if (!__any(progressed)) __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x % 64 == 0) atomicExch(&ws->overrun, 1u);
}

template <bool Lower>
int trs_solve(gkomi_stream_t s, int64_t n, int64_t nrhs, const int32_t* row_ptrs,
              const int32_t* col_idxs, const double* vals, int unit_diag, const double* b,
              int64_t b_stride, double* x, int64_t x_stride, void* workspace,
              size_t workspace_bytes)
{
    if (n < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || x_stride < nrhs) return GKOMI_EINVAL;
    if (workspace == nullptr || workspace_bytes < sizeof(trs_workspace)) return GKOMI_EWORKSPACE;
    if (x == b) return GKOMI_EINVAL;  // x is used as the ready flags
    const int64_t chunks = ceildiv(n, block);
    if (chunks > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    hipStream_t stream = to_stream(s);
    trs_workspace* ws = static_cast<trs_workspace*>(workspace);
    for (int64_t j = 0; j < nrhs; ++j) {
        hipLaunchKernelGGL(trs_prepare_kernel, dim3(grid_for(n, block)), dim3(block), 0, stream, n,
                           x + j, x_stride, ws);
        hipLaunchKernelGGL(trs_syncfree_kernel<Lower>, dim3(static_cast<unsigned>(chunks)),
                           dim3(block), 0, stream, n, row_ptrs, col_idxs, vals, unit_diag != 0,
                           b + j, b_stride, x + j, x_stride, ws);
        int err = check_launch();
        if (err) return err;
    }
    return GKOMI_SUCCESS;
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_trs_workspace_bytes(void) { return 64; }

extern "C" int gkomi_lower_trs_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                             const int32_t* row_ptrs, const int32_t* col_idxs,
                                             const double* vals, int unit_diag, const double* b,
                                             int64_t b_stride, double* x, int64_t x_stride,
                                             void* workspace, size_t workspace_bytes)
{
    return trs_solve<true>(s, n, nrhs, row_ptrs, col_idxs, vals, unit_diag, b, b_stride, x,
                           x_stride, workspace, workspace_bytes);
}

extern "C" int gkomi_upper_trs_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                             const int32_t* row_ptrs, const int32_t* col_idxs,
                                             const double* vals, int unit_diag, const double* b,
                                             int64_t b_stride, double* x, int64_t x_stride,
                                             void* workspace, size_t workspace_bytes)
{
    return trs_solve<false>(s, n, nrhs, row_ptrs, col_idxs, vals, unit_diag, b, b_stride, x,
                            x_stride, workspace, workspace_bytes);
}

// nonzero if a solve on this workspace gave up waiting (blocking read)
extern "C" int gkomi_trs_check_overrun(gkomi_stream_t s, const void* workspace, int* host_flag)
{
    if (workspace == nullptr || host_flag == nullptr) return GKOMI_EINVAL;
    trs_workspace h{};
    hipStream_t stream = to_stream(s);
    int err = static_cast<int>(
        hipMemcpyAsync(&h, workspace, sizeof(h), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    *host_flag = static_cast<int>(h.overrun);
    return err;
}
