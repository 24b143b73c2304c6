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
//  * rows are handed out in chunks of 512 through an atomic ticket, so every
//    row a chunk depends on belongs to a workgroup that has already started:
//    no dependence on dispatch order or placement.
//  * a finished row is ALSO published in an LDS copy of the chunk's x; a
//    dependency on a row of the same chunk (the common case: r-1, r-grid) is
//    an LDS poll (~0.1 us per hand-off) instead of a memory round trip (>1 us).
//  * polls of out-of-chunk rows are software-pipelined: issued, then the wave
//    runs a few LDS-only passes for its other lanes, and only then is the
//    answer looked at -- a lane waiting for memory does not make the in-wave
//    chains of its neighbours advance at memory latency.
//  * no lane ever spins inside a loop another lane of its wave needs to leave:
//    each pass consumes the dependencies that are ready, a finished row stores
//    at once, and the wave leaves together (`__all`).  Spins are bounded; on
//    overrun the kernel raises a flag in the workspace instead of hanging.
// Per row the subtractions happen in storage order -> bit-identical to the
// reference.  Latency-bound (dependency chain x hand-off latency), as SURVEY
// 8(d) says; bytes 12*nnz + 4(n+1) + 16n.
#include "common.hpp"

#include <cstdlib>

namespace gkomi {
namespace {

constexpr int block = 512;
constexpr int prep_block = 256;
constexpr int window = 6;      // dependency entries of a row parked in LDS
constexpr int lds_passes = 4;  // at most this many LDS-only passes between two looks at memory
constexpr int max_idle = 8;    // x s_sleep(4) = 256 cycles each
constexpr unsigned long long sentinel_bits = 0x7ff8dead0badbeefull;
constexpr long long default_max_rounds = 1ll << 23;

struct trs_workspace {
    unsigned int ticket;
    unsigned int overrun;
};

__global__ __launch_bounds__(prep_block) void trs_prepare_kernel(int64_t n, double* __restrict__ x,
                                                                int64_t x_stride,
                                                                trs_workspace* __restrict__ ws)
{
    const int64_t gid = blockIdx.x * static_cast<int64_t>(prep_block) + threadIdx.x;
    if (gid == 0) ws->ticket = 0;  // overrun is sticky: zeroed by whoever creates the workspace
    for (int64_t i = gid; i < n; i += static_cast<int64_t>(gridDim.x) * prep_block) {
        reinterpret_cast<unsigned long long*>(x)[i * x_stride] = sentinel_bits;
    }
}

// Positions: pos = row (lower) or n-1-row (upper) is the solve order; a
// dependency of pos always has a smaller position.  LDS per workgroup:
// 512 x (6 x 12 B window + 8 B result) = 40 KB -> 4 workgroups = 32 waves/CU.
template <bool Lower>
__global__ __launch_bounds__(block) void trs_syncfree_kernel(
    int32_t n, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, bool unit_diag, const double* __restrict__ b,
    int64_t b_stride, double* x, int64_t x_stride, trs_workspace* ws, long long max_rounds)
{
    __shared__ unsigned long long s_x[block];  // results of this chunk, sentinel = pending
    __shared__ int s_pos[window][block];       // dependency positions of the row
    __shared__ double s_val[window][block];
    const int tid = threadIdx.x;
    // the ticket travels through s_x[0] (40960 B of LDS exactly: 4 workgroups per CU)
    if (tid == 0) s_x[0] = atomicAdd(&ws->ticket, 1u);
    __syncthreads();
    const int chunk_base = static_cast<int>(s_x[0]) * block;  // chunks * block fits: host check
    __syncthreads();
    s_x[tid] = sentinel_bits;
    __syncthreads();
    const int pos = chunk_base + tid;
    const int row = Lower ? pos : n - 1 - pos;
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(x);

    // Row state.  Storage entries [k, end) not yet looked at; the window holds
    // the dependency entries found so far, wj..wn-1 not yet subtracted.
    // (cur_pos, cur_val) = window entry wj, with the whole lane state folded
    // into cur_pos so that a pass tests one register:
    //   >= chunk_base     dependency on a row of this chunk (LDS poll)
    //   0..chunk_base-1   dependency on an earlier chunk, not fetched yet
    //   st_fetched        fetched ahead: cur_val already holds val * x[col]
    //   st_complete       nothing left to subtract: divide and publish
    //   st_done           published (or no row at all)
    // Only refill() reads the matrix from memory and it waits for its own
    // loads, so the pass loop depends on LDS alone.
    constexpr int st_fetched = -1, st_complete = -2, st_done = -3;
    int k = 0, end = 0, wj = 0, wn = 0, cur_pos = st_done;
    unsigned int far_mask = 0;  // window entries that live in earlier chunks, not fetched yet
    double sum = 0.0, diag = 1.0, cur_val = 0.0;
    auto refill = [&]() {  // requires wj == wn, k < end
        int c[window];
        double v[window];
#pragma unroll
        for (int e = 0; e < window; ++e) {  // clamped: all loads in flight together
            const int ke = min(k + e, end - 1);
            c[e] = col_idxs[ke];
            v[e] = vals[ke];
        }
        const int cnt = min(window, end - k);
        wj = wn = 0;
        far_mask = 0;
#pragma unroll
        for (int e = 0; e < window; ++e) {
            if (e < cnt) {
                if (c[e] == row) {
                    diag = v[e];
                } else if (Lower ? c[e] < row : c[e] > row) {
                    const int p = Lower ? c[e] : n - 1 - c[e];
                    s_pos[wn][tid] = p;
                    s_val[wn][tid] = v[e];
                    if (p < chunk_base) far_mask |= 1u << wn;
                    ++wn;
                }  // else: other triangle, not part of the solve
            }
        }
        k += cnt;
        // clamped loads past `cnt` are never used, hence never waited for: drain
        // them here (vmcnt(0)) or their registers stay "pending" and the hot
        // loop below gets conservative waits that also wait for polls and stores
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    auto load_cur = [&]() {
        while (wj == wn && k < end) refill();
        if (wj < wn) {
            cur_pos = s_pos[wj][tid];
            cur_val = s_val[wj][tid];
        } else {
            cur_pos = st_complete;
        }
    };
    if (pos < n) {
        k = row_ptrs[row];
        end = row_ptrs[row + 1];
        sum = b[static_cast<int64_t>(row) * b_stride];
        load_cur();
        // a use of `sum` here: the load of b is waited for once, before the
        // loop, instead of by a conservative s_waitcnt vmcnt(0) inside it
        asm volatile("" : "+v"(sum));
    }

    // Out-of-chunk dependencies are FETCHED ahead of their turn: the answer
    // val * x[col] is parked in the window (st_fetched) and subtracted when its
    // turn comes, so the storage-order subtraction chain never waits for
    // memory behind an LDS hand-off (upper solves meet the nearest, i.e.
    // latest, dependency first).  One poll per lane in flight.
    unsigned long long polled = sentinel_bits;
    int pe = -1;  // window entry the outstanding poll is for, -1 = none
    int idle = 1;
    for (long long round = 0; round < max_rounds; ++round) {
        bool progressed = false;
        // (1) the answer to the poll of the previous round (wave-uniform test
        // first: a wave without polls must not wait for its own write-through
        // stores here)
        if (__any(pe >= 0) && pe >= 0) {
            if (polled != sentinel_bits) {
                const double xv = __longlong_as_double(static_cast<long long>(polled));
                far_mask &= ~(1u << pe);
                if (pe == wj) {  // its turn already: the first pass below subtracts it
                    cur_val *= xv;
                    cur_pos = st_fetched;
                } else {
                    s_val[pe][tid] *= xv;
                    s_pos[pe][tid] = st_fetched;
                }
                progressed = true;
            }
            pe = -1;
        }
        // nothing moved last round: wait a little BEFORE asking again, so that
        // the answer is fresh when it is looked at.  (Skipping this sleep for
        // waves that have polls to send is slower -- 992 vs 821 us on the 108^3
        // factor: more polls in the memory queues delay every hand-off.)
        if (idle > 1 && !__any(progressed)) {
            for (int i = 1; i < idle; ++i) __builtin_amdgcn_s_sleep(4);
        }
        // (2) next out-of-chunk dependency of the window: ask memory now, look
        // next round.  (Letting only the wave's earliest waiting row ask while
        // the wave is stuck was tried and is 40x slower: at a grid-line boundary
        // the later rows of a wave are needed long before its earlier ones.)
        const bool ask = far_mask != 0;
        if (ask) {
            pe = __ffs(far_mask) - 1;
            const int ppos = pe == wj ? cur_pos : s_pos[pe][tid];
            const int col = Lower ? ppos : n - 1 - ppos;
            polled = __hip_atomic_load(xb + static_cast<int64_t>(col) * x_stride, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        // (3) LDS-only passes: a chain inside the chunk advances one row per
        // pass, so the pass length IS the hand-off latency -> branch-light
        for (int p = 0; p < lds_passes; ++p) {
            const bool want = cur_pos >= chunk_base;
            if (!__any(want || cur_pos == st_fetched || cur_pos == st_complete)) break;
            const unsigned long long bits =
                __hip_atomic_load(&s_x[want ? cur_pos - chunk_base : tid], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
            bool moved = false;
            bool take = want && bits != sentinel_bits;
            const double prod = cur_val * __longlong_as_double(static_cast<long long>(bits));
            while (take || cur_pos == st_fetched) {  // (the only load_cur() of the loop)
                sum -= take ? prod : cur_val;
                take = false;
                ++wj;
                load_cur();
                moved = true;
            }
            if (cur_pos == st_complete) {
                const double r = unit_diag ? sum : sum / diag;
                unsigned long long out = static_cast<unsigned long long>(__double_as_longlong(r));
                if (out == sentinel_bits) out ^= 1ull;  // a result must never read as "pending"
                __hip_atomic_store(&s_x[tid], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(xb + static_cast<int64_t>(row) * x_stride, out,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cur_pos = st_done;
                moved = true;
            }
            progressed |= moved;
            if (!__any(moved)) break;  // nothing arrived: look at memory again
        }
        if (__all(cur_pos == st_done)) return;
        // back off while nothing moves: thousands of resident waves far behind
        // the dependency front would otherwise flood the memory system with
        // polls (MI355X_MICROARCH.md "polling-cost")
        idle = __any(progressed) ? 1 : min(idle * 2, max_idle);
    }
    if ((tid & 63) == 0) atomicExch(&ws->overrun, 1u);
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
    if (n > INT32_MAX - block) return GKOMI_ENOTSUPPORTED;  // rows are int32 indices
    const int64_t chunks = ceildiv(n, block);
    // GKOMI_TRS_MAX_ROUNDS: test hook that makes a solve give up early (tests/test_trs_ilu_gpu.py)
    const char* env_rounds = getenv("GKOMI_TRS_MAX_ROUNDS");
    const long long max_rounds = env_rounds != nullptr && env_rounds[0] != 0 ? atoll(env_rounds) : default_max_rounds;
    hipStream_t stream = to_stream(s);
    trs_workspace* ws = static_cast<trs_workspace*>(workspace);
    for (int64_t j = 0; j < nrhs; ++j) {
        hipLaunchKernelGGL(trs_prepare_kernel, dim3(grid_for(n, prep_block)), dim3(prep_block), 0,
                           stream, n, x + j, x_stride, ws);
        hipLaunchKernelGGL(trs_syncfree_kernel<Lower>, dim3(static_cast<unsigned>(chunks)),
                           dim3(block), 0, stream, static_cast<int32_t>(n), row_ptrs, col_idxs, vals,
                           unit_diag != 0,
                           b + j, b_stride, x + j, x_stride, ws, max_rounds);
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
