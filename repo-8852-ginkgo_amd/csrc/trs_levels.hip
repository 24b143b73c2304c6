// Sparse triangular solves with an ANALYSIS phase = solver::LowerTrs/UpperTrs::generate
// (the reference runs hipsparseXcsrsv2_analysis there and keeps the result in its
// SolveStruct: hip/solver/common_trs_kernels.hip.hpp:61-253; the CUDA sync-free
// variant cuda/solver/common_trs_kernels.cuh:374-455).  Numerical contract =
// reference/solver/lower_trs_kernels.cpp:90-120, upper_trs_kernels.cpp:90-123.
//
// Why: the analysis-free kernel of trs.hip hands rows to lanes in storage order,
// so the 64 lanes of a wave hold a dependency CHAIN (r needs r-1): one useful
// lane per pass.  Here `generate` computes the dependency level of every row,
// sorts the rows by level (stable) and stores the factor once more in that order
// as 64-row slices, column-major inside a slice (SELL-64 of the dependencies
// only, the diagonal apart).  A wave then owns 64 rows that become ready
// TOGETHER: its loads are coalesced, every lane has work in every pass, and
// the critical path is one memory hand-off per LEVEL.
//
// The solve stays sync-free, in "position space": the plan carries its own copy
// xp of the solution in level order, and the stored column indices are
// POSITIONS in that order.  xp[pos] doubles as its own ready flag (sentinel NaN
// payload, ONE agent-scope store to publish, agent-scope loads to poll --
// MI355X_MICROARCH.md's data-tagged granule); since the dependencies of 64
// consecutive positions are themselves (nearly) consecutive positions of the
// level before, a poll instruction touches a handful of cache lines instead of
// 64.  Slices are handed out by an atomic ticket in level order (every
// dependency of a slice lives in a slice with a smaller-or-equal ticket, i.e. in
// a wave that has started); a lane publishes the moment its row is complete (a
// slice may straddle levels); no lane waits inside a loop another lane of its
// wave must leave; spins are bounded and raise a STICKY flag instead of hanging.
//
// Who polls: many waves are resident, the front is a few of them.  A wave
// far behind the front must not poll its dependencies (the fabric would carry
// nothing else), so it first watches ONE position per wave -- a scout: the last
// row of the level 16 before its own at a 4-us cadence, then the last row of
// the level 2 before its own at a 0.25-us cadence -- and only then polls its
// real dependencies.  Scouts are hints: correctness never depends on them.
// Per row the subtractions run in storage order -> bit-identical to the reference.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

#include "sort_scan.hpp"

namespace gkomi {
namespace {

constexpr int slice = 64;        // rows per wave
constexpr int solve_block = 1024; // 16 slices per ticket
constexpr int default_scout_far = 16;  // levels between a slice and its far / near scout
constexpr int default_scout_near = 2;

// tuning knobs (tools/tune_trs.py): an environment variable overrides the default
inline int tuning(const char* name, int fallback)
{
    const char* v = getenv(name);
    return v != nullptr && v[0] != 0 ? atoi(v) : fallback;
}
constexpr unsigned long long sentinel_bits = 0x7ff8dead0badbeefull;
constexpr long long default_max_rounds = 1ll << 22;

struct plan_header {
    int64_t n;
    int64_t nslices;
    int64_t nlevels;
    int64_t entries;       // SELL slots (multiple of 64)
    unsigned int ticket;
    unsigned int overrun;  // sticky: zeroed when the plan is built, never by a solve
    int32_t lower;
    int32_t pad_;
};
static_assert(sizeof(plan_header) <= 256, "the plan header has 256 bytes");

struct plan_layout {
    size_t perm, slice_off, scout, diag, xp, cols, vals, level_start, total;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

plan_layout make_plan_layout(int64_t nslices, int64_t entries)
{
    plan_layout l{};
    size_t off = 256;
    l.perm = off; off += align_up(sizeof(int32_t) * nslices * slice, 256);
    l.slice_off = off; off += align_up(sizeof(int32_t) * (nslices + 1), 256);
    l.scout = off; off += align_up(sizeof(int32_t) * 2 * (nslices + 1), 256);
    l.diag = off; off += align_up(sizeof(double) * nslices * slice, 256);
    l.xp = off; off += align_up(sizeof(double) * nslices * slice, 256);
    l.cols = off; off += align_up(sizeof(int32_t) * entries, 256);
    l.vals = off; off += align_up(sizeof(double) * entries, 256);
    // first position of every level (nlevels + 1 <= rows + 1 entries): the single-workgroup solve of small factors
    l.level_start = off; off += align_up(sizeof(int32_t) * (nslices * slice + 1), 256);
    l.total = off;
    return l;
}

// symbolic workspace: levels[n] | sorted levels[n] | rows[n] | perm[n] | dep count[n] |
// slice_len[nslices + 1] | slice_off[nslices + 1] | flags | sort / scan scratch
struct symbolic_layout {
    size_t level, level_sorted, rows, perm, cnt, slice_len, slice_off, flags, tmp, tmp_bytes, total;
};

symbolic_layout make_symbolic_layout(int64_t n)
{
    symbolic_layout l{};
    const int64_t nslices = ceildiv(n, slice);
    const size_t vec = align_up(sizeof(int32_t) * static_cast<size_t>(n > 0 ? n : 1), 256);
    size_t off = 0;
    l.level = off; off += vec;
    l.level_sorted = off; off += vec;
    l.rows = off; off += vec;
    l.perm = off; off += vec;
    l.cnt = off; off += vec;
    l.slice_len = off; off += align_up(sizeof(int32_t) * (nslices + 1), 256);
    l.slice_off = off; off += align_up(sizeof(int32_t) * (nslices + 1), 256);
    l.flags = off; off += 256;
    const size_t sort_bytes = radix_sort_workspace_bytes(n > 0 ? n : 1, sizeof(uint32_t), true);
    const size_t scan_bytes = scan_workspace_bytes(nslices + 1);
    l.tmp_bytes = align_up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes, 256) + 256;
    l.tmp = off; off += l.tmp_bytes;
    l.total = off;
    return l;
}

template <bool Lower>
__device__ __forceinline__ bool is_dep(int col, int row)
{
    return Lower ? col < row : col > row;
}

// level[row] = 1 + max level of its dependencies (0 without any): chaotic
// in-place relaxation, monotone from 0, run until a whole batch changes nothing.
// It also leaves the number of dependencies of the row in cnt.
template <bool Lower>
__global__ __launch_bounds__(256) void trs_relax_levels_kernel(
    int32_t n, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    int32_t* level, int32_t* __restrict__ cnt, int32_t* __restrict__ changed)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int begin = row < n ? row_ptrs[row] : 0;
    const int end = row < n ? row_ptrs[row + 1] : 0;
    int mine = row < n ? level[row] : 0;
    bool any_change = false;
    // a wave repeats its sweep while it still moves (the lanes of a wave read
    // each other's levels one sweep late: a dependency chain inside the wave
    // would otherwise need one LAUNCH per link)
    for (int rep = 0; rep < 64; ++rep) {
        int lvl = 0, deps = 0;
        for (int k = begin; k < end; ++k) {
            const int col = col_idxs[k];
            if (is_dep<Lower>(col, row) && col >= 0 && col < n) {
                lvl = max(lvl, __hip_atomic_load(level + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1);
                ++deps;
            }
        }
        const bool moved = row < n && lvl != mine;
        if (row < n && rep == 0) cnt[row] = deps;
        if (moved) {
            mine = lvl;
            __hip_atomic_store(level + row, lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            any_change = true;
        }
        if (!__any(moved)) break;
    }
    if (any_change) *changed = 1;
}

__global__ __launch_bounds__(256) void trs_iota_kernel(int32_t n, int32_t* __restrict__ rows)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) rows[i] = i;
}

// inv[row] = position of the row in level order
__global__ __launch_bounds__(256) void trs_invert_perm_kernel(int32_t n, const int32_t* __restrict__ perm,
                                                             int32_t* __restrict__ inv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}

// one wave per slice: longest dependency list among its rows; entry nslices = 0
__global__ __launch_bounds__(256) void trs_slice_len_kernel(
    int32_t n, int32_t nslices, const int32_t* __restrict__ perm, const int32_t* __restrict__ cnt,
    int32_t* __restrict__ slice_len, const int32_t* __restrict__ level_sorted,
    int32_t* __restrict__ nlevels)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s > nslices) return;
    if (s == nslices) {
        if (lane == 0) {
            slice_len[s] = 0;
            *nlevels = n > 0 ? level_sorted[n - 1] + 1 : 0;
        }
        return;
    }
    const int i = s * slice + lane;
    int m = i < n ? cnt[perm[i]] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if (lane == 0) {
        slice_len[s] = m * slice;
        atomicMax(nlevels + 1, m * slice);  // flags[2]: the longest slice (integer max: order-free)
    }
}

// first position in level order whose level is >= want (n if none)
__device__ __forceinline__ int first_position_of_level(const int32_t* __restrict__ level_sorted, int n, int want)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (level_sorted[mid] >= want) {
            hi = mid;
        } else {
            lo = mid + 1;
        }
    }
    return lo;
}

// numeric phase: the factor once more, rows in level order, dependencies only
// (as POSITIONS in level order), column-major inside every 64-row slice; -1
// pads; the diagonal (last stored occurrence, like the reference's loop) apart.
// scout[2 s], scout[2 s + 1]: the positions slice s watches before it polls its
// dependencies -- the last row of the level `scout_far` / `scout_near` levels
// before the slice's first level (-1: none).
template <bool Lower>
__global__ __launch_bounds__(256) void trs_fill_plan_kernel(
    int32_t n, int32_t nslices, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const int32_t* __restrict__ perm_in, const int32_t* __restrict__ inv,
    const int32_t* __restrict__ slice_off_in, const int32_t* __restrict__ level_sorted,
    int32_t* __restrict__ perm, int32_t* __restrict__ slice_off, int32_t* __restrict__ scout,
    double* __restrict__ diag, int32_t* __restrict__ cols, double* __restrict__ pvals, int scout_far,
    int scout_near)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s > nslices) return;
    if (s == nslices) {
        if (lane == 0) slice_off[s] = slice_off_in[s];
        return;
    }
    const int off = slice_off_in[s];
    const int len = (slice_off_in[s + 1] - off) / slice;
    if (lane == 0) {
        slice_off[s] = off;
        const int lvl = level_sorted[s * slice];
        scout[2 * s] = lvl >= scout_far ? first_position_of_level(level_sorted, n, lvl - scout_far + 1) - 1 : -1;
        scout[2 * s + 1] = lvl >= scout_near ? first_position_of_level(level_sorted, n, lvl - scout_near + 1) - 1 : -1;
    }
    const int i = s * slice + lane;
    const int row = i < n ? perm_in[i] : -1;
    perm[i] = row;
    double d = 1.0;
    int e = 0;
    if (row >= 0) {
        for (int k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const int col = col_idxs[k];
            if (col == row) d = vals[k];
            if (is_dep<Lower>(col, row) && col >= 0 && col < n) {
                cols[off + e * slice + lane] = inv[col];
                pvals[off + e * slice + lane] = vals[k];
                ++e;
            }
        }
    }
    diag[i] = d;
    for (; e < len; ++e) {
        cols[off + e * slice + lane] = -1;
        pvals[off + e * slice + lane] = 0.0;
    }
}

__global__ __launch_bounds__(256) void trs_plan_prepare_kernel(int64_t slots, double* __restrict__ xp,
                                                              plan_header* __restrict__ hdr)
{
    const int64_t gid = blockIdx.x * 256ll + threadIdx.x;
    if (gid == 0) hdr->ticket = 0;
    for (int64_t i = gid; i < slots; i += static_cast<int64_t>(gridDim.x) * 256) {
        reinterpret_cast<unsigned long long*>(xp)[i] = sentinel_bits;
    }
}

__device__ __forceinline__ unsigned long long poll(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Tried and dropped (profiles/r02_trs.log): running the whole solve on ONE XCD (first wave elects
// its XCC_ID, the others leave, persistent waves take slices by ticket) so that the hand-off goes
// through that XCD's L2 -- 557 vs 587 us on the 108^3 factor, both slower than this version: the
// pass a wave repeats per hand-off, not the fabric round trip, is what a level costs, and one
// ticket per slice is 20 k atomics on one word.
// window = dependency entries a lane keeps in registers: 4 when no row of the
// factor has more (the pass a wave repeats per hand-off is then half as long),
// 8 otherwise.
template <int window>
__global__ __launch_bounds__(solve_block) void trs_level_solve_kernel(
    plan_header* hdr, const int32_t* __restrict__ perm, const int32_t* __restrict__ slice_off,
    const int32_t* __restrict__ scout, const double* __restrict__ diag,
    const int32_t* __restrict__ cols, const double* __restrict__ pvals, double* xp, bool unit_diag,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ x, int64_t x_stride,
    int naps_between_polls, long long max_rounds)
{
    const int lane = threadIdx.x & 63;
    // one ticket per workgroup = solve_block / 64 consecutive slices, one per wave (a ticket per
    // wave would put 20 k atomics on one word: ~11 ns each, more than the whole solve should take)
    __shared__ unsigned int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    unsigned long long* xb = reinterpret_cast<unsigned long long*>(xp);
    const int64_t nslices = hdr->nslices;
    bool gave_up = false;
    {
        const int64_t s = static_cast<int64_t>(s_ticket) * (solve_block / slice) + (threadIdx.x >> 6);
        if (s >= nslices) return;
        const int64_t mine = s * slice + lane;  // my position in level order
        const int row = perm[mine];
        const int off = slice_off[s];
        const int len = (slice_off[s + 1] - off) / slice;
        double sum = row >= 0 ? b[row * b_stride] : 0.0;
        const double d = diag[mine];
        bool done = row < 0;
        auto publish = [&]() {
            const double xr = unit_diag ? sum : sum / d;
            // the flag-carrying copy: ONE 8-byte store into the XCD's L2, then the caller's x
            __hip_atomic_store(xb + mine, static_cast<unsigned long long>(__double_as_longlong(xr)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x[row * x_stride] = xr;
            done = true;
        };
        // every lane walks its own row through a register window of `window`
        // dependencies (a wave-wide window would deadlock: a lane of a slice that
        // straddles levels may wait for a longer row of its own wave)
        int c[window];
        double v[window];
        unsigned long long xs[window];
        int next = 0;      // first entry of the row not yet in the window
        int cur = window;  // window position to consume next; window = empty
        auto fill = [&]() {  // coalesced where the lanes walk together: 64 consecutive slots per entry
#pragma unroll
            for (int w = 0; w < window; ++w) {
                const int e = min(next + w, len - 1);
                c[w] = cols[off + e * slice + lane];
                v[w] = pvals[off + e * slice + lane];
                if (next + w >= len) c[w] = -1;
            }
            next += window;
            cur = 0;
        };
        auto ask = [&](bool all) {  // polls of the window, all in flight together
#pragma unroll
            for (int w = 0; w < window; ++w) {
                if (all) {
                    xs[w] = c[w] >= 0 ? poll(xb + c[w]) : 0ull;
                } else if (w >= cur && c[w] >= 0 && xs[w] == sentinel_bits) {
                    xs[w] = poll(xb + c[w]);
                }
            }
        };
        // the matrix data of the first window travels while the wave waits its turn
        if (!done && len > 0) fill();
        // scouts (wave-uniform): far one slowly, near one quickly; bounded, hints only
        {
            const int far = scout[2 * s], near = scout[2 * s + 1];
            if (far >= 0) {
                for (int naps = 0; naps < (1 << 14) && poll(xb + far) == sentinel_bits; ++naps) {
                    __builtin_amdgcn_s_sleep(127);
                }
            }
            if (near >= 0) {
                for (int naps = 0; naps < (1 << 16) && poll(xb + near) == sentinel_bits; ++naps) {
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        if (!done && len > 0) ask(true);
        long long rounds = 0;
        // consume, in storage order, what has arrived; refill / publish on the way
        auto advance = [&]() {
            if (!done && cur == window) {
                if (next >= len) {
                    publish();  // no (further) dependency
                } else {
                    fill();
                    ask(true);
                }
            }
#pragma unroll
            for (int w = 0; w < window; ++w) {
                if (!done && cur == w) {
                    if (c[w] < 0) {  // the row has no further dependency: publish at once
                        publish();
                    } else if (xs[w] != sentinel_bits) {
                        sum -= v[w] * __longlong_as_double(static_cast<long long>(xs[w]));
                        ++cur;
                    }
                }
            }
        };
        while (true) {
            advance();
            if (__all(done)) break;
            if (__all(done || cur == window)) continue;  // only refills / publishes pending: no wait
            if (++rounds > max_rounds) {
                gave_up = true;
                break;
            }
            for (int i = 0; i < naps_between_polls; ++i) __builtin_amdgcn_s_sleep(1);
            if (!done) ask(false);
        }
    }
    if (gave_up && lane == 0) atomicExch(&hdr->overrun, 1u);  // unsolved rows keep the sentinel NaN
}

// level_start[l] = first position (level order) of level l, level_start[nlevels] = n; every level 0 .. nlevels - 1 holds a row
__global__ __launch_bounds__(256) void trs_level_start_kernel(int32_t n, int32_t nlevels, const int32_t* __restrict__ level_sorted,
                                                             int32_t* __restrict__ level_start)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p == 0) level_start[nlevels] = n;
    if (p >= n) return;
    if (p == 0 || level_sorted[p] != level_sorted[p - 1]) level_start[level_sorted[p]] = p;
}

// ---- small factors: the whole solve in ONE workgroup -----------------------------------------------------------------
// A factor of a few thousand rows (the reference's own test matrices: ani4, 3081 rows, ~140 levels of ~22 rows) gives the
// chip-wide kernels nothing to spread: every level is a hand-off through memory (1.2 us each: 170 us per solve).  Here one
// workgroup of 512 threads owns the factor: x lives in LDS in level order, every thread keeps the dependencies of its
// (at most R) rows in registers -- loaded once, coalesced, from the plan's SELL-64 slices -- and a level costs the LDS reads
// of its rows, the division and one workgroup barrier.  Rows of a level are independent; inside a row the subtractions run
// in storage order and the quotient is the IEEE one: the reference's bits (reference/solver/lower_trs_kernels.cpp:90-120).
constexpr int small_block = 512;
constexpr int small_max_rows = 4096;   // 8 positions per thread x up to 8 dependencies each: 192 VGPRs of matrix data

template <int R, int D>
__global__ __launch_bounds__(small_block) void trs_small_solve_kernel(
    const plan_header* __restrict__ hdr, const int32_t* __restrict__ perm, const int32_t* __restrict__ slice_off,
    const double* __restrict__ diag, const int32_t* __restrict__ cols, const double* __restrict__ pvals,
    const int32_t* __restrict__ level_start, bool unit_diag, const double* __restrict__ b, int64_t b_stride,
    double* __restrict__ x, int64_t x_stride)
{
    // LDS: x in level order (it starts as the right-hand side: a row's cell holds b until the row is solved), the
    // diagonal, the first position of every level.  Registers: the dependencies of the thread's R rows -- values as
    // doubles, positions packed two per register (positions < 4096; 0xffff = no dependency): R = 8, D = 8 is 128 + 32
    // registers of matrix data at two waves per SIMD (everything in registers spilled: 0.45 us per level instead of 0.15).
    __shared__ double xs[R * small_block];
    __shared__ double ds[R * small_block];
    __shared__ int32_t ls[R * small_block + 1];
    const int n = static_cast<int>(hdr->n), nlevels = static_cast<int>(hdr->nlevels);
    const int tid = threadIdx.x;
    for (int l = tid; l <= nlevels; l += small_block) ls[l] = level_start[l];
    unsigned cp[R][D / 2];
    double v[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = tid + r * small_block;
#pragma unroll
        for (int e = 0; e < D / 2; ++e) cp[r][e] = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < D; ++e) v[r][e] = 0.0;
        if (p < n) {
            const int s = p >> 6, lane = p & 63;
            const int off = slice_off[s];
            const int len = (slice_off[s + 1] - off) / slice;
#pragma unroll
            for (int e = 0; e < D; ++e) {
                if (e < len) {
                    const int c = cols[off + e * slice + lane];
                    const unsigned half = c >= 0 ? static_cast<unsigned>(c) : 0xffffu;
                    cp[r][e / 2] = (e & 1) ? ((cp[r][e / 2] & 0x0000ffffu) | (half << 16)) : ((cp[r][e / 2] & 0xffff0000u) | half);
                    v[r][e] = pvals[off + e * slice + lane];
                }
            }
            ds[p] = diag[p];
            xs[p] = b[perm[p] * b_stride];
        }
    }
    __syncthreads();
    // (Tried: skipping the workgroup barrier between levels that live in one wave -- LDS executes a wave's instructions in
    // order -- 12 % slower: the bookkeeping costs every wave more than the barriers it saves.)
    const int wave_first = (tid & ~63);   // first position of this wave's block in round r: wave_first + r * small_block
    int lo = 0, hi = nlevels > 0 ? ls[1] : 0;
    for (int l = 0; l < nlevels; ++l) {
        const int hi_next = ls[min(l + 2, nlevels)];   // the next level's bound is asked for now, not behind the barrier
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int first = wave_first + r * small_block;
            if (hi <= first || lo >= first + 64) continue;   // wave-uniform: none of my 64 positions is in this level
            const int p = tid + r * small_block;
            if (p >= lo && p < hi) {
                // all reads go out together (a padding slot reads cell 0 and is dropped: a branch per slot would put one
                // LDS round trip per dependency on the level's critical path)
                double xd[D];
                bool has[D];
#pragma unroll
                for (int e = 0; e < D; ++e) {
                    const unsigned c = (cp[r][e / 2] >> (16 * (e & 1))) & 0xffffu;
                    has[e] = c != 0xffffu;
                    xd[e] = xs[has[e] ? c : 0u];
                }
                double acc = xs[p];
                const double d = ds[p];
                // the products first (independent), then the subtractions in storage order; a padding slot subtracts
                // +0.0, which changes nothing (also not a -0.0): no select in the dependent chain
                double t[D];
#pragma unroll
                for (int e = 0; e < D; ++e) t[e] = has[e] ? v[r][e] * xd[e] : 0.0;
#pragma unroll
                for (int e = 0; e < D; ++e) acc -= t[e];
                // (acc / 1.0 == acc for every acc: the unit diagonal ParILU stores explicitly costs no division)
                xs[p] = (unit_diag || d == 1.0) ? acc : acc / d;
            }
        }
        lo = hi;
        hi = hi_next;
        __syncthreads();
    }
    // x leaves LDS in one sweep (a store inside the loop would make every barrier wait for it)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = tid + r * small_block;
        if (p < n) x[perm[p] * x_stride] = xs[p];
    }
}

template <int R>
void launch_small(hipStream_t stream, int64_t max_deps, const plan_header* hdr, const plan_layout& pl, char* p, bool unit_diag,
                  const double* b, int64_t b_stride, double* x, int64_t x_stride)
{
#define GKOMI_SMALL(DEPS)                                                                                                  \
    hipLaunchKernelGGL((trs_small_solve_kernel<R, DEPS>), dim3(1), dim3(small_block), 0, stream, hdr,                       \
                       reinterpret_cast<const int32_t*>(p + pl.perm), reinterpret_cast<const int32_t*>(p + pl.slice_off),  \
                       reinterpret_cast<const double*>(p + pl.diag), reinterpret_cast<const int32_t*>(p + pl.cols),         \
                       reinterpret_cast<const double*>(p + pl.vals), reinterpret_cast<const int32_t*>(p + pl.level_start),  \
                       unit_diag, b, b_stride, x, x_stride)
    if (max_deps <= 4) {
        GKOMI_SMALL(4);
    } else {
        GKOMI_SMALL(8);
    }
#undef GKOMI_SMALL
}

template <bool Lower>
int analyse_symbolic(hipStream_t stream, int64_t n, const int32_t* row_ptrs, const int32_t* col_idxs,
                     void* workspace, size_t workspace_bytes, int64_t* host_out)
{
    const symbolic_layout l = make_symbolic_layout(n);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    char* ws = static_cast<char*>(workspace);
    int32_t* level = reinterpret_cast<int32_t*>(ws + l.level);
    int32_t* level_sorted = reinterpret_cast<int32_t*>(ws + l.level_sorted);
    int32_t* rows = reinterpret_cast<int32_t*>(ws + l.rows);
    int32_t* perm = reinterpret_cast<int32_t*>(ws + l.perm);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws + l.cnt);
    int32_t* slice_len = reinterpret_cast<int32_t*>(ws + l.slice_len);
    int32_t* slice_off = reinterpret_cast<int32_t*>(ws + l.slice_off);
    int32_t* flags = reinterpret_cast<int32_t*>(ws + l.flags);  // [0] changed, [1] nlevels
    const int32_t n32 = static_cast<int32_t>(n);
    const int32_t nslices = static_cast<int32_t>(ceildiv(n, slice));
    host_out[0] = host_out[1] = host_out[2] = host_out[3] = 0;
    if (n == 0) return GKOMI_SUCCESS;
    int err = static_cast<int>(hipMemsetAsync(level, 0, sizeof(int32_t) * n, stream));
    if (err) return err;
    const dim3 grid(static_cast<unsigned>(ceildiv(n, 256)));
    // the fixed point is reached after at most (longest dependency path) sweeps;
    // a sweep usually settles many levels (rows are visited roughly in order)
    constexpr int batch = 16;
    for (int64_t sweeps = 0; sweeps <= n + batch; sweeps += batch) {
        err = static_cast<int>(hipMemsetAsync(flags, 0, 4 * sizeof(int32_t), stream));
        if (err) return err;
        for (int i = 0; i < batch; ++i) {
            hipLaunchKernelGGL(trs_relax_levels_kernel<Lower>, grid, dim3(256), 0, stream, n32,
                               row_ptrs, col_idxs, level, cnt, flags);
        }
        int32_t changed = 0;
        err = static_cast<int>(hipMemcpyAsync(&changed, flags, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        if (err) return err;
        err = static_cast<int>(hipStreamSynchronize(stream));
        if (err) return err;
        if (!changed) break;
    }
    hipLaunchKernelGGL(trs_iota_kernel, grid, dim3(256), 0, stream, n32, rows);
    size_t tmp_bytes = l.tmp_bytes;
    // stable: rows of one level keep their storage order (deterministic layout)
    // (levels are non-negative: their order as unsigned 32-bit keys is their order; sort_scan.hip)
    err = radix_sort_u32(stream, n, reinterpret_cast<const uint32_t*>(level), reinterpret_cast<uint32_t*>(level_sorted),
                         reinterpret_cast<const uint32_t*>(rows), reinterpret_cast<uint32_t*>(perm), 32, ws + l.tmp, tmp_bytes);
    if (err) return err;
    hipLaunchKernelGGL(trs_invert_perm_kernel, grid, dim3(256), 0, stream, n32, perm, rows);  // rows := inverse
    hipLaunchKernelGGL(trs_slice_len_kernel, dim3(static_cast<unsigned>(ceildiv(nslices + 1, 4))), dim3(256),
                       0, stream, n32, nslices, perm, cnt, slice_len, level_sorted, flags + 1);
    tmp_bytes = l.tmp_bytes;
    err = exclusive_sum_i32(stream, slice_len, slice_off, nslices + 1, ws + l.tmp, tmp_bytes);
    if (err) return err;
    int32_t h[3] = {0, 0, 0};
    err = static_cast<int>(hipMemcpyAsync(&h[0], slice_off + nslices, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipMemcpyAsync(&h[1], flags + 1, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipMemcpyAsync(&h[2], flags + 2, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    if (err) return err;
    if (h[0] < 0) return GKOMI_ENOTSUPPORTED;  // more than 2^31 slots
    host_out[0] = nslices;
    host_out[1] = h[0];
    host_out[2] = h[1];
    host_out[3] = h[2] / slice;  // longest dependency list of a row
    return check_launch();
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_trs_symbolic_workspace_bytes(int64_t n)
{
    if (n < 0 || n > INT32_MAX - 1024) return 0;
    return make_symbolic_layout(n).total;
}

extern "C" int gkomi_trs_analyse_symbolic_i32(gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
                                              const int32_t* col_idxs, int lower, void* workspace,
                                              size_t workspace_bytes, int64_t* host_out)
{
    if (n < 0 || host_out == nullptr) return GKOMI_EINVAL;
    if (n > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    return lower ? analyse_symbolic<true>(to_stream(s), n, row_ptrs, col_idxs, workspace, workspace_bytes, host_out)
                 : analyse_symbolic<false>(to_stream(s), n, row_ptrs, col_idxs, workspace, workspace_bytes, host_out);
}

extern "C" size_t gkomi_trs_plan_bytes(int64_t nslices, int64_t entries)
{
    if (nslices < 0 || entries < 0) return 0;
    return make_plan_layout(nslices, entries).total;
}

extern "C" int gkomi_trs_analyse_numeric_f64_i32(gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
                                                 const int32_t* col_idxs, const double* vals, int lower,
                                                 const void* symbolic_workspace, int64_t nslices,
                                                 int64_t entries, int64_t nlevels, void* plan,
                                                 size_t plan_bytes)
{
    if (n < 0 || nslices != ceildiv(n, slice) || entries < 0 || entries % slice != 0) return GKOMI_EINVAL;
    const plan_layout pl = make_plan_layout(nslices, entries);
    if (plan == nullptr || plan_bytes < pl.total || symbolic_workspace == nullptr) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    plan_header h{};
    h.n = n; h.nslices = nslices; h.nlevels = nlevels; h.entries = entries;
    h.ticket = 0; h.overrun = 0; h.lower = lower ? 1 : 0;
    int err = static_cast<int>(hipMemcpyAsync(plan, &h, sizeof(h), hipMemcpyHostToDevice, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));  // h lives on this stack frame
    if (err || n == 0) return err;
    const symbolic_layout sl = make_symbolic_layout(n);
    const char* sw = static_cast<const char*>(symbolic_workspace);
    char* p = static_cast<char*>(plan);
    const dim3 grid(static_cast<unsigned>(ceildiv(nslices + 1, 4)));
#define GKOMI_FILL(LOWER)                                                                             \
    hipLaunchKernelGGL(trs_fill_plan_kernel<LOWER>, grid, dim3(256), 0, stream, static_cast<int32_t>(n), \
                       static_cast<int32_t>(nslices), row_ptrs, col_idxs, vals,                         \
                       reinterpret_cast<const int32_t*>(sw + sl.perm),                                  \
                       reinterpret_cast<const int32_t*>(sw + sl.rows),                                  \
                       reinterpret_cast<const int32_t*>(sw + sl.slice_off),                             \
                       reinterpret_cast<const int32_t*>(sw + sl.level_sorted),                          \
                       reinterpret_cast<int32_t*>(p + pl.perm), reinterpret_cast<int32_t*>(p + pl.slice_off), \
                       reinterpret_cast<int32_t*>(p + pl.scout),                                        \
                       reinterpret_cast<double*>(p + pl.diag), reinterpret_cast<int32_t*>(p + pl.cols),  \
                       reinterpret_cast<double*>(p + pl.vals), tuning("GKOMI_TRS_FAR", default_scout_far),  \
                       tuning("GKOMI_TRS_NEAR", default_scout_near))
    if (lower) GKOMI_FILL(true); else GKOMI_FILL(false);
#undef GKOMI_FILL
    err = check_launch();
    if (err) return err;
    hipLaunchKernelGGL(trs_level_start_kernel, dim3(static_cast<unsigned>(ceildiv(n, 256))), dim3(256), 0, stream,
                       static_cast<int32_t>(n), static_cast<int32_t>(nlevels), reinterpret_cast<const int32_t*>(sw + sl.level_sorted),
                       reinterpret_cast<int32_t*>(p + pl.level_start));
    return check_launch();
}

extern "C" int gkomi_trs_solve_plan_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, void* plan,
                                        int64_t nslices, int64_t entries, int64_t max_deps, int unit_diag,
                                        const double* b, int64_t b_stride, double* x, int64_t x_stride)
{
    if (n < 0 || nrhs < 0 || b_stride < nrhs || x_stride < nrhs) return GKOMI_EINVAL;
    if (n == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (plan == nullptr || nslices != ceildiv(n, slice)) return GKOMI_EINVAL;
    if (x == b) return GKOMI_EINVAL;  // x carries the ready flags
    const plan_layout pl = make_plan_layout(nslices, entries);
    char* p = static_cast<char*>(plan);
    plan_header* hdr = reinterpret_cast<plan_header*>(p);
    hipStream_t stream = to_stream(s);
    // a factor of a few thousand rows: one workgroup, x in LDS (see trs_small_solve_kernel); GKOMI_TRS_SMALL=0: tuning hook
    static const bool small_on = [] {
        const char* e = getenv("GKOMI_TRS_SMALL");
        return e == nullptr || e[0] != '0';
    }();
    if (small_on && n <= small_max_rows && max_deps >= 0 && max_deps <= 8) {
        for (int64_t j = 0; j < nrhs; ++j) {
            if (n <= 1 * small_block) {
                launch_small<1>(stream, max_deps, hdr, pl, p, unit_diag != 0, b + j, b_stride, x + j, x_stride);
            } else if (n <= 2 * small_block) {
                launch_small<2>(stream, max_deps, hdr, pl, p, unit_diag != 0, b + j, b_stride, x + j, x_stride);
            } else if (n <= 4 * small_block) {
                launch_small<4>(stream, max_deps, hdr, pl, p, unit_diag != 0, b + j, b_stride, x + j, x_stride);
            } else {
                launch_small<8>(stream, max_deps, hdr, pl, p, unit_diag != 0, b + j, b_stride, x + j, x_stride);
            }
            const int err = check_launch();
            if (err) return err;
        }
        return GKOMI_SUCCESS;
    }
    const unsigned groups = static_cast<unsigned>(ceildiv(nslices, solve_block / slice));
    const char* env_rounds = getenv("GKOMI_TRS_MAX_ROUNDS");
    const long long max_rounds = env_rounds != nullptr && env_rounds[0] != 0 ? atoll(env_rounds) : default_max_rounds;
    const int naps = tuning("GKOMI_TRS_NAP", 1);
    for (int64_t j = 0; j < nrhs; ++j) {
        hipLaunchKernelGGL(trs_plan_prepare_kernel, dim3(grid_for(nslices * slice, 256)), dim3(256), 0, stream,
                           nslices * slice, reinterpret_cast<double*>(p + pl.xp), hdr);
#define GKOMI_SOLVE(WINDOW)                                                                                   \
    hipLaunchKernelGGL((trs_level_solve_kernel<WINDOW>), dim3(groups), dim3(solve_block), 0, stream, hdr,      \
                       reinterpret_cast<const int32_t*>(p + pl.perm),                                         \
                       reinterpret_cast<const int32_t*>(p + pl.slice_off),                                    \
                       reinterpret_cast<const int32_t*>(p + pl.scout),                                        \
                       reinterpret_cast<const double*>(p + pl.diag),                                          \
                       reinterpret_cast<const int32_t*>(p + pl.cols),                                         \
                       reinterpret_cast<const double*>(p + pl.vals), reinterpret_cast<double*>(p + pl.xp),    \
                       unit_diag != 0, b + j, b_stride, x + j, x_stride, naps, max_rounds)
        if (max_deps >= 0 && max_deps <= 4) {
            GKOMI_SOLVE(4);
        } else {
            GKOMI_SOLVE(8);
        }
#undef GKOMI_SOLVE
        const int err = check_launch();
        if (err) return err;
    }
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_trs_plan_check_overrun(gkomi_stream_t s, const void* plan, int* host_flag)
{
    if (plan == nullptr || host_flag == nullptr) return GKOMI_EINVAL;
    plan_header h{};
    hipStream_t stream = to_stream(s);
    int err = static_cast<int>(hipMemcpyAsync(&h, plan, sizeof(h), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    *host_flag = static_cast<int>(h.overrun);
    return err;
}


// Which solve `generate` should prepare for a factor of n rows whose symbolic analysis found nlevels levels and rows of at
// most max_deps dependencies: 1 = the level plan (gkomi_trs_analyse_numeric + gkomi_trs_solve_plan: level-scheduled
// waves for wide levels, ONE workgroup with x in LDS for factors of up to 4096 rows), 0 = the analysis-free kernel
// (chains and narrow bands of large factors: in-workgroup LDS hand-offs).  One rule for the shims, the mirror and
// gkomi.solvers.
extern "C" int64_t gkomi_trs_use_plan(int64_t n, int64_t nlevels, int64_t max_deps)
{
    if (n <= 0) return 0;
    if (n <= small_max_rows && max_deps >= 0 && max_deps <= 8) return 1;
    return n >= 64 * (nlevels > 0 ? nlevels : 1) ? 1 : 0;
}

// ... and between the brick plan of a box-grid factor (levels_estimate levels, coarse_levels brick-to-brick hand-offs on
// the longest path: gkomi_trs_bricks_levels_estimate / info[1]) and the level plan: us per level inside bricks 0.17, per
// hand-off 5.0, per level of the level plan 1.7 (profiles/r02_trs_bricks.md) -- or 0.2 when the factor is small enough for
// the single-workgroup solve (profiles/r04_trs_small.log).
extern "C" int64_t gkomi_trs_prefer_bricks(int64_t n, int64_t levels_estimate, int64_t coarse_levels)
{
    if (levels_estimate <= 16) return 0;
    const double per_level = n <= small_max_rows ? 0.5 : 1.7;   // measured: profiles/r04_trs_small.log
    return 0.17 * levels_estimate + 5.0 * coarse_levels < per_level * levels_estimate ? 1 : 0;
}
