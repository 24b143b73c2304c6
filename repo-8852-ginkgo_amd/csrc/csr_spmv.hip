// CSR SpMV for gfx950.  Replaces gko::kernels::hip::csr::{spmv,
// advanced_spmv} (core/matrix/csr_kernels.hpp:58-75); numerical contract =
// reference/matrix/csr_kernels.cpp:75-128.
//
// Two kernels:
//
//  * stream  -- a workgroup owns a contiguous block of rows, streams that
//    block's (vals, col_idxs) range with 16-B/8-B per-lane coalesced loads,
//    gathers b[col], writes the per-nonzero products into an LDS tile and
//    then lets one thread per row add its row's products LEFT TO RIGHT.  No
//    atomics, no tree: the summation order is the reference's
//    (`c += val*b` in storage order, unfused), so the result is bit-identical
//    to the reference executor for any row-length distribution.  Rows longer
//    than the tile are carried across tiles in a register.
//
//  * vector  -- Ginkgo's "classical" idea re-cut for 64-wide waves: a
//    sub-wave of 2..64 lanes per row, strided over the row, xor-shuffle tree.
//    Used for long rows, where one thread adding a whole row serialises.
//
// HBM traffic per row of the 5-pt stencil: 5*(8+4) + 4 + 8 (b, once, the rest
// from L2) + 8 = 80 B.
#include "common.hpp"

#include <type_traits>

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <atomic>
#include <mutex>

#include "internal.hpp"

namespace gkomi {
namespace {

constexpr int num_xcd = 8;
// over-read of the nonzero-split kernel: rows up to split_max_over + 1 entries
// are summed from LDS alone
constexpr int split_max_over = 64;
// most rows that may start in one tile before the nonzero-split kernel hands rows out by index (see the kernel)
constexpr int split_rows_limit = 2048;
// load-balanced kernel: row segments longer than this are reduced by a whole wave
constexpr int balanced_coop_min = 128;

// Blocks are dealt round-robin to the 8 XCDs (MI355X_MICROARCH.md §Workgroup
// dispatch).  Give each XCD one contiguous chunk of row blocks so that the
// b entries shared by neighbouring row blocks hit in that XCD's L2.  The
// launch grid is 8*per, surplus ids exit; pure speed, never correctness.
__device__ __forceinline__ int xcd_chunked_block(int bid, int per)
{
    // groups of 8*per consecutive ids; inside a group XCD k owns `per`
    // consecutive logical blocks.  per = ceil(nblocks / 8) gives every XCD one
    // contiguous eighth of the matrix; a smaller per keeps the 8 XCDs streaming
    // from neighbouring addresses.
    const int group = bid / (num_xcd * per);
    const int w = bid - group * num_xcd * per;
    return group * num_xcd * per + (w % num_xcd) * per + w / num_xcd;
}

// Dot = true adds the CG epilogue: partial[logical block] = sum over the
// block's rows of b[row] * c[row] (fixed-order tree, no atomics), and the whole
// launch is skipped when stop_status[0] says the solver has stopped.
typedef double nt_double2 __attribute__((ext_vector_type(2)));
typedef int nt_int2 __attribute__((ext_vector_type(2)));
// pairs of the nonzero-split kernel: 16/8 bytes per lane, but only element-aligned
// when the pair was pulled back to the end of the arrays
typedef double split_double2 __attribute__((ext_vector_type(2), aligned(8)));
typedef int split_int2 __attribute__((ext_vector_type(2), aligned(4)));

// Index types of the boundary's instantiations that exist here (include/ginkgo/core/base/types.hpp:544-560:
// {int32, int64}): positions in the nonzero arrays are `pos` (int for int32 matrices: the address arithmetic of
// the kernel of record stays 32-bit; int64_t for int64 ones: nnz > 2^31 on a 288 GB part), a pair of column
// indices is 8 or 16 bytes.
template <typename I>
struct index_traits;
template <>
struct index_traits<int32_t> {
    using pos = int;
    typedef int vec2 __attribute__((ext_vector_type(2)));                  // aligned pairs (stream kernel)
    typedef int split_vec2 __attribute__((ext_vector_type(2), aligned(4)));  // element-aligned pairs (split kernel)
};
template <>
struct index_traits<int64_t> {
    using pos = int64_t;
    typedef long vec2 __attribute__((ext_vector_type(2)));
    typedef long split_vec2 __attribute__((ext_vector_type(2), aligned(8)));
};
template <typename I>
struct index_pair {
    I x, y;
};

// Pad: one spare LDS slot per 32 products, so that rows of even length (stride
// 8, 16, 32 doubles between neighbouring lanes in the row-sum phase) spread
// over the banks instead of hitting the same one (2-way instead of 32-way).
template <bool Pad>
__device__ __forceinline__ int lds_slot(int i)
{
    return Pad ? i + (i >> 5) : i;
}

template <typename I, int Block, int RowsPerThread, int Tile, bool Advanced, bool Swizzle,
          bool Dot = false, bool NT = false, bool Pad = false>
__global__ __launch_bounds__(Block) void csr_stream_kernel(
    I nrows, const I* __restrict__ row_ptrs,
    const I* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p,
    const double* __restrict__ beta_p, int nblocks, int per_xcd,
    double* __restrict__ dot_partial = nullptr,
    const uint8_t* __restrict__ stop_status = nullptr,
    const double* __restrict__ dot_w = nullptr,
    double* __restrict__ dot_partial2 = nullptr)
{
    using pos_t = typename index_traits<I>::pos;
    using ivec2 = typename index_traits<I>::vec2;
    constexpr int rows_per_block = Block * RowsPerThread;
    constexpr int pairs = Tile / (2 * Block);
    static_assert(Tile % (2 * Block) == 0, "tile must be a whole number of pair sweeps");
    __shared__ __attribute__((aligned(16))) double prod[Pad ? Tile + Tile / 32 : Tile];

    const int logical =
        Swizzle ? xcd_chunked_block(blockIdx.x, per_xcd) : blockIdx.x;
    if (logical >= nblocks) return;
    if (Dot && status_has_stopped_uniform(stop_status)) return;
    // rhs column handled by this grid row
    b += blockIdx.y;
    c += blockIdx.y;

    const int tid = threadIdx.x;
    const I r0 = static_cast<I>(logical) * rows_per_block;
    const I r1 = min(r0 + rows_per_block, nrows);
    const pos_t p0 = row_ptrs[r0];
    const pos_t p1 = row_ptrs[r1];
    const pos_t nnz_total = row_ptrs[nrows];

    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }

    pos_t ra[RowsPerThread], rb[RowsPerThread];
    double sum[RowsPerThread];
    // Dot: partials of w . c (w = b unless given: CG's p.q, BiCGSTAB's rr.v and s.t) and, on request, of c . c
    // (BiCGSTAB's t.t).  w's entries for my rows are asked for up here, not at the end of the workgroup.
    const double* w = Dot && dot_w != nullptr ? dot_w : b;
    const int64_t w_stride = Dot && dot_w != nullptr ? 1 : b_stride;
    double wv[RowsPerThread];
#pragma unroll
    for (int i = 0; i < RowsPerThread; ++i) {
        const I row = r0 + tid + i * Block;
        wv[i] = 0.0;
        if (row < r1) {
            ra[i] = row_ptrs[row];
            rb[i] = row_ptrs[row + 1];
            // advanced: c = beta*c first, then accumulate (reference :119-126)
            sum[i] = Advanced ? c[row * c_stride] * beta : 0.0;
            if (Dot) wv[i] = w[row * w_stride];
        } else {
            ra[i] = rb[i] = p1;
            sum[i] = 0.0;
        }
    }

    for (pos_t t0 = p0 & ~pos_t{1}; t0 < p1; t0 += Tile) {
        double2 v[pairs];
        index_pair<I> ci[pairs];
        // 1) issue every streaming load of this tile
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            const pos_t k = t0 + 2 * (tid + u * Block);
            v[u] = make_double2(0.0, 0.0);
            ci[u] = index_pair<I>{0, 0};
            if (k < p1) {
                if (k + 1 < nnz_total) {
                    if (NT) {  // read-once streams: do not keep them in the caches b lives in
                        const nt_double2 tv = __builtin_nontemporal_load(
                            reinterpret_cast<const nt_double2*>(vals + k));
                        const ivec2 tc = __builtin_nontemporal_load(
                            reinterpret_cast<const ivec2*>(col_idxs + k));
                        v[u] = make_double2(tv.x, tv.y);
                        ci[u] = index_pair<I>{static_cast<I>(tc.x), static_cast<I>(tc.y)};
                    } else {
                        v[u] = *reinterpret_cast<const double2*>(vals + k);
                        const ivec2 tc = *reinterpret_cast<const ivec2*>(col_idxs + k);
                        ci[u] = index_pair<I>{static_cast<I>(tc.x), static_cast<I>(tc.y)};
                    }
                } else {
                    v[u].x = vals[k];
                    ci[u].x = col_idxs[k];
                }
            }
        }
        // 2) gather b, branch-free so that all gathers are in flight
        //    together: every ci is either a loaded (valid) column -- possibly
        //    of a neighbouring block's row -- or 0, so the load is always in
        //    bounds; products outside [p0, p1) land in LDS slots nobody reads.
        double2 xv[pairs];
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            xv[u].x = b[ci[u].x * b_stride];
            xv[u].y = b[ci[u].y * b_stride];
        }
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            double2 pr;
            if (Advanced) {
                // reference order: (valpha * val) * b
                pr.x = (alpha * v[u].x) * xv[u].x;
                pr.y = (alpha * v[u].y) * xv[u].y;
            } else {
                pr.x = v[u].x * xv[u].x;
                pr.y = v[u].y * xv[u].y;
            }
            if (Pad) {  // a pair never straddles a pad slot, but loses the 16-B alignment
                const int slot = lds_slot<true>(2 * (tid + u * Block));
                prod[slot] = pr.x;
                prod[slot + 1] = pr.y;
            } else {
                *reinterpret_cast<double2*>(prod + 2 * (tid + u * Block)) = pr;
            }
        }
        __syncthreads();
        // 3) one thread per row: add this tile's part of the row in order
        const pos_t t1 = t0 + Tile;
#pragma unroll
        for (int i = 0; i < RowsPerThread; ++i) {
            const int lo = static_cast<int>(max(ra[i], t0) - t0);
            const int hi = static_cast<int>(min(rb[i], t1) - t0);
            for (int k = lo; k < hi; ++k) {
                sum[i] += prod[lds_slot<Pad>(k)];
            }
        }
        if (t1 < p1) __syncthreads();
    }

    double pq = 0.0, qq = 0.0;
#pragma unroll
    for (int i = 0; i < RowsPerThread; ++i) {
        const I row = r0 + tid + i * Block;
        if (row < r1) {
            c[row * c_stride] = sum[i];
            if (Dot) {
                pq += wv[i] * sum[i];
                qq += sum[i] * sum[i];
            }
        }
    }
    if (Dot) {
        __shared__ double red[Block / wave_size];
        __syncthreads();
        const double total = block_reduce_sum<Block>(pq, red);
        if (tid == 0) dot_partial[logical] = total;
        if (dot_partial2 != nullptr) {
            __syncthreads();
            const double total2 = block_reduce_sum<Block>(qq, red);
            if (tid == 0) dot_partial2[logical] = total2;
        }
    }
}


// sum += prod[lo .. hi) left to right (hi <= cap, the number of valid LDS slots),
// eight LDS reads in flight per step instead of one read per dependent add
__device__ __forceinline__ double add_products(double sum, const double* prod, int lo, int hi, int cap)
{
    for (int k = lo; k < hi; k += 8) {
        double p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = prod[min(k + u, cap - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum = k + u < hi ? sum + p[u] : sum;
    }
    return sum;
}

// ---- nonzero-split stream kernel (needs srow) ---------------------------------
//
// The stream kernel above cannot issue a single streaming load before
// row_ptrs[r0] has come back: three dependent memory round trips (row_ptrs ->
// vals/col_idxs -> b) per workgroup, which is what a cold 15-us launch spends
// its fill and drain on.  Here the work is cut by NONZEROS: workgroup t owns the
// rows that START inside nonzeros [t*Tile, (t+1)*Tile), so the addresses of its
// streaming loads depend on nothing but blockIdx and go out first.  Which rows
// those are comes from `srow` (srow[t] = first row with row_ptrs[row] >=
// t*Tile), built once per matrix -- the role of Csr::srow_ of the reference's
// load_balance strategy (csr.hpp:395-459, 1139-1157).  A row that starts in the
// tile may end behind it: the workgroup also reads the next `over` nonzeros
// (over >= longest row - 1; the kernel is selected for short rows only), so
// every row is summed by ONE thread, left to right, from its own products:
// bit-identical to the reference for any row lengths, no atomics, no carries.
// col_idxs are requested before vals: vector-memory results return in order,
// so the gathers of b start while the values are still in flight.
// tools/dot_probe.hip only: leaves parts of the dot epilogue out (bit 0 the status check, bit 1 the load of the other
// factor, bit 2 the reduction across the workgroup) to price them one by one.  0 in the product.
#ifndef GKOMI_DOT_PROBE
#define GKOMI_DOT_PROBE 0
#endif
template <typename I, int Block, int Tile, int MaxOver, bool Advanced, bool Swizzle,
          bool Dot = false, bool NT = false, bool ColsFirst = true>
__global__ __launch_bounds__(Block) void csr_split_kernel(
    I nrows, typename index_traits<I>::pos nnz, const I* __restrict__ row_ptrs,
    const I* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p,
    const double* __restrict__ beta_p, const I* __restrict__ srow,
    int ntiles, int per_xcd, int over,
    double* __restrict__ dot_partial = nullptr,
    const uint8_t* __restrict__ stop_status = nullptr,
    const double* __restrict__ dot_w = nullptr,
    double* __restrict__ dot_partial2 = nullptr
#ifdef GKOMI_TIMELINE
    , unsigned long long* __restrict__ stamps = nullptr
#endif
    )
{
    using pos_t = typename index_traits<I>::pos;
    using isplit2 = typename index_traits<I>::split_vec2;
    constexpr int pairs = Tile / (2 * Block);
    static_assert(Tile % (2 * Block) == 0, "tile must be a whole number of pair sweeps");
    static_assert(MaxOver <= 2 * Block && MaxOver % 2 == 0, "one extra pair per lane at most");
    // tools/spmv_timeline.hip only: per-workgroup phase stamps (100 MHz
    // s_memrealtime, comparable across CUs) into a buffer nothing else reads
#ifdef GKOMI_TIMELINE
#define GKOMI_STAMP(slot)                                                           \
    do {                                                                            \
        if (stamps != nullptr && threadIdx.x == 0) {                                \
            asm volatile("" ::: "memory");                                          \
            stamps[8 * (Swizzle ? xcd_chunked_block(blockIdx.x, per_xcd) : blockIdx.x) + (slot)] = \
                __builtin_amdgcn_s_memrealtime();                                   \
            asm volatile("" ::: "memory");                                          \
        }                                                                           \
    } while (0)
#else
#define GKOMI_STAMP(slot) do { } while (0)
#endif
    __shared__ __attribute__((aligned(16))) double prod[Tile + MaxOver];

    const int logical =
        Swizzle ? xcd_chunked_block(blockIdx.x, per_xcd) : blockIdx.x;
    if (logical >= ntiles) return;
    b += blockIdx.y;
    c += blockIdx.y;
    const int tid = threadIdx.x;
    const pos_t t0 = static_cast<pos_t>(logical) * Tile;
    GKOMI_STAMP(0);
#ifdef GKOMI_TIMELINE
    if (stamps != nullptr && threadIdx.x == 0) {
        stamps[8 * logical + 5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID[3:0]
        stamps[8 * logical + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    }
#endif
    // a solve that has stopped needs nothing from this launch; asked through the scalar cache (a byte load queues in
    // the vector-memory pipe in front of every workgroup's streaming loads: 25 us of a 323-us launch on the 256^3
    // matrix -- wherever the check stands in the source, the compiler sinks the loads behind its branch)
    if (Dot && !(GKOMI_DOT_PROBE & 1) && status_has_stopped_uniform(stop_status)) return;
    // the rows that start in this tile (scalar loads, back long before the
    // streaming loads below)
    // (the top bit of every srow entry: the matrix is in sparse-rows mode, below)
    constexpr I srow_mark = static_cast<I>(static_cast<typename std::make_unsigned<I>::type>(1) << (8 * sizeof(I) - 1));
    const I srow_first = srow[logical];
    const I row_begin = srow_first & ~srow_mark;
    const I row_end = srow[logical + 1] & ~srow_mark;
    // srow's last entry: the most rows that start in any one tile (gkomi_csr_make_srow).  A tile of a matrix with long
    // runs of EMPTY rows may own hundreds of thousands of them (the non-local block of a distributed matrix, a selection
    // matrix): its workgroup would walk them 256 at a time, two dependent loads each -- 1.3 ms on a matrix whose other
    // tiles take 50 us together.  Beyond split_rows_limit, rows behind the first 2 * Block of their tile are handed
    // out by ROW INDEX instead (step 5 below): every workgroup looks at an equal share of the rows.  make_srow marks
    // EVERY entry's top bit then -- no load beyond the two every workgroup needs anyway (reading the statistic itself
    // here cost the 1M-row matrix's cold apply 1 %: 16.68 -> 16.85 us).
    const bool sparse_rows = srow_first < 0;

    // 1) streaming loads: addresses known from the block index alone.  Branch-
    //    free (a branch per load makes the compiler drain the loads before it):
    //    a pair that would start at or behind the last nonzero reads the last
    //    two nonzeros instead -- valid columns, products nobody adds -- and the
    //    lane that owns nonzero nnz-1 of an odd nnz picks it out of that pair.
    double2 v[pairs + 1];
    index_pair<I> ci[pairs + 1];
    auto load_cols = [&](int u, pos_t k) {
        const isplit2* src = reinterpret_cast<const isplit2*>(col_idxs + min(k, nnz - 2));
        const isplit2 tc = NT ? __builtin_nontemporal_load(src) : *src;
        ci[u] = index_pair<I>{static_cast<I>(k == nnz - 1 ? tc.y : tc.x), static_cast<I>(tc.y)};
    };
    auto load_vals = [&](int u, pos_t k) {
        const split_double2* src = reinterpret_cast<const split_double2*>(vals + min(k, nnz - 2));
        const split_double2 tv = NT ? __builtin_nontemporal_load(src) : *src;
        v[u] = make_double2(k == nnz - 1 ? tv.y : tv.x, tv.y);
    };
    // the pairs behind the tile (rows that start in the tile and end behind
    // it); lanes past `over` repeat the last useful pair (one request)
    const pos_t k_over = t0 + Tile + 2 * min(tid, max(over / 2 - 1, 0));
    if (ColsFirst) {
#pragma unroll
        for (int u = 0; u < pairs; ++u) load_cols(u, t0 + 2 * (tid + u * Block));
        load_cols(pairs, k_over);
#pragma unroll
        for (int u = 0; u < pairs; ++u) load_vals(u, t0 + 2 * (tid + u * Block));
        load_vals(pairs, k_over);
    } else {
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            load_vals(u, t0 + 2 * (tid + u * Block));
            load_cols(u, t0 + 2 * (tid + u * Block));
        }
        load_vals(pairs, k_over);
        load_cols(pairs, k_over);
    }

    // 2) row_ptrs of the rows that start in this tile
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // two rounds of rows are kept in registers (a 1536-nonzero tile of the
    // 5-pt stencil starts 307 rows); more rounds re-read row_ptrs on demand
    pos_t ra[2], rb[2];
    double c0[2];
    // dot epilogue: the other factor of w . (A b) for my rows, asked for NOW -- behind the barrier it is one more
    // dependent round trip at the end of every workgroup (28 us of a 326-us launch on the 256^3 matrix,
    // profiles/r03_p3_cg_kernels.md)
    const double* w = Dot && dot_w != nullptr ? dot_w : b;
    const int64_t w_stride = Dot && dot_w != nullptr ? 1 : b_stride;
    double wv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // (kept under its own branch up here: loaded unconditionally, the
        // compiler sinks the loads into the row loop behind the barrier)
        const I row = row_begin + tid + i * Block;
        ra[i] = rb[i] = t0;
        c0[i] = 0.0;
        wv[i] = 0.0;
        if (row < row_end) {
            ra[i] = NT ? __builtin_nontemporal_load(row_ptrs + row) : row_ptrs[row];
            rb[i] = NT ? __builtin_nontemporal_load(row_ptrs + row + 1) : row_ptrs[row + 1];
            // advanced: c = beta*c first, then accumulate (reference :119-126)
            if (Advanced) c0[i] = c[row * c_stride];
            if (Dot) wv[i] = (GKOMI_DOT_PROBE & 2) ? 1.0 : w[row * w_stride];
        }
    }

    // 3) gather b (every loaded column is a stored one: always in bounds), products -> LDS
    double2 xv[pairs + 1];
#pragma unroll
    for (int u = 0; u <= pairs; ++u) {
        xv[u].x = b[ci[u].x * b_stride];
        xv[u].y = b[ci[u].y * b_stride];
    }
#pragma unroll
    for (int u = 0; u <= pairs; ++u) {
        double2 pr;
        if (Advanced) {
            pr.x = (alpha * v[u].x) * xv[u].x;  // reference order: (valpha * val) * b
            pr.y = (alpha * v[u].y) * xv[u].y;
        } else {
            pr.x = v[u].x * xv[u].x;
            pr.y = v[u].y * xv[u].y;
        }
        if (u < pairs) {
            *reinterpret_cast<double2*>(prod + 2 * (tid + u * Block)) = pr;
        } else {
            // unconditional (lanes past `over` repeat the last useful pair and
            // store the same product to the same slot): a branch here makes the
            // compiler sink the pair's loads into it, behind all the others
            *reinterpret_cast<double2*>(prod + static_cast<int>(k_over - t0)) = pr;
        }
    }
    GKOMI_STAMP(1);
    __syncthreads();
    GKOMI_STAMP(2);

    // 4) one thread per row, left to right
    double pq = 0.0, qq = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const I row = row_begin + tid + i * Block;
        const bool active = row < row_end;
        double sum = 0.0;
        if (active) {
            sum = Advanced ? c0[i] * beta : 0.0;
            const pos_t hi = rb[i] - t0;
            sum = add_products(sum, prod, static_cast<int>(ra[i] - t0), static_cast<int>(min(hi, pos_t{Tile} + over)), Tile + MaxOver);
            // a row longer than the caller's hint promised: finish it from memory
            for (pos_t k = max(ra[i] - t0, pos_t{Tile} + over); k < hi; ++k) {
                const double val = Advanced ? alpha * vals[t0 + k] : vals[t0 + k];
                sum += val * b[col_idxs[t0 + k] * b_stride];
            }
            if (Dot) {
                pq += wv[i] * sum;
                qq += sum * sum;
            }
        }
        if (active) c[row * c_stride] = sum;
    }
    for (I row = row_begin + tid + 2 * Block; row < row_end && !sparse_rows; row += Block) {
        double sum = Advanced ? c[row * c_stride] * beta : 0.0;
        const pos_t lo = row_ptrs[row] - t0;
        const pos_t hi = row_ptrs[row + 1] - t0;
        sum = add_products(sum, prod, static_cast<int>(lo), static_cast<int>(min(hi, pos_t{Tile} + over)), Tile + MaxOver);
        for (pos_t k = max(lo, pos_t{Tile} + over); k < hi; ++k) {
            const double val = Advanced ? alpha * vals[t0 + k] : vals[t0 + k];
            sum += val * b[col_idxs[t0 + k] * b_stride];
        }
        c[row * c_stride] = sum;
        if (Dot) {
            pq += w[row * w_stride] * sum;
            qq += sum * sum;
        }
    }
    // 5) sparse-rows mode: my share of the ROWS; those that are not among the first 2 * Block of the tile they start in
    //    (row_ptrs[row] / Tile) were left out above and are summed here from memory, left to right like the others
    if (sparse_rows) {
        const int64_t share = (static_cast<int64_t>(nrows) + ntiles - 1) / ntiles;
        const int64_t first = static_cast<int64_t>(logical) * share;
        const int64_t last = min(first + share, static_cast<int64_t>(nrows));
        for (int64_t row = first + tid; row < last; row += Block) {
            const pos_t lo = row_ptrs[row];
            const pos_t hi = row_ptrs[row + 1];
            if (row - static_cast<int64_t>(srow[lo / Tile] & ~srow_mark) < 2 * Block) continue;
            double sum = Advanced ? c[row * c_stride] * beta : 0.0;
            for (pos_t k = lo; k < hi; ++k) {
                const double val = Advanced ? alpha * vals[k] : vals[k];
                sum += val * b[col_idxs[k] * b_stride];
            }
            c[row * c_stride] = sum;
            if (Dot) {
                pq += w[row * w_stride] * sum;
                qq += sum * sum;
            }
        }
    }
    GKOMI_STAMP(3);
    if (Dot && (GKOMI_DOT_PROBE & 4)) {
        if (tid == 0) dot_partial[logical] = pq;
    } else if (Dot) {
        __shared__ double red[Block / wave_size];
        __syncthreads();
        const double total = block_reduce_sum<Block>(pq, red);
        if (tid == 0) dot_partial[logical] = total;
        if (dot_partial2 != nullptr) {
            __syncthreads();
            const double total2 = block_reduce_sum<Block>(qq, red);
            if (tid == 0) dot_partial2[logical] = total2;
        }
    }
#undef GKOMI_STAMP
}

// srow[t] = first row in [0, nrows] whose row_ptrs entry is >= t * tile
// (lower bound; nrows if there is none), for t = 0 .. ntiles.
template <typename I>
__global__ __launch_bounds__(256) void csr_make_srow_kernel(
    I nrows, const I* __restrict__ row_ptrs, int tile, int64_t ntiles,
    I* __restrict__ srow)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t > ntiles) return;
    if (t == 0) srow[ntiles + 1] = 0;  // the statistic csr_srow_most_rows_kernel leaves behind the tile starts
    const int64_t target = t * tile;
    I lo = 0, hi = nrows;  // answer in [lo, hi]
    while (lo < hi) {
        const I mid = lo + (hi - lo) / 2;
        if (row_ptrs[mid] >= target) {
            hi = mid;
        } else {
            lo = mid + 1;
        }
    }
    srow[t] = lo;
}

// srow[ntiles + 1] = the most rows that start in one tile (initialised to 0 by the kernel above)
template <typename I>
__global__ __launch_bounds__(256) void csr_srow_most_rows_kernel(int64_t ntiles, I* __restrict__ srow)
{
    __shared__ long long red[256 / wave_size];
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    long long rows = t < ntiles ? static_cast<long long>(srow[t + 1]) - static_cast<long long>(srow[t]) : 0;
    for (int d = 1; d < wave_size; d *= 2) rows = max(rows, __shfl_xor(rows, d));
    if ((threadIdx.x & (wave_size - 1)) == 0) red[threadIdx.x / wave_size] = rows;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / wave_size; ++w) rows = max(rows, red[w]);
        if (rows > 0) {
            if (sizeof(I) == 8) {
                atomicMax(reinterpret_cast<long long*>(srow + ntiles + 1), rows);
            } else {
                atomicMax(reinterpret_cast<int*>(srow + ntiles + 1), static_cast<int>(rows));
            }
        }
    }
}

// ... and beyond split_rows_limit every entry's top bit says so (the kernel learns it from the entries it reads anyway)
template <typename I>
__global__ __launch_bounds__(256) void csr_srow_mark_kernel(int64_t ntiles, I* __restrict__ srow)
{
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t <= ntiles && srow[ntiles + 1] > split_rows_limit) srow[t] |= static_cast<I>(static_cast<typename std::make_unsigned<I>::type>(1) << (8 * sizeof(I) - 1));
}

// ---- several right-hand sides at once ---------------------------------------
//
// The reference's classical kernel (and the stream kernel above with
// gridDim.y = nrhs) re-reads the matrix for every column of b.  Here a
// workgroup stages its row block's (val, col) tile in LDS once (12 B per
// nonzero) and every row thread walks its row, gathering the NR contiguous
// values b(col, j0 .. j0+NR-1) of the row-major b per nonzero (one 16/32-B
// load when aligned) into NR register accumulators -- the matrix is read once
// per NR columns and the per-(row, column) summation order is still the
// reference's left-to-right one (bit-exact).  The row walk is unrolled by four
// with clamped indices so that 4*NR gathers are in flight per thread.
template <int NR, bool Advanced, bool Vec>
__global__ __launch_bounds__(256) void csr_spmm_kernel(
    int nrows, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p,
    const double* __restrict__ beta_p)
{
    constexpr int Block = 256;
    constexpr int Tile = 1536;
    constexpr int pairs = Tile / (2 * Block);
    constexpr int unroll = NR >= 8 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) double s_val[Tile];
    __shared__ __attribute__((aligned(8))) int32_t s_col[Tile];

    b += blockIdx.y * NR;
    c += blockIdx.y * NR;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * Block;
    const int r1 = min(r0 + Block, nrows);
    const int p0 = row_ptrs[r0];
    const int p1 = row_ptrs[r1];
    const int nnz_total = row_ptrs[nrows];
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const int row = r0 + tid;
    int ra = p1, rb = p1;
    double sum[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) sum[j] = 0.0;
    if (row < r1) {
        ra = row_ptrs[row];
        rb = row_ptrs[row + 1];
        if (Advanced) {
#pragma unroll
            for (int j = 0; j < NR; ++j) sum[j] = c[row * c_stride + j] * beta;
        }
    }
    for (int t0 = p0 & ~1; t0 < p1; t0 += Tile) {
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            const int k = t0 + 2 * (tid + u * Block);
            double2 v = make_double2(0.0, 0.0);
            int2 ci = make_int2(0, 0);
            if (k < p1) {
                if (k + 1 < nnz_total) {
                    v = *reinterpret_cast<const double2*>(vals + k);
                    ci = *reinterpret_cast<const int2*>(col_idxs + k);
                } else {
                    v.x = vals[k];
                    ci.x = col_idxs[k];
                }
            }
            *reinterpret_cast<double2*>(s_val + 2 * (tid + u * Block)) = v;
            *reinterpret_cast<int2*>(s_col + 2 * (tid + u * Block)) = ci;
        }
        __syncthreads();
        const int t1 = t0 + Tile;
        const int lo = max(ra, t0);
        const int hi = min(rb, t1);
        for (int k = lo; k < hi; k += unroll) {
            double v[unroll];
            double x[unroll][NR];
#pragma unroll
            for (int u = 0; u < unroll; ++u) {
                const int idx = min(k + u, hi - 1) - t0;
                v[u] = s_val[idx];
                const double* src = b + s_col[idx] * b_stride;
                if (Vec) {
#pragma unroll
                    for (int j = 0; j < NR; j += 2) {
                        const double2 t = *reinterpret_cast<const double2*>(src + j);
                        x[u][j] = t.x;
                        x[u][j + 1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NR; ++j) x[u][j] = src[j];
                }
            }
#pragma unroll
            for (int u = 0; u < unroll; ++u) {
                if (k + u < hi) {
                    const double av = Advanced ? alpha * v[u] : v[u];
#pragma unroll
                    for (int j = 0; j < NR; ++j) sum[j] += av * x[u][j];
                }
            }
        }
        if (t1 < p1) __syncthreads();
    }
    if (row < r1) {
        if (Vec) {
#pragma unroll
            for (int j = 0; j < NR; j += 2) {
                *reinterpret_cast<double2*>(c + row * c_stride + j) = make_double2(sum[j], sum[j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NR; ++j) c[row * c_stride + j] = sum[j];
        }
    }
}


// ---- load-balanced kernel ("load_balance", csr.hpp:356-524) -------------------
//
// For matrices whose row lengths vary wildly the work is split by NONZEROS, not
// rows: workgroup b owns nonzeros [b*Tile, (b+1)*Tile).  The (vals, col_idxs)
// loads depend only on b, so they are issued first; while they are in flight
// the wave finds the rows that intersect the tile with a 64-ary search over
// row_ptrs (4 dependent probes for 16M rows instead of 24 binary-search steps).
// Products go through the same LDS tile as the stream kernel; one thread per
// intersecting row adds its part left to right and issues one fp64 atomic
// (`c` is pre-set to 0 or beta*c by the caller-side fill/scale, like the
// reference's dense::fill + atomic_add).  A row spanning k tiles gets k atomics.

// largest r in [0, nrows] with row_ptrs[r] <= target (row containing nonzero
// `target`, skipping empty rows); wave-cooperative, result in every lane
__device__ __forceinline__ int wave_find_row(const int32_t* __restrict__ row_ptrs,
                                             int nrows, int target)
{
    int lo = 0, hi = nrows;  // invariant: row_ptrs[lo] <= target, answer in [lo, hi]
    const int lane = threadIdx.x & 63;
    while (hi - lo > 0) {
        const int span = hi - lo;
        const int stepw = (span + 63) / 64;
        const int probe = min(lo + (lane + 1) * stepw, hi);
        const bool le = row_ptrs[probe] <= target;
        // lanes probe increasing positions: count how many are still <= target
        const unsigned long long mask = __ballot(le);
        const int cnt = __popcll(mask);
        const int new_lo = cnt == 0 ? lo : min(lo + cnt * stepw, hi);
        const int new_hi = cnt == 64 ? hi : min(lo + (cnt + 1) * stepw, hi) - 1;
        // new_hi: the first failing probe sits at lo + (cnt+1)*stepw, the answer is below it
        lo = new_lo;
        hi = max(new_hi, new_lo);
        if (stepw == 1) break;
    }
    return lo;
}

// Row segments of a tile: a segment of at most CoopMin products is added by ONE
// thread, left to right (the reference's order: a short row that lies inside one
// tile gets the reference's bits).  A longer one would be a dependent chain of up
// to Tile additions on one lane while the other 255 idle (1536 x ~10 cycles = 6 us
// per workgroup: what bounded the power-law class at 0.8 TB/s): it goes on a list
// in LDS and is reduced by a whole wave -- lane-strided partial sums, each in index
// order, then the fixed xor tree; the role of the reference's segmented scan
// (common/cuda_hip/components/segment_scan.hpp.inc:43-64 inside
// common/cuda_hip/matrix/csr_kernels.hpp.inc:98), same bits on every run.
// Rows that lie inside the tile are stored (c holds 0 or beta*c from the launch in
// front and nobody else touches them); only a row cut by a tile boundary needs the
// atomic.
struct long_segment {
    int row, from, to;  // products [from, to) of the tile's LDS image
};

// sum of prod[from, to) by the whole wave, result in every lane
__device__ __forceinline__ double wave_segment_sum(const double* prod, int from, int to)
{
    const int lane = threadIdx.x & (wave_size - 1);
    double acc0 = 0.0, acc1 = 0.0;
    int k = from + lane;
    for (; k + wave_size < to; k += 2 * wave_size) {
        acc0 += prod[k];
        acc1 += prod[k + wave_size];
    }
    if (k < to) acc0 += prod[k];
    return wave_reduce_sum(acc0 + acc1);
}

// Column windows (Window = true): when the gathers of b are spread over more than an XCD's 4 MiB L2 (uniformly random
// columns of a 1M-column matrix: 8 MB of b), half of them miss it and every miss moves a 128-B line across the fabric
// for 8 useful bytes -- 10.2 M fabric requests per launch, 8.6 M of them for b, 1.3 GB at 7.6 TB/s: the fabric is
// saturated by the gather (profiles/r04_gather_pmc.md).  The launch is then repeated once per window of columns
// [col_lo, col_hi) that does fit (the matrix stream is read again, 12 B per nonzero, but b is read from L2): a product
// outside the window is +0.0 in that pass, the row's partial sums of the passes add up in c.  Pass 0 stores / adds to
// the pre-set c as the single pass does, later passes add (Accum).
template <int Block, int Tile, bool Advanced, bool NT, int CoopMin, bool Window, bool Accum>
__global__ __launch_bounds__(Block) void csr_balanced_kernel(
    int nrows, int nnz_total, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p, const int32_t* __restrict__ srow, int col_lo, int col_hi)
{
    constexpr int pairs = Tile / (2 * Block);
    constexpr int max_long = Tile / (CoopMin + 1) + 2;
    __shared__ __attribute__((aligned(16))) double prod[Tile];
    __shared__ int s_rows[2];
    __shared__ int s_nlong;
    __shared__ long_segment s_long[max_long];
    b += blockIdx.y;
    c += blockIdx.y;
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * Tile;
    const int t1 = min(t0 + Tile, nnz_total);
    const double alpha = Advanced ? alpha_p[0] : 1.0;
    double2 v[pairs];
    int2 ci[pairs];
    if (tid == 0) s_nlong = 0;
#pragma unroll
    for (int u = 0; u < pairs; ++u) {
        const int k = t0 + 2 * (tid + u * Block);
        v[u] = make_double2(0.0, 0.0);
        ci[u] = make_int2(Window ? col_lo : 0, Window ? col_lo : 0);
        if (k + 1 < t1) {
            if (NT) {
                const nt_double2 tv = __builtin_nontemporal_load(reinterpret_cast<const nt_double2*>(vals + k));
                const nt_int2 tc = __builtin_nontemporal_load(reinterpret_cast<const nt_int2*>(col_idxs + k));
                v[u] = make_double2(tv.x, tv.y);
                ci[u] = make_int2(tc.x, tc.y);
            } else {
                v[u] = *reinterpret_cast<const double2*>(vals + k);
                ci[u] = *reinterpret_cast<const int2*>(col_idxs + k);
            }
        } else if (k < t1) {
            v[u].x = vals[k];
            ci[u].x = col_idxs[k];
        }
    }
    // rows intersecting the tile: from the matrix's srow (the rows that START in the tile, plus the one that runs
    // into it: three scalar loads), else wave 0 searches row_ptrs while the loads fly
    int first, last;
    if (srow != nullptr) {
        const int s0 = srow[blockIdx.x] & 0x7fffffff;  // (top bit: the split kernel's sparse-rows mark)
        first = (s0 < nrows && row_ptrs[s0] == t0) ? s0 : s0 - 1;
        last = min(srow[blockIdx.x + 1] & 0x7fffffff, nrows) - 1;
    } else {
        if (tid < 64) {
            const int f = wave_find_row(row_ptrs, nrows, t0);
            const int l = wave_find_row(row_ptrs, nrows, t1 - 1);
            if (tid == 0) {
                s_rows[0] = f;
                s_rows[1] = l;
            }
        }
    }
    // with srow the first round of rows asks for its row_ptrs now, not behind the barrier
    int ra0 = 0, rb0 = 0;
    if (srow != nullptr && first + tid <= last) {
        ra0 = row_ptrs[first + tid];
        rb0 = row_ptrs[first + tid + 1];
    }
    double2 xv[pairs];
    bool in0[pairs], in1[pairs];
#pragma unroll
    for (int u = 0; u < pairs; ++u) {
        // outside the window: the gather goes to the window's first entry (one hot line) and the product is
        // replaced by +0.0 below (not multiplied by zero: b may hold Inf / NaN there)
        in0[u] = !Window || (ci[u].x >= col_lo && ci[u].x < col_hi);
        in1[u] = !Window || (ci[u].y >= col_lo && ci[u].y < col_hi);
        xv[u].x = b[(in0[u] ? ci[u].x : col_lo) * b_stride];
        xv[u].y = b[(in1[u] ? ci[u].y : col_lo) * b_stride];
    }
#pragma unroll
    for (int u = 0; u < pairs; ++u) {
        double2 pr;
        pr.x = Advanced ? (alpha * v[u].x) * xv[u].x : v[u].x * xv[u].x;
        pr.y = Advanced ? (alpha * v[u].y) * xv[u].y : v[u].y * xv[u].y;
        if (Window) {
            pr.x = in0[u] ? pr.x : 0.0;
            pr.y = in1[u] ? pr.y : 0.0;
        }
        *reinterpret_cast<double2*>(prod + 2 * (tid + u * Block)) = pr;
    }
    __syncthreads();
    if (srow == nullptr) {
        first = s_rows[0];
        last = min(s_rows[1], nrows - 1);
    }
    const int count = t1 - t0;
    // a row inside the tile: c[row] (0 or beta * c, or the passes before) + sum, by the one thread that owns it;
    // a row cut by the tile: one atomic per tile it runs through
    auto emit = [&](int row, bool whole, double sum) {
        double* dst = c + row * c_stride;
        if (whole) {
            *dst = (Advanced || Accum) ? *dst + sum : sum;
        } else {
            unsafeAtomicAdd(dst, sum);
        }
    };
    // A tile that owns a long run of EMPTY rows (the non-local block of a distributed matrix, selection matrices):
    // walking the rows 256 at a time costs two dependent loads per round -- 1 ms for 250 000 rows.  Empty rows need
    // nothing from this kernel, so such a tile goes by its NONZEROS instead: every position looks up the last row
    // that starts at or before it (18 steps for 250 000 rows), and the positions where a row starts (and the tile's
    // first position: the row that runs into it) sum that row's segment as below.
    const bool by_nonzeros = last - first + 1 > 4 * Block;
    for (int j = tid; by_nonzeros && j < count; j += Block) {
        const int k = t0 + j;
        int r_lo = first, r_hi = last;  // row_ptrs[r_lo] <= k throughout
        while (r_lo < r_hi) {
            const int mid = r_lo + (r_hi - r_lo + 1) / 2;
            if (row_ptrs[mid] <= k) {
                r_lo = mid;
            } else {
                r_hi = mid - 1;
            }
        }
        const int ra = row_ptrs[r_lo];
        if (ra == k || j == 0) {
            const int rb = row_ptrs[r_lo + 1];
            const int hi = min(rb, t1) - t0;
            if (hi - j > CoopMin) {
                const int slot = atomicAdd(&s_nlong, 1);
                s_long[slot] = long_segment{r_lo, j, hi};
            } else {
                emit(r_lo, ra >= t0 && rb <= t1, add_products(0.0, prod, j, hi, count));
            }
        }
    }
    for (int row = first + tid, round = 0; row <= last && !by_nonzeros; row += Block, ++round) {
        const int ra = (srow != nullptr && round == 0) ? ra0 : row_ptrs[row];
        const int rb = (srow != nullptr && round == 0) ? rb0 : row_ptrs[row + 1];
        const int lo = max(ra, t0) - t0;
        const int hi = min(rb, t1) - t0;
        if (lo < hi) {
            if (hi - lo > CoopMin) {
                const int slot = atomicAdd(&s_nlong, 1);
                s_long[slot] = long_segment{row, lo, hi};
            } else {
                emit(row, ra >= t0 && rb <= t1, add_products(0.0, prod, lo, hi, count));
            }
        }
    }
    if (CoopMin >= Tile) return;
    __syncthreads();
    const int nlong = s_nlong;
    for (int i = tid / wave_size; i < nlong; i += Block / wave_size) {
        const long_segment seg = s_long[i];
        const double sum = wave_segment_sum(prod, seg.from, seg.to);
        if ((tid & (wave_size - 1)) == 0) emit(seg.row, seg.to - seg.from == row_ptrs[seg.row + 1] - row_ptrs[seg.row], sum);
    }
}

// Variant of the stream kernel's staging for arrays whose base pointers are
// not 16-/8-byte aligned is not needed: misaligned inputs take the vector
// kernel, which only uses natural-width loads.

template <int SubWave, bool Advanced>
__global__ __launch_bounds__(256) void csr_vector_kernel(
    int nrows, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p,
    const double* __restrict__ beta_p)
{
    b += blockIdx.y;
    c += blockIdx.y;
    const int sub_lane = threadIdx.x % SubWave;
    const int64_t subs_per_block = 256 / SubWave;
    const int64_t first = blockIdx.x * subs_per_block + threadIdx.x / SubWave;
    const int64_t step = static_cast<int64_t>(gridDim.x) * subs_per_block;
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    // all lanes of a wave run the same number of iterations (shuffles need
    // the partner lanes active)
    const int64_t rounds = (nrows + step - 1) / step;
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t row = first + it * step;
        double acc = 0.0;
        if (row < nrows) {
            const int end = row_ptrs[row + 1];
            for (int k = row_ptrs[row] + sub_lane; k < end; k += SubWave) {
                const double val = Advanced ? alpha * vals[k] : vals[k];
                acc += val * b[col_idxs[k] * b_stride];
            }
        }
        acc = subwave_reduce_sum<SubWave>(acc);
        if (row < nrows && sub_lane == 0) {
            c[row * c_stride] =
                Advanced ? c[row * c_stride] * beta + acc : acc;
        }
    }
}

template <typename I>
__global__ __launch_bounds__(256) void csr_max_row_nnz_kernel(
    int64_t nrows, const I* __restrict__ row_ptrs,
    I* __restrict__ result)
{
    __shared__ I smax[4];
    I m = 0;
    for (int64_t row = blockIdx.x * 256 + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * 256) {
        m = max(m, static_cast<I>(row_ptrs[row + 1] - row_ptrs[row]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
        // integer max: order-independent, exact
        if (sizeof(I) == 8) {
            atomicMax(reinterpret_cast<long long*>(result), static_cast<long long>(m));
        } else {
            atomicMax(reinterpret_cast<int*>(result), static_cast<int>(m));
        }
    }
}

template <int Block, int RowsPerThread, int Tile, bool NT = false, bool Pad = false, typename I = int32_t>
int launch_stream(hipStream_t stream, bool swizzle, int chunk, I nrows, int nrhs,
                  const I* row_ptrs, const I* col_idxs,
                  const double* vals, const double* b, int64_t b_stride,
                  double* c, int64_t c_stride, const double* alpha,
                  const double* beta)
{
    constexpr int rows_per_block = Block * RowsPerThread;
    const int nblocks = static_cast<int>(ceildiv(nrows, rows_per_block));
    int per = static_cast<int>(ceildiv(nblocks, num_xcd));
    if (chunk > 0 && chunk < per) per = chunk;
    const bool advanced = alpha != nullptr;
    const bool swz = swizzle && nblocks >= 2 * num_xcd;
    const int groups = static_cast<int>(ceildiv(nblocks, per * num_xcd));
    dim3 grid(swz ? groups * per * num_xcd : nblocks, nrhs);
#define GKOMI_LAUNCH(ADV, SWZ)                                                 \
    hipLaunchKernelGGL(                                                        \
        (csr_stream_kernel<I, Block, RowsPerThread, Tile, ADV, SWZ, false, NT, Pad>), \
        grid,                                                                  \
        dim3(Block), 0, stream, nrows, row_ptrs, col_idxs, vals, b, b_stride,  \
        c, c_stride, alpha, beta, nblocks, per)
    if (advanced) {
        if (swz) GKOMI_LAUNCH(true, true); else GKOMI_LAUNCH(true, false);
    } else {
        if (swz) GKOMI_LAUNCH(false, true); else GKOMI_LAUNCH(false, false);
    }
#undef GKOMI_LAUNCH
    return check_launch();
}

template <int NR>
int launch_spmm(hipStream_t stream, int nrows, int slices,
                const int32_t* row_ptrs, const int32_t* col_idxs,
                const double* vals, const double* b, int64_t b_stride,
                double* c, int64_t c_stride, const double* alpha,
                const double* beta)
{
    dim3 grid(static_cast<unsigned>(ceildiv(nrows, 256)), slices);
    const bool vec = reinterpret_cast<uintptr_t>(b) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(c) % 16 == 0 &&
                     b_stride % 2 == 0 && c_stride % 2 == 0;
#define GKOMI_LAUNCH(ADV, VEC)                                                 \
    hipLaunchKernelGGL((csr_spmm_kernel<NR, ADV, VEC>), grid, dim3(256), 0,    \
                       stream, nrows, row_ptrs, col_idxs, vals, b, b_stride,   \
                       c, c_stride, alpha, beta)
    if (alpha != nullptr) {
        if (vec) GKOMI_LAUNCH(true, true); else GKOMI_LAUNCH(true, false);
    } else {
        if (vec) GKOMI_LAUNCH(false, true); else GKOMI_LAUNCH(false, false);
    }
#undef GKOMI_LAUNCH
    return check_launch();
}

template <int SubWave>
int launch_vector(hipStream_t stream, int nrows, int nrhs,
                  const int32_t* row_ptrs, const int32_t* col_idxs,
                  const double* vals, const double* b, int64_t b_stride,
                  double* c, int64_t c_stride, const double* alpha,
                  const double* beta)
{
    const int64_t subs_per_block = 256 / SubWave;
    dim3 grid(grid_for(nrows, static_cast<int>(subs_per_block), 8192), nrhs);
    if (alpha != nullptr) {
        hipLaunchKernelGGL((csr_vector_kernel<SubWave, true>), grid, dim3(256),
                           0, stream, nrows, row_ptrs, col_idxs, vals, b,
                           b_stride, c, c_stride, alpha, beta);
    } else {
        hipLaunchKernelGGL((csr_vector_kernel<SubWave, false>), grid,
                           dim3(256), 0, stream, nrows, row_ptrs, col_idxs,
                           vals, b, b_stride, c, c_stride, alpha, beta);
    }
    return check_launch();
}

template <int Block, int Tile, bool NT, typename I = int32_t>
int launch_split(hipStream_t stream, bool swizzle, I nrows, typename index_traits<I>::pos nnz,
                 const I* row_ptrs, const I* col_idxs,
                 const double* vals, const double* b, int64_t b_stride,
                 double* c, int64_t c_stride, const double* alpha,
                 const double* beta, const I* srow, int over, int chunk = 0)
{
    constexpr int MaxOver = split_max_over;
    const int ntiles = static_cast<int>(nnz / Tile + 1);
    int per = static_cast<int>(ceildiv(ntiles, num_xcd));
    if (chunk > 0 && chunk < per) per = chunk;  // XCD k takes `chunk` consecutive tiles of every 8 * chunk
    const bool swz = swizzle && ntiles >= 2 * num_xcd;
    const int groups = static_cast<int>(ceildiv(ntiles, per * num_xcd));
    dim3 grid(swz ? groups * per * num_xcd : ntiles, 1);
#define GKOMI_LAUNCH(ADV, SWZ)                                                 \
    hipLaunchKernelGGL(                                                        \
        (csr_split_kernel<I, Block, Tile, MaxOver, ADV, SWZ, false, NT, true>), \
        grid, dim3(Block), 0, stream, nrows, nnz, row_ptrs, col_idxs, vals, b, \
        b_stride, c, c_stride, alpha, beta, srow, ntiles, per, over)
    if (alpha != nullptr) {
        if (swz) GKOMI_LAUNCH(true, true); else GKOMI_LAUNCH(true, false);
    } else {
        if (swz) GKOMI_LAUNCH(false, true); else GKOMI_LAUNCH(false, false);
    }
#undef GKOMI_LAUNCH
    return check_launch();
}

// column window of the load-balanced kernel: what of b an XCD's 4 MiB L2 keeps next to the matrix streams that pass
// through it (GKOMI_CSR_COLBLOCK_KIB: tuning hook)
inline int64_t colblock_window_bytes()
{
    static const int64_t bytes = [] {
        const char* e = std::getenv("GKOMI_CSR_COLBLOCK_KIB");
        const long long kib = e != nullptr ? std::atoll(e) : 0;
        // 4 MiB: two windows for the 8 MB of b of a 1M-column matrix -- 126 vs 140 us on the power-law class, 144 vs
        // 172 us on uniformly random columns; 3 windows of 2.7 MiB 153 / 166, 4 of 2 MiB 191 / 198: every window is
        // one more pass over the matrix at ~45 us (profiles/r04_colblock_sweep.md)
        return static_cast<int64_t>(kib > 0 ? kib : 4096) << 10;
    }();
    return bytes;
}

// number of column windows for a b of ncols rows of b_stride doubles (1 = no windows)
inline int colblock_passes(int64_t ncols, int64_t b_stride)
{
    const int64_t bytes = 8 * ncols * b_stride;
    // a b that nearly fits is left alone: two passes cost a second read of the matrix
    if (bytes <= colblock_window_bytes() * 5 / 4) return 1;
    return static_cast<int>(std::min<int64_t>(ceildiv(bytes, colblock_window_bytes()), 64));
}

template <int Tile, bool Advanced, bool NT>
int launch_balanced(hipStream_t stream, int nrows, int ncols, int nrhs, int nnz, const int32_t* row_ptrs,
                    const int32_t* col_idxs, const double* vals, const double* b, int64_t b_stride, double* c,
                    int64_t c_stride, const double* alpha, const int32_t* srow, int passes, bool serial)
{
    constexpr int Block = 256;
    dim3 grid(static_cast<unsigned>(ceildiv(nnz, Tile)), nrhs);
#define GKOMI_BALANCED(COOP, WIN, ACC, LO, HI)                                                                     \
    hipLaunchKernelGGL((csr_balanced_kernel<Block, Tile, Advanced, NT, COOP, WIN, ACC>), grid, dim3(Block), 0,    \
                       stream, nrows, nnz, row_ptrs, col_idxs, vals, b, b_stride, c, c_stride, alpha, srow, LO, HI)
    if (serial) {  // every segment by one thread: the kernel of rounds 1-3, kept for A/B timings
        GKOMI_BALANCED(Tile, false, false, 0, 0);
    } else if (passes <= 1) {
        GKOMI_BALANCED(balanced_coop_min, false, false, 0, 0);
    } else {
        const int width = static_cast<int>(ceildiv(ncols, passes));
        for (int p = 0; p < passes; ++p) {
            const int lo = p * width, hi = std::min(ncols, lo + width);
            if (lo >= hi) break;
            if (p == 0) {
                GKOMI_BALANCED(balanced_coop_min, true, false, lo, hi);
            } else {
                GKOMI_BALANCED(balanced_coop_min, true, true, lo, hi);
            }
        }
    }
#undef GKOMI_BALANCED
    return check_launch();
}

// pages of b (page_cols columns each, at most 4096 pages) that the gathers of one tile touch, one tile per workgroup,
// summed over the sampled tiles (out[0]) with their count (out[1]).  Pages, not max - min or a deviation from the mean:
// a banded tile with one far column (an arrow matrix) touches two pages, a tile of uniformly random columns all of them.
__global__ __launch_bounds__(256) void csr_col_spread_kernel(int64_t nnz, const int32_t* __restrict__ col_idxs, int tile,
                                                              int64_t tile_step, int page_cols, double* __restrict__ out)
{
    __shared__ unsigned bitmap[4096 / 32];
    __shared__ double red[256 / wave_size];
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * tile_step * tile;
    const int count = static_cast<int>(min(static_cast<int64_t>(tile), nnz - t0));
    if (count <= 0) return;
    if (threadIdx.x < 4096 / 32) bitmap[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 256) {
        const int page = min(max(col_idxs[t0 + i], 0) / page_cols, 4095);
        atomicOr(&bitmap[page >> 5], 1u << (page & 31));
    }
    __syncthreads();
    const double pages = block_reduce_sum<256>(threadIdx.x < 4096 / 32 ? static_cast<double>(__popc(bitmap[threadIdx.x])) : 0.0, red);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(out, pages);
        unsafeAtomicAdd(out + 1, 1.0);
    }
}

}  // namespace
}  // namespace gkomi


namespace gkomi {

// q = A p with the dot(p, q) partials epilogue, for the fused CG path
// (cg_solver.hip).  One partial per row block; returns their count.
int csr_spmv_dot_launch(hipStream_t stream, int nrows, int64_t nnz,
                        const int32_t* row_ptrs, const int32_t* col_idxs,
                        const double* vals, const double* p, double* q,
                        double* partial, const uint8_t* stop_status,
                        bool swizzle, const double* dot_w, double* partial2, bool nontemporal)
{
    constexpr int Block = 256, Tile = 1536;
    const int nblocks = static_cast<int>(ceildiv(nrows, Block));
    const int per = static_cast<int>(ceildiv(nblocks, num_xcd));
    const bool swz = swizzle && nblocks >= 2 * num_xcd;
    dim3 grid(swz ? per * num_xcd : nblocks, 1);
    // not Infinity-Cache resident (the matrix alone, or the solve's working set: the driver says so):
    // the matrix streams from HBM every iteration, read it with nontemporal loads (see csr_auto_swizzle)
    const bool nt = nontemporal || !swizzle;
#define GKOMI_STREAM_DOT(SWZ, NT)                                                              \
    hipLaunchKernelGGL((csr_stream_kernel<int32_t, Block, 1, Tile, false, SWZ, true, NT>), grid, dim3(Block), 0, stream, \
                       nrows, row_ptrs, col_idxs, vals, p, int64_t{1}, q, int64_t{1}, nullptr, nullptr, nblocks, \
                       per, partial, stop_status, dot_w, partial2)
    if (swz) {
        if (nt) GKOMI_STREAM_DOT(true, true); else GKOMI_STREAM_DOT(true, false);
    } else {
        if (nt) GKOMI_STREAM_DOT(false, true); else GKOMI_STREAM_DOT(false, false);
    }
#undef GKOMI_STREAM_DOT
    return check_launch();
}

int csr_spmv_dot_num_partials(int nrows)
{
    return static_cast<int>(ceildiv(nrows, 256));
}

// The same for a matrix with its srow: the nonzero-split kernel with the dot-product epilogue, one
// partial per tile (on the 256^3 7-point matrix 300 vs 332 us per SpMV).  XCD chunking and nontemporal
// streams as the caller decides (spmv_dot_plan: by the size of the matrix and of the solve's working set).
// Returns the number of partials, <= 0 when not applicable (the tile must be one the kernel is built for).
int csr_split_dot_num_partials(int64_t nnz, int64_t tile)
{
    if (tile != 1536 && tile != 2048 && tile != 3072) return 0;
    if (nnz < 2 || nnz > INT32_MAX - 2 * tile - 1024) return 0;
    return static_cast<int>(nnz / tile + 1);
}

namespace {
template <int Tile>
int launch_split_dot(hipStream_t stream, int nrows, int nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
                     const double* vals, const double* p, double* q, double* partial, const uint8_t* stop_status,
                     const int32_t* srow, int over, bool swizzle, bool nt, const double* dot_w, double* partial2)
{
    constexpr int Block = 256;
    const int ntiles = nnz / Tile + 1;
    // Infinity-Cache resident: one contiguous eighth of the tiles per XCD; streaming from HBM: 16 consecutive
    // tiles per XCD (see the automatic strategy of gkomi_csr_spmv_srow_f64_i32)
    const int per = swizzle ? static_cast<int>(ceildiv(ntiles, num_xcd)) : 16;
    const bool swz = ntiles >= 2 * num_xcd;
    const int groups = static_cast<int>(ceildiv(ntiles, per * num_xcd));
    dim3 grid(swz ? groups * per * num_xcd : ntiles, 1);
#define GKOMI_SPLIT_DOT(SWZ, NT)                                                                          \
    hipLaunchKernelGGL((csr_split_kernel<int32_t, Block, Tile, split_max_over, false, SWZ, true, NT, true>), grid, \
                       dim3(Block), 0, stream, nrows, nnz, row_ptrs, col_idxs, vals, p, int64_t{1}, q,    \
                       int64_t{1}, nullptr, nullptr, srow, ntiles, per, over, partial, stop_status, dot_w, partial2)
    if (swz) {
        if (nt) GKOMI_SPLIT_DOT(true, true); else GKOMI_SPLIT_DOT(true, false);
    } else {
        if (nt) GKOMI_SPLIT_DOT(false, true); else GKOMI_SPLIT_DOT(false, false);
    }
#undef GKOMI_SPLIT_DOT
    return check_launch();
}

// out[j] = raw[j * run .. (j + 1) * run) added in index order within a lane's stride, then across the
// wave's fixed tree: one wave per output, the same bits on every run
__global__ __launch_bounds__(1024) void compress_partials_kernel(const double* __restrict__ raw, int nraw,
                                                                  double* __restrict__ out, int nout, int run,
                                                                  const double* __restrict__ raw2,
                                                                  double* __restrict__ out2,
                                                                  const uint8_t* __restrict__ stop_status)
{
    if (stop_status != nullptr && status_has_stopped(stop_status[0])) return;
    const int wave = (blockIdx.x * 1024 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nout) return;
    const int lo = wave * run, hi = min(lo + run, nraw);
    double a = 0.0, b = 0.0;
    for (int i = lo + lane; i < hi; i += 64) {
        a += raw[i];
        if (raw2 != nullptr) b += raw2[i];
    }
    a = wave_reduce_sum(a);
    if (lane == 0) out[wave] = a;
    if (raw2 != nullptr) {
        b = wave_reduce_sum(b);
        if (lane == 0) out2[wave] = b;
    }
}
}  // namespace

int compress_partials_launch(hipStream_t stream, const double* raw, int nraw, double* out, int nout,
                             const double* raw2, double* out2, const uint8_t* stop_status)
{
    if (nraw <= 0 || nout <= 0) return GKOMI_EINVAL;
    const int run = static_cast<int>(ceildiv(nraw, nout));
    hipLaunchKernelGGL(compress_partials_kernel, dim3(static_cast<unsigned>(ceildiv(nout, 16))), dim3(1024), 0,
                       stream, raw, nraw, out, nout, run, raw2, out2, stop_status);
    return check_launch();
}

int csr_split_dot_launch(hipStream_t stream, int nrows, int64_t nnz, const int32_t* row_ptrs,
                         const int32_t* col_idxs, const double* vals, const double* p, double* q,
                         double* partial, const uint8_t* stop_status, const int32_t* srow, int64_t tile,
                         int over, bool swizzle, bool nontemporal, const double* dot_w, double* partial2)
{
    if (csr_split_dot_num_partials(nnz, tile) <= 0) return GKOMI_ENOTSUPPORTED;
    const int z = static_cast<int>(nnz);
    if (tile == 1536) {
        return launch_split_dot<1536>(stream, nrows, z, row_ptrs, col_idxs, vals, p, q, partial, stop_status, srow,
                                      over, swizzle, nontemporal, dot_w, partial2);
    } else if (tile == 2048) {
        return launch_split_dot<2048>(stream, nrows, z, row_ptrs, col_idxs, vals, p, q, partial, stop_status, srow,
                                      over, swizzle, nontemporal, dot_w, partial2);
    }
    return launch_split_dot<3072>(stream, nrows, z, row_ptrs, col_idxs, vals, p, q, partial, stop_status, srow,
                                  over, swizzle, nontemporal, dot_w, partial2);
}

// swizzle heuristic shared by the automatic strategy: XCD-chunked row blocks
// pay while the matrix is Infinity-Cache resident (measured 13.7 vs 15.1 us
// warm at 1M rows) and cost ~4 % once it streams from HBM (57.3 vs 59.6 us at
// 4M rows) -- profiles/README.md.  The same threshold selects nontemporal
// loads of vals / col_idxs: 54.6 vs 59.0 us at 4M rows (5.86 TB/s), but 15.6 vs
// 13.8 us on a resident 1M-row matrix (profiles/r01_tune_nt.log).  Crossover
// measured warm (profiles/r01_tune_pad.log): 212 MB and 308 MB matrices still
// gain from the cache (36.3 vs 39.9, 56.2 vs 58.5 us), 320 MB and 337 MB ones
// from the nontemporal loads (54.2 vs 59.1, 69.6 vs 70.7 us) -> 288 MiB.
bool csr_auto_swizzle(int64_t nrows, int64_t nnz)
{
    if (nnz < 0) return true;
    return 12 * nnz + 20 * nrows < (int64_t{288} << 20);
}

// Is this matrix still in the Infinity Cache?  Only the caller knows its working set (GKOMI_CSR_STREAMING says so),
// but a Csr::apply through the reference's interface cannot carry that flag.  What the library does know is what ITS
// OWN SpMVs have streamed since this matrix was applied last: a running byte count over all CSR applies of the process
// and, per matrix (keyed by its values array), the count at its last apply.  More than the cache in between = evicted for
// sure -> nontemporal streams (cold 18.3 -> 16.6 us on the 1M-row 5-pt matrix); an unknown matrix or a
// smaller distance keeps the cached variant.  Other kernels' traffic is not counted: the estimate errs towards "resident".
std::atomic<int64_t> csr_evicted_applies{0};
bool csr_probably_evicted(const void* key, int64_t bytes)
{
    struct entry {
        const void* key;
        uint64_t stamp;
    };
    static std::mutex lock;
    static uint64_t clock = 0;
    static entry table[64] = {};
    std::lock_guard<std::mutex> guard(lock);
    entry* slot = nullptr;
    entry* oldest = &table[0];
    for (entry& e : table) {
        if (e.key == key) slot = &e;
        if (e.stamp < oldest->stamp) oldest = &e;
    }
    bool evicted = false;
    if (slot != nullptr) {
        evicted = clock - slot->stamp + static_cast<uint64_t>(bytes) > static_cast<uint64_t>(infinity_cache_bytes);
    } else {
        slot = oldest;
        slot->key = key;
    }
    clock += static_cast<uint64_t>(bytes);
    slot->stamp = clock;
    if (evicted) csr_evicted_applies.fetch_add(1, std::memory_order_relaxed);
    return evicted;
}

}  // namespace gkomi


extern "C" int64_t gkomi_csr_srow_tile(void) { return 1536; }

// 12 B per nonzero: beyond ~20 M nonzeros the matrix alone fills the 256 MiB Infinity Cache
// and the larger the stream, the larger the tile that pays: 2048 from there, 3072 beyond ~60 M
// (256^3 7-point, 117 M nonzeros: 309 / 300 / 313 us with tiles of 2048 / 3072 / 4096; 160^3, 28.5 M:
// ~80 us with all of them -- tools/p3_probe.py)
extern "C" int64_t gkomi_csr_srow_tile_for(int64_t nnz)
{
    return nnz > 60000000 ? 3072 : (nnz > 20000000 ? 2048 : 1536);
}

extern "C" int64_t gkomi_csr_srow_entries(int64_t nnz, int64_t tile)
{
    if (nnz < 0 || tile <= 0) return 0;
    return nnz / tile + 3;  // tile starts 0 .. nnz / tile + 1, then the most rows starting in one tile
}

extern "C" int gkomi_csr_make_srow_i32(gkomi_stream_t stream_, int64_t nrows, int64_t nnz,
                                       const int32_t* row_ptrs, int64_t tile, int32_t* srow,
                                       int64_t nsrow)
{
    using namespace gkomi;
    if (nrows < 0 || nnz < 0 || tile <= 0 || tile % 2 != 0) return GKOMI_EINVAL;
    if (nrows > INT32_MAX - 1024 || nnz > INT32_MAX - 2 * tile - 1024 || tile > (1 << 20)) {
        return GKOMI_ENOTSUPPORTED;
    }
    if (nsrow < gkomi_csr_srow_entries(nnz, tile)) return GKOMI_EWORKSPACE;
    const int ntiles = static_cast<int>(nnz / tile) + 1;
    hipLaunchKernelGGL(csr_make_srow_kernel<int32_t>, dim3(static_cast<unsigned>(ceildiv(ntiles + 1, 256))),
                       dim3(256), 0, to_stream(stream_), static_cast<int>(nrows), row_ptrs,
                       static_cast<int>(tile), ntiles, srow);
    hipLaunchKernelGGL(csr_srow_most_rows_kernel<int32_t>, dim3(static_cast<unsigned>(ceildiv(ntiles, 256))), dim3(256), 0,
                       to_stream(stream_), static_cast<int64_t>(ntiles), srow);
    hipLaunchKernelGGL(csr_srow_mark_kernel<int32_t>, dim3(static_cast<unsigned>(ceildiv(ntiles + 1, 256))), dim3(256), 0,
                       to_stream(stream_), static_cast<int64_t>(ntiles), srow);
    return check_launch();
}

extern "C" int gkomi_csr_spmv_f64_i32(
    gkomi_stream_t stream_, int64_t nrows, int64_t ncols, int64_t nrhs,
    int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha, const double* beta, int strategy,
    int64_t max_row_nnz_hint)
{
    return gkomi_csr_spmv_srow_f64_i32(stream_, nrows, ncols, nrhs, nnz, row_ptrs, col_idxs, vals,
                                       b, b_stride, c, c_stride, alpha, beta, strategy,
                                       max_row_nnz_hint, nullptr, 0);
}

extern "C" int gkomi_csr_spmv_srow_f64_i32(
    gkomi_stream_t stream_, int64_t nrows, int64_t ncols, int64_t nrhs,
    int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha, const double* beta, int strategy,
    int64_t max_row_nnz_hint, const int32_t* srow, int64_t srow_tile)
{
    using namespace gkomi;
    if (nrows < 0 || ncols < 0 || nrhs < 0) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrows > INT32_MAX - 1024 || nrhs > 65535 || nnz > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    // empty output: no-op (hip/matrix/csr_kernels.hip.cpp:291-292)
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(stream_);
    const int n = static_cast<int>(nrows);
    const int r = static_cast<int>(nrhs);

    int kind = strategy & 0xff;
    int variant = (strategy >> 8) & 0xff;
    bool no_swizzle = (strategy >> 16) & 1;
    const int chunk_code = (strategy >> 17) & 0x7f;  // XCD chunk = 2^(code-1) row blocks, 0 = one eighth each
    int chunk = chunk_code ? 1 << (chunk_code - 1) : 0;
    if (const char* e = getenv("GKOMI_CSR_XCD_CHUNK")) {  // experiments (tools/p3_chunk_sweep.py): any chunk size
        if (e[0] != 0 && chunk_code == 127) chunk = std::max(1, atoi(e));
    }
    int auto_chunk = 0;  // the automatic strategy's choice for matrices that stream from HBM
    const bool automatic = kind == GKOMI_CSR_AUTO;
    const bool aligned = (reinterpret_cast<uintptr_t>(vals) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(col_idxs) % 8 == 0);
    const bool split_ok = srow != nullptr && aligned && nnz >= 2 &&
                          nnz <= INT32_MAX - 2 * srow_tile - 1024 &&
                          (srow_tile == 1024 || srow_tile == 1536 || srow_tile == 2048 || srow_tile == 3072);
    if (kind == GKOMI_CSR_SPLIT && !split_ok) return srow == nullptr ? GKOMI_EINVAL : GKOMI_ENOTSUPPORTED;
    if (kind == GKOMI_CSR_AUTO) {
        // the role of Csr::automatical (csr.hpp:526-705): short rows stream
        // (cut by nonzeros when the matrix carries its srow, by rows otherwise);
        // long rows one sub-wave per row; when a few rows
        // dwarf the average (max > 64 x mean) split by nonzeros instead
        // (an unknown row length too: a row longer than the 64 nonzeros read behind its tile is
        // finished from memory by its thread -- correct for any matrix; callers that know better say so)
        // (GKOMI_CSR_COLBLOCK changes nothing here: matrices of short rows keep the kernels that give the
        // reference's bits; the flag adds column windows to the load-balanced kernel where that one is chosen
        // anyway, or with an explicit GKOMI_CSR_BALANCED)
        // (a matrix with fewer nonzeros than rows -- the non-local block of a distributed matrix -- has too few tiles
        // for its rows: 86 workgroups for 2 M rows; the row-cut kernel's grid follows the rows)
        if (split_ok && r == 1 && max_row_nnz_hint <= split_max_over + 1 && nnz >= nrows) {
            kind = GKOMI_CSR_SPLIT;
        } else if (max_row_nnz_hint < 0 || max_row_nnz_hint <= 256) {
            kind = GKOMI_CSR_STREAM;
        } else if (nnz > 0 && max_row_nnz_hint > 64 * (nnz / nrows + 1)) {
            kind = GKOMI_CSR_BALANCED;
        } else {
            kind = GKOMI_CSR_VECTOR;
        }
    }
    if (kind == GKOMI_CSR_BALANCED && (!aligned || nnz < 0)) kind = GKOMI_CSR_VECTOR;
    if (kind == GKOMI_CSR_STREAM && !aligned) kind = GKOMI_CSR_VECTOR;
    // nontemporal matrix streams: for a matrix that will not be found in the
    // Infinity Cache at its next use -- larger than the cache, or the caller
    // says so (GKOMI_CSR_STREAMING: its working set between two applies is)
    bool nt = false;
    if (automatic) {
        // 256 threads, 256 rows, 1536-nonzero tile: fastest measured
        no_swizzle = !csr_auto_swizzle(nrows, nnz);
        // (the tracker is asked on every automatic apply, also when the answer is not needed: it keeps the byte count)
        const bool evicted = nnz > 0 && csr_probably_evicted(vals, 12 * nnz + 20 * nrows);
        nt = no_swizzle || ((strategy & GKOMI_CSR_STREAMING) != 0) || evicted;
        // A matrix that streams from HBM, cut by nonzeros: consecutive tiles on consecutive XCDs (no swizzle) pull
        // the x lines of neighbouring rows into up to three L2s (1.19x the algorithmic traffic on the 256^3 7-point
        // matrix); one contiguous eighth per XCD reads every line once but is 6 % slower there (eight far-apart
        // streams).  16 consecutive tiles per XCD keep a row's near neighbours in one L2 and the streams together:
        // 282 vs 291 (no swizzle) vs 308 us (eighths), tools/p3_chunk_sweep.py, profiles/r03_p3_chunk_sweep.log.
        if (no_swizzle && kind == GKOMI_CSR_SPLIT && chunk == 0) {
            no_swizzle = false;
            auto_chunk = 16;
        }
        // rows of 32+ entries: the padded LDS tile (row-sum reads of
        // neighbouring lanes are 32+ doubles apart) is 6-15 % faster, shorter
        // rows lose 1-2 % to the split LDS stores (profiles/r01_tune_pad.log)
        const bool pad = max_row_nnz_hint >= 32;
        variant = kind == GKOMI_CSR_STREAM ? (nt ? (pad ? 16 : 14) : (pad ? 15 : 5)) : 0;
    }

#define GKOMI_ARGS                                                            \
    n, r, row_ptrs, col_idxs, vals, b, b_stride, c, c_stride, alpha, beta
    if (kind == GKOMI_CSR_STREAM && r >= 2 && (automatic || variant == 20)) {
        // several right-hand sides: read the matrix once per 8 (then 4,
        // then 2) columns; a last odd column goes through the single-column kernel
        int done = 0;
        if (r - done >= 8) {
            const int slices = (r - done) / 8;
            const int err = launch_spmm<8>(stream, n, slices, row_ptrs, col_idxs, vals, b + done, b_stride,
                                           c + done, c_stride, alpha, beta);
            if (err) return err;
            done += 8 * slices;
        }
        if (r - done >= 4) {
            const int slices = (r - done) / 4;
            const int err = launch_spmm<4>(stream, n, slices, row_ptrs, col_idxs, vals, b + done, b_stride,
                                           c + done, c_stride, alpha, beta);
            if (err) return err;
            done += 4 * slices;
        }
        if (r - done >= 2) {
            const int err = launch_spmm<2>(stream, n, 1, row_ptrs, col_idxs, vals, b + done, b_stride,
                                           c + done, c_stride, alpha, beta);
            if (err) return err;
            done += 2;
        }
        if (r - done == 1) {
            return launch_stream<256, 1, 1536>(stream, !no_swizzle, chunk, n, 1, row_ptrs, col_idxs, vals,
                                               b + done, b_stride, c + done, c_stride, alpha, beta);
        }
        return GKOMI_SUCCESS;
    }
    if (kind == GKOMI_CSR_SPLIT) {
        if (automatic) variant = nt ? 2 : 0;
        // longest row - 1 nonzeros may lie behind the tile a row starts in
        int over = split_max_over;
        if (max_row_nnz_hint >= 1 && max_row_nnz_hint <= split_max_over) {
            over = static_cast<int>((max_row_nnz_hint - 1 + 1) / 2 * 2);
        } else if (max_row_nnz_hint == 0) {
            over = 0;
        }
        const int z = static_cast<int>(nnz);
        for (int j = 0; j < r; ++j) {  // explicit strategy with several columns: one launch each
            int err = GKOMI_EINVAL;
#define GKOMI_SPLIT_ARGS                                                            \
    stream, !no_swizzle, n, z, row_ptrs, col_idxs, vals, b + j, b_stride, c + j,    \
        c_stride, alpha, beta, srow, over, chunk > 0 ? chunk : auto_chunk
            // variant bit 2: nontemporal streams.  (Write-through stores of c, 8 or 16 bytes wide, and
            // nontemporal stores of c were measured and dropped: 17.2 / 16.9 vs 16.6 us cold,
            // profiles/r02_tune_split.log, r02_tune_ntstore.log.)
#define GKOMI_SPLIT_TILE(BLOCK, TILE)                                               \
    if (variant & 2) {                                                              \
        err = launch_split<BLOCK, TILE, true>(GKOMI_SPLIT_ARGS);                    \
    } else {                                                                        \
        err = launch_split<BLOCK, TILE, false>(GKOMI_SPLIT_ARGS);                   \
    }
            if (srow_tile == 1536) {
                GKOMI_SPLIT_TILE(256, 1536)
            } else if (srow_tile == 1024) {
                GKOMI_SPLIT_TILE(256, 1024)
            } else if (srow_tile == 3072) {
                GKOMI_SPLIT_TILE(256, 3072)
            } else {
                GKOMI_SPLIT_TILE(256, 2048)
            }
#undef GKOMI_SPLIT_TILE
#undef GKOMI_SPLIT_ARGS
            if (err) return err;
        }
        return GKOMI_SUCCESS;
    }
    if (kind == GKOMI_CSR_STREAM) {
        switch (variant) {
        case 1: return launch_stream<256, 2, 4096>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 2: return launch_stream<512, 1, 4096>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 3: return launch_stream<256, 4, 8192>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 4: return launch_stream<128, 1, 1024>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 5: return launch_stream<256, 1, 1536>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 6: return launch_stream<512, 1, 3072>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 7: return launch_stream<1024, 1, 6144>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 8: return launch_stream<64, 1, 384>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 9: return launch_stream<192, 1, 1152>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 10: return launch_stream<320, 1, 1920>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 11: return launch_stream<128, 1, 768>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 12: return launch_stream<256, 1, 1024>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 13: return launch_stream<384, 1, 2304>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 14: return launch_stream<256, 1, 1536, true>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 15: return launch_stream<256, 1, 1536, false, true>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        case 16: return launch_stream<256, 1, 1536, true, true>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        default: return launch_stream<256, 1, 2048>(stream, !no_swizzle, chunk, GKOMI_ARGS);
        }
    }
    if (kind == GKOMI_CSR_BALANCED) {
        // c = 0 (or beta * c) first, then atomics: reference load_balance does
        // dense::fill + atomic_add (hip/matrix/csr_kernels.hip.cpp:293-309)
        int err = alpha == nullptr
                      ? gkomi_dense_fill_f64(stream_, nrows, nrhs, c, c_stride, 0.0)
                      : gkomi_dense_scale_f64(stream_, nrows, nrhs, beta, 1, c, c_stride);
        if (err) return err;
        if (nnz == 0) return GKOMI_SUCCESS;
        // nontemporal matrix streams by the rule of the stream kernels; variant 1: every segment by one thread;
        // GKOMI_CSR_COLBLOCK: one pass per column window of b (see csr_balanced_kernel)
        const int passes = (strategy & GKOMI_CSR_COLBLOCK) != 0 && variant != 1 ? colblock_passes(ncols, b_stride) : 1;
        // (nontemporal matrix streams in the windowed passes, so that they do not push the window of b out of L2, were
        // measured and lost: 138 vs 126 us on the power-law class -- the 170 MB matrix is Infinity-Cache resident
        // between the passes and nontemporal loads give that up; profiles/r04_colblock_sweep.md)
        const bool bnt = automatic ? nt : (!csr_auto_swizzle(nrows, nnz) || (strategy & GKOMI_CSR_STREAMING) != 0);
        const int z = static_cast<int>(nnz), nc = static_cast<int>(ncols);
        // the matrix's srow spares the search for the rows of a tile when it was built for a tile of this kernel
        const bool with_srow = srow != nullptr && variant != 1 && nnz >= 2 &&
                               (srow_tile == 1536 || srow_tile == 2048 || srow_tile == 3072);
        const int32_t* sr = with_srow ? srow : nullptr;
#define GKOMI_BALANCED_TILE(TILE)                                                                                 \
    (alpha != nullptr                                                                                              \
         ? (bnt ? launch_balanced<TILE, true, true>(stream, n, nc, r, z, row_ptrs, col_idxs, vals, b, b_stride, c, \
                                                    c_stride, alpha, sr, passes, variant == 1)                     \
                : launch_balanced<TILE, true, false>(stream, n, nc, r, z, row_ptrs, col_idxs, vals, b, b_stride,  \
                                                     c, c_stride, alpha, sr, passes, variant == 1))                \
         : (bnt ? launch_balanced<TILE, false, true>(stream, n, nc, r, z, row_ptrs, col_idxs, vals, b, b_stride,   \
                                                     c, c_stride, alpha, sr, passes, variant == 1)                 \
                : launch_balanced<TILE, false, false>(stream, n, nc, r, z, row_ptrs, col_idxs, vals, b, b_stride,  \
                                                      c, c_stride, alpha, sr, passes, variant == 1)))
        if (with_srow && srow_tile == 2048) return GKOMI_BALANCED_TILE(2048);
        if (with_srow && srow_tile == 3072) return GKOMI_BALANCED_TILE(3072);
        return GKOMI_BALANCED_TILE(1536);
#undef GKOMI_BALANCED_TILE
    }
    if (kind == GKOMI_CSR_VECTOR) {
        int64_t len = max_row_nnz_hint;
        if (variant != 0) len = variant;  // explicit sub-wave width for tests
        if (len < 0) len = 8;
        if (len <= 2) return launch_vector<2>(stream, GKOMI_ARGS);
        if (len <= 4) return launch_vector<4>(stream, GKOMI_ARGS);
        if (len <= 8) return launch_vector<8>(stream, GKOMI_ARGS);
        if (len <= 16) return launch_vector<16>(stream, GKOMI_ARGS);
        if (len <= 32) return launch_vector<32>(stream, GKOMI_ARGS);
        return launch_vector<64>(stream, GKOMI_ARGS);
    }
#undef GKOMI_ARGS
    return GKOMI_EINVAL;
}


extern "C" int gkomi_csr_max_row_nnz_i32(gkomi_stream_t stream_,
                                          int64_t nrows,
                                          const int32_t* row_ptrs,
                                          int32_t* result)
{
    using namespace gkomi;
    hipStream_t stream = to_stream(stream_);
    int err = static_cast<int>(hipMemsetAsync(result, 0, sizeof(int32_t), stream));
    if (err) return err;
    if (nrows <= 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(csr_max_row_nnz_kernel<int32_t>, dim3(grid_for(nrows, 256)),
                       dim3(256), 0, stream, nrows, row_ptrs, result);
    return check_launch();
}


extern "C" int gkomi_csr_analyse_gather_i32(gkomi_stream_t stream_, int64_t ncols, int64_t nnz, const int32_t* col_idxs,
                                            double* scratch, int* host_flags, int64_t* host_footprint_bytes)
{
    using namespace gkomi;
    if (host_flags == nullptr || scratch == nullptr || nnz < 0 || ncols < 0) return GKOMI_EINVAL;
    *host_flags = 0;
    if (host_footprint_bytes != nullptr) *host_footprint_bytes = 0;
    if (nnz == 0 || ncols == 0) return GKOMI_SUCCESS;
    hipStream_t stream = to_stream(stream_);
    constexpr int tile = 1536;
    const int64_t ntiles = ceildiv(nnz, tile);
    const int64_t step = std::max<int64_t>(1, ntiles / 4096);   // at most ~4096 evenly spaced tiles
    const int64_t sampled = ceildiv(ntiles, step);
    int err = static_cast<int>(hipMemsetAsync(scratch, 0, 2 * sizeof(double), stream));
    if (err) return err;
    // 64 KiB pages of b (8192 columns), coarser when b has more than 4096 of them
    const int page_cols = static_cast<int>(std::max<int64_t>(8192, ceildiv(ncols, 4096)));
    hipLaunchKernelGGL(csr_col_spread_kernel, dim3(static_cast<unsigned>(sampled)), dim3(256), 0, stream, nnz, col_idxs,
                       tile, step, page_cols, scratch);
    err = check_launch();
    if (err) return err;
    double h[2] = {0.0, 0.0};
    err = static_cast<int>(hipMemcpyAsync(h, scratch, sizeof(h), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    if (err) return err;
    const double pages = h[1] > 0.0 ? h[0] / h[1] : 0.0;
    // bytes of b (one column) a tile's gathers range over; a single page says nothing (a tile's rows span < a page)
    const int64_t footprint = pages <= 1.0 ? 0 : std::min<int64_t>(static_cast<int64_t>(pages * page_cols * 8.0), 8 * ncols);
    if (host_footprint_bytes != nullptr) *host_footprint_bytes = footprint;
    // windows pay when a tile's gathers range over more of b than an XCD's L2 keeps AND b itself is larger than that
    if (footprint > (int64_t{3} << 20) && colblock_passes(ncols, 1) > 1) *host_flags = GKOMI_CSR_COLBLOCK;
    return GKOMI_SUCCESS;
}


// ---- <double, int64>: the instantiation a 288 GB part needs (nnz > 2^31; include/ginkgo/core/base/types.hpp:544-560
// lists it next to <double, int32>).  Same kernels (templates over the index type), 16 B per nonzero.  The automatic
// strategy cuts by nonzeros when the matrix carries its srow (rows of any length are correct: a row longer than the 64
// nonzeros read behind its tile is finished from memory), by rows otherwise; "classical" / "load_balance" requests are
// served by the row-cut stream kernel (bit-exact for any row lengths, no sub-wave / atomic variants for this index type).
extern "C" int gkomi_csr_make_srow_i64(gkomi_stream_t stream_, int64_t nrows, int64_t nnz, const int64_t* row_ptrs,
                                       int64_t tile, int64_t* srow, int64_t nsrow)
{
    using namespace gkomi;
    if (nrows < 0 || nnz < 0 || tile <= 0 || tile % 2 != 0) return GKOMI_EINVAL;
    if (tile > (1 << 20) || nnz / tile + 2 > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    if (nsrow < gkomi_csr_srow_entries(nnz, tile)) return GKOMI_EWORKSPACE;
    const int64_t ntiles = nnz / tile + 1;
    hipLaunchKernelGGL(csr_make_srow_kernel<int64_t>, dim3(static_cast<unsigned>(ceildiv(ntiles + 1, 256))), dim3(256), 0,
                       to_stream(stream_), nrows, row_ptrs, static_cast<int>(tile), ntiles, srow);
    hipLaunchKernelGGL(csr_srow_most_rows_kernel<int64_t>, dim3(static_cast<unsigned>(ceildiv(ntiles, 256))), dim3(256), 0,
                       to_stream(stream_), ntiles, srow);
    hipLaunchKernelGGL(csr_srow_mark_kernel<int64_t>, dim3(static_cast<unsigned>(ceildiv(ntiles + 1, 256))), dim3(256), 0,
                       to_stream(stream_), ntiles, srow);
    return check_launch();
}

extern "C" int gkomi_csr_max_row_nnz_i64(gkomi_stream_t stream_, int64_t nrows, const int64_t* row_ptrs, int64_t* result)
{
    using namespace gkomi;
    hipStream_t stream = to_stream(stream_);
    int err = static_cast<int>(hipMemsetAsync(result, 0, sizeof(int64_t), stream));
    if (err) return err;
    if (nrows <= 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(csr_max_row_nnz_kernel<int64_t>, dim3(grid_for(nrows, 256)), dim3(256), 0, stream, nrows, row_ptrs,
                       result);
    return check_launch();
}

extern "C" int gkomi_csr_spmv_srow_f64_i64(gkomi_stream_t stream_, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
                                           const int64_t* row_ptrs, const int64_t* col_idxs, const double* vals,
                                           const double* b, int64_t b_stride, double* c, int64_t c_stride,
                                           const double* alpha, const double* beta, int strategy,
                                           int64_t max_row_nnz_hint, const int64_t* srow, int64_t srow_tile)
{
    using namespace gkomi;
    if (nrows < 0 || ncols < 0 || nrhs < 0 || nnz < 0) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrhs > 65535 || ceildiv(nrows, 256) + 8 > INT32_MAX || nnz / 1536 + 2 > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    // pairs of values and of 8-byte column indices are loaded with one 16-byte instruction each
    if (reinterpret_cast<uintptr_t>(vals) % 16 != 0 || reinterpret_cast<uintptr_t>(col_idxs) % 16 != 0) {
        return GKOMI_ENOTSUPPORTED;
    }
    hipStream_t stream = to_stream(stream_);
    const int kind = strategy & 0xff;
    const bool split_ok = srow != nullptr && nnz >= 2 && (srow_tile == 1536 || srow_tile == 2048 || srow_tile == 3072);
    if (kind == GKOMI_CSR_SPLIT && !split_ok) return srow == nullptr ? GKOMI_EINVAL : GKOMI_ENOTSUPPORTED;
    // nontemporal matrix streams for a matrix the Infinity Cache will not hold at its next use (csr_auto_swizzle's rule
    // at 16 B per nonzero), or when the caller says so
    const bool resident = 16 * nnz + 24 * nrows < (int64_t{288} << 20);
    const bool nt = !resident || (strategy & GKOMI_CSR_STREAMING) != 0;
    const int r = static_cast<int>(nrhs);
    if (split_ok && (kind == GKOMI_CSR_SPLIT || (kind == GKOMI_CSR_AUTO && r == 1 && nnz >= nrows))) {  // (fewer nonzeros than rows: by rows)
        int over = split_max_over;
        if (max_row_nnz_hint >= 1 && max_row_nnz_hint <= split_max_over) {
            over = static_cast<int>(max_row_nnz_hint / 2 * 2);
        } else if (max_row_nnz_hint == 0) {
            over = 0;
        }
        // streams from HBM: 16 consecutive tiles per XCD (see the int32 entry); resident: one eighth each
        const int chunk = resident ? 0 : 16;
        for (int j = 0; j < r; ++j) {
            int err;
#define GKOMI_SPLIT64(TILE)                                                                                          \
    err = nt ? launch_split<256, TILE, true, int64_t>(stream, true, nrows, nnz, row_ptrs, col_idxs, vals, b + j, b_stride, \
                                                      c + j, c_stride, alpha, beta, srow, over, chunk)               \
             : launch_split<256, TILE, false, int64_t>(stream, true, nrows, nnz, row_ptrs, col_idxs, vals, b + j,    \
                                                       b_stride, c + j, c_stride, alpha, beta, srow, over, chunk)
            if (srow_tile == 1536) {
                GKOMI_SPLIT64(1536);
            } else if (srow_tile == 2048) {
                GKOMI_SPLIT64(2048);
            } else {
                GKOMI_SPLIT64(3072);
            }
#undef GKOMI_SPLIT64
            if (err) return err;
        }
        return GKOMI_SUCCESS;
    }
    if (nt) {
        return launch_stream<256, 1, 1536, true, false, int64_t>(stream, resident, 0, nrows, r, row_ptrs, col_idxs, vals, b,
                                                                  b_stride, c, c_stride, alpha, beta);
    }
    return launch_stream<256, 1, 1536, false, false, int64_t>(stream, resident, 0, nrows, r, row_ptrs, col_idxs, vals, b,
                                                               b_stride, c, c_stride, alpha, beta);
}

extern "C" int gkomi_csr_spmv_f64_i64(gkomi_stream_t stream_, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
                                      const int64_t* row_ptrs, const int64_t* col_idxs, const double* vals, const double* b,
                                      int64_t b_stride, double* c, int64_t c_stride, const double* alpha,
                                      const double* beta, int strategy, int64_t max_row_nnz_hint)
{
    return gkomi_csr_spmv_srow_f64_i64(stream_, nrows, ncols, nrhs, nnz, row_ptrs, col_idxs, vals, b, b_stride, c, c_stride,
                                       alpha, beta, strategy, max_row_nnz_hint, nullptr, 0);
}

/* diagnostics: how many automatic applies of this process found their matrix evicted (csr_probably_evicted) */
extern "C" int64_t gkomi_diag_csr_evicted_applies(void) { return gkomi::csr_evicted_applies.load(); }
