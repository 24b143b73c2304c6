// COO SpMV tile kernel for gfx950, several right-hand sides per pass and
// atomic-free paths for row-sorted matrices.  Replaces
// gko::kernels::hip::coo::{spmv, advanced_spmv, spmv2, advanced_spmv2}
// (core/matrix/coo_kernels.hpp; common/cuda_hip/matrix/coo_kernels.hpp.inc:57-222
// is the reference's segmented-scan + atomics kernel).
//
// One workgroup owns a tile of nonzeros (1536 for one column, 1024 / 512 for
// 2 / 4 columns per pass): 16 B per nonzero streamed with 16-/8-B per-lane
// loads, the NR columns of b gathered per nonzero (contiguous in the row-major
// b), products into LDS column by column.  The thread that owns the first
// element of a row segment adds the segment left to right -- the reference's
// order inside a row (reference/matrix/coo_kernels.cpp:92-131) -- and the
// (row, sums) pairs are compacted so that consecutive lanes write consecutive
// rows.  What happens with a segment depends on the variant:
//
//   any order (CMode < 0): one fp64 atomic per segment and column on top of a
//     c that the caller zeroed / scaled (the reference's composition: fill or
//     scale, then spmv2).
//
//   row-sorted (CMode 0 / 1 / 2: c = Ab, c = beta c + alpha Ab, c += alpha Ab),
//   no atomics, no fill / scale launch before, every sum in a fixed order:
//
//     short rows (Halo = 64; the caller vouches for rows of at most 64
//       nonzeros): a row belongs to the tile in which it ENDS.  The tile also
//       loads the 64 nonzeros in front of it, so a row that started in the
//       previous tile is summed from its first element -- every row in the
//       reference's order, one launch.  Rows without nonzeros are written by
//       the tile that sees the gap in the row indices.
//
//     any row length (Halo = 1): a row whose nonzeros all lie in one tile is
//       stored by that tile; a row cut by a tile boundary leaves one partial
//       sum per tile in a carry slot of the workspace and a second, small
//       launch (one wave per tile) adds the slots of a chain in tile order.
//
// Algorithmic bytes per column: 16 nnz + 8 ncols + 8 nrows (sorted; the
// any-order composition adds the 16 nrows of the fill and the atomics).
#include "common.hpp"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <climits>
#include <cstdlib>

#define GKOMI_TRY(expr)          \
    do {                         \
        const int err_ = (expr); \
        if (err_) return err_;   \
    } while (0)

namespace gkomi {
namespace {

constexpr int block = 256;
constexpr int max_nr = 8;
constexpr int halo_rows = 64;   // longest row the one-launch sorted variant takes
constexpr int min_tile = 512;
constexpr int coop_min = 128;    // row segments longer than this are summed by a whole wave

// nonzeros per thread / per workgroup for NR columns per pass (LDS: 8 NR + 4 B per nonzero)
template <int NR>
struct tile_shape {
    static constexpr int items = NR == 1 ? 6 : (NR == 2 ? 4 : 2);  // 8 columns: 2 too (37 KB of LDS)
    static constexpr int tile = block * items;
};

// partial sum of a row that is cut by a tile boundary
struct carry_slot {
    int32_t row;
    int32_t flags;  // bit 0: valid, bit 1: the row goes on in the next tile
    double sum[max_nr];
};
constexpr int carry_valid = 1, carry_continues = 2;

struct sorted_header {
    int32_t unsorted;     // gkomi_coo_analyse_rows_i32's findings
    int32_t max_row_nnz;  // capped at halo_rows + 1
    int32_t violation;    // sticky: a row longer than the caller's hint met the short-row kernel
    int32_t pad_;
};

template <int CMode>
__device__ __forceinline__ double combine(double old, double beta, double sum)
{
    return CMode == 0 ? sum : (CMode == 1 ? beta * old + sum : old + sum);
}

// element (row, j) of a row-major matrix; Off32: every byte offset fits in 32
// bits (the host checks), which keeps the address arithmetic to one multiply
template <bool Off32, typename T>
__device__ __forceinline__ T* at(T* base, int row, int64_t stride, int j = 0)
{
    if (Off32) {
        return base + (static_cast<uint32_t>(row) * static_cast<uint32_t>(stride) + static_cast<uint32_t>(j));
    }
    return base + (row * stride + j);
}

// rows [from, to) have no nonzeros: c = 0 or beta c, by the whole wave for
// every lane that `wants` it (uniform control flow required)
template <int NR, int CMode, bool Off32>
__device__ __forceinline__ void wave_fill_empty_rows(bool wants, int from, int to, double* __restrict__ c,
                                                     int64_t c_stride, double beta)
{
    unsigned long long todo = __ballot(wants);
    const int lane = threadIdx.x & (wave_size - 1);
    while (todo != 0) {
        const int src = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1;
        const int g0 = __shfl(from, src), g1 = __shfl(to, src);
        for (int64_t i = lane; i < static_cast<int64_t>(g1 - g0) * NR; i += wave_size) {
            double* p = at<Off32>(c, g0 + static_cast<int>(i / NR), c_stride, static_cast<int>(i % NR));
            *p = CMode == 1 ? beta * *p : 0.0;
        }
    }
}

// Halo: nonzeros in front of the tile that are loaded too (64: short-row
// variant; 1: only the row index, for the carry variant; unused for atomics)
template <int NR, int CMode, int Halo, bool Off32>
__global__ __launch_bounds__(block) void coo_tile_kernel(
    int64_t nnz, int64_t nrows, const int32_t* __restrict__ row_idxs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c, int64_t c_stride,
    const double* __restrict__ alpha_p, const double* __restrict__ beta_p,
    carry_slot* __restrict__ carries, sorted_header* __restrict__ hdr)
{
    constexpr int items = tile_shape<NR>::items, tile = tile_shape<NR>::tile;
    constexpr bool owner = CMode >= 0;
    constexpr bool with_halo = owner && Halo > 1;
    constexpr bool with_carries = owner && Halo == 1;
    constexpr bool fills_gaps = owner && CMode != 2;
    constexpr bool b_vec = NR % 2 == 0;
    constexpr int off = Halo == 1 ? 2 : Halo;  // LDS slot of the tile's first nonzero (even: 16-B stores)
    constexpr int cap = off + tile - 1;        // last product slot
    __shared__ __attribute__((aligned(16))) double prod[NR][off + tile];
    __shared__ __attribute__((aligned(8))) int32_t rowid[off + tile + 2];
    __shared__ int32_t seg_start[tile];
    __shared__ int wave_heads[block / wave_size];
    // segments of more than coop_min products (Halo = 64 vouches for rows of at most 64)
    constexpr bool coop = !with_halo;
    __shared__ int s_nlong;
    __shared__ int s_long[coop ? tile / (coop_min + 1) + 2 : 1];
    if (threadIdx.x == 0) s_nlong = 0;
    b += static_cast<int64_t>(blockIdx.y) * NR;
    c += static_cast<int64_t>(blockIdx.y) * NR;
    const double alpha = alpha_p != nullptr ? alpha_p[0] : 1.0;
    const double beta = CMode == 1 ? beta_p[0] : 0.0;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * tile;
    const int count = static_cast<int>(min(static_cast<int64_t>(tile), nnz - base));
    const int tid = threadIdx.x;
    const int lane = tid & (wave_size - 1);
    const int wave = tid / wave_size;
    {
        constexpr int pairs = items / 2;
        const double* tile_vals = vals + base;
        const int32_t* tile_rows = row_idxs + base;
        const int32_t* tile_cols = col_idxs + base;
        double2 v[pairs];
        int2 r[pairs], cc[pairs];
        if (count == tile) {  // every tile but the last: nothing between the loads
#pragma unroll
            for (int u = 0; u < pairs; ++u) {
                const unsigned e = 2 * (tid + u * block);
                v[u] = *reinterpret_cast<const double2*>(tile_vals + e);
                r[u] = *reinterpret_cast<const int2*>(tile_rows + e);
                cc[u] = *reinterpret_cast<const int2*>(tile_cols + e);
            }
        } else {
#pragma unroll
            for (int u = 0; u < pairs; ++u) {
                const int e = 2 * (tid + u * block);
                v[u] = make_double2(0.0, 0.0);
                r[u] = make_int2(0, 0);
                cc[u] = make_int2(0, 0);
                if (e + 1 < count) {
                    v[u] = *reinterpret_cast<const double2*>(tile_vals + e);
                    r[u] = *reinterpret_cast<const int2*>(tile_rows + e);
                    cc[u] = *reinterpret_cast<const int2*>(tile_cols + e);
                } else if (e < count) {
                    v[u].x = tile_vals[e];
                    r[u].x = tile_rows[e];
                    cc[u].x = tile_cols[e];
                }
            }
        }
        // the neighbours: the last wave loads the halo (or the row index in front
        // of the tile) and the row index behind the tile.  -1 / -2: no such element
        double hv = 0.0;
        int hr = -1, hc = 0;
        int next_row = -2;
        const int h = tid - (block - wave_size);  // lane of the last wave
        if (owner && h >= 0) {
            if (with_halo) {
                const int64_t g = base - Halo + h;
                if (g >= 0) {
                    hv = vals[g];
                    hr = row_idxs[g];
                    hc = col_idxs[g];
                }
            } else if (h == wave_size - 1 && base > 0) {
                hr = row_idxs[base - 1];
            }
            if (h == 0 && base + count < nnz) next_row = row_idxs[base + count];
        }
        double x0[pairs][NR], x1[pairs][NR];
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            const double* s0 = at<Off32>(b, cc[u].x, b_stride);
            const double* s1 = at<Off32>(b, cc[u].y, b_stride);
            if (b_vec) {
#pragma unroll
                for (int j = 0; j < NR; j += 2) {
                    const double2 t0 = *reinterpret_cast<const double2*>(s0 + j);
                    const double2 t1 = *reinterpret_cast<const double2*>(s1 + j);
                    x0[u][j] = t0.x;
                    x0[u][j + 1 < NR ? j + 1 : j] = t0.y;
                    x1[u][j] = t1.x;
                    x1[u][j + 1 < NR ? j + 1 : j] = t1.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    x0[u][j] = s0[j];
                    x1[u][j] = s1[j];
                }
            }
        }
        if (owner && h >= 0) {
            if (with_halo) {
                const double av = alpha * hv;
                const double* sh = at<Off32>(b, hc, b_stride);
#pragma unroll
                for (int j = 0; j < NR; ++j) prod[j][h] = av * sh[j];
                rowid[h] = hr;
            } else if (h == wave_size - 1) {
                rowid[off - 1] = hr;
            }
            if (h == 0) rowid[off + count] = next_row;
        }
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            const int e = 2 * (tid + u * block);
            const double a0 = alpha * v[u].x, a1 = alpha * v[u].y;  // alpha == 1: exact
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                *reinterpret_cast<double2*>(&prod[j][off + e]) = make_double2(a0 * x0[u][j], a1 * x1[u][j]);
            }
            if (count == tile) {
                *reinterpret_cast<int2*>(rowid + off + e) = r[u];
            } else {  // the last tile: slot `count` belongs to the row index behind the tile
                if (e < count) rowid[off + e] = r[u].x;
                if (e + 1 < count) rowid[off + e + 1] = r[u].y;
            }
        }
    }
    __syncthreads();
    // The tile as CSR in LDS: the positions where a row segment starts, in order.
    // Nonzero 0 always starts one (of this tile), whatever the tile in front holds.
    const int first = off + tid * items;
    const int end = off + count;
    int ids[items + 1];
#pragma unroll
    for (int u = 0; u <= items; ++u) ids[u] = rowid[first + u - 1];
    int nheads = 0;
#pragma unroll
    for (int u = 0; u < items; ++u) {
        nheads += (first + u < end && (ids[u + 1] != ids[u] || first + u == off)) ? 1 : 0;
    }
    int incl = nheads;
#pragma unroll
    for (int d = 1; d < wave_size; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == wave_size - 1) wave_heads[wave] = incl;
    __syncthreads();
    int offset = incl - nheads;
    int nseg = 0;
#pragma unroll
    for (int w = 0; w < block / wave_size; ++w) {
        const int wc = wave_heads[w];
        if (w < wave) offset += wc;
        nseg += wc;
    }
#pragma unroll
    for (int u = 0; u < items; ++u) {
        if (first + u < end && (ids[u + 1] != ids[u] || first + u == off)) seg_start[offset++] = first + u;
    }
    __syncthreads();
    // one thread per segment: its products added left to right -- the reference's
    // order inside a row -- and consecutive lanes write consecutive rows
    carry_slot* my_carries = with_carries
                                 ? carries + 2 * (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x)
                                 : nullptr;
    // what becomes of a finished segment sum; called by whole waves (the empty-row fill is wave-cooperative),
    // `active` lanes hold a segment
    auto finish = [&](bool active, int i, int row, int before, bool starts_here, const double (&sum)[NR]) {
        bool mine = active;
        if (owner) {
            const bool ends_here = i + 1 < nseg || rowid[end] != row;
            if (with_halo) {
                mine = active && ends_here;  // the tile in which the row ends owns it
            } else {
                mine = active && starts_here && ends_here;
                if (active && !mine) {
                    carry_slot* slot = my_carries + (starts_here ? 1 : 0);
                    slot->row = row;
                    slot->flags = carry_valid | (ends_here ? 0 : carry_continues);
#pragma unroll
                    for (int j = 0; j < NR; ++j) slot->sum[j] = sum[j];
                }
                // the slots this tile does not use are marked so on every launch
                if (active && i == 0 && starts_here) my_carries[0].flags = 0;
                if (active && i == nseg - 1 && (ends_here || !starts_here)) my_carries[1].flags = 0;
            }
        }
        if (mine) {
            double* dst = at<Off32>(c, row, c_stride);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (owner) {
                    dst[j] = combine<CMode>(CMode == 0 ? 0.0 : dst[j], beta, sum[j]);
                } else {
                    unsafeAtomicAdd(dst + j, sum[j]);
                }
            }
        }
        if (fills_gaps) {
            // rows without nonzeros in front of a row that starts here, and behind
            // the last nonzero of the matrix
            wave_fill_empty_rows<NR, CMode, Off32>(active && starts_here && row - before > 1, before + 1, row, c,
                                                   c_stride, beta);
            wave_fill_empty_rows<NR, CMode, Off32>(active && i == nseg - 1 && base + count == nnz && row + 1 < nrows,
                                                   row + 1, static_cast<int>(nrows), c, c_stride, beta);
        }
    };
    for (int ib = 0; ib < nseg; ib += block) {
        const int i = ib + tid;
        bool active = i < nseg;
        int from = active ? seg_start[i] : off;
        const int to = !active ? from : (i + 1 < nseg ? seg_start[i + 1] : end);
        const int row = rowid[from];
        const int before = owner ? rowid[from - 1] : -1;
        const bool starts_here = before != row;
        if (with_halo && active && i == 0 && !starts_here) {  // back to the row's first nonzero
            while (from > 0 && rowid[from - 1] == row) --from;
            if (from == 0 && base > Halo) atomicOr(&hdr->violation, 1);  // longer than the caller said
        }
        // a long segment is a dependent chain of additions on one lane (up to a whole tile: ~6 us per
        // workgroup): it waits for a whole wave (below)
        if (coop && active && to - from > coop_min) {
            s_long[atomicAdd(&s_nlong, 1)] = i;
            active = false;
        }
        double sum[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) sum[j] = 0.0;
        for (int k = from; active && k < to; k += 4) {
            double p[4][NR];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < NR; ++j) p[u][j] = prod[j][min(k + u, cap)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < NR; ++j) sum[j] = k + u < to ? sum[j] + p[u][j] : sum[j];
            }
        }
        finish(active, i, row, before, starts_here, sum);
    }
    if (!coop) return;
    // long segments: one wave each -- lane-strided partial sums in index order, then the fixed xor tree (the
    // role of the reference's segmented scan, common/cuda_hip/matrix/coo_kernels.hpp.inc:57-120); same bits every run
    __syncthreads();
    const int nlong = s_nlong;
    for (int q = wave; q < nlong; q += block / wave_size) {
        const int i = s_long[q];
        const int from = seg_start[i];
        const int to = i + 1 < nseg ? seg_start[i + 1] : end;
        const int row = rowid[from];
        const int before = owner ? rowid[from - 1] : -1;
        double sum[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            double a0 = 0.0, a1 = 0.0;
            int k = from + lane;
            for (; k + wave_size < to; k += 2 * wave_size) {
                a0 += prod[j][k];
                a1 += prod[j][k + wave_size];
            }
            if (k < to) a0 += prod[j][k];
            sum[j] = wave_reduce_sum(a0 + a1);
        }
        finish(lane == 0, i, row, before, before != row, sum);
    }
}

// One wave per tile: the tile where a cut row starts adds the partial sums of
// the tiles the row runs through, in tile order, and stores the row.
template <int NR, int CMode>
__global__ __launch_bounds__(block) void coo_carry_kernel(
    int64_t ntiles, const carry_slot* __restrict__ carries, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ beta_p)
{
    const int lane = threadIdx.x & (wave_size - 1);
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * (block / wave_size) + threadIdx.x / wave_size;
    if (tile >= ntiles) return;
    carries += 2 * static_cast<int64_t>(blockIdx.y) * ntiles;
    c += static_cast<int64_t>(blockIdx.y) * NR;
    const carry_slot& start = carries[2 * tile + 1];
    if (!(start.flags & carry_valid)) return;
    const int row = start.row;
    double total[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) total[j] = start.sum[j];
    bool done = false;
    for (int64_t t = tile + 1; !done; t += wave_size) {
        const int64_t mine = t + lane;
        bool ok = false, cont = false;
        double part[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) part[j] = 0.0;
        if (mine < ntiles) {
            const carry_slot& s = carries[2 * mine];
            ok = (s.flags & carry_valid) && s.row == row;
            cont = ok && (s.flags & carry_continues);
            if (ok) {
#pragma unroll
                for (int j = 0; j < NR; ++j) part[j] = s.sum[j];
            }
        }
        // lanes [0, stop) go on into the next tile; lane `stop` ends the chain
        const unsigned long long going = __ballot(cont);
        const int stop = going == ~0ull ? wave_size : __ffsll(static_cast<long long>(~going)) - 1;
        const bool included = lane < stop || (lane == stop && ok);
        done = stop < wave_size;
#pragma unroll
        for (int j = 0; j < NR; ++j) total[j] += wave_reduce_sum(included ? part[j] : 0.0);
    }
    if (lane == 0) {
        const double beta = CMode == 1 ? beta_p[0] : 0.0;
        double* at = c + row * c_stride;
#pragma unroll
        for (int j = 0; j < NR; ++j) at[j] = combine<CMode>(CMode == 0 ? 0.0 : at[j], beta, total[j]);
    }
}

// sortedness and the longest run of equal row indices (capped at halo_rows + 1)
__global__ __launch_bounds__(block) void coo_analyse_rows_kernel(int64_t nnz,
                                                                 const int32_t* __restrict__ row_idxs,
                                                                 sorted_header* hdr)
{
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    bool bad = false;
    int longest = 0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < nnz; i += step) {
        const int row = row_idxs[i];
        if (i + 1 < nnz) {
            const int next = row_idxs[i + 1];
            bad |= row > next;
            if (next == row) continue;
        }
        // i ends a run: its length, counted backwards
        int len = 1;
        while (len <= halo_rows && i - len >= 0 && row_idxs[i - len] == row) ++len;
        longest = max(longest, len);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off, 64));
    const bool any_bad = __ballot(bad) != 0;
    if ((threadIdx.x & (wave_size - 1)) == 0) {
        atomicMax(&hdr->max_row_nnz, longest);
        if (any_bad) atomicOr(&hdr->unsorted, 1);
    }
}

__global__ void coo_clear_header_kernel(sorted_header* hdr)
{
    hdr->unsorted = 0;
    hdr->max_row_nnz = 0;
    hdr->violation = 0;
}

inline bool aligned_to(const void* p, size_t a) { return reinterpret_cast<uintptr_t>(p) % a == 0; }

struct coo_args {
    int64_t nrows, nnz;
    const int32_t* rows;
    const int32_t* cols;
    const double* vals;
    const double* b;
    int64_t b_stride;
    double* c;
    int64_t c_stride;
    const double* alpha;
    const double* beta;
    carry_slot* carries;
    sorted_header* hdr;
    bool off32;  // every element of b and c within 4 GiB of its base: 32-bit offsets
};

inline bool offsets_fit_32(int64_t nrows, int64_t ncols, int64_t b_stride, int64_t c_stride)
{
    constexpr int64_t limit = (int64_t{1} << 32) / 8 - 16;
    return ncols >= 0 && nrows * c_stride < limit && ncols * b_stride < limit;
}

template <int NR, int CMode, int Halo>
int launch_tile(hipStream_t s, int groups, const coo_args& a, int64_t col)
{
    const int64_t ntiles = ceildiv(a.nnz, tile_shape<NR>::tile);
    if (ntiles > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    dim3 grid(static_cast<unsigned>(ntiles), static_cast<unsigned>(groups));
    if (a.off32) {
        hipLaunchKernelGGL((coo_tile_kernel<NR, CMode, Halo, true>), grid, dim3(block), 0, s, a.nnz, a.nrows,
                           a.rows, a.cols, a.vals, a.b + col, a.b_stride, a.c + col, a.c_stride, a.alpha,
                           a.beta, a.carries, a.hdr);
    } else {
        hipLaunchKernelGGL((coo_tile_kernel<NR, CMode, Halo, false>), grid, dim3(block), 0, s, a.nnz, a.nrows,
                           a.rows, a.cols, a.vals, a.b + col, a.b_stride, a.c + col, a.c_stride, a.alpha,
                           a.beta, a.carries, a.hdr);
    }
    if (CMode >= 0 && Halo == 1 && ntiles > 1) {
        dim3 fix(static_cast<unsigned>(ceildiv(ntiles, block / wave_size)), static_cast<unsigned>(groups));
        hipLaunchKernelGGL((coo_carry_kernel<NR, (CMode >= 0 ? CMode : 0)>), fix, dim3(block), 0, s, ntiles,
                           a.carries, a.c + col, a.c_stride, a.beta);
    }
    return check_launch();
}

// columns in passes of 8, 4, 2 and 1 (max_nr: test / tuning hook GKOMI_COO_MAX_NR)
template <int CMode, int Halo>
int tile_passes(hipStream_t s, int64_t nrhs, const coo_args& a)
{
    static const int cap = [] {
        const char* e = std::getenv("GKOMI_COO_MAX_NR");
        const int v = e != nullptr ? std::atoi(e) : max_nr;
        return v >= 1 && v <= max_nr ? v : max_nr;
    }();
    const bool vec2 = aligned_to(a.b, 16) && a.b_stride % 2 == 0;
    int64_t done = 0;
#define GKOMI_PASS(NRV)                                                                     \
    if (nrhs - done >= NRV && NRV <= cap && (NRV == 1 || vec2)) {                           \
        const int groups = static_cast<int>((nrhs - done) / NRV);                           \
        const int err = launch_tile<NRV, CMode, Halo>(s, groups, a, done);                  \
        if (err) return err;                                                                \
        done += static_cast<int64_t>(NRV) * groups;                                         \
    }
    // four columns of atomics per segment land in one 32-B sector and serialise
    // (97 us vs 2 x 32 us on P2, profiles/r02_coo.log): pairs at most there
    if (CMode >= 0) {
        GKOMI_PASS(8)
        GKOMI_PASS(4)
    }
    GKOMI_PASS(2)
    GKOMI_PASS(1)
#undef GKOMI_PASS
    return GKOMI_SUCCESS;
}

}  // namespace

// the any-order multi-column pass of formats.hip's gkomi_coo_spmv2_f64_i32
int coo_tile_atomic_launch(hipStream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz, const int32_t* rows,
                           const int32_t* cols, const double* vals, const double* b, int64_t b_stride,
                           double* c, int64_t c_stride, const double* alpha)
{
    if (!(aligned_to(vals, 16) && aligned_to(rows, 8) && aligned_to(cols, 8))) return GKOMI_ENOTSUPPORTED;
    const coo_args a{nrows, nnz, rows, cols, vals, b, b_stride, c, c_stride, alpha, nullptr, nullptr, nullptr,
                     ncols > 0 && offsets_fit_32(nrows, ncols, b_stride, c_stride)};
    return tile_passes<-1, 1>(s, nrhs, a);
}

}  // namespace gkomi

using namespace gkomi;

extern "C" size_t gkomi_coo_sorted_workspace_bytes(int64_t nnz, int64_t nrhs)
{
    if (nnz < 0 || nrhs < 0) return 0;
    // two carry slots per tile and group of columns, for the smallest tile and one column per group
    const int64_t ntiles = ceildiv(nnz, min_tile);
    const int64_t groups = nrhs > 1 ? nrhs : 1;
    return sizeof(sorted_header) + static_cast<size_t>(2 * ntiles * groups) * sizeof(carry_slot);
}

extern "C" int gkomi_coo_analyse_rows_i32(gkomi_stream_t s, int64_t nnz, const int32_t* row_idxs,
                                          void* workspace, size_t workspace_bytes, int* host_sorted,
                                          int64_t* host_max_row_nnz)
{
    if (nnz < 0 || host_sorted == nullptr) return GKOMI_EINVAL;
    if (workspace == nullptr || workspace_bytes < sizeof(sorted_header)) return GKOMI_EWORKSPACE;
    *host_sorted = 1;
    if (host_max_row_nnz != nullptr) *host_max_row_nnz = nnz > 0 ? 1 : 0;
    hipStream_t stream = to_stream(s);
    sorted_header* hdr = static_cast<sorted_header*>(workspace);
    hipLaunchKernelGGL(coo_clear_header_kernel, dim3(1), dim3(1), 0, stream, hdr);
    if (nnz >= 2) {
        hipLaunchKernelGGL(coo_analyse_rows_kernel, dim3(grid_for(nnz, block)), dim3(block), 0, stream, nnz,
                           row_idxs, hdr);
    }
    GKOMI_TRY(check_launch());
    if (nnz < 2) return GKOMI_SUCCESS;
    sorted_header h{};
    GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h, hdr, sizeof(h), hipMemcpyDeviceToHost, stream)));
    GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    *host_sorted = h.unsorted ? 0 : 1;
    if (host_max_row_nnz != nullptr) *host_max_row_nnz = h.max_row_nnz;
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_coo_sorted_check(gkomi_stream_t s, const void* workspace, int* host_flag)
{
    if (workspace == nullptr || host_flag == nullptr) return GKOMI_EINVAL;
    sorted_header h{};
    hipStream_t stream = to_stream(s);
    GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&h, workspace, sizeof(h), hipMemcpyDeviceToHost, stream)));
    GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    *host_flag = h.violation;
    return GKOMI_SUCCESS;
}

namespace {
template <int Halo>
int sorted_modes(hipStream_t stream, int cmode, int64_t nrhs, const coo_args& a)
{
    switch (cmode) {
    case 0: return tile_passes<0, Halo>(stream, nrhs, a);
    case 1: return tile_passes<1, Halo>(stream, nrhs, a);
    default: return tile_passes<2, Halo>(stream, nrhs, a);
    }
}

int sorted_apply(gkomi_stream_t s, int cmode, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
                 const int32_t* rows, const int32_t* cols, const double* vals, const double* b,
                 int64_t b_stride, double* c, int64_t c_stride, const double* alpha, const double* beta,
                 int64_t max_row_nnz_hint, void* workspace, size_t workspace_bytes)
{
    if (nrows < 0 || ncols < 0 || nrhs < 0 || nnz < 0) return GKOMI_EINVAL;
    if (nrhs > 65535 || nrows > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    if (nnz == 0) {  // every row is an empty row
        if (cmode == 0) return gkomi_dense_fill_f64(s, nrows, nrhs, c, c_stride, 0.0);
        if (cmode == 1) return gkomi_dense_scale_f64(s, nrows, nrhs, beta, 1, c, c_stride);
        return GKOMI_SUCCESS;
    }
    if (workspace == nullptr || workspace_bytes < gkomi_coo_sorted_workspace_bytes(nnz, nrhs)) {
        return GKOMI_EWORKSPACE;
    }
    if (!(aligned_to(vals, 16) && aligned_to(rows, 8) && aligned_to(cols, 8))) return GKOMI_ENOTSUPPORTED;
    sorted_header* hdr = static_cast<sorted_header*>(workspace);
    const coo_args a{nrows, nnz, rows, cols, vals, b, b_stride, c, c_stride, alpha, beta,
                     reinterpret_cast<carry_slot*>(hdr + 1), hdr, offsets_fit_32(nrows, ncols, b_stride, c_stride)};
    hipStream_t stream = to_stream(s);
    if (max_row_nnz_hint >= 1 && max_row_nnz_hint <= halo_rows) {
        return sorted_modes<halo_rows>(stream, cmode, nrhs, a);
    }
    return sorted_modes<1>(stream, cmode, nrhs, a);
}
}  // namespace

extern "C" int gkomi_coo_spmv_sorted_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
    const int32_t* row_idxs, const int32_t* col_idxs, const double* vals, const double* b,
    int64_t b_stride, double* c, int64_t c_stride, const double* alpha, const double* beta,
    int64_t max_row_nnz_hint, void* workspace, size_t workspace_bytes)
{
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    return sorted_apply(s, alpha == nullptr ? 0 : 1, nrows, ncols, nrhs, nnz, row_idxs, col_idxs, vals, b,
                        b_stride, c, c_stride, alpha, beta, max_row_nnz_hint, workspace, workspace_bytes);
}

extern "C" int gkomi_coo_spmv2_sorted_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
    const int32_t* row_idxs, const int32_t* col_idxs, const double* vals, const double* b,
    int64_t b_stride, double* c, int64_t c_stride, const double* alpha, int64_t max_row_nnz_hint,
    void* workspace, size_t workspace_bytes)
{
    return sorted_apply(s, 2, nrows, ncols, nrhs, nnz, row_idxs, col_idxs, vals, b, b_stride, c, c_stride,
                        alpha, nullptr, max_row_nnz_hint, workspace, workspace_bytes);
}
