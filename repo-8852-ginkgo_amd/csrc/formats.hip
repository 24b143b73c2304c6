// ELL / SELL-P / COO / Hybrid SpMV for gfx950.  Replaces
// gko::kernels::hip::{ell,sellp,coo}::{spmv, advanced_spmv, spmv2,
// advanced_spmv2} (core/matrix/{ell,sellp,coo}_kernels.hpp) and the Hybrid
// composition of core/matrix/hybrid.cpp:133-159.
//
// ELL / SELL-P: column-major storage makes the per-lane loads coalesced by
// construction (8 B values + 4 B columns per lane); one thread owns one row
// and accumulates in a register in storage order -> bit-identical to
// reference/matrix/ell_kernels.cpp:57-157, sellp_kernels.cpp:57-131.
// Algorithmic bytes: ELL 12*stride*K + 8*ncols + 8*nrows; SELL-P
// 12*slice_size*total_cols + 16*num_slices + 8*ncols + 8*nrows.
//
// COO (any order of the entries): 16 B/nonzero streamed with 8-/16-B per-lane
// loads into an LDS tile of products + row ids; one thread per row segment adds
// it left to right and issues ONE fp64 atomic per segment.  For row-sorted
// input every row that lies inside one tile is summed exactly in the
// reference's order (coo_kernels.cpp:92-131); rows cut by a tile boundary get
// two atomics (commutative: still deterministic); only rows spread over three
// or more tiles depend on arrival order.  The tile kernel lives in
// coo_spmv.hip (with the atomic-free entries for sorted matrices); the kernel
// in this file is the round-1 version, kept for arrays that are not 16-/8-B
// aligned.
#include "common.hpp"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>

namespace gkomi {
// coo_spmv.hip: c += alpha A b with the tile read once per group of columns
int coo_tile_atomic_launch(hipStream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz, const int32_t* rows,
                           const int32_t* cols, const double* vals, const double* b, int64_t b_stride,
                           double* c, int64_t c_stride, const double* alpha);
namespace {

constexpr int block = 256;
constexpr int wave_size = 64;

// Dot = true (single column, plain apply) adds the epilogue of the fused
// Krylov drivers: dot_partial[block] = sum over the block's rows of
// w(row) * c(row) (w = b unless dot_w is given), on request also of c(row)^2,
// and nothing at all once *stop_status says the solve has stopped.  Same
// row -> thread map and block sum as the CSR kernel's epilogue
// (csr_spmv.hip), so the partials -- and with them the iterates of a fused
// solve -- are bit-identical across CSR / ELL / SELL-P.
template <bool Advanced, bool Dot = false>
__global__ __launch_bounds__(block) void ell_spmv_kernel(
    int64_t nrows, int64_t num_stored, int64_t stride,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p,
    const double* __restrict__ beta_p, double* __restrict__ dot_partial = nullptr,
    const uint8_t* __restrict__ stop_status = nullptr,
    const double* __restrict__ dot_w = nullptr, double* __restrict__ dot_partial2 = nullptr)
{
    if (Dot && status_has_stopped_uniform(stop_status)) return;
    double pq = 0.0, qq = 0.0;
    const double* w = Dot && dot_w != nullptr ? dot_w : b;
    b += blockIdx.y;
    c += blockIdx.y;
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         row < nrows; row += step) {
        double result = Advanced ? c[row * c_stride] * beta : 0.0;
        // 8 columns at a time: all 16 loads, then all 8 gathers in flight, then the in-order accumulation.  A batch
        // that reaches behind the last column repeats it (clamped index: the same lines again, nothing added) --
        // a remainder loop of one column per two dependent round trips made the 7-column matrix of the 108^3
        // system 20 % slower than its CSR form (25.0 vs 20.5 us)
        for (int64_t i = 0; i < num_stored; i += 8) {
            double v[8];
            int32_t col[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t at = row + min(i + u, num_stored - 1) * stride;
                v[u] = vals[at];
                col[u] = col_idxs[at];
            }
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = b[max(col[u], 0) * b_stride];  // padding (-1) reads b[0], unused
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i + u < num_stored && col[u] != -1) {
                    result += Advanced ? (alpha * v[u]) * x[u] : v[u] * x[u];
                }
            }
        }
        c[row * c_stride] = result;
        if (Dot) {
            pq += w[row] * result;
            qq += result * result;
        }
    }
    if (Dot) {
        __shared__ double red[block / wave_size];
        const double total = block_reduce_sum<block>(pq, red);
        if (threadIdx.x == 0) dot_partial[blockIdx.x] = total;
        if (dot_partial2 != nullptr) {
            __syncthreads();
            const double total2 = block_reduce_sum<block>(qq, red);
            if (threadIdx.x == 0) dot_partial2[blockIdx.x] = total2;
        }
    }
}

template <bool Advanced, bool Dot = false>
__global__ __launch_bounds__(block) void sellp_spmv_kernel(
    int64_t nrows, int64_t slice_size, const uint64_t* __restrict__ slice_sets,
    const uint64_t* __restrict__ slice_lengths,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p,
    const double* __restrict__ beta_p, double* __restrict__ dot_partial = nullptr,
    const uint8_t* __restrict__ stop_status = nullptr,
    const double* __restrict__ dot_w = nullptr, double* __restrict__ dot_partial2 = nullptr)
{
    if (Dot && status_has_stopped_uniform(stop_status)) return;
    double pq = 0.0, qq = 0.0;
    const double* w = Dot && dot_w != nullptr ? dot_w : b;
    b += blockIdx.y;
    c += blockIdx.y;
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         row < nrows; row += step) {
        const int64_t slice = row / slice_size;
        const int64_t local = row % slice_size;
        const int64_t len = static_cast<int64_t>(slice_lengths[slice]);
        const int64_t base =
            static_cast<int64_t>(slice_sets[slice]) * slice_size + local;
        double result = Advanced ? c[row * c_stride] * beta : 0.0;
        for (int64_t i = 0; i < len; i += 8) {  // (as the ELL kernel: clamped batches of 8)
            double v[8];
            int32_t col[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t at = base + min(i + u, len - 1) * slice_size;
                v[u] = vals[at];
                col[u] = col_idxs[at];
            }
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = b[max(col[u], 0) * b_stride];  // padding (-1) reads b[0], unused
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i + u < len && col[u] != -1) {
                    result += Advanced ? (alpha * v[u]) * x[u] : v[u] * x[u];
                }
            }
        }
        c[row * c_stride] = result;
        if (Dot) {
            pq += w[row] * result;
            qq += result * result;
        }
    }
    if (Dot) {
        __shared__ double red[block / wave_size];
        const double total = block_reduce_sum<block>(pq, red);
        if (threadIdx.x == 0) dot_partial[blockIdx.x] = total;
        if (dot_partial2 != nullptr) {
            __syncthreads();
            const double total2 = block_reduce_sum<block>(qq, red);
            if (threadIdx.x == 0) dot_partial2[blockIdx.x] = total2;
        }
    }
}

// ---- ELL / SELL-P, several right-hand sides -------------------------------------
// One thread per row with NR register accumulators: the matrix is read once per
// NR columns of b (gridDim.y = nrhs re-reads it per column, like the
// reference's kernels); b(col, j..j+NR-1) is contiguous in the row-major b.
// Same per-(row, column) order as the single-column kernels -> bit-identical.
template <int NR, bool Advanced, bool Vec>
__device__ __forceinline__ void ell_like_row(const int32_t* __restrict__ col_idxs,
                                             const double* __restrict__ vals, int64_t base,
                                             int64_t step_elems, int64_t len,
                                             const double* __restrict__ b, int64_t b_stride,
                                             double* __restrict__ c_row, double alpha, double beta)
{
    constexpr int unroll = NR >= 8 ? 2 : 4;
    double acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = Advanced ? c_row[j] * beta : 0.0;
    for (int64_t i = 0; i < len; i += unroll) {
        double v[unroll];
        int32_t col[unroll];
#pragma unroll
        for (int u = 0; u < unroll; ++u) {
            const int64_t at = base + min(i + u, len - 1) * step_elems;
            v[u] = vals[at];
            col[u] = col_idxs[at];
        }
        double x[unroll][NR];
#pragma unroll
        for (int u = 0; u < unroll; ++u) {
            const double* src = b + max(col[u], 0) * b_stride;  // padding (-1) reads row 0, unused
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
            if (i + u < len && col[u] != -1) {
                const double av = Advanced ? alpha * v[u] : v[u];
#pragma unroll
                for (int j = 0; j < NR; ++j) acc[j] += av * x[u][j];
            }
        }
    }
    if (Vec) {
#pragma unroll
        for (int j = 0; j < NR; j += 2) {
            *reinterpret_cast<double2*>(c_row + j) = make_double2(acc[j], acc[j + 1]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; ++j) c_row[j] = acc[j];
    }
}

// Sellp = false: ELL (slice_size = stride, slice arrays unused)
template <int NR, bool Advanced, bool Vec, bool Sellp>
__global__ __launch_bounds__(block) void ell_like_multi_kernel(
    int64_t nrows, int64_t num_stored, int64_t slice_size,
    const uint64_t* __restrict__ slice_sets, const uint64_t* __restrict__ slice_lengths,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c, int64_t c_stride,
    const double* __restrict__ alpha_p, const double* __restrict__ beta_p)
{
    b += blockIdx.y * NR;
    c += blockIdx.y * NR;
    double alpha = 1.0, beta = 0.0;
    if (Advanced) {
        alpha = alpha_p[0];
        beta = beta_p[0];
    }
    const int64_t step = static_cast<int64_t>(gridDim.x) * block;
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += step) {
        int64_t base = row, len = num_stored;
        if (Sellp) {
            const int64_t slice = row / slice_size;
            len = static_cast<int64_t>(slice_lengths[slice]);
            base = static_cast<int64_t>(slice_sets[slice]) * slice_size + row % slice_size;
        }
        ell_like_row<NR, Advanced, Vec>(col_idxs, vals, base, slice_size, len, b, b_stride,
                                        c + row * c_stride, alpha, beta);
    }
}

template <int NR, bool Sellp>
int launch_ell_like_multi(hipStream_t s, int64_t nrows, int slices, int64_t num_stored,
                          int64_t slice_size, const uint64_t* slice_sets,
                          const uint64_t* slice_lengths, const int32_t* col_idxs,
                          const double* vals, const double* b, int64_t b_stride, double* c,
                          int64_t c_stride, const double* alpha, const double* beta)
{
    dim3 grid(grid_for(nrows, block, 1 << 20), static_cast<unsigned>(slices));
    const bool vec = reinterpret_cast<uintptr_t>(b) % 16 == 0 && reinterpret_cast<uintptr_t>(c) % 16 == 0 &&
                     b_stride % 2 == 0 && c_stride % 2 == 0;
#define GKOMI_ELLM(ADV, VEC)                                                                      \
    hipLaunchKernelGGL((ell_like_multi_kernel<NR, ADV, VEC, Sellp>), grid, dim3(block), 0, s,     \
                       nrows, num_stored, slice_size, slice_sets, slice_lengths, col_idxs, vals,  \
                       b, b_stride, c, c_stride, alpha, beta)
    if (alpha != nullptr) {
        if (vec) GKOMI_ELLM(true, true); else GKOMI_ELLM(true, false);
    } else {
        if (vec) GKOMI_ELLM(false, true); else GKOMI_ELLM(false, false);
    }
#undef GKOMI_ELLM
    return check_launch();
}

// columns [0, done) in passes of 8, 4 and 2; returns how many were handled (the
// caller finishes an odd last column with the single-column kernel)
template <bool Sellp>
int ell_like_multi(hipStream_t s, int64_t nrows, int64_t nrhs, int64_t num_stored,
                   int64_t slice_size, const uint64_t* slice_sets, const uint64_t* slice_lengths,
                   const int32_t* col_idxs, const double* vals, const double* b, int64_t b_stride,
                   double* c, int64_t c_stride, const double* alpha, const double* beta,
                   int64_t* handled)
{
    int64_t done = 0;
#define GKOMI_PASS(NRV)                                                                            \
    if (nrhs - done >= NRV) {                                                                      \
        const int slices = static_cast<int>((nrhs - done) / NRV);                                  \
        const int err = launch_ell_like_multi<NRV, Sellp>(s, nrows, slices, num_stored, slice_size, \
                                                          slice_sets, slice_lengths, col_idxs,     \
                                                          vals, b + done, b_stride, c + done,      \
                                                          c_stride, alpha, beta);                  \
        if (err) return err;                                                                       \
        done += static_cast<int64_t>(NRV) * slices;                                                \
    }
    GKOMI_PASS(8)
    GKOMI_PASS(4)
    GKOMI_PASS(2)
#undef GKOMI_PASS
    *handled = done;
    return 0;
}

constexpr int coo_items = 6;
constexpr int coo_tile = block * coo_items;  // 1536 nonzeros per workgroup

template <bool Scaled, bool Vec>
__global__ __launch_bounds__(block) void coo_spmv2_kernel(
    int64_t nnz, const int32_t* __restrict__ row_idxs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ b, int64_t b_stride, double* __restrict__ c,
    int64_t c_stride, const double* __restrict__ alpha_p)
{
    __shared__ __attribute__((aligned(16))) double prod[coo_tile];
    __shared__ __attribute__((aligned(8))) int32_t rowid[coo_tile];
    __shared__ int wave_heads[block / wave_size];
    b += blockIdx.y;
    c += blockIdx.y;
    const double alpha = Scaled ? alpha_p[0] : 1.0;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * coo_tile;
    const int count = static_cast<int>(min(static_cast<int64_t>(coo_tile), nnz - base));
    const int tid = threadIdx.x;
    if (Vec) {
        constexpr int pairs = coo_items / 2;
        double2 v[pairs];
        int2 r[pairs], cc[pairs];
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            const int e = 2 * (tid + u * block);
            v[u] = make_double2(0.0, 0.0);
            r[u] = make_int2(0, 0);
            cc[u] = make_int2(0, 0);
            if (e + 1 < count) {
                v[u] = *reinterpret_cast<const double2*>(vals + base + e);
                r[u] = *reinterpret_cast<const int2*>(row_idxs + base + e);
                cc[u] = *reinterpret_cast<const int2*>(col_idxs + base + e);
            } else if (e < count) {
                v[u].x = vals[base + e];
                r[u].x = row_idxs[base + e];
                cc[u].x = col_idxs[base + e];
            }
        }
        double2 x[pairs];
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            x[u].x = b[cc[u].x * b_stride];
            x[u].y = b[cc[u].y * b_stride];
        }
#pragma unroll
        for (int u = 0; u < pairs; ++u) {
            const int e = 2 * (tid + u * block);
            double2 pr;
            pr.x = Scaled ? (alpha * v[u].x) * x[u].x : v[u].x * x[u].x;
            pr.y = Scaled ? (alpha * v[u].y) * x[u].y : v[u].y * x[u].y;
            *reinterpret_cast<double2*>(prod + e) = pr;
            *reinterpret_cast<int2*>(rowid + e) = r[u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < coo_items; ++u) {
            const int e = tid + u * block;
            if (e < count) {
                const double val = vals[base + e];
                const double x = b[col_idxs[base + e] * b_stride];
                prod[e] = Scaled ? (alpha * val) * x : val * x;
                rowid[e] = row_idxs[base + e];
            }
        }
    }
    __syncthreads();
    // segment heads inside this thread's run of coo_items elements: the head
    // adds its segment left to right (the reference's order inside a row)
    const int first = tid * coo_items;
    double sums[coo_items];
    int rows[coo_items];
    int nheads = 0;
#pragma unroll
    for (int u = 0; u < coo_items; ++u) {
        const int e = first + u;
        rows[u] = -1;
        sums[u] = 0.0;
        if (e < count) {
            const int row = rowid[e];
            if (e == 0 || rowid[e - 1] != row) {
                double sum = prod[e];
                int k = e + 1;
                while (k < count && rowid[k] == row) {
                    sum += prod[k];
                    ++k;
                }
                rows[u] = row;
                sums[u] = sum;
                ++nheads;
            }
        }
    }
    // compact (row, sum) into LDS so that consecutive lanes issue the atomics
    // of consecutive segments: one coalesced atomic instruction per 64
    // segments instead of scattered ones (43.5 -> 21.4 us on P2,
    // tools/coo_experiment.hip)
    const int lane = tid & (wave_size - 1);
    const int wave = tid / wave_size;
    int incl = nheads;
#pragma unroll
    for (int d = 1; d < wave_size; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == wave_size - 1) wave_heads[wave] = incl;
    __syncthreads();  // also: every thread is done reading prod / rowid
    int offset = incl - nheads;
    int total = 0;
#pragma unroll
    for (int w = 0; w < block / wave_size; ++w) {
        const int wc = wave_heads[w];
        if (w < wave) offset += wc;
        total += wc;
    }
#pragma unroll
    for (int u = 0; u < coo_items; ++u) {
        if (rows[u] >= 0) {
            prod[offset] = sums[u];
            rowid[offset] = rows[u];
            ++offset;
        }
    }
    __syncthreads();
    for (int i = tid; i < total; i += block) {
        unsafeAtomicAdd(c + rowid[i] * c_stride, prod[i]);
    }
}

inline bool aligned_to(const void* p, size_t a)
{
    return reinterpret_cast<uintptr_t>(p) % a == 0;
}

int coo_launch(hipStream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz, const int32_t* rows,
               const int32_t* cols, const double* vals, const double* b,
               int64_t b_stride, double* c, int64_t c_stride,
               const double* alpha)
{
    if (nnz == 0 || nrhs == 0) return GKOMI_SUCCESS;
    const int64_t nblocks = ceildiv(nnz, coo_tile);
    if (nblocks > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    static const bool old1 = std::getenv("GKOMI_COO_OLD1") != nullptr;  // tuning hook: the round-1 kernel
    if (nrhs >= 2 || !old1) {
        // the tile kernel of coo_spmv.hip (21.3 vs 25.1 us on P2 for one column; the
        // tile once per 2 columns for more); it wants aligned arrays, the kernel
        // above takes the rest
        const int err = coo_tile_atomic_launch(s, nrows, ncols, nrhs, nnz, rows, cols, vals, b, b_stride, c,
                                               c_stride, alpha);
        if (err != GKOMI_ENOTSUPPORTED) return err;
    }
    dim3 grid(static_cast<unsigned>(nblocks), static_cast<unsigned>(nrhs));
    const bool vec = aligned_to(vals, 16) && aligned_to(rows, 8) && aligned_to(cols, 8);
#define GKOMI_COO(SC, VE)                                                     \
    hipLaunchKernelGGL((coo_spmv2_kernel<SC, VE>), grid, dim3(block), 0, s,   \
                       nnz, rows, cols, vals, b, b_stride, c, c_stride, alpha)
    if (alpha != nullptr) {
        if (vec) GKOMI_COO(true, true); else GKOMI_COO(true, false);
    } else {
        if (vec) GKOMI_COO(false, true); else GKOMI_COO(false, false);
    }
#undef GKOMI_COO
    return check_launch();
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int gkomi_ell_spmv_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs,
    int64_t num_stored_per_row, int64_t stride, const int32_t* col_idxs,
    const double* vals, const double* b, int64_t b_stride, double* c,
    int64_t c_stride, const double* alpha, const double* beta)
{
    if (nrows < 0 || ncols < 0 || nrhs < 0 || num_stored_per_row < 0) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (stride < nrows || b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    if (nrhs > 65535) return GKOMI_ENOTSUPPORTED;
    if (nrhs >= 2) {
        int64_t done = 0;
        const int err = ell_like_multi<false>(to_stream(s), nrows, nrhs, num_stored_per_row, stride,
                                              nullptr, nullptr, col_idxs, vals, b, b_stride, c, c_stride,
                                              alpha, beta, &done);
        if (err) return err;
        if (done == nrhs) return GKOMI_SUCCESS;
        b += done;
        c += done;
        nrhs -= done;
    }
    dim3 grid(grid_for(nrows, block, 1 << 20), static_cast<unsigned>(nrhs));
    if (alpha != nullptr) {
        hipLaunchKernelGGL(ell_spmv_kernel<true>, grid, dim3(block), 0, to_stream(s), nrows,
                           num_stored_per_row, stride, col_idxs, vals, b, b_stride, c, c_stride,
                           alpha, beta);
    } else {
        hipLaunchKernelGGL(ell_spmv_kernel<false>, grid, dim3(block), 0, to_stream(s), nrows,
                           num_stored_per_row, stride, col_idxs, vals, b, b_stride, c, c_stride,
                           alpha, beta);
    }
    return check_launch();
}

extern "C" int gkomi_sellp_spmv_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs,
    int64_t slice_size, const uint64_t* slice_sets,
    const uint64_t* slice_lengths, const int32_t* col_idxs, const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha, const double* beta)
{
    if (nrows < 0 || ncols < 0 || nrhs < 0 || slice_size <= 0) return GKOMI_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    if (nrhs > 65535) return GKOMI_ENOTSUPPORTED;
    if (nrhs >= 2) {
        int64_t done = 0;
        const int err = ell_like_multi<true>(to_stream(s), nrows, nrhs, 0, slice_size, slice_sets,
                                             slice_lengths, col_idxs, vals, b, b_stride, c, c_stride,
                                             alpha, beta, &done);
        if (err) return err;
        if (done == nrhs) return GKOMI_SUCCESS;
        b += done;
        c += done;
        nrhs -= done;
    }
    dim3 grid(grid_for(nrows, block, 1 << 20), static_cast<unsigned>(nrhs));
    if (alpha != nullptr) {
        hipLaunchKernelGGL(sellp_spmv_kernel<true>, grid, dim3(block), 0, to_stream(s), nrows,
                           slice_size, slice_sets, slice_lengths, col_idxs, vals, b, b_stride, c,
                           c_stride, alpha, beta);
    } else {
        hipLaunchKernelGGL(sellp_spmv_kernel<false>, grid, dim3(block), 0, to_stream(s), nrows,
                           slice_size, slice_sets, slice_lengths, col_idxs, vals, b, b_stride, c,
                           c_stride, alpha, beta);
    }
    return check_launch();
}

extern "C" int gkomi_coo_spmv2_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
    const int32_t* row_idxs, const int32_t* col_idxs, const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha)
{
    if (nrows < 0 || ncols < 0 || nrhs < 0 || nnz < 0) return GKOMI_EINVAL;
    if (nrhs > 65535) return GKOMI_ENOTSUPPORTED;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    if (b_stride < nrhs || c_stride < nrhs) return GKOMI_EINVAL;
    return coo_launch(to_stream(s), nrows, ncols, nrhs, nnz, row_idxs, col_idxs, vals, b, b_stride, c,
                      c_stride, alpha);
}

extern "C" int gkomi_coo_spmv_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
    const int32_t* row_idxs, const int32_t* col_idxs, const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha, const double* beta)
{
    if ((alpha == nullptr) != (beta == nullptr)) return GKOMI_EINVAL;
    if (nrows < 0 || nrhs < 0) return GKOMI_EINVAL;
    if (nrows == 0 || nrhs == 0) return GKOMI_SUCCESS;
    int err;
    if (alpha == nullptr) {
        // spmv = dense::fill(c, 0) + spmv2 (reference/matrix/coo_kernels.cpp:63-71)
        err = gkomi_dense_fill_f64(s, nrows, nrhs, c, c_stride, 0.0);
    } else {
        // advanced_spmv = dense::scale(beta, c) + advanced_spmv2 (:77-88)
        err = gkomi_dense_scale_f64(s, nrows, nrhs, beta, 1, c, c_stride);
    }
    if (err) return err;
    return gkomi_coo_spmv2_f64_i32(s, nrows, ncols, nrhs, nnz, row_idxs, col_idxs, vals, b,
                                   b_stride, c, c_stride, alpha);
}

extern "C" int gkomi_hybrid_spmv_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs,
    int64_t ell_num_stored_per_row, int64_t ell_stride,
    const int32_t* ell_col_idxs, const double* ell_vals, int64_t coo_nnz,
    const int32_t* coo_row_idxs, const int32_t* coo_col_idxs,
    const double* coo_vals, const double* b, int64_t b_stride, double* c,
    int64_t c_stride, const double* alpha, const double* beta)
{
    // ell->apply(b, x); coo->apply2(b, x)  (core/matrix/hybrid.cpp:133-159)
    int err = gkomi_ell_spmv_f64_i32(s, nrows, ncols, nrhs, ell_num_stored_per_row, ell_stride,
                                     ell_col_idxs, ell_vals, b, b_stride, c, c_stride, alpha, beta);
    if (err) return err;
    return gkomi_coo_spmv2_f64_i32(s, nrows, ncols, nrhs, coo_nnz, coo_row_idxs, coo_col_idxs,
                                   coo_vals, b, b_stride, c, c_stride, alpha);
}

namespace gkomi {

// The SpMV + dot epilogue for a system matrix behind the library's own ELL /
// SELL-P callbacks (the fused drivers of cg_solver.hip / krylov.hip): the
// number of partials one launch writes, or 0 when `op` is some other operator
// and the driver has to follow A.apply with a separate partials kernel.
int op_spmv_dot_num_partials(gkomi_matrix_apply_fn op, const void* ctx)
{
    int64_t nrows = 0;
    if (op == &gkomi_ell_matrix_apply_cb) {
        const auto* m = static_cast<const gkomi_ell_ctx*>(ctx);
        if (m->nrows != m->ncols || m->stride < m->nrows) return 0;
        nrows = m->nrows;
    } else if (op == &gkomi_sellp_matrix_apply_cb) {
        const auto* m = static_cast<const gkomi_sellp_ctx*>(ctx);
        if (m->nrows != m->ncols || m->slice_size <= 0) return 0;
        nrows = m->nrows;
    } else {
        return 0;
    }
    if (nrows <= 0 || ceildiv(nrows, block) > (int64_t{1} << 20)) return 0;
    return static_cast<int>(ceildiv(nrows, block));
}

int op_spmv_dot_launch(hipStream_t stream, gkomi_matrix_apply_fn op, const void* ctx,
                       const double* in, double* out, double* partial,
                       const uint8_t* stop_status, const double* dot_w, double* partial2)
{
    const int g = op_spmv_dot_num_partials(op, ctx);
    if (g <= 0) return GKOMI_ENOTSUPPORTED;
    if (op == &gkomi_ell_matrix_apply_cb) {
        const auto* m = static_cast<const gkomi_ell_ctx*>(ctx);
        hipLaunchKernelGGL((ell_spmv_kernel<false, true>), dim3(g), dim3(block), 0, stream, m->nrows,
                           m->num_stored_per_row, m->stride, m->col_idxs, m->vals, in, int64_t{1}, out,
                           int64_t{1}, nullptr, nullptr, partial, stop_status, dot_w, partial2);
    } else {
        const auto* m = static_cast<const gkomi_sellp_ctx*>(ctx);
        hipLaunchKernelGGL((sellp_spmv_kernel<false, true>), dim3(g), dim3(block), 0, stream, m->nrows,
                           m->slice_size, m->slice_sets, m->slice_lengths, m->col_idxs, m->vals, in,
                           int64_t{1}, out, int64_t{1}, nullptr, nullptr, partial, stop_status, dot_w,
                           partial2);
    }
    return check_launch();
}

}  // namespace gkomi
