// ParILU(0) and its setup kernels for gfx950.  Replaces
// gko::kernels::hip::factorization::{add_diagonal_elements,
// initialize_row_ptrs_l_u, initialize_l_u}
// (core/factorization/factorization_kernels.hpp),
// par_ilu_factorization::compute_l_u_factors
// (core/factorization/par_ilu_kernels.hpp:54) and csr::transpose; semantics =
// reference/factorization/factorization_kernels.cpp:84-250,
// reference/factorization/par_ilu_kernels.cpp:54-120,
// reference/matrix/csr_kernels.cpp:551-586.  Driver order:
// core/factorization/par_ilu.cpp:74-163.
//
// The setup kernels are integer/scatter work and bit-exact.  The sweep is the
// asynchronous fixed-point iteration of Chow & Patel: one thread per stored
// entry of A (COO), sparse dot of L(row,:) with U(:,col) by a two-pointer
// merge, every read may see an old or a new neighbour -- the result converges
// to the reference's single sequential sweep (= ILU(0)); parity is by
// tolerance, like test/factorization/par_ilu_kernels.cpp:277-309.
#include "common.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;

// missing[row] = 1 if row < ncols has no stored diagonal; missing[nrows] = 0
__global__ __launch_bounds__(block) void missing_diagonal_kernel(
    int64_t nrows, int64_t ncols, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, int32_t* __restrict__ missing)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row <= nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        int32_t m = 0;
        if (row < nrows && row < ncols) {
            m = 1;
            const int32_t end = row_ptrs[row + 1];
            for (int32_t k = row_ptrs[row]; k < end; ++k) {
                if (col_idxs[k] == row) {
                    m = 0;
                    break;
                }
            }
        }
        missing[row] = m;
    }
}

// added[row] = number of diagonals inserted before row (exclusive scan of missing)
__global__ __launch_bounds__(block) void add_diagonal_kernel(
    int64_t nrows, int64_t ncols, const int32_t* __restrict__ old_row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const int32_t* __restrict__ added, int32_t* __restrict__ new_cols,
    double* __restrict__ new_vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        const int32_t start = old_row_ptrs[row], end = old_row_ptrs[row + 1];
        const bool insert = added[row + 1] != added[row];
        int32_t shift = added[row];
        bool handled = !insert;
        // reference :118-150: the zero goes in front of the first col > row
        for (int32_t old = start; old < end; ++old) {
            const int32_t col = col_idxs[old];
            if (!handled && col > row) {
                new_vals[old + shift] = 0.0;
                new_cols[old + shift] = static_cast<int32_t>(row);
                ++shift;
                handled = true;
            }
            new_vals[old + shift] = vals[old];
            new_cols[old + shift] = col;
        }
        if (!handled) {
            new_vals[end + shift] = 0.0;
            new_cols[end + shift] = static_cast<int32_t>(row);
        }
    }
}

__global__ __launch_bounds__(block) void shift_row_ptrs_kernel(int64_t nrows,
                                                              int32_t* __restrict__ row_ptrs,
                                                              const int32_t* __restrict__ added)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row <= nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        row_ptrs[row] += added[row];
    }
}

__global__ __launch_bounds__(block) void count_l_u_kernel(int64_t n,
                                                         const int32_t* __restrict__ row_ptrs,
                                                         const int32_t* __restrict__ col_idxs,
                                                         int32_t* __restrict__ l_counts,
                                                         int32_t* __restrict__ u_counts)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row <= n;
         row += static_cast<int64_t>(gridDim.x) * block) {
        int32_t l = 0, u = 0;
        if (row < n) {
            const int32_t end = row_ptrs[row + 1];
            for (int32_t k = row_ptrs[row]; k < end; ++k) {
                l += col_idxs[k] < row;
                u += col_idxs[k] > row;
            }
            ++l;  // the diagonal is always stored
            ++u;
        }
        l_counts[row] = l;
        u_counts[row] = u;
    }
}

__global__ __launch_bounds__(block) void initialize_l_u_kernel(
    int64_t n, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, const int32_t* __restrict__ l_row_ptrs,
    int32_t* __restrict__ l_cols, double* __restrict__ l_vals,
    const int32_t* __restrict__ u_row_ptrs, int32_t* __restrict__ u_cols,
    double* __restrict__ u_vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < n;
         row += static_cast<int64_t>(gridDim.x) * block) {
        int64_t il = l_row_ptrs[row];
        int64_t iu = u_row_ptrs[row] + 1;
        double diag = 1.0;
        const int32_t end = row_ptrs[row + 1];
        for (int32_t k = row_ptrs[row]; k < end; ++k) {
            const int32_t col = col_idxs[k];
            const double v = vals[k];
            if (col < row) {
                l_cols[il] = col;
                l_vals[il] = v;
                ++il;
            } else if (col == row) {
                diag = v;
            } else {
                u_cols[iu] = col;
                u_vals[iu] = v;
                ++iu;
            }
        }
        l_cols[l_row_ptrs[row + 1] - 1] = static_cast<int32_t>(row);
        u_cols[u_row_ptrs[row]] = static_cast<int32_t>(row);
        l_vals[l_row_ptrs[row + 1] - 1] = 1.0;
        u_vals[u_row_ptrs[row]] = diag;
    }
}

__global__ __launch_bounds__(block) void par_ilu_sweep_kernel(
    int64_t nnz, const int32_t* __restrict__ coo_rows, const int32_t* __restrict__ coo_cols,
    const double* __restrict__ coo_vals, const int32_t* __restrict__ l_row_ptrs,
    const int32_t* __restrict__ l_cols, double* l_vals, const int32_t* __restrict__ ut_row_ptrs,
    const int32_t* __restrict__ ut_cols, double* ut_vals)
{
    for (int64_t el = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; el < nnz;
         el += static_cast<int64_t>(gridDim.x) * block) {
        const int32_t row = coo_rows[el], col = coo_cols[el];
        int32_t rl = l_row_ptrs[row], ru = ut_row_ptrs[col];
        const int32_t el_end = l_row_ptrs[row + 1], eu_end = ut_row_ptrs[col + 1];
        double sum = coo_vals[el], last = 0.0;
        while (rl < el_end && ru < eu_end) {
            const int32_t cl = l_cols[rl], cu = ut_cols[ru];
            if (cl == cu) {
                last = l_vals[rl] * ut_vals[ru];
                sum -= last;
            } else {
                last = 0.0;
            }
            if (cl <= cu) ++rl;
            if (cu <= cl) ++ru;
        }
        sum += last;  // undo the last operation
        if (row > col) {
            const double w = sum / ut_vals[eu_end - 1];
            if (isfinite(w)) l_vals[rl - 1] = w;
        } else {
            if (isfinite(sum)) ut_vals[ru - 1] = sum;
        }
    }
}

// ---- ParIC: L-only setup + fixed-point sweeps ------------
// factorization::initialize_row_ptrs_l / initialize_l
// (reference/factorization/factorization_kernels.cpp:251-318) and
// par_ic_factorization::{init_factor, compute_factor}
// (reference/factorization/par_ic_kernels.cpp:55-124).

__global__ __launch_bounds__(block) void count_l_kernel(int64_t n,
                                                       const int32_t* __restrict__ row_ptrs,
                                                       const int32_t* __restrict__ col_idxs,
                                                       int32_t* __restrict__ l_counts)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row <= n;
         row += static_cast<int64_t>(gridDim.x) * block) {
        int32_t l = 0;
        if (row < n) {
            const int32_t end = row_ptrs[row + 1];
            for (int32_t k = row_ptrs[row]; k < end; ++k) l += col_idxs[k] < row;
            ++l;  // the diagonal is always stored
        }
        l_counts[row] = l;
    }
}

__global__ __launch_bounds__(block) void initialize_l_kernel(
    int64_t n, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, const int32_t* __restrict__ l_row_ptrs,
    int32_t* __restrict__ l_cols, double* __restrict__ l_vals, bool diag_sqrt)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < n;
         row += static_cast<int64_t>(gridDim.x) * block) {
        int64_t il = l_row_ptrs[row];
        double diag = 1.0;
        const int32_t end = row_ptrs[row + 1];
        for (int32_t k = row_ptrs[row]; k < end; ++k) {
            const int32_t col = col_idxs[k];
            if (col < row) {
                l_cols[il] = col;
                l_vals[il] = vals[k];
                ++il;
            } else if (col == row) {
                diag = vals[k];
            }
        }
        if (diag_sqrt) {
            diag = sqrt(diag);
            if (!isfinite(diag)) diag = 1.0;
        }
        l_cols[l_row_ptrs[row + 1] - 1] = static_cast<int32_t>(row);
        l_vals[l_row_ptrs[row + 1] - 1] = diag;
    }
}

__global__ __launch_bounds__(block) void par_ic_init_factor_kernel(
    int64_t n, const int32_t* __restrict__ l_row_ptrs, const int32_t* __restrict__ l_cols,
    double* __restrict__ l_vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < n;
         row += static_cast<int64_t>(gridDim.x) * block) {
        for (int32_t nz = l_row_ptrs[row]; nz < l_row_ptrs[row + 1]; ++nz) {
            if (l_cols[nz] == row) {
                const double d = sqrt(l_vals[nz]);
                l_vals[nz] = isfinite(d) ? d : 1.0;
            }
        }
    }
}

// one thread per stored entry of L; asynchronous (entries read whatever the
// other threads have written so far), like the reference's GPU sweeps
__global__ __launch_bounds__(block) void par_ic_sweep_kernel(
    int64_t l_nnz, const int32_t* __restrict__ l_row_idxs, const double* __restrict__ a_vals,
    const int32_t* __restrict__ l_row_ptrs, const int32_t* __restrict__ l_cols, double* l_vals)
{
    for (int64_t nz = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; nz < l_nnz;
         nz += static_cast<int64_t>(gridDim.x) * block) {
        const int32_t row = l_row_idxs[nz], col = l_cols[nz];
        int32_t lb = l_row_ptrs[row], hb = l_row_ptrs[col];
        const int32_t le = l_row_ptrs[row + 1], he = l_row_ptrs[col + 1];
        double sum = 0.0;
        while (lb < le && hb < he) {
            const int32_t l_col = l_cols[lb], lh_row = l_cols[hb];
            if (l_col == lh_row && l_col < col) sum += l_vals[lb] * l_vals[hb];
            lb += (l_col <= lh_row);
            hb += (lh_row <= l_col);
        }
        double nv = a_vals[nz] - sum;
        if (row == col) {
            nv = sqrt(nv);
        } else {
            nv = nv / l_vals[he - 1];
        }
        if (isfinite(nv)) l_vals[nz] = nv;
    }
}

// ---- transpose ------------------------------------------------------------------

__global__ __launch_bounds__(block) void count_cols_kernel(int64_t nnz,
                                                          const int32_t* __restrict__ col_idxs,
                                                          int32_t* __restrict__ counts)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; i < nnz;
         i += static_cast<int64_t>(gridDim.x) * block) {
        atomicAdd(counts + col_idxs[i], 1);
    }
}

__global__ __launch_bounds__(block) void scatter_transpose_kernel(
    int64_t nrows, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, int32_t* __restrict__ cursor,
    int32_t* __restrict__ t_cols, double* __restrict__ t_vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        const int32_t end = row_ptrs[row + 1];
        for (int32_t k = row_ptrs[row]; k < end; ++k) {
            const int32_t dest = atomicAdd(cursor + col_idxs[k], 1);
            t_cols[dest] = static_cast<int32_t>(row);
            t_vals[dest] = vals[k];
        }
    }
}

// the atomic cursor fills each transposed row in arrival order; sorting every
// row by its (unique) column index restores the reference's order
__global__ __launch_bounds__(block) void sort_rows_kernel(int64_t nrows,
                                                         const int32_t* __restrict__ row_ptrs,
                                                         int32_t* __restrict__ cols,
                                                         double* __restrict__ vals)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        const int32_t begin = row_ptrs[row], end = row_ptrs[row + 1];
        for (int32_t i = begin + 1; i < end; ++i) {
            const int32_t c = cols[i];
            const double v = vals[i];
            int32_t j = i - 1;
            while (j >= begin && cols[j] > c) {
                cols[j + 1] = cols[j];
                vals[j + 1] = vals[j];
                --j;
            }
            cols[j + 1] = c;
            vals[j + 1] = v;
        }
    }
}

__global__ __launch_bounds__(block) void is_sorted_kernel(int64_t nrows,
                                                         const int32_t* __restrict__ row_ptrs,
                                                         const int32_t* __restrict__ cols,
                                                         int* __restrict__ unsorted)
{
    for (int64_t row = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x; row < nrows;
         row += static_cast<int64_t>(gridDim.x) * block) {
        for (int32_t k = row_ptrs[row] + 1; k < row_ptrs[row + 1]; ++k) {
            if (cols[k - 1] > cols[k]) {
                *unsorted = 1;  // benign race: everybody writes the same value
                break;
            }
        }
    }
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

// csr::sort_by_column_index / is_sorted_by_column_index
// (reference/matrix/csr_kernels.cpp:969-1009): one thread per row, stable
// insertion sort (rows on this path are short; the reference's std::sort leaves
// the order of duplicate columns unspecified)
extern "C" int gkomi_csr_sort_by_column_index_f64_i32(gkomi_stream_t s, int64_t nrows,
                                                      const int32_t* row_ptrs, int32_t* col_idxs,
                                                      double* vals)
{
    if (nrows < 0) return GKOMI_EINVAL;
    if (nrows == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(sort_rows_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), nrows, row_ptrs, col_idxs, vals);
    return check_launch();
}

extern "C" int gkomi_csr_is_sorted_by_column_index_i32(gkomi_stream_t s, int64_t nrows,
                                                       const int32_t* row_ptrs,
                                                       const int32_t* col_idxs, void* workspace,
                                                       size_t workspace_bytes, int* host_is_sorted)
{
    if (nrows < 0 || host_is_sorted == nullptr) return GKOMI_EINVAL;
    *host_is_sorted = 1;
    if (nrows == 0) return GKOMI_SUCCESS;
    if (workspace == nullptr || workspace_bytes < sizeof(int)) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    int* flag = static_cast<int*>(workspace);
    int err = static_cast<int>(hipMemsetAsync(flag, 0, sizeof(int), stream));
    if (err) return err;
    hipLaunchKernelGGL(is_sorted_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block), 0, stream,
                       nrows, row_ptrs, col_idxs, flag);
    err = check_launch();
    if (err) return err;
    int unsorted = 0;
    err = static_cast<int>(hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    *host_is_sorted = unsorted ? 0 : 1;
    return err;
}

extern "C" size_t gkomi_factorization_workspace_bytes(int64_t nrows)
{
    if (nrows < 0) return 0;
    return sizeof(int32_t) * static_cast<size_t>(nrows + 1) +
           gkomi_prefix_sum_workspace_bytes(nrows + 1) + 64;
}

// phase 1 of add_diagonal_elements: how many diagonal entries are missing
// (count_missing_elements, factorization_kernels.cpp:54-80).  Leaves the
// exclusive scan of the per-row flags in the workspace for phase 2.
extern "C" int gkomi_factorization_count_missing_diagonal_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, const int32_t* row_ptrs,
    const int32_t* col_idxs, void* workspace, size_t workspace_bytes, int64_t* host_missing)
{
    if (nrows < 0 || ncols < 0) return GKOMI_EINVAL;
    if (workspace == nullptr || workspace_bytes < gkomi_factorization_workspace_bytes(nrows)) {
        return GKOMI_EWORKSPACE;
    }
    hipStream_t stream = to_stream(s);
    int32_t* added = static_cast<int32_t*>(workspace);
    char* scan_ws = static_cast<char*>(workspace) + sizeof(int32_t) * (nrows + 1);
    scan_ws += (64 - reinterpret_cast<uintptr_t>(scan_ws) % 64) % 64;
    hipLaunchKernelGGL(missing_diagonal_kernel, dim3(grid_for(nrows + 1, block)), dim3(block), 0,
                       stream, nrows, ncols, row_ptrs, col_idxs, added);
    int err = check_launch();
    if (err) return err;
    err = gkomi_prefix_sum_i32(s, added, nrows + 1, scan_ws,
                               gkomi_prefix_sum_workspace_bytes(nrows + 1));
    if (err) return err;
    if (host_missing != nullptr) {
        int32_t total = 0;
        err = static_cast<int>(hipMemcpyAsync(&total, added + nrows, sizeof(int32_t),
                                              hipMemcpyDeviceToHost, stream));
        if (err) return err;
        err = static_cast<int>(hipStreamSynchronize(stream));
        *host_missing = total;
    }
    return err;
}

// phase 2: writes the widened (col_idxs, values) and shifts row_ptrs in place
extern "C" int gkomi_factorization_add_diagonal_elements_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, int32_t* new_col_idxs, double* new_vals, const void* workspace)
{
    if (nrows < 0 || ncols < 0 || workspace == nullptr) return GKOMI_EINVAL;
    if (nrows == 0) return GKOMI_SUCCESS;
    hipStream_t stream = to_stream(s);
    const int32_t* added = static_cast<const int32_t*>(workspace);
    hipLaunchKernelGGL(add_diagonal_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block), 0,
                       stream, nrows, ncols, row_ptrs, col_idxs, vals, added, new_col_idxs,
                       new_vals);
    hipLaunchKernelGGL(shift_row_ptrs_kernel, dim3(grid_for(nrows + 1, block)), dim3(block), 0,
                       stream, nrows, row_ptrs, added);
    return check_launch();
}

extern "C" int gkomi_factorization_initialize_row_ptrs_l_u_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs, const int32_t* col_idxs,
    int32_t* l_row_ptrs, int32_t* u_row_ptrs, void* workspace, size_t workspace_bytes)
{
    if (n < 0) return GKOMI_EINVAL;
    hipLaunchKernelGGL(count_l_u_kernel, dim3(grid_for(n + 1, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), n, row_ptrs, col_idxs, l_row_ptrs, u_row_ptrs);
    int err = check_launch();
    if (err) return err;
    err = gkomi_prefix_sum_i32(s, l_row_ptrs, n + 1, workspace, workspace_bytes);
    if (err) return err;
    return gkomi_prefix_sum_i32(s, u_row_ptrs, n + 1, workspace, workspace_bytes);
}

extern "C" int gkomi_factorization_initialize_l_u_f64_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, const int32_t* l_row_ptrs, int32_t* l_col_idxs, double* l_vals,
    const int32_t* u_row_ptrs, int32_t* u_col_idxs, double* u_vals)
{
    if (n < 0) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(initialize_l_u_kernel, dim3(grid_for(n, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), n, row_ptrs, col_idxs, vals, l_row_ptrs, l_col_idxs, l_vals,
                       u_row_ptrs, u_col_idxs, u_vals);
    return check_launch();
}

extern "C" int gkomi_par_ilu_compute_l_u_factors_f64_i32(
    gkomi_stream_t s, int64_t iterations, int64_t nnz, const int32_t* coo_row_idxs,
    const int32_t* coo_col_idxs, const double* coo_vals, const int32_t* l_row_ptrs,
    const int32_t* l_col_idxs, double* l_vals, const int32_t* ut_row_ptrs,
    const int32_t* ut_col_idxs, double* ut_vals)
{
    if (iterations < 0 || nnz < 0) return GKOMI_EINVAL;
    if (nnz == 0) return GKOMI_SUCCESS;
    // "Auto" (0): the reference's GPU backends run several asynchronous sweeps
    // (hip/factorization/par_ilu_kernels.hip.cpp:69 uses 10)
    if (iterations == 0) iterations = 10;
    for (int64_t it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(par_ilu_sweep_kernel, dim3(grid_for(nnz, block, 1 << 16)), dim3(block), 0,
                           to_stream(s), nnz, coo_row_idxs, coo_col_idxs, coo_vals, l_row_ptrs,
                           l_col_idxs, l_vals, ut_row_ptrs, ut_col_idxs, ut_vals);
    }
    return check_launch();
}

extern "C" int gkomi_factorization_initialize_row_ptrs_l_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs, const int32_t* col_idxs,
    int32_t* l_row_ptrs, void* workspace, size_t workspace_bytes)
{
    if (n < 0) return GKOMI_EINVAL;
    hipLaunchKernelGGL(count_l_kernel, dim3(grid_for(n + 1, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), n, row_ptrs, col_idxs, l_row_ptrs);
    int err = check_launch();
    if (err) return err;
    return gkomi_prefix_sum_i32(s, l_row_ptrs, n + 1, workspace, workspace_bytes);
}

extern "C" int gkomi_factorization_initialize_l_f64_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs, const int32_t* col_idxs,
    const double* vals, const int32_t* l_row_ptrs, int32_t* l_col_idxs, double* l_vals,
    int diag_sqrt)
{
    if (n < 0) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(initialize_l_kernel, dim3(grid_for(n, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), n, row_ptrs, col_idxs, vals, l_row_ptrs, l_col_idxs, l_vals,
                       diag_sqrt != 0);
    return check_launch();
}

extern "C" int gkomi_par_ic_init_factor_f64_i32(gkomi_stream_t s, int64_t n,
                                                const int32_t* l_row_ptrs,
                                                const int32_t* l_col_idxs, double* l_vals)
{
    if (n < 0) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(par_ic_init_factor_kernel, dim3(grid_for(n, block, 1 << 16)), dim3(block), 0,
                       to_stream(s), n, l_row_ptrs, l_col_idxs, l_vals);
    return check_launch();
}

extern "C" int gkomi_par_ic_compute_factor_f64_i32(gkomi_stream_t s, int64_t iterations,
                                                   int64_t l_nnz, const int32_t* l_row_idxs,
                                                   const double* a_lower_vals,
                                                   const int32_t* l_row_ptrs,
                                                   const int32_t* l_col_idxs, double* l_vals)
{
    if (iterations < 0 || l_nnz < 0) return GKOMI_EINVAL;
    if (l_nnz == 0) return GKOMI_SUCCESS;
    if (iterations == 0) iterations = 10;  // "Auto", as for ParILU
    for (int64_t it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(par_ic_sweep_kernel, dim3(grid_for(l_nnz, block, 1 << 16)), dim3(block), 0,
                           to_stream(s), l_nnz, l_row_idxs, a_lower_vals, l_row_ptrs, l_col_idxs,
                           l_vals);
    }
    return check_launch();
}

extern "C" size_t gkomi_csr_transpose_workspace_bytes(int64_t ncols)
{
    if (ncols < 0) return 0;
    return sizeof(int32_t) * static_cast<size_t>(ncols + 1) +
           gkomi_prefix_sum_workspace_bytes(ncols + 1) + 64;
}

extern "C" int gkomi_csr_transpose_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                                           int64_t nnz, const int32_t* row_ptrs,
                                           const int32_t* col_idxs, const double* vals,
                                           int32_t* t_row_ptrs, int32_t* t_col_idxs,
                                           double* t_vals, void* workspace,
                                           size_t workspace_bytes)
{
    if (nrows < 0 || ncols < 0 || nnz < 0) return GKOMI_EINVAL;
    if (workspace == nullptr || workspace_bytes < gkomi_csr_transpose_workspace_bytes(ncols)) {
        return GKOMI_EWORKSPACE;
    }
    hipStream_t stream = to_stream(s);
    int32_t* cursor = static_cast<int32_t*>(workspace);
    char* scan_ws = static_cast<char*>(workspace) + sizeof(int32_t) * (ncols + 1);
    scan_ws += (64 - reinterpret_cast<uintptr_t>(scan_ws) % 64) % 64;
    int err = static_cast<int>(
        hipMemsetAsync(t_row_ptrs, 0, sizeof(int32_t) * static_cast<size_t>(ncols + 1), stream));
    if (err) return err;
    if (nnz > 0) {
        hipLaunchKernelGGL(count_cols_kernel, dim3(grid_for(nnz, block)), dim3(block), 0, stream,
                           nnz, col_idxs, t_row_ptrs);
        err = check_launch();
        if (err) return err;
    }
    err = gkomi_prefix_sum_i32(s, t_row_ptrs, ncols + 1, scan_ws,
                               gkomi_prefix_sum_workspace_bytes(ncols + 1));
    if (err) return err;
    if (nnz == 0 || nrows == 0) return GKOMI_SUCCESS;
    err = static_cast<int>(hipMemcpyAsync(cursor, t_row_ptrs,
                                          sizeof(int32_t) * static_cast<size_t>(ncols + 1),
                                          hipMemcpyDeviceToDevice, stream));
    if (err) return err;
    hipLaunchKernelGGL(scatter_transpose_kernel, dim3(grid_for(nrows, block, 1 << 16)), dim3(block),
                       0, stream, nrows, row_ptrs, col_idxs, vals, cursor, t_col_idxs, t_vals);
    hipLaunchKernelGGL(sort_rows_kernel, dim3(grid_for(ncols, block, 1 << 16)), dim3(block), 0,
                       stream, ncols, t_row_ptrs, t_col_idxs, t_vals);
    return check_launch();
}
