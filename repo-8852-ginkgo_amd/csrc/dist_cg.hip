// Row-partitioned distributed apply and CG, native: what
// experimental::distributed::Matrix::apply_impl (core/distributed/matrix.cpp:263-335)
// and Cg::apply_dense_impl on distributed vectors (core/solver/cg.cpp:107-193 with
// the reductions of core/distributed/vector.cpp:317-409) do, for one right-hand
// side, over a `gkomi_comm` (RCCL over xGMI, csrc/comm.hip).
//
// apply:  pack the rows the neighbours need (row_gather) -> halo exchange on a
//         side stream  ||  local SpMV  -> non-local part, visiting ONLY the rows
//         that have off-rank entries (the reference applies the whole non-local
//         CSR with alpha = beta = 1, which rewrites every local row; rows
//         without off-rank entries come out unchanged, so skipping them is
//         bit-identical).
// CG:     the fused one-GPU iteration of cg_solver.hip (K1 criterion + p update,
//         K2 SpMV + p.q partials, K3 x, r update + r.r partials) with the two
//         scalars it needs made global: the workgroup partials are summed into a
//         small device buffer and all-reduced in place -- rho = r.z and tau^2 =
//         r.r travel together in ONE two-element all-reduce, beta = p.q in a
//         second one; K1 / K3 then read the reduced value where they re-added
//         partials.  All scalars stay on the device, every rank sees the same
//         bits, the criterion is evaluated on the device each iteration and the
//         host looks every `check_every` iterations.
#include "cg_fused.hpp"

#include "internal.hpp"
#include "sort_scan.hpp"

namespace gkomi {
namespace {

#define GKOMI_TRY(expr)          \
    do {                         \
        int err_ = (expr);       \
        if (err_) return err_;   \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// out[0] = sum a[0..na) (+ sum b[0..nb)), fixed order, one workgroup; out[1]
// likewise from a2 when given.  The operand of the all-reduce.
__global__ __launch_bounds__(fblock) void dist_sum_partials_kernel(
    const double* __restrict__ a, int na, const double* __restrict__ b, int nb,
    const double* __restrict__ a2, int na2, double* __restrict__ out, const cg_scalars* scal)
{
    __shared__ double smem[fblock / wave_size];
    if (scal != nullptr && status_has_stopped(scal->status)) return;
    double t = sum_partials(a, na, smem);
    if (b != nullptr) t += sum_partials(b, nb, smem);
    if (threadIdx.x == 0) out[0] = t;
    if (a2 != nullptr) {
        const double t2 = sum_partials(a2, na2, smem);
        if (threadIdx.x == 0) out[1] = t2;
    }
}

// Non-local part of the apply on the rows that have off-rank entries:
// x[row] = 1 * x[row] + sum (1 * val) * halo[col], in storage order
// (reference/matrix/csr_kernels.cpp:102-128 with alpha = beta = 1).  Dot: also the
// change of w . x it causes, one partial per workgroup.
template <bool Dot>
__global__ __launch_bounds__(256) void dist_nonlocal_rows_kernel(
    int nl_rows, const int32_t* __restrict__ row_idxs, const int32_t* __restrict__ row_ptrs,
    const int32_t* __restrict__ col_idxs, const double* __restrict__ vals,
    const double* __restrict__ halo, double* __restrict__ x, const double* __restrict__ w,
    double* __restrict__ partial, const cg_scalars* scal)
{
    __shared__ double red[256 / wave_size];
    if (Dot && scal != nullptr && status_has_stopped(scal->status)) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    double delta = 0.0;
    if (k < nl_rows) {
        const int row = row_idxs[k];
        const double old = x[row];
        double sum = old * 1.0;
        for (int e = row_ptrs[k]; e < row_ptrs[k + 1]; ++e) {
            sum += (1.0 * vals[e]) * halo[col_idxs[e]];
        }
        x[row] = sum;
        if (Dot) delta = w[row] * sum - w[row] * old;
    }
    if (Dot) {
        const double total = block_reduce_sum<256>(delta, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = total;
    }
}

// flag[i] = row i of the non-local block has entries
__global__ __launch_bounds__(256) void dist_flag_rows_kernel(int n, const int32_t* __restrict__ row_ptrs,
                                                            int32_t* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = row_ptrs[i + 1] > row_ptrs[i] ? 1 : 0;
    if (i == n) flag[i] = 0;
}

__global__ __launch_bounds__(256) void dist_compact_rows_kernel(int n, const int32_t* __restrict__ row_ptrs,
                                                               const int32_t* __restrict__ pos,
                                                               int32_t* __restrict__ row_idxs,
                                                               int32_t* __restrict__ compact_ptrs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && row_ptrs[i + 1] > row_ptrs[i]) {
        row_idxs[pos[i]] = i;
        compact_ptrs[pos[i]] = row_ptrs[i];
    }
    if (i == n) compact_ptrs[pos[n]] = row_ptrs[n];
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

struct gkomi_dist_ctx {
    hipStream_t comm_stream = nullptr;
    hipEvent_t packed = nullptr;
    hipEvent_t received = nullptr;
};

extern "C" int gkomi_dist_ctx_create(gkomi_dist_ctx** out)
{
    if (out == nullptr) return GKOMI_EINVAL;
    gkomi_dist_ctx* c = new gkomi_dist_ctx;
    int err = static_cast<int>(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    if (!err) err = static_cast<int>(hipEventCreateWithFlags(&c->packed, hipEventDisableTiming));
    if (!err) err = static_cast<int>(hipEventCreateWithFlags(&c->received, hipEventDisableTiming));
    if (err) {
        delete c;
        return err;
    }
    *out = c;
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_dist_ctx_destroy(gkomi_dist_ctx* c)
{
    if (c == nullptr) return GKOMI_SUCCESS;
    (void)hipEventDestroy(c->packed);
    (void)hipEventDestroy(c->received);
    (void)hipStreamDestroy(c->comm_stream);
    delete c;
    return GKOMI_SUCCESS;
}

extern "C" size_t gkomi_dist_nonlocal_rows_workspace_bytes(int64_t n_local)
{
    if (n_local < 0) return 0;
    const size_t scan_bytes = scan_workspace_bytes(n_local + 1);
    return 2 * align_up(sizeof(int32_t) * (n_local + 1), 256) + align_up(scan_bytes, 256) + 256;
}

// rows of the non-local block (CSR over all n_local rows) that have entries:
// row_idxs_out[k] (ascending) and compact_ptrs_out[k] (k = 0 .. count, into the
// same col_idxs / vals); both arrays sized n_local + 1.  Blocking: *host_count.
extern "C" int gkomi_dist_nonlocal_rows_i32(gkomi_stream_t s, int64_t n_local, const int32_t* row_ptrs,
                                            int32_t* row_idxs_out, int32_t* compact_ptrs_out,
                                            void* workspace, size_t workspace_bytes, int64_t* host_count)
{
    if (n_local < 0 || host_count == nullptr) return GKOMI_EINVAL;
    if (n_local > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    if (workspace == nullptr || workspace_bytes < gkomi_dist_nonlocal_rows_workspace_bytes(n_local)) {
        return GKOMI_EWORKSPACE;
    }
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    const size_t vec = align_up(sizeof(int32_t) * (n_local + 1), 256);
    int32_t* flag = reinterpret_cast<int32_t*>(ws);
    int32_t* pos = reinterpret_cast<int32_t*>(ws + vec);
    void* tmp = ws + 2 * vec;
    size_t tmp_bytes = workspace_bytes - 2 * vec;
    const int n = static_cast<int>(n_local);
    const dim3 grid(static_cast<unsigned>(ceildiv(n_local + 1, 256)));
    hipLaunchKernelGGL(dist_flag_rows_kernel, grid, dim3(256), 0, stream, n, row_ptrs, flag);
    GKOMI_TRY(exclusive_sum_i32(stream, flag, pos, n_local + 1, tmp, tmp_bytes));
    hipLaunchKernelGGL(dist_compact_rows_kernel, grid, dim3(256), 0, stream, n, row_ptrs, pos, row_idxs_out,
                       compact_ptrs_out);
    int32_t count = 0;
    GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&count, pos + n_local, sizeof(int32_t), hipMemcpyDeviceToHost, stream)));
    GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    *host_count = count;
    return check_launch();
}

namespace {

int check_matrix(const gkomi_dist_matrix* A, const gkomi_comm* comm)
{
    if (A == nullptr || comm == nullptr || comm->allreduce_sum_f64 == nullptr || comm->alltoallv == nullptr) {
        return GKOMI_EINVAL;
    }
    if (A->n_local < 0 || A->n_local > INT32_MAX - 1024 || A->l_nnz < 0 || A->l_nnz > INT32_MAX) {
        return GKOMI_ENOTSUPPORTED;
    }
    if (A->send_counts == nullptr || A->send_offsets == nullptr || A->recv_counts == nullptr ||
        A->recv_offsets == nullptr) {
        return GKOMI_EINVAL;
    }
    return GKOMI_SUCCESS;
}

// the halo exchange: pack on the compute stream, the wire on the side stream
int start_exchange(hipStream_t stream, const gkomi_comm* comm, gkomi_dist_ctx* ctx,
                   const gkomi_dist_matrix* A, const double* b)
{
    if (A->send_total > 0) {
        GKOMI_TRY(gkomi_dense_row_gather_f64_i32(stream, A->send_total, 1, A->gather_idxs, b, 1, A->send_buf, 1));
    }
    GKOMI_TRY(static_cast<int>(hipEventRecord(ctx->packed, stream)));
    GKOMI_TRY(static_cast<int>(hipStreamWaitEvent(ctx->comm_stream, ctx->packed, 0)));
    // every rank takes part, whatever it sends (matrix.cpp:263-303): the collective
    // is the same on all ranks even when one of them has no neighbours
    GKOMI_TRY(comm->alltoallv(comm->self, ctx->comm_stream, A->send_buf, A->send_counts, A->send_offsets,
                              A->recv_buf, A->recv_counts, A->recv_offsets, static_cast<int>(sizeof(double))));
    GKOMI_TRY(static_cast<int>(hipEventRecord(ctx->received, ctx->comm_stream)));
    return GKOMI_SUCCESS;
}

int nonlocal_part(hipStream_t stream, gkomi_dist_ctx* ctx, const gkomi_dist_matrix* A, double* x,
                  const double* w, double* partial, const cg_scalars* scal)
{
    GKOMI_TRY(static_cast<int>(hipStreamWaitEvent(stream, ctx->received, 0)));
    if (A->nl_rows <= 0) return GKOMI_SUCCESS;
    const dim3 grid(static_cast<unsigned>(ceildiv(A->nl_rows, 256)));
    if (partial != nullptr) {
        hipLaunchKernelGGL(dist_nonlocal_rows_kernel<true>, grid, dim3(256), 0, stream,
                           static_cast<int>(A->nl_rows), A->nl_row_idxs, A->nl_row_ptrs, A->nl_col_idxs,
                           A->nl_vals, A->recv_buf, x, w, partial, scal);
    } else {
        hipLaunchKernelGGL(dist_nonlocal_rows_kernel<false>, grid, dim3(256), 0, stream,
                           static_cast<int>(A->nl_rows), A->nl_row_idxs, A->nl_row_ptrs, A->nl_col_idxs,
                           A->nl_vals, A->recv_buf, x, w, partial, scal);
    }
    return check_launch();
}

struct dist_layout {
    size_t r, z, p, q, scalars, part_a, part_b, part_c, part_d, red_a, red_b, red, small, total;
};

dist_layout make_dist_layout(int64_t n, int64_t nl_rows)
{
    dist_layout l{};
    const size_t vec = align_up(sizeof(double) * static_cast<size_t>(n > 0 ? n : 1), 256);
    size_t off = 0;
    l.r = off; off += vec;
    l.z = off; off += vec;
    l.p = off; off += vec;
    l.q = off; off += vec;
    l.scalars = off; off += 256;
    l.part_a = off; off += align_up(sizeof(double) * max_parts, 256);
    l.part_b = off; off += align_up(sizeof(double) * max_parts, 256);
    // the local block's SpMV + dot launch: the row-cut kernel leaves a partial per 256 rows, the nonzero-split one per tile
    l.part_c = off; off += align_up(sizeof(double) * (spmv_dot_partials_room(n) + 1), 256);
    l.part_d = off; off += align_up(sizeof(double) * (static_cast<size_t>(ceildiv(nl_rows, 256)) + 1), 256);
    l.red_a = off; off += 256;
    l.red_b = off; off += 256;
    l.red = off; off += align_up(gkomi_dense_reduction_workspace_bytes(n, 1) + 8, 256);
    l.small = off; off += 256;
    l.total = off;
    return l;
}

}  // namespace

extern "C" int gkomi_dist_matrix_apply_f64(gkomi_stream_t s, const gkomi_comm* comm, gkomi_dist_ctx* ctx,
                                           const gkomi_dist_matrix* A, const double* b, double* x)
{
    GKOMI_TRY(check_matrix(A, comm));
    if (ctx == nullptr) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(s);
    GKOMI_TRY(start_exchange(stream, comm, ctx, A, b));
    if (A->n_local > 0) {
        GKOMI_TRY(gkomi_csr_spmv_srow_f64_i32(s, A->n_local, A->n_local, 1, A->l_nnz, A->l_row_ptrs,
                                              A->l_col_idxs, A->l_vals, b, 1, x, 1, nullptr, nullptr, 0,
                                              A->l_max_row_nnz, A->l_srow, A->l_srow_tile));
    }
    return nonlocal_part(stream, ctx, A, x, nullptr, nullptr, nullptr);
}

extern "C" size_t gkomi_dist_cg_workspace_bytes(int64_t n_local, int64_t nl_rows)
{
    if (n_local < 0 || nl_rows < 0) return 0;
    return make_dist_layout(n_local, nl_rows).total;
}

extern "C" int gkomi_dist_cg_solve_f64(gkomi_stream_t s, const gkomi_comm* comm, gkomi_dist_ctx* ctx,
                                       const gkomi_dist_matrix* A, gkomi_apply_fn precond, void* precond_ctx,
                                       const double* b, double* x, int64_t max_iters, double reduction_factor,
                                       int baseline, int check_every, void* workspace, size_t workspace_bytes,
                                       double* host_info)
{
    GKOMI_TRY(check_matrix(A, comm));
    if (ctx == nullptr || max_iters < 0 || baseline < 0 || baseline > 2) return GKOMI_EINVAL;
    const int64_t n = A->n_local;
    const dist_layout l = make_dist_layout(n, A->nl_rows);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    if (reinterpret_cast<uintptr_t>(A->l_vals) % 16 != 0 || reinterpret_cast<uintptr_t>(A->l_col_idxs) % 8 != 0 ||
        reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(workspace) % 16 != 0) {
        return GKOMI_ENOTSUPPORTED;  // the fused kernels move 16 B per lane
    }
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    double* r = reinterpret_cast<double*>(ws + l.r);
    double* z = reinterpret_cast<double*>(ws + l.z);
    double* p = reinterpret_cast<double*>(ws + l.p);
    double* q = reinterpret_cast<double*>(ws + l.q);
    cg_scalars* scal = reinterpret_cast<cg_scalars*>(ws + l.scalars);
    double* part_a = reinterpret_cast<double*>(ws + l.part_a);
    double* part_b = reinterpret_cast<double*>(ws + l.part_b);
    double* part_c = reinterpret_cast<double*>(ws + l.part_c);
    double* part_d = reinterpret_cast<double*>(ws + l.part_d);
    double* red_a = reinterpret_cast<double*>(ws + l.red_a);  // {r.z, r.r}
    double* red_b = reinterpret_cast<double*>(ws + l.red_b);  // {p.q}
    void* red = ws + l.red;
    const size_t red_bytes = gkomi_dense_reduction_workspace_bytes(n, 1) + 8;
    double* small = reinterpret_cast<double*>(ws + l.small);
    double* orig_tau = small;
    double* prev_rho = small + 1;
    double* rho = small + 2;
    uint8_t* stop_status = reinterpret_cast<uint8_t*>(small + 8);

    // cg::initialize, r = b - A x, baseline norm (cg.cpp:137-142, residual_norm.cpp:119-189)
    GKOMI_TRY(gkomi_cg_initialize_f64(s, n, 1, b, 1, r, 1, z, 1, p, 1, q, 1, prev_rho, rho, stop_status));
    GKOMI_TRY(gkomi_dist_matrix_apply_f64(s, comm, ctx, A, x, q));
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, 1, rho, 1, 1.0));
    GKOMI_TRY(gkomi_dense_sub_scaled_f64(s, n, 1, rho, 1, q, 1, r, 1));
    if (baseline == 2) {
        GKOMI_TRY(gkomi_dense_fill_f64(s, 1, 1, orig_tau, 1, 1.0));
    } else {
        // Vector::compute_norm2: local sum of squares, all-reduce, square root
        GKOMI_TRY(gkomi_dense_compute_squared_norm2_f64(s, n, 1, baseline == 0 ? b : r, 1, orig_tau, red, red_bytes));
        GKOMI_TRY(comm->allreduce_sum_f64(comm->self, s, orig_tau, 1));
        GKOMI_TRY(gkomi_dense_compute_sqrt_f64(s, 1, 1, orig_tau, 1));
    }
    hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(1), 0, stream, scal, orig_tau, baseline == 2 ? 1 : 0);
    GKOMI_TRY(check_launch());

    const int g = vec_grid(n);
    const int nd = static_cast<int>(ceildiv(A->nl_rows, 256));
    // the local block is a CSR matrix like any other driver's: with its srow it runs the nonzero-split kernel the
    // apply runs (csr_spmv.hip), nontemporal when matrix + vectors of an iteration exceed the Infinity Cache
    sysmat local = make_csr_sysmat(n, A->l_nnz, A->l_row_ptrs, A->l_col_idxs, A->l_vals, GKOMI_CSR_AUTO, A->l_max_row_nnz);
    local.srow = A->l_srow;
    local.srow_tile = A->l_srow != nullptr ? A->l_srow_tile : 0;
    local.note_working_set(static_cast<int64_t>(sizeof(double)) * n * 6);
    const spmv_dot_plan spmv(local);
    if (n > 0 && !spmv.fused()) return GKOMI_ENOTSUPPORTED;  // (aligned CSR arrays were checked above)
    const int nb = spmv.num_partials;
    const double* zz = precond == nullptr ? r : z;
    if (check_every < 1) check_every = 1;
    // partials of r.z and r.r -> {rho, tau^2} in one all-reduce
    auto reduce_rho_tau = [&](const cg_scalars* gate) -> int {
        hipLaunchKernelGGL(dist_sum_partials_kernel, dim3(1), dim3(fblock), 0, stream, part_a, g,
                           static_cast<const double*>(nullptr), 0, precond == nullptr ? part_a : part_b, g,
                           red_a, gate);
        GKOMI_TRY(check_launch());
        return comm->allreduce_sum_f64(comm->self, s, red_a, 2);
    };
    if (precond != nullptr) GKOMI_TRY(precond(precond_ctx, s, r, z));
    hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, zz,
                       static_cast<const cg_scalars*>(nullptr), part_a, precond == nullptr ? nullptr : part_b);
    GKOMI_TRY(check_launch());
    GKOMI_TRY(reduce_rho_tau(nullptr));

    cg_scalars polled{};
    long long it = 0, iterations = -1;
    int converged = 0;
    bool done = false;
    while (!done) {
        for (int c = 0; c < check_every; ++c, ++it) {
            // K1: criterion on the reduced scalars, p = z + (rho / prev_rho) p
            hipLaunchKernelGGL(cg_fused_step1_kernel, dim3(g), dim3(fblock), 0, stream, n, p, zz, red_a, 1,
                               red_a + 1, 1, scal, it, static_cast<long long>(max_iters), reduction_factor);
            // q = A p: halo of p on the wire while the local block runs
            GKOMI_TRY(start_exchange(stream, comm, ctx, A, p));
            if (n > 0) {
                GKOMI_TRY(spmv.launch(stream, p, q, part_c, &scal->status));
            }
            GKOMI_TRY(nonlocal_part(stream, ctx, A, q, p, nd > 0 ? part_d : nullptr, scal));
            // beta = p.q: local partials (+ what the non-local rows added), all-reduce
            hipLaunchKernelGGL(dist_sum_partials_kernel, dim3(1), dim3(fblock), 0, stream, part_c, n > 0 ? nb : 0,
                               nd > 0 ? part_d : static_cast<const double*>(nullptr), nd,
                               static_cast<const double*>(nullptr), 0, red_b, static_cast<const cg_scalars*>(scal));
            GKOMI_TRY(comm->allreduce_sum_f64(comm->self, s, red_b, 1));
            // K3: x += (rho / beta) p, r -= (rho / beta) q, partials of r.r
            hipLaunchKernelGGL(cg_fused_step2_kernel, dim3(g), dim3(fblock), 0, stream, n, x, r, p, q, red_b, 1,
                               scal, it, precond == nullptr ? part_a : part_b);
            if (precond != nullptr) {
                GKOMI_TRY(precond(precond_ctx, s, r, z));
                hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, z,
                                   static_cast<const cg_scalars*>(scal), part_a, static_cast<double*>(nullptr));
            }
            GKOMI_TRY(reduce_rho_tau(scal));
            if (it >= max_iters) {
                ++it;
                break;
            }
        }
        GKOMI_TRY(check_launch());
        GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&polled, scal, sizeof(cg_scalars), hipMemcpyDeviceToHost, stream)));
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
        if (polled.status & GKOMI_STATUS_ID_MASK) {
            done = true;
            iterations = polled.stop_iter;
            converged = (polled.status & GKOMI_STATUS_CONVERGED) ? 1 : 0;
        }
    }
    if (host_info != nullptr) {
        host_info[0] = static_cast<double>(iterations);
        host_info[1] = static_cast<double>(converged);
        host_info[2] = polled.tau;
        host_info[3] = polled.orig_tau;
    }
    return precond_status(precond, precond_ctx, s);
}
