// Cross-file host entry points inside the library (not part of the C ABI).
#pragma once
#include "common.hpp"

namespace gkomi {

int csr_spmv_dot_launch(hipStream_t stream, int nrows, int64_t nnz,
                        const int32_t* row_ptrs, const int32_t* col_idxs,
                        const double* vals, const double* p, double* q,
                        double* partial, const uint8_t* stop_status,
                        bool swizzle, const double* dot_w = nullptr,
                        double* partial2 = nullptr, bool nontemporal = false);
int csr_spmv_dot_num_partials(int nrows);
// the nonzero-split kernel with the dot epilogue, for matrices with an srow that stream from HBM
int csr_split_dot_num_partials(int64_t nnz, int64_t tile);
int csr_split_dot_launch(hipStream_t stream, int nrows, int64_t nnz, const int32_t* row_ptrs,
                         const int32_t* col_idxs, const double* vals, const double* p, double* q,
                         double* partial, const uint8_t* stop_status, const int32_t* srow, int64_t tile,
                         int over, bool swizzle, bool nontemporal, const double* dot_w = nullptr,
                         double* partial2 = nullptr);
bool csr_auto_swizzle(int64_t nrows, int64_t nnz);
// `raw` partial sums -> at most spmv_dot_max_partials, in place order (each output = the sum of a run of
// consecutive inputs, added in index order by one wave): what the consumers of a fused driver re-add
constexpr int spmv_dot_max_partials = 4096;
// ... but a launch that leaves up to twice as many is not followed by the compression: its consumers are the <= 512
// workgroups of a system of that size (fused_vec_grid), 8192 partials are 64 KB of L2 reads each, and the extra
// launch costs 4.8 us (108^3 system: 5711 tiles; profiles/r03_p3_cg_kernels.md)
constexpr int spmv_dot_uncompressed_partials = 8192;
int compress_partials_launch(hipStream_t stream, const double* raw, int nraw, double* out, int nout,
                             const double* raw2, double* out2, const uint8_t* stop_status);
// room a fused driver keeps for the partials of one SpMV + dot launch (doubles): the row-cut
// kernel leaves one per 256 rows, the nonzero-split kernel one per tile -- up to n / 64 + 1 (rows of up
// to ~24 nonzeros on average with the 1536 tile), in front of them the compressed ones
inline size_t spmv_dot_partials_room(int64_t n)
{
    return static_cast<size_t>(spmv_dot_max_partials) + static_cast<size_t>(n / 64 + 2);
}
// the Infinity Cache: a solve whose working set between two applies of A is larger reads A nontemporally
constexpr int64_t infinity_cache_bytes = int64_t{256} << 20;

// One single-launch ("persistent") solver kernel at a time per process: two of them would each
// hold some CUs and wait for the rest (runtime.hip).  try_acquire is non-blocking.
bool persistent_try_acquire();
void persistent_release();
int device_cu_count();  // CUs of the current device, 0 = unknown

// ---- the host's view of a running solve, without stopping it ---------------------------------------
// The fused drivers evaluate their stopping criterion on the device; the host only has to learn, sooner
// or later, that it fired, and must not run too far ahead of the device meanwhile.  A blocking look
// (copy + synchronize) leaves the GPU idle for the host's turnaround -- ~85 us per look on this pool,
// 21 us per GMRES iteration at a look every 4 iterations (profiles/r03_poll_gap.md).  Instead the ONE
// thread that evaluates the criterion also stores {iteration reached, iteration stopped} into a line of
// pinned, fine-grained host memory (system-scope release store, one PCIe write per iteration), and the
// host reads that line between launches: no copy, no synchronisation, the queue never drains.  Nothing
// depends on those stores for correctness: if they never show up, host_watch::wait notices that the
// stream has drained and says so, and the driver looks at device memory the old way.
struct host_watch_line {
    // ONE 64-bit word, so that the host never sees `done` of one publication with `stop_iter` of another (two
    // words may arrive in either order: at a GMRES restart boundary a host that saw `done` without the stop ran an
    // extra update + residual on a solve that had stopped):  low half = done + 1 (the criterion has been
    // evaluated for every iteration <= done; 0: none yet), high half = stop_iter + 1 (0: not stopped)
    unsigned long long word;
    long long pad_[7];
    static long long done_of(unsigned long long w) { return static_cast<long long>(w & 0xffffffffull) - 1; }
    static long long stop_of(unsigned long long w) { return static_cast<long long>(w >> 32) - 1; }
};
struct host_watch {
    host_watch_line* host = nullptr;  // as the host reads it
    host_watch_line* dev = nullptr;   // as kernels write it; nullptr = not available, drivers poll device memory
    int slot = -1;
    host_watch();                     // a free line of this thread's pinned block, reset to {-1, -1}
    ~host_watch();
    host_watch(const host_watch&) = delete;
    host_watch& operator=(const host_watch&) = delete;
    // Returns true once the device has evaluated iteration `target` or has stopped; false if the stream
    // drained without either becoming visible (the caller then reads device memory and stops using this).
    bool wait(hipStream_t stream, long long target);
    long long stop_iter() const;
};
#ifdef __HIPCC__
__device__ __forceinline__ void host_watch_publish(host_watch_line* w, long long done, long long stop_iter)
{
    if (w == nullptr) return;
    // relaxed on purpose: a system-scope RELEASE writes back the XCD's dirty L2 lines first (microseconds, in
    // the kernel's critical thread).  One word: both values of one publication arrive together.
    const unsigned long long word = (static_cast<unsigned long long>(done + 1) & 0xffffffffull) |
                                    (static_cast<unsigned long long>(stop_iter >= 0 ? stop_iter + 1 : 0) << 32);
    __hip_atomic_store(&w->word, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
// how far the host lets itself run ahead of the last iteration it has seen evaluated: enough to keep the
// queue from draining (a launch costs the host ~4 us), few enough that the launches issued after the
// criterion fired (they return at once, but a preconditioner's do not) stay cheap
constexpr long long host_watch_lag = 3;

constexpr int fused_vec_block = 1024, fused_vec_max_parts = 1024;
// Workgroups of the fused vector kernels (1024 lanes, 16 B per lane and sweep): one per 2048 rows, at most fused_vec_max_parts.
// Two are resident per CU; a grid between one and two rounds of resident workgroups would run a second, mostly empty
// round -- on the 108^3 system (616 workgroups on 256 CUs) K1 + K3 of a CG iteration (cg_fused.hpp) took 4 us longer than with 512
// workgroups that each sweep twice (47.6 -> 43.6 us per iteration).  Beyond two rounds (the 256^3 system) the full
// fused_vec_max_parts stay: 1024 workgroups 532 us per iteration, 512 540 (profiles/r03_p3_cg_kernels.md).
inline int fused_vec_grid(int64_t n)
{
    static const int64_t resident = 2 * static_cast<int64_t>(device_cu_count());
    int64_t g = ceildiv(n / 2 + 1, fused_vec_block);
    if (g > fused_vec_max_parts) {
        g = fused_vec_max_parts;
    } else if (g > resident && resident > 0) {
        g = resident;
    }
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

// block-Jacobi apply with the partials of r . z and r . r in the same launch (jacobi.hip); > 0: number of partials,
// 0: not for this preconditioner, < 0: -(error) - 1000
int jacobi_apply_dot_launch(gkomi_stream_t s, const gkomi_jacobi_ctx* c, const double* in, double* out, double* part_rz,
                            double* part_rr, size_t room, const uint8_t* stop_status);

// a brick solve as one link of a chain on contiguous vectors (trs_bricks.hip): see the definition
int trs_bricks_solve_chained(gkomi_stream_t s, gkomi_trs_bricks* h, void* plan, int unit_diag, const double* b, double* x,
                             bool x_is_armed, double* arm, double* rearm_b);

// end-of-solve health check of a preconditioner callback (blocking): 0 or an error such as
// GKOMI_ETRS_OVERRUN when one of the ILU's triangular solves gave up (precond.hip)
int precond_status(gkomi_apply_fn precond, void* ctx, gkomi_stream_t s);

// The system matrix of a solver driver: CSR arrays (op == nullptr) or any
// format behind a gkomi_matrix_apply_fn (Ell, Sellp, Coo, Hybrid, a user LinOp).
struct sysmat {
    int64_t n = 0, nnz = 0;
    const int32_t* row_ptrs = nullptr;
    const int32_t* col_idxs = nullptr;
    const double* vals = nullptr;
    int strategy = 0;
    int64_t hint = -1;
    const int32_t* srow = nullptr;  // tile start rows of the nonzero-split kernel (Csr::srow_), or none
    int64_t srow_tile = 0;
    gkomi_matrix_apply_fn op = nullptr;
    void* ctx = nullptr;
    bool is_csr() const { return op == nullptr; }
    // out = A in (alpha == beta == nullptr) or out = alpha A in + beta out; stride == nrhs
    int apply(gkomi_stream_t s, int64_t nrhs, const double* alpha, const double* in,
              const double* beta, double* out) const
    {
        if (op != nullptr) return op(ctx, s, nrhs, alpha, in, nrhs, beta, out, nrhs);
        return gkomi_csr_spmv_srow_f64_i32(s, n, n, nrhs, nnz, row_ptrs, col_idxs, vals, in, nrhs, out,
                                           nrhs, alpha, beta, strategy, hint, srow, srow_tile);
    }
    int64_t storage_bytes() const { return is_csr() ? 12 * nnz + 4 * (n + 1) : 0; }
    // What the driver knows and a single apply cannot: between two applies of A the solve moves
    // `vector_bytes` of vectors besides the matrix.  Beyond the Infinity Cache the matrix will not be
    // found there at its next use -> nontemporal matrix streams (GKOMI_CSR_STREAMING), which also
    // leaves the cache to the vectors (not a bench-only flag).
    void note_working_set(int64_t vector_bytes)
    {
        if (is_csr() && (strategy & 0xff) == GKOMI_CSR_AUTO && storage_bytes() + vector_bytes > infinity_cache_bytes) {
            strategy |= GKOMI_CSR_STREAMING;
        }
    }
};

inline sysmat make_csr_sysmat(int64_t n, int64_t nnz, const int32_t* row_ptrs,
                              const int32_t* col_idxs, const double* vals, int strategy,
                              int64_t hint)
{
    sysmat A;
    A.n = n; A.nnz = nnz; A.row_ptrs = row_ptrs; A.col_idxs = col_idxs; A.vals = vals;
    A.strategy = strategy; A.hint = hint;
    return A;
}

// a CSR matrix behind the library's own callback is CSR to every driver (its srow, strategy and row
// statistic travel in the record); anything else stays an operator
inline sysmat make_op_sysmat(int64_t n, gkomi_matrix_apply_fn op, void* ctx)
{
    sysmat A;
    if (op == &gkomi_csr_matrix_apply_cb && ctx != nullptr) {
        const auto* m = static_cast<const gkomi_csr_ctx*>(ctx);
        if (m->nrows == n && m->ncols == n) {
            A = make_csr_sysmat(n, m->nnz, m->row_ptrs, m->col_idxs, m->vals, static_cast<int>(m->strategy),
                                m->max_row_nnz_hint);
            A.srow = m->srow;
            A.srow_tile = m->srow != nullptr ? m->srow_tile : 0;
            return A;
        }
    }
    A.n = n; A.op = op; A.ctx = ctx;
    return A;
}

// ELL / SELL-P behind the library's own callbacks (formats.hip): partials per
// launch of the SpMV + dot epilogue, 0 for any other operator
int op_spmv_dot_num_partials(gkomi_matrix_apply_fn op, const void* ctx);
int op_spmv_dot_launch(hipStream_t stream, gkomi_matrix_apply_fn op, const void* ctx,
                       const double* in, double* out, double* partial,
                       const uint8_t* stop_status, const double* dot_w, double* partial2);

// How a fused single-rhs driver gets out = A in together with the per-block
// partials of w . out (w = in unless given) and, on request, of out . out in
// ONE launch: the CSR stream kernel's epilogue, the ELL / SELL-P kernels'
// epilogue, or not at all (num_partials == 0: any other operator -- the driver
// follows A.apply with its own partials kernel).  A CSR matrix behind
// gkomi_csr_matrix_apply_cb counts as CSR.
struct spmv_dot_plan {
    int num_partials = 0;  // what the consumers re-add (<= spmv_dot_max_partials once compressed)
    int raw_partials = 0;  // what the launch leaves (behind the compressed ones when there are more)
    bool csr = false, swizzle = false, split = false, nontemporal = false;
    int over = 0;
    sysmat A;
    explicit spmv_dot_plan(const sysmat& A_) : A(A_)
    {
        if (A.n <= 0 || A.n > INT32_MAX) return;
        if (A.is_csr()) {
            if (reinterpret_cast<uintptr_t>(A.vals) % 16 != 0 ||
                reinterpret_cast<uintptr_t>(A.col_idxs) % 8 != 0) {
                return;
            }
            const int kind = A.strategy & 0xff;
            csr = true;
            swizzle = csr_auto_swizzle(A.n, A.nnz);
            nontemporal = !swizzle || (A.strategy & GKOMI_CSR_STREAMING) != 0;
            raw_partials = csr_spmv_dot_num_partials(static_cast<int>(A.n));
            // the matrix carries its srow: cut by nonzeros (rows of up to 65 entries are summed from
            // the tile; a longer row is finished from memory, so an unknown row length is safe too)
            const int nsplit = A.srow != nullptr && (kind == GKOMI_CSR_AUTO || kind == GKOMI_CSR_SPLIT) && A.nnz >= 2 && A.hint <= 65
                                   ? csr_split_dot_num_partials(A.nnz, A.srow_tile)
                                   : 0;
            if (nsplit > 0 && static_cast<size_t>(nsplit) + spmv_dot_max_partials <= spmv_dot_partials_room(A.n)) {
                split = true;
                raw_partials = nsplit;
                over = A.hint < 0 ? 64 : (A.hint <= 1 ? 0 : static_cast<int>(A.hint / 2 * 2));
            }
        } else if (A.ctx != nullptr) {
            raw_partials = op_spmv_dot_num_partials(A.op, A.ctx);
        }
        num_partials = raw_partials > spmv_dot_uncompressed_partials ? spmv_dot_max_partials : raw_partials;
    }
    bool fused() const { return num_partials > 0; }
    // `partial` (and `partial2`) must hold spmv_dot_partials_room(n) doubles
    int launch(hipStream_t stream, const double* in, double* out, double* partial,
               const uint8_t* stop_status, const double* dot_w = nullptr,
               double* partial2 = nullptr) const
    {
        const bool squeeze = raw_partials > num_partials;
        double* raw = squeeze ? partial + spmv_dot_max_partials : partial;
        double* raw2 = partial2 != nullptr && squeeze ? partial2 + spmv_dot_max_partials : partial2;
        int err;
        if (split) {
            err = csr_split_dot_launch(stream, static_cast<int>(A.n), A.nnz, A.row_ptrs, A.col_idxs, A.vals, in, out,
                                       raw, stop_status, A.srow, A.srow_tile, over, swizzle, nontemporal, dot_w, raw2);
        } else if (csr) {
            err = csr_spmv_dot_launch(stream, static_cast<int>(A.n), A.nnz, A.row_ptrs, A.col_idxs,
                                      A.vals, in, out, raw, stop_status, swizzle, dot_w, raw2, nontemporal);
        } else {
            err = op_spmv_dot_launch(stream, A.op, A.ctx, in, out, raw, stop_status, dot_w, raw2);
        }
        if (err != 0 || !squeeze) return err;
        return compress_partials_launch(stream, raw, raw_partials, partial, num_partials, raw2, partial2, stop_status);
    }
};

}  // namespace gkomi
