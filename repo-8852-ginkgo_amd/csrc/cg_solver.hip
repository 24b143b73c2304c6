// Native CG driver for gfx950: Cg::apply_dense_impl (core/solver/cg.cpp:107-193)
// with the criteria Combined(Iteration, ResidualNorm)
// (core/stop/combined.cpp:40, core/stop/residual_norm.cpp:119-228).
//
// mode 0 replays the reference's kernel sequence one launch per kernel and
// checks the criterion on the host every iteration (a blocking 2-byte D2H copy
// per iteration, as hip/stop/residual_norm_kernels.hip.cpp:119-120 does).
//
// mode 1 is the MI355X design: three launches per iteration, every scalar on
// the device, no per-iteration host round trip.
//   K1 step1 : every workgroup re-adds the <=1024 partials of rho = r.z and
//              tau^2 = r.r left by K3 (same order in every workgroup, so all
//              agree bit for bit), evaluates the criterion, and -- unless
//              stopped -- p = z + (rho/prev_rho) p.
//   K2 spmv  : q = A p (stream kernel) + partials of beta = p.q.
//   K3 step2 : re-adds the beta partials, x += (rho/beta) p, r -= (rho/beta) q,
//              and leaves the partials of r.r for the next K1.
// Once the criterion fires, K1 records the iteration and sets the
// stopping_status; all later launches return immediately, so x, r and the
// iteration count are exactly those of the iteration that stopped.  The host
// polls the status every `check_every` iterations.
// HBM traffic per iteration (n rows, Identity): K1 3n, K2 matrix + 2n (+n
// for p in the epilogue, L2-resident), K3 6n values -- vs 18n + matrix in the
// reference's accounting (core/solver/cg.cpp:148-156).
#include "cg_persistent.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace gkomi {
namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct workspace_layout {
    size_t r, z, p, q, scalars, part_a, part_b, part_c, red, small, pcg_ctl, pcg_slots, total;
};

workspace_layout make_layout(int64_t n, int64_t nrhs)
{
    workspace_layout l{};
    const size_t vec = align_up(sizeof(double) * static_cast<size_t>(n) * nrhs, 256);
    size_t off = 0;
    l.r = off; off += vec;
    l.z = off; off += vec;
    l.p = off; off += vec;
    l.q = off; off += vec;
    l.scalars = off; off += 256;
    // (room for one partial per workgroup of a block-Jacobi apply that carries the dots, see the fused loop)
    l.part_a = off; off += align_up(sizeof(double) * spmv_dot_partials_room(n), 256);   // r.z / r.r
    l.part_b = off; off += align_up(sizeof(double) * spmv_dot_partials_room(n), 256);   // r.r with a preconditioner
    l.part_c = off; off += align_up(sizeof(double) * spmv_dot_partials_room(n), 256);    // p.q (internal.hpp)
    l.red = off; off += align_up(gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8, 256);
    // mode 0 scalars: alpha-free set {prev_rho, rho, beta, tau, orig_tau, one, neg_one} x nrhs,
    // then stop_status[nrhs] and 2 flag bytes
    l.small = off; off += align_up(sizeof(double) * 8 * static_cast<size_t>(nrhs) + nrhs + 16, 256);
    // the persistent single-launch solve: its control word and two banks of one slot per workgroup
    l.pcg_ctl = off; off += align_up(sizeof(pcg_control), 256);
    l.pcg_slots = off; off += align_up(sizeof(pcg_slot) * 2 * (max_parts + pcg_copies) * pcg_max_stride, 256);
    l.total = off;
    return l;
}

#define GKOMI_TRY(expr)          \
    do {                         \
        int err_ = (expr);       \
        if (err_) return err_;   \
    } while (0)

std::atomic<int64_t> pcg_solves{0};  // solves finished by the single-launch kernel (diagnostics, tests)
// GKOMI_CG_PERSISTENT at start-up, gkomi_cg_persistent_enable afterwards
std::atomic<int> pcg_mode{[] {
    const char* e = std::getenv("GKOMI_CG_PERSISTENT");
    return e == nullptr ? 1 : std::atoi(e);
}()};

int identity_or_precond(gkomi_apply_fn precond, void* ctx, gkomi_stream_t s,
                        int64_t n, int64_t nrhs, const double* r, double* z)
{
    if (precond == nullptr) {
        // matrix::Identity::apply copies (core/matrix/identity.cpp)
        return gkomi_dense_copy_f64(s, n, nrhs, r, nrhs, z, nrhs);
    }
    return precond(ctx, s, r, z);
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int64_t gkomi_cg_persistent_solves(void) { return pcg_solves.load(); }
extern "C" int gkomi_cg_persistent_enable(int mode)
{
    if (mode < 0 || mode > 2) return GKOMI_EINVAL;
    pcg_mode.store(mode);
    return GKOMI_SUCCESS;
}

extern "C" size_t gkomi_cg_workspace_bytes(int64_t n, int64_t nrhs)
{
    if (n < 0 || nrhs <= 0) return 0;
    return make_layout(n, nrhs).total;
}

namespace {
// The whole solve in one launch (cg_persistent.hpp) when the vectors AND the matrix fit the
// register files.  Returns 1 when the solve is finished (*result = the final scalars), 0 when the
// path does not apply or gave up (state restored for the three-launch iteration: r = b - A x from
// whatever x the kernel left, p = 0, scalars re-initialised), -(1000 + code) on a HIP / library error.
#define PCG_TRY(expr)                        \
    do {                                     \
        const int err_ = (expr);             \
        if (err_) return -(1000 + err_);     \
    } while (0)
int persistent_cg(gkomi_stream_t s, int64_t n, const sysmat& A, const spmv_dot_plan& spmv, gkomi_apply_fn precond,
                  const double* b, double* x, double* r, double* p, double* q, const double* one,
                  const double* neg_one, const double* orig_tau, int baseline, int64_t max_iters,
                  double reduction_factor, cg_scalars* scal, void* ctl_mem, void* slot_mem, cg_scalars* result)
{
    hipStream_t stream = to_stream(s);
    cg_scalars polled{};
    // The whole solve in one launch (cg_persistent.hpp) when the vectors AND the matrix fit the
    // register files: Identity preconditioner, aligned CSR, rows of at most 7 nonzeros (the
    // caller's max_row_nnz_hint says so; the kernel checks), one workgroup per CU.
    // GKOMI_CG_PERSISTENT=0 turns it off, =2 also takes matrices that do not fit (they stream
    // from memory every iteration: 33 vs 34.8 us per iteration on P2, not worth the rendezvous).
    const int persistent_mode = pcg_mode.load();
    static const int cus = device_cu_count();
    const int64_t pcg_chunk = cus > 0 ? ceildiv(n, cus) : 0;
    // an ELL system matrix behind the library's callback: its rows go into the registers just the same
    const gkomi_ell_ctx* ell = !A.is_csr() && A.op == &gkomi_ell_matrix_apply_cb && A.ctx != nullptr
                                   ? static_cast<const gkomi_ell_ctx*>(A.ctx)
                                   : nullptr;
    if (ell != nullptr && (ell->nrows != n || ell->ncols != n || ell->stride < n)) ell = nullptr;
    const int64_t pcg_hint = ell != nullptr ? ell->num_stored_per_row
                                            : spmv.A.hint;  // (a CSR matrix behind its callback carries its own)
    const bool pcg_fits_matrix =
        pcg_hint >= 1 && pcg_hint <= 7 && ceildiv(pcg_chunk, 512) <= 8;
    const bool pcg_fits_vectors = ceildiv(pcg_chunk, pcg_block) <= pcg_max_rows_per_thread;
    if (persistent_mode >= 1 && precond == nullptr && (spmv.csr || ell != nullptr) && cus >= 8 &&
        cus <= max_parts &&
        n >= 64 * static_cast<int64_t>(cus) &&
        (pcg_fits_matrix || (persistent_mode >= 2 && pcg_fits_vectors && ell == nullptr)) &&
        persistent_try_acquire()) {  // one persistent solve at a time per process
        struct release_guard {
            ~release_guard() { persistent_release(); }
        } release;
        const sysmat& M = spmv.A;
        const int32_t* m_row_ptrs = ell != nullptr ? nullptr : M.row_ptrs;
        const int32_t* m_col_idxs = ell != nullptr ? ell->col_idxs : M.col_idxs;
        const double* m_vals = ell != nullptr ? ell->vals : M.vals;
        const int ell_stored = ell != nullptr ? static_cast<int>(ell->num_stored_per_row) : 0;
        const int64_t ell_stride = ell != nullptr ? ell->stride : 0;
        pcg_control* ctl = static_cast<pcg_control*>(ctl_mem);
        pcg_slot* slots = static_cast<pcg_slot*>(slot_mem);
        const int chunk = static_cast<int>(ceildiv(n, cus));
        const int rows_per_thread = static_cast<int>(ceildiv(chunk, pcg_block));
        const long long max_polls = [] {
            const char* e = std::getenv("GKOMI_MEET_MAX_POLLS");  // test hook: how long a meeting waits
            return e != nullptr && e[0] != 0 ? std::max(1ll, atoll(e)) : 1ll << 22;
        }();
        static const int stride = [] {
            const char* e = std::getenv("GKOMI_PCG_STRIDE");  // slot spacing in 16-B units (tuning)
            const int v = e != nullptr ? std::atoi(e) : pcg_default_stride;
            return v >= 1 && v <= pcg_max_stride ? v : pcg_default_stride;
        }();
        static const bool resident_on = [] {
            const char* e = std::getenv("GKOMI_PCG_RESIDENT");
            return e == nullptr || e[0] != '0';
        }();
        static const int nap = [] {
            const char* e = std::getenv("GKOMI_PCG_NAP");
            const int v = e != nullptr ? std::atoi(e) : 1;
            return v >= 0 && v <= 64 ? v : 1;
        }();
        hipLaunchKernelGGL(pcg_clear_kernel, dim3(1), dim3(256), 0, stream, slots, stride, 2 * (cus + pcg_copies), ctl);
#define GKOMI_PCG(R, KR, BLOCK)                                                                        \
hipLaunchKernelGGL((cg_persistent_kernel<R, KR, BLOCK, (KR == 7 && R == 8)>), dim3(cus), dim3(BLOCK), 0, stream, \
                   static_cast<int>(n), chunk, m_row_ptrs, m_col_idxs, m_vals, x, r, p, q, slots,  \
                   stride, nap, ctl, scal, static_cast<long long>(max_iters), reduction_factor,    \
                   max_polls, ell_stored, ell_stride)
        // rows of at most 5 nonzeros, up to 8 rows per thread of a 512-thread workgroup (256
        // registers each): the matrix stays in registers
        const int rows_per_thread_512 = static_cast<int>(ceildiv(chunk, 512));
        const bool resident = pcg_fits_matrix && resident_on;
        if (resident && pcg_hint <= 5) {
            if (rows_per_thread_512 <= 2) {
                GKOMI_PCG(2, 5, 512);
            } else if (rows_per_thread_512 <= 4) {
                GKOMI_PCG(4, 5, 512);
            } else {
                GKOMI_PCG(8, 5, 512);
            }
        } else if (resident) {
            if (rows_per_thread_512 <= 2) {
                GKOMI_PCG(2, 7, 512);
            } else if (rows_per_thread_512 <= 4) {
                GKOMI_PCG(4, 7, 512);
            } else {
                GKOMI_PCG(8, 7, 512);  // x in LDS: 7 nonzeros x 8 rows of values fill the registers
            }
        } else if (rows_per_thread <= 1) {
            GKOMI_PCG(1, 0, 1024);
        } else if (rows_per_thread <= 2) {
            GKOMI_PCG(2, 0, 1024);
        } else if (rows_per_thread <= 4) {
            GKOMI_PCG(4, 0, 1024);
        } else {
            GKOMI_PCG(8, 0, 1024);
        }
#undef GKOMI_PCG
        PCG_TRY(check_launch());
        pcg_control hctl{};
        PCG_TRY(static_cast<int>(hipMemcpyAsync(&hctl, ctl, sizeof(pcg_control), hipMemcpyDeviceToHost, stream)));
        PCG_TRY(static_cast<int>(hipMemcpyAsync(&polled, scal, sizeof(cg_scalars), hipMemcpyDeviceToHost, stream)));
        PCG_TRY(static_cast<int>(hipStreamSynchronize(stream)));
#ifdef GKOMI_PCG_PROFILE
        fprintf(stderr, "pcg phases (us per iteration, workgroup 0): meet rho %.2f | p + barrier %.2f | inv %.2f | "
                        "spmv %.2f | meet pq %.2f | update %.2f over %lld iterations\n",
                hctl.ticks[0] * 0.01 / (polled.stop_iter + 1), hctl.ticks[1] * 0.01 / (polled.stop_iter + 1),
                hctl.ticks[2] * 0.01 / (polled.stop_iter + 1), hctl.ticks[3] * 0.01 / (polled.stop_iter + 1),
                hctl.ticks[4] * 0.01 / (polled.stop_iter + 1), hctl.ticks[5] * 0.01 / (polled.stop_iter + 1),
                polled.stop_iter);
        fprintf(stderr, "pcg meeting (us, mean over %llu meetings): own slot seen %.2f | last slot seen %.2f | total "
                        "published %.2f | workgroup 101 has the total %.2f\n",
                hctl.pad2_[4], hctl.pad2_[0] * 0.01 / hctl.pad2_[4], hctl.pad2_[1] * 0.01 / hctl.pad2_[4],
                hctl.pad2_[2] * 0.01 / hctl.pad2_[4], hctl.pad2_[3] * 0.01 / hctl.pad2_[4]);
#endif
        if (hctl.overrun == 0 && (polled.status & GKOMI_STATUS_ID_MASK)) {
            pcg_solves.fetch_add(1);
            *result = polled;
            return 1;
        } else {
            // a meeting timed out (workgroups not resident together?): x is a valid guess, start over
            // from r = b - A x with the three-launch iteration
            PCG_TRY(gkomi_dense_copy_f64(s, n, 1, b, 1, r, 1));
            PCG_TRY(A.apply(s, 1, neg_one, x, one, r));
            PCG_TRY(gkomi_dense_fill_f64(s, n, 1, p, 1, 0.0));
            hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(1), 0, stream, scal, orig_tau,
                               baseline == 2 ? 1 : 0);
        }
    }
    return 0;
}
#undef PCG_TRY

int cg_solve_impl(gkomi_stream_t s, int64_t n, int64_t nrhs, const sysmat& A_, gkomi_apply_fn precond,
                  void* precond_ctx, const double* b, double* x, int64_t max_iters,
                  double reduction_factor, int baseline, int mode, int check_every, void* workspace,
                  size_t workspace_bytes, double* host_info)
{
    // what this solve moves between two applies of A decides how A is read (internal.hpp)
    sysmat A = A_;
    A.note_working_set(static_cast<int64_t>(sizeof(double)) * n * nrhs * 6);
    if (mode == 1 && n == 0) mode = 0;  // the fused path needs rows
    // the fused kernels move 16 B per lane through x and the workspace vectors: anything else
    // (a view at an odd offset) takes the reference sequence, like the other fused drivers
    if (mode == 1 && nrhs == 1 &&
        (reinterpret_cast<uintptr_t>(x) % 16 != 0 || reinterpret_cast<uintptr_t>(workspace) % 16 != 0)) {
        mode = 0;
    }
    if (n < 0 || nrhs <= 0 || max_iters < 0) return GKOMI_EINVAL;
    if (baseline < 0 || baseline > 2 || (mode != 0 && mode != 1)) return GKOMI_EINVAL;
    if (mode == 1 && nrhs != 1) return GKOMI_ENOTSUPPORTED;
    if (n > INT32_MAX - 1024) return GKOMI_ENOTSUPPORTED;
    const workspace_layout l = make_layout(n, nrhs);
    if (workspace == nullptr || workspace_bytes < l.total) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* ws = static_cast<char*>(workspace);
    double* r = reinterpret_cast<double*>(ws + l.r);
    double* z = reinterpret_cast<double*>(ws + l.z);
    double* p = reinterpret_cast<double*>(ws + l.p);
    double* q = reinterpret_cast<double*>(ws + l.q);
    void* red = ws + l.red;
    const size_t red_bytes = gkomi_dense_reduction_workspace_bytes(n, nrhs) + 8;
    double* small = reinterpret_cast<double*>(ws + l.small);
    double* prev_rho = small;
    double* rho = small + nrhs;
    double* beta = small + 2 * nrhs;
    double* tau = small + 3 * nrhs;
    double* orig_tau = small + 4 * nrhs;
    double* one = small + 5 * nrhs;
    double* neg_one = small + 6 * nrhs;
    uint8_t* stop_status = reinterpret_cast<uint8_t*>(small + 8 * nrhs);
    uint8_t* dev_flags = stop_status + nrhs + (8 - nrhs % 8) % 8;

    // cg::initialize, then r = b - A x (advanced apply), cg.cpp:137-142
    GKOMI_TRY(gkomi_cg_initialize_f64(s, n, nrhs, b, nrhs, r, nrhs, z, nrhs, p,
                                      nrhs, q, nrhs, prev_rho, rho, stop_status));
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, one, nrhs, 1.0));
    GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, neg_one, nrhs, -1.0));
    GKOMI_TRY(A.apply(s, nrhs, neg_one, x, one, r));
    // criterion generate: baseline norm (residual_norm.cpp:119-189)
    if (baseline == 0) {
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, b, nrhs, orig_tau, red, red_bytes));
    } else if (baseline == 1) {
        GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, orig_tau, red, red_bytes));
    } else {
        GKOMI_TRY(gkomi_dense_fill_f64(s, 1, nrhs, orig_tau, nrhs, 1.0));
    }

    long long iterations = -1;
    int converged = 0;

    if (mode == 0) {
        uint8_t host_flags[2] = {0, 0};
        long long iter = -1;
        while (true) {
            GKOMI_TRY(identity_or_precond(precond, precond_ctx, s, n, nrhs, r, z));
            GKOMI_TRY(gkomi_dense_compute_dot_f64(s, n, nrhs, r, nrhs, z, nrhs, rho, red, red_bytes));
            ++iter;
            bool stop = false;
            if (iter >= max_iters) {
                GKOMI_TRY(gkomi_set_all_statuses(s, nrhs, id_iteration, 1, stop_status));
                stop = true;
            } else {
                GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, tau, red, red_bytes));
                GKOMI_TRY(gkomi_residual_norm_f64(s, nrhs, tau, orig_tau, reduction_factor,
                                                  id_residual, 1, stop_status, dev_flags,
                                                  host_flags));
                stop = host_flags[0] != 0;
                converged = stop ? 1 : 0;
            }
            if (stop) break;
            GKOMI_TRY(gkomi_cg_step_1_f64(s, n, nrhs, p, nrhs, z, nrhs, rho, prev_rho, stop_status));
            GKOMI_TRY(A.apply(s, nrhs, nullptr, p, nullptr, q));
            GKOMI_TRY(gkomi_dense_compute_dot_f64(s, n, nrhs, p, nrhs, q, nrhs, beta, red, red_bytes));
            GKOMI_TRY(gkomi_cg_step_2_f64(s, n, nrhs, x, nrhs, r, nrhs, p, nrhs, q, nrhs, beta, rho,
                                          stop_status));
            std::swap(prev_rho, rho);
        }
        iterations = iter;
        if (host_info != nullptr) {
            // final recurrence residual norms for the report
            GKOMI_TRY(gkomi_dense_compute_norm2_f64(s, n, nrhs, r, nrhs, tau, red, red_bytes));
            for (int64_t j = 0; j < nrhs; ++j) {
                GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 2 + 2 * j, tau + j, sizeof(double),
                                                          hipMemcpyDeviceToHost, stream)));
                GKOMI_TRY(static_cast<int>(hipMemcpyAsync(host_info + 3 + 2 * j, orig_tau + j,
                                                          sizeof(double), hipMemcpyDeviceToHost, stream)));
            }
        }
        GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
    } else {
        cg_scalars* scal = reinterpret_cast<cg_scalars*>(ws + l.scalars);
        double* part_a = reinterpret_cast<double*>(ws + l.part_a);
        double* part_b = reinterpret_cast<double*>(ws + l.part_b);
        double* part_c = reinterpret_cast<double*>(ws + l.part_c);
        const int g = vec_grid(n);
        // q = A p with the p.q partials in the same launch for CSR / ELL /
        // SELL-P; any other operator: its apply, then a partials kernel
        const spmv_dot_plan spmv(A);
        const int nb = spmv.fused() ? spmv.num_partials : g;
        // (misaligned CSR arrays: spmv.fused() is false, apply + partials kernel like any operator)
        cg_scalars polled{};  // per call: concurrent solves on other streams / threads do not share it
        if (check_every < 1) check_every = 1;
        hipLaunchKernelGGL(cg_init_scalars_kernel, dim3(1), dim3(1), 0, stream, scal, orig_tau,
                           baseline == 2 ? 1 : 0);
        GKOMI_TRY(check_launch());
        // The whole solve in one launch when vectors and matrix fit the register files (persistent_cg
        // above); otherwise, or when it gave up, the three-launch iteration below.
        {
            const int done = persistent_cg(s, n, A, spmv, precond, b, x, r, p, q, one, neg_one, orig_tau, baseline,
                                           max_iters, reduction_factor, scal, ws + l.pcg_ctl, ws + l.pcg_slots,
                                           &polled);
            if (done < 0) return -done - 1000;
            if (done == 1) {
                if (host_info != nullptr) {
                    host_info[0] = static_cast<double>(polled.stop_iter);
                    host_info[1] = (polled.status & GKOMI_STATUS_CONVERGED) ? 1.0 : 0.0;
                    host_info[2] = polled.tau;
                    host_info[3] = polled.orig_tau;
                }
                return precond_status(precond, precond_ctx, s);
            }
        }
        // partials of r.z (and r.r) for the first check
        const double* zz = precond == nullptr ? r : z;
        // The library's own block-Jacobi behind the callback: z = M^-1 r leaves the partials of r.z and r.r itself
        // (jacobi_apply_kernel<..., Dot>: every lane holds both factors of its row) -- one launch and a pass over
        // r and z less per iteration (profiles/r03_p3_cg_kernels.md).  `ng` = how many partials K1 re-adds: the
        // apply's workgroups then, the vector grid otherwise.
        const gkomi_jacobi_ctx* jac =
            precond == &gkomi_jacobi_apply_cb ? static_cast<const gkomi_jacobi_ctx*>(precond_ctx) : nullptr;
        int ng = g;
        auto precondition = [&](bool first) -> int {  // z = M^-1 r and the partials of r.z (and, first / fused, r.r)
            if (jac != nullptr) {
                const int got = jacobi_apply_dot_launch(s, jac, r, z, part_a, part_b, spmv_dot_partials_room(n),
                                                        &scal->status);
                if (got < 0) return -got - 1000;
                if (got > 0) {
                    ng = got;
                    return 0;
                }
                jac = nullptr;  // not for this one (scalar Jacobi): the general way from here on
            }
            GKOMI_TRY(precond(precond_ctx, s, r, z));
            hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, z,
                               first ? static_cast<const cg_scalars*>(nullptr) : static_cast<const cg_scalars*>(scal),
                               part_a, first ? part_b : static_cast<double*>(nullptr));
            return check_launch();
        };
        if (precond != nullptr) {
            GKOMI_TRY(precondition(true));
        } else {
            hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, r, zz,
                               static_cast<const cg_scalars*>(nullptr), part_a, static_cast<double*>(nullptr));
            GKOMI_TRY(check_launch());
        }
        const double* tau_part = precond == nullptr ? part_a : part_b;
        long long it = 0;
        bool done = false;
        // The host does not look at device memory while the solve runs: K1's first thread reports the
        // iteration it has evaluated (and the one at which the criterion fired) into pinned host memory
        // (host_watch, internal.hpp), and the host keeps at most min(check_every, host_watch_lag)
        // iterations ahead of what it has seen -- the queue never drains for a look.  Without that
        // line (or if its stores never show up) the old way: a blocking look every check_every iterations.
        host_watch watch;
        // launches issued after the criterion fired return at once -- unless a preconditioner's are among them
        const long long lag = std::min<long long>(check_every, precond == nullptr ? 4 * host_watch_lag : host_watch_lag);
        auto issue = [&](long long i) -> int {
            // (with the Jacobi apply's partials both sums have ng terms; K3's r.r partials, g of them, otherwise)
            hipLaunchKernelGGL(cg_fused_step1_kernel, dim3(g), dim3(fblock), 0, stream, n, p, zz,
                               part_a, ng, tau_part, jac != nullptr ? ng : g, scal, i,
                               static_cast<long long>(max_iters), reduction_factor, watch.dev);
            if (spmv.fused()) {
                GKOMI_TRY(spmv.launch(stream, p, q, part_c, &scal->status));
            } else {
                GKOMI_TRY(A.apply(s, 1, nullptr, p, nullptr, q));
                hipLaunchKernelGGL(cg_dot2_partials_kernel, dim3(g), dim3(fblock), 0, stream, n, p, q,
                                   static_cast<const cg_scalars*>(scal), part_c,
                                   static_cast<double*>(nullptr));
            }
            hipLaunchKernelGGL(cg_fused_step2_kernel, dim3(g), dim3(fblock), 0, stream, n, x, r, p,
                               q, part_c, nb, scal, i, precond == nullptr ? part_a : part_b);
            if (precond != nullptr) GKOMI_TRY(precondition(false));
            return check_launch();
        };
        while (!done) {
            bool look = false;
            if (watch.dev != nullptr) {
                GKOMI_TRY(issue(it));
                const bool last = it >= max_iters;  // the launch with it == max_iters stops for sure
                ++it;
                if (last) {
                    look = true;
                } else if (it - 1 >= lag) {
                    if (!watch.wait(stream, it - 1 - lag)) {
                        watch.dev = nullptr;
                        look = true;
                    } else {
                        look = watch.stop_iter() >= 0;
                    }
                }
            } else {
                for (int c = 0; c < check_every; ++c, ++it) {
                    GKOMI_TRY(issue(it));
                    if (it >= max_iters) {
                        ++it;
                        break;
                    }
                }
                look = true;
            }
            if (!look) continue;
            GKOMI_TRY(static_cast<int>(hipMemcpyAsync(&polled, scal, sizeof(cg_scalars),
                                                      hipMemcpyDeviceToHost, stream)));
            GKOMI_TRY(static_cast<int>(hipStreamSynchronize(stream)));
            const cg_scalars* h = &polled;
            if (h->status & GKOMI_STATUS_ID_MASK) {
                done = true;
                iterations = h->stop_iter;
                converged = (h->status & GKOMI_STATUS_CONVERGED) ? 1 : 0;
                if (host_info != nullptr) {
                    host_info[2] = h->tau;
                    host_info[3] = h->orig_tau;
                }
            }
        }
    }
    if (host_info != nullptr) {
        host_info[0] = static_cast<double>(iterations);
        host_info[1] = static_cast<double>(converged);
    }
    return precond_status(precond, precond_ctx, s);
}
}  // namespace

extern "C" int gkomi_cg_solve_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz,
    const int32_t* row_ptrs, const int32_t* col_idxs, const double* vals,
    int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond,
    void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int mode, int check_every,
    void* workspace, size_t workspace_bytes, double* host_info)
{
    return cg_solve_impl(s, n, nrhs,
                         make_csr_sysmat(n, nnz, row_ptrs, col_idxs, vals, spmv_strategy, max_row_nnz_hint),
                         precond, precond_ctx, b, x, max_iters, reduction_factor, baseline, mode,
                         check_every, workspace, workspace_bytes, host_info);
}

// the system matrix behind a callback, fused (single rhs): three launches per
// iteration for ELL / SELL-P / CSR behind the library's callbacks (SpMV + dot
// epilogue), apply + 3 for any other operator
extern "C" int gkomi_cg_solve_fused_op_f64(gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix,
                                           void* matrix_ctx, gkomi_apply_fn precond, void* precond_ctx,
                                           const double* b, double* x, int64_t max_iters,
                                           double reduction_factor, int baseline, int64_t check_every,
                                           void* workspace, size_t workspace_bytes, double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return cg_solve_impl(s, n, 1, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                         max_iters, reduction_factor, baseline, 1,
                         static_cast<int>(check_every < 1 ? 1 : (check_every > 1 << 20 ? 1 << 20 : check_every)),
                         workspace, workspace_bytes, host_info);
}

// the system matrix behind a callback: the reference kernel sequence (mode 0)
extern "C" int gkomi_cg_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                     gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                     gkomi_apply_fn precond, void* precond_ctx, const double* b,
                                     double* x, int64_t max_iters, double reduction_factor,
                                     int baseline, void* workspace, size_t workspace_bytes,
                                     double* host_info)
{
    if (matrix == nullptr) return GKOMI_EINVAL;
    return cg_solve_impl(s, n, nrhs, make_op_sysmat(n, matrix, matrix_ctx), precond, precond_ctx, b, x,
                         max_iters, reduction_factor, baseline, 0, 1, workspace, workspace_bytes,
                         host_info);
}
