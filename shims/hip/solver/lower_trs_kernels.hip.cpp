// hip/solver/lower_trs_kernels.hip.cpp (upper_trs: the same with lower = 0):
// should_perform_transpose / generate / solve (core/solver/lower_trs_kernels.hpp).
// generate = the dependency-level analysis (the hipsparseXcsrsv2_analysis of
// hip/solver/common_trs_kernels.hip.hpp:61-253), kept in the SolveStruct; solve = the
// brick plan for factors of grid problems (csrc/trs_bricks.hip), else the level-scheduled kernel,
// or the analysis-free one for factors whose levels are narrow.
#include "../gkomi_bindings.hpp"
#include <cstdlib>

namespace gko {
namespace kernels {
namespace hip {
namespace lower_trs {

struct gkomi_solve_struct : solver::SolveStruct {
    array<char> symbolic, plan, workspace;
    int64_t nslices{0}, entries{0}, nlevels{0}, max_deps{-1};
    bool planned{false};
    unsigned long long solves{0};  // since generate: when the sticky give-up flag is looked at
    gkomi_trs_bricks* bricks{nullptr};  // host side of the brick plan; its device plan is `plan`
    explicit gkomi_solve_struct(std::shared_ptr<const Executor> exec) : symbolic(exec), plan(exec), workspace(exec) {}
    ~gkomi_solve_struct() { gkomi_trs_bricks_destroy(bricks); }
};

void should_perform_transpose(std::shared_ptr<const HipExecutor> exec, bool& do_transpose) { do_transpose = false; }

void generate(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* matrix,
              std::shared_ptr<solver::SolveStruct>& solve_struct, bool unit_diag, const solver::trisolve_algorithm algorithm,
              const size_type num_rhs)
{
    auto st = std::make_shared<gkomi_solve_struct>(exec);
    const int64_t n = static_cast<int64_t>(matrix->get_size()[0]);
    st->workspace.resize_and_reset(gkomi_trs_workspace_bytes());
    st->workspace.fill(0);  // the analysis-free kernel's ticket and its sticky give-up flag
    // many levels on a box grid: bricks solved out of LDS (one LDS step per level, a memory hand-off per brick level).
    // The brick analysis runs on the device and estimates the factor's levels from the box geometry it found, so a
    // factor that takes the brick plan skips the level analysis altogether.
    if (n >= 2) {
        const int err = gkomi_trs_bricks_create_i32(GKOMI_NULL_STREAM, n, matrix->get_const_row_ptrs(), matrix->get_const_col_idxs(), /*lower=*/1,
                                                    0, 0, 0, &st->bricks);
        if (err != GKOMI_SUCCESS && err != GKOMI_ENOTSUPPORTED) GKOMI_CALL(err);
        int64_t info[8] = {};
        if (st->bricks != nullptr) GKOMI_CALL(gkomi_trs_bricks_info(st->bricks, info));
        const int64_t levels = st->bricks != nullptr ? gkomi_trs_bricks_levels_estimate(st->bricks) : 0;
        if (st->bricks != nullptr && gkomi_trs_prefer_bricks(n, levels, info[1]) != 0) {
            st->nlevels = levels;
            st->plan.resize_and_reset(gkomi_trs_bricks_plan_bytes(st->bricks));
            GKOMI_CALL(gkomi_trs_bricks_numeric_f64_i32(GKOMI_NULL_STREAM, st->bricks, matrix->get_const_row_ptrs(), matrix->get_const_col_idxs(),
                                                        matrix->get_const_values(), st->plan.get_data(), st->plan.get_num_elems()));
            solve_struct = st;
            return;
        }
        gkomi_trs_bricks_destroy(st->bricks);
        st->bricks = nullptr;
    }
    st->symbolic.resize_and_reset(gkomi_trs_symbolic_workspace_bytes(n));
    int64_t out[4] = {};
    GKOMI_CALL(gkomi_trs_analyse_symbolic_i32(GKOMI_NULL_STREAM, n, matrix->get_const_row_ptrs(), matrix->get_const_col_idxs(), /*lower=*/1,
                                              st->symbolic.get_data(), st->symbolic.get_num_elems(), out));
    st->nslices = out[0]; st->entries = out[1]; st->nlevels = out[2]; st->max_deps = out[3];
    // wide levels: the level-scheduled solve; small factors: one workgroup with x in LDS; chains and narrow bands of large
    // factors: the in-workgroup hand-offs of the other kernel (gkomi_trs_use_plan)
    st->planned = gkomi_trs_use_plan(n, st->nlevels, st->max_deps) != 0;
    if (st->planned) {
        st->plan.resize_and_reset(gkomi_trs_plan_bytes(st->nslices, st->entries));
        GKOMI_CALL(gkomi_trs_analyse_numeric_f64_i32(GKOMI_NULL_STREAM, n, matrix->get_const_row_ptrs(), matrix->get_const_col_idxs(),
                                                     matrix->get_const_values(), 1, st->symbolic.get_const_data(), st->nslices, st->entries,
                                                     st->nlevels, st->plan.get_data(), st->plan.get_num_elems()));
    }
    solve_struct = st;
}

void solve(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* matrix, const solver::SolveStruct* solve_struct,
           bool unit_diag, const solver::trisolve_algorithm algorithm, matrix::Dense<double>* trans_b, matrix::Dense<double>* trans_x,
           const matrix::Dense<double>* b, matrix::Dense<double>* x)
{
    auto st = const_cast<gkomi_solve_struct*>(dynamic_cast<const gkomi_solve_struct*>(solve_struct));
    if (st == nullptr) GKO_NOT_SUPPORTED("solve needs the SolveStruct of this backend's generate()");
    const int64_t n = static_cast<int64_t>(matrix->get_size()[0]);
    if (st->bricks != nullptr) {
        GKOMI_CALL(gkomi_trs_bricks_solve_f64(GKOMI_NULL_STREAM, st->bricks, st->plan.get_data(), b->get_size()[1], unit_diag, b->get_const_values(),
                                              b->get_stride(), x->get_values(), x->get_stride()));
    } else if (st->planned) {
        GKOMI_CALL(gkomi_trs_solve_plan_f64(GKOMI_NULL_STREAM, n, b->get_size()[1], st->plan.get_data(), st->nslices, st->entries, st->max_deps,
                                            unit_diag, b->get_const_values(), b->get_stride(), x->get_values(), x->get_stride()));
    } else {
        GKOMI_CALL(gkomi_lower_trs_solve_f64_i32(GKOMI_NULL_STREAM, n, b->get_size()[1], matrix->get_const_row_ptrs(), matrix->get_const_col_idxs(),
                                                 matrix->get_const_values(), unit_diag, b->get_const_values(), b->get_stride(), x->get_values(),
                                                 x->get_stride(), st->workspace.get_data(), st->workspace.get_num_elems()));
    }
    // A solve whose bounded waits ran out leaves NaNs in x and a STICKY flag (the reference's nan_produced
    // guard, cuda/solver/common_trs_kernels.cuh:444-449, which is checked on every solve).  Behind this
    // binding the caller is the reference's core/, not one of the native drivers (those look at the flag at
    // the end of a solve): the flag is read after EVERY solve -- a blocking 4-byte copy, as the reference's own
    // guard is.  GKOMI_TRS_CHECK_EVERY=N reads it after the first solve and then every Nth (sticky: a later
    // look still reports an earlier give-up).
    static const unsigned long long every = [] {
        const char* e = std::getenv("GKOMI_TRS_CHECK_EVERY");
        return e != nullptr && std::atoll(e) > 0 ? static_cast<unsigned long long>(std::atoll(e)) : 1ull;
    }();
    if (st->solves++ % every == 0) {
        int gave_up = 0;
        if (st->bricks != nullptr) {
            GKOMI_CALL(gkomi_trs_bricks_check_overrun(GKOMI_NULL_STREAM, st->plan.get_const_data(), &gave_up));
        } else if (st->planned) {
            GKOMI_CALL(gkomi_trs_plan_check_overrun(GKOMI_NULL_STREAM, st->plan.get_const_data(), &gave_up));
        } else {
            GKOMI_CALL(gkomi_trs_check_overrun(GKOMI_NULL_STREAM, st->workspace.get_const_data(), &gave_up));
        }
        if (gave_up != 0) GKOMI_CALL(GKOMI_ETRS_OVERRUN);
    }
}

}  // namespace lower_trs
}  // namespace hip
}  // namespace kernels
}  // namespace gko
