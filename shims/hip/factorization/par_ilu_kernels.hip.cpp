// hip/factorization/par_ilu_kernels.hip.cpp + factorization_kernels.hip.cpp:
// par_ilu_factorization::compute_l_u_factors (core/factorization/par_ilu_kernels.hpp:54) and
// factorization::{add_diagonal_elements, initialize_row_ptrs_l_u, initialize_l_u, initialize_row_ptrs_l,
// initialize_l} (core/factorization/factorization_kernels.hpp), the chain core/factorization/par_ilu.cpp:74-163 runs.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace factorization {

void add_diagonal_elements(std::shared_ptr<const HipExecutor> exec, matrix::Csr<double, int32>* mtx, bool is_sorted)
{
    // two phases like the reference's kernels: count the rows without a diagonal entry (blocking: the arrays
    // are resized in between), then rebuild row_ptrs in place and scatter into the new arrays
    const int64_t nrows = static_cast<int64_t>(mtx->get_size()[0]), ncols = static_cast<int64_t>(mtx->get_size()[1]);
    array<char> tmp(exec, gkomi_factorization_workspace_bytes(nrows));
    int64_t missing = 0;
    GKOMI_CALL(gkomi_factorization_count_missing_diagonal_i32(GKOMI_NULL_STREAM, nrows, ncols, mtx->get_const_row_ptrs(),
                                                              mtx->get_const_col_idxs(), tmp.get_data(), tmp.get_num_elems(), &missing));
    if (missing == 0) return;
    const size_type new_nnz = mtx->get_num_stored_elements() + static_cast<size_type>(missing);
    array<int32> new_cols(exec, new_nnz);
    array<double> new_vals(exec, new_nnz);
    GKOMI_CALL(gkomi_factorization_add_diagonal_elements_f64_i32(GKOMI_NULL_STREAM, nrows, ncols, mtx->get_row_ptrs(),
                                                                 mtx->get_const_col_idxs(), mtx->get_const_values(), new_cols.get_data(),
                                                                 new_vals.get_data(), tmp.get_data()));
    // the reference swaps the new arrays in through matrix::CsrBuilder (core/matrix/csr_builder.hpp)
    matrix::CsrBuilder<double, int32> builder{mtx};
    builder.get_col_idx_array() = std::move(new_cols);
    builder.get_value_array() = std::move(new_vals);
}

void initialize_row_ptrs_l_u(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* system_matrix,
                             int32* l_row_ptrs, int32* u_row_ptrs)
{
    const int64_t n = static_cast<int64_t>(system_matrix->get_size()[0]);
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(n + 1));
    GKOMI_CALL(gkomi_factorization_initialize_row_ptrs_l_u_i32(GKOMI_NULL_STREAM, n, system_matrix->get_const_row_ptrs(),
                                                               system_matrix->get_const_col_idxs(), l_row_ptrs, u_row_ptrs, tmp.get_data(),
                                                               tmp.get_num_elems()));
}

void initialize_l_u(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* system_matrix,
                    matrix::Csr<double, int32>* l_factor, matrix::Csr<double, int32>* u_factor)
{
    GKOMI_CALL(gkomi_factorization_initialize_l_u_f64_i32(
        GKOMI_NULL_STREAM, static_cast<int64_t>(system_matrix->get_size()[0]), system_matrix->get_const_row_ptrs(),
        system_matrix->get_const_col_idxs(), system_matrix->get_const_values(), l_factor->get_const_row_ptrs(), l_factor->get_col_idxs(),
        l_factor->get_values(), u_factor->get_const_row_ptrs(), u_factor->get_col_idxs(), u_factor->get_values()));
}

void initialize_row_ptrs_l(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* system_matrix, int32* l_row_ptrs)
{
    const int64_t n = static_cast<int64_t>(system_matrix->get_size()[0]);
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(n + 1));
    GKOMI_CALL(gkomi_factorization_initialize_row_ptrs_l_i32(GKOMI_NULL_STREAM, n, system_matrix->get_const_row_ptrs(),
                                                             system_matrix->get_const_col_idxs(), l_row_ptrs, tmp.get_data(), tmp.get_num_elems()));
}

void initialize_l(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* system_matrix,
                  matrix::Csr<double, int32>* l_factor, bool diag_sqrt)
{
    GKOMI_CALL(gkomi_factorization_initialize_l_f64_i32(GKOMI_NULL_STREAM, static_cast<int64_t>(system_matrix->get_size()[0]),
                                                        system_matrix->get_const_row_ptrs(), system_matrix->get_const_col_idxs(),
                                                        system_matrix->get_const_values(), l_factor->get_const_row_ptrs(),
                                                        l_factor->get_col_idxs(), l_factor->get_values(), diag_sqrt ? 1 : 0));
}

}  // namespace factorization

namespace par_ilu_factorization {

void compute_l_u_factors(std::shared_ptr<const HipExecutor> exec, size_type iterations, const matrix::Coo<double, int32>* system_matrix,
                         matrix::Csr<double, int32>* l_factor, matrix::Csr<double, int32>* u_factor)
{
    // u_factor arrives TRANSPOSED (CSC of U as CSR, core/factorization/par_ilu.cpp:128-150); iterations == 0 = the
    // kernel's default number of sweeps
    GKOMI_CALL(gkomi_par_ilu_compute_l_u_factors_f64_i32(
        GKOMI_NULL_STREAM, static_cast<int64_t>(iterations), static_cast<int64_t>(system_matrix->get_num_stored_elements()),
        system_matrix->get_const_row_idxs(), system_matrix->get_const_col_idxs(), system_matrix->get_const_values(),
        l_factor->get_const_row_ptrs(), l_factor->get_const_col_idxs(), l_factor->get_values(), u_factor->get_const_row_ptrs(),
        u_factor->get_const_col_idxs(), u_factor->get_values()));
}

}  // namespace par_ilu_factorization
}  // namespace hip
}  // namespace kernels
}  // namespace gko
