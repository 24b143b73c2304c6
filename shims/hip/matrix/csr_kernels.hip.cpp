// hip/matrix/csr_kernels.hip.cpp of a reference tree that binds libgkomi.so:
// csr::spmv / csr::advanced_spmv (core/matrix/csr_kernels.hpp:58-75).
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace csr {

// strategy objects keep their API (csr.hpp:170-705); their name selects the kernel
inline int strategy_code(const matrix::Csr<double, int32>* a)
{
    const auto name = a->get_strategy()->get_name();
    return name == "classical"      ? GKOMI_CSR_VECTOR
           : name == "load_balance" ? GKOMI_CSR_BALANCED
           : name == "merge_path"   ? GKOMI_CSR_STREAM
                                    : GKOMI_CSR_AUTO;  // automatical, sparselib, cusparse
}

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* a,
          const matrix::Dense<double>* b, matrix::Dense<double>* c)
{
    // Csr::srow_ (csr.hpp:1265-1266) is the tile start-row array of the nonzero-split kernel when
    // the matrix was built with GKOMI's make_srow (gkomi_csr_make_srow_i32 in Csr::make_srow);
    // pass NULL / 0 to keep the reference's own srow contents untouched.
    GKOMI_CALL(gkomi_csr_spmv_srow_f64_i32(
        GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
        a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
        b->get_stride(), c->get_values(), c->get_stride(), nullptr, nullptr, strategy_code(a),
        /*max_row_nnz_hint=*/-1, a->get_num_srow_elements() ? a->get_const_srow() : nullptr, gkomi_csr_srow_tile_for(static_cast<int64_t>(a->get_num_stored_elements()))));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha,
                   const matrix::Csr<double, int32>* a, const matrix::Dense<double>* b,
                   const matrix::Dense<double>* beta, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_csr_spmv_srow_f64_i32(
        GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
        a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
        b->get_stride(), c->get_values(), c->get_stride(), alpha->get_const_values(), beta->get_const_values(),
        strategy_code(a), -1, a->get_num_srow_elements() ? a->get_const_srow() : nullptr, gkomi_csr_srow_tile_for(static_cast<int64_t>(a->get_num_stored_elements()))));
}

}  // namespace csr
}  // namespace hip
}  // namespace kernels
}  // namespace gko
