// hip/matrix/coo_kernels.hip.cpp: coo::spmv / advanced_spmv / spmv2 / advanced_spmv2
// (core/matrix/coo_kernels.hpp).  These are the any-order entries (one fp64 atomic per row segment, like the
// reference's HIP kernels); a Coo whose rows are known to be sorted takes gkomi_coo_spmv*_sorted_f64_i32 --
// the reference's Coo keeps no such statistic, a maintainer would cache it next to the arrays like Csr's srow.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace coo {

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Coo<double, int32>* a, const matrix::Dense<double>* b,
          matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_coo_spmv_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
                                      a->get_const_row_idxs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
                                      b->get_stride(), c->get_values(), c->get_stride(), nullptr, nullptr));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, const matrix::Coo<double, int32>* a,
                   const matrix::Dense<double>* b, const matrix::Dense<double>* beta, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_coo_spmv_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
                                      a->get_const_row_idxs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
                                      b->get_stride(), c->get_values(), c->get_stride(), alpha->get_const_values(),
                                      beta->get_const_values()));
}

void spmv2(std::shared_ptr<const HipExecutor> exec, const matrix::Coo<double, int32>* a, const matrix::Dense<double>* b,
           matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_coo_spmv2_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
                                       a->get_const_row_idxs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
                                       b->get_stride(), c->get_values(), c->get_stride(), nullptr));
}

void advanced_spmv2(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, const matrix::Coo<double, int32>* a,
                    const matrix::Dense<double>* b, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_coo_spmv2_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
                                       a->get_const_row_idxs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
                                       b->get_stride(), c->get_values(), c->get_stride(), alpha->get_const_values()));
}

}  // namespace coo
}  // namespace hip
}  // namespace kernels
}  // namespace gko
