// hip/matrix/ell_kernels.hip.cpp: ell::spmv / advanced_spmv (core/matrix/ell_kernels.hpp, the
// <double, double, double, int32> instantiation of the mixed-precision signature).
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace ell {

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Ell<double, int32>* a, const matrix::Dense<double>* b,
          matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_ell_spmv_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1],
                                      a->get_num_stored_elements_per_row(), a->get_stride(), a->get_const_col_idxs(),
                                      a->get_const_values(), b->get_const_values(), b->get_stride(), c->get_values(), c->get_stride(),
                                      nullptr, nullptr));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, const matrix::Ell<double, int32>* a,
                   const matrix::Dense<double>* b, const matrix::Dense<double>* beta, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_ell_spmv_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1],
                                      a->get_num_stored_elements_per_row(), a->get_stride(), a->get_const_col_idxs(),
                                      a->get_const_values(), b->get_const_values(), b->get_stride(), c->get_values(), c->get_stride(),
                                      alpha->get_const_values(), beta->get_const_values()));
}

}  // namespace ell
}  // namespace hip
}  // namespace kernels
}  // namespace gko
