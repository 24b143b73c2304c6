// hip/solver/cg_kernels.hip.cpp (the HIP instantiation of common/unified/solver/cg_kernels.cpp):
// cg::initialize / step_1 / step_2 (core/solver/cg_kernels.hpp:54-80).
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace cg {

// stopping_status is one uint8 (stopping_status.hpp:144-147)
inline uint8_t* raw(array<stopping_status>* s) { return reinterpret_cast<uint8_t*>(s->get_data()); }
inline const uint8_t* raw(const array<stopping_status>* s) { return reinterpret_cast<const uint8_t*>(s->get_const_data()); }

void initialize(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* b, matrix::Dense<double>* r,
                matrix::Dense<double>* z, matrix::Dense<double>* p, matrix::Dense<double>* q, matrix::Dense<double>* prev_rho,
                matrix::Dense<double>* rho, array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cg_initialize_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], b->get_const_values(), b->get_stride(),
                                       r->get_values(), r->get_stride(), z->get_values(), z->get_stride(), p->get_values(), p->get_stride(),
                                       q->get_values(), q->get_stride(), prev_rho->get_values(), rho->get_values(), raw(stop_status)));
}

void step_1(std::shared_ptr<const HipExecutor> exec, matrix::Dense<double>* p, const matrix::Dense<double>* z,
            const matrix::Dense<double>* rho, const matrix::Dense<double>* prev_rho, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cg_step_1_f64(GKOMI_NULL_STREAM, p->get_size()[0], p->get_size()[1], p->get_values(), p->get_stride(),
                                   z->get_const_values(), z->get_stride(), rho->get_const_values(), prev_rho->get_const_values(),
                                   raw(stop_status)));
}

void step_2(std::shared_ptr<const HipExecutor> exec, matrix::Dense<double>* x, matrix::Dense<double>* r, const matrix::Dense<double>* p,
            const matrix::Dense<double>* q, const matrix::Dense<double>* beta, const matrix::Dense<double>* rho,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cg_step_2_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_values(), x->get_stride(), r->get_values(),
                                   r->get_stride(), p->get_const_values(), p->get_stride(), q->get_const_values(), q->get_stride(),
                                   beta->get_const_values(), rho->get_const_values(), raw(stop_status)));
}

}  // namespace cg
}  // namespace hip
}  // namespace kernels
}  // namespace gko
