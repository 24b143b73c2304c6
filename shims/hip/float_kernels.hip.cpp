// The <float, int32> instantiations of the kernels libgkomi.so provides in single precision
// (GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE, include/ginkgo/core/base/types.hpp:544-560): csr::spmv / advanced_spmv,
// the dense BLAS-1 kernels, cg::initialize / step_1 / step_2, residual_norm.  In a reference tree these are the same
// function templates as the double ones, instantiated by the macros at the end of each hip/**/*_kernels.hip.cpp; here
// they are written out as overloads next to the double ones (same parameter lists, float in the value positions).
#include "gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {

namespace csr {

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<float, int32>* a, const matrix::Dense<float>* b,
          matrix::Dense<float>* c)
{
    GKOMI_CALL(gkomi_csr_spmv_f32_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
                                      a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
                                      b->get_stride(), c->get_values(), c->get_stride(), nullptr, nullptr));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* alpha, const matrix::Csr<float, int32>* a,
                   const matrix::Dense<float>* b, const matrix::Dense<float>* beta, matrix::Dense<float>* c)
{
    GKOMI_CALL(gkomi_csr_spmv_f32_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
                                      a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
                                      b->get_stride(), c->get_values(), c->get_stride(), alpha->get_const_values(),
                                      beta->get_const_values()));
}

}  // namespace csr

namespace dense {

inline void ensure_f32(array<char>& tmp, size_type need)
{
    if (tmp.get_num_elems() < need) tmp.resize_and_reset(need);
}

void fill(std::shared_ptr<const HipExecutor> exec, matrix::Dense<float>* x, float value)
{
    GKOMI_CALL(gkomi_dense_fill_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_values(), x->get_stride(), value));
}

void copy(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* in, matrix::Dense<float>* out)
{
    GKOMI_CALL(gkomi_dense_copy_f32(GKOMI_NULL_STREAM, in->get_size()[0], in->get_size()[1], in->get_const_values(), in->get_stride(),
                                    out->get_values(), out->get_stride()));
}

void scale(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* alpha, matrix::Dense<float>* x)
{
    GKOMI_CALL(gkomi_dense_scale_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(), alpha->get_size()[1],
                                     x->get_values(), x->get_stride()));
}

void inv_scale(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* alpha, matrix::Dense<float>* x)
{
    GKOMI_CALL(gkomi_dense_inv_scale_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(),
                                         alpha->get_size()[1], x->get_values(), x->get_stride()));
}

void add_scaled(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* alpha, const matrix::Dense<float>* x,
                matrix::Dense<float>* y)
{
    GKOMI_CALL(gkomi_dense_add_scaled_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(), alpha->get_size()[1],
                                          x->get_const_values(), x->get_stride(), y->get_values(), y->get_stride()));
}

void sub_scaled(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* alpha, const matrix::Dense<float>* x,
                matrix::Dense<float>* y)
{
    GKOMI_CALL(gkomi_dense_sub_scaled_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(), alpha->get_size()[1],
                                          x->get_const_values(), x->get_stride(), y->get_values(), y->get_stride()));
}

void compute_dot(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* x, const matrix::Dense<float>* y,
                 matrix::Dense<float>* result, array<char>& tmp)
{
    ensure_f32(tmp, gkomi_dense_reduction_workspace_bytes_f32(x->get_size()[0], x->get_size()[1]));
    GKOMI_CALL(gkomi_dense_compute_dot_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_const_values(), x->get_stride(),
                                           y->get_const_values(), y->get_stride(), result->get_values(), tmp.get_data(), tmp.get_num_elems()));
}

void compute_norm2(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* x, matrix::Dense<float>* result, array<char>& tmp)
{
    ensure_f32(tmp, gkomi_dense_reduction_workspace_bytes_f32(x->get_size()[0], x->get_size()[1]));
    GKOMI_CALL(gkomi_dense_compute_norm2_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_const_values(), x->get_stride(),
                                             result->get_values(), tmp.get_data(), tmp.get_num_elems()));
}

}  // namespace dense

namespace cg {

inline uint8_t* raw_f32(array<stopping_status>* s) { return reinterpret_cast<uint8_t*>(s->get_data()); }
inline const uint8_t* raw_f32(const array<stopping_status>* s) { return reinterpret_cast<const uint8_t*>(s->get_const_data()); }

void initialize(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* b, matrix::Dense<float>* r, matrix::Dense<float>* z,
                matrix::Dense<float>* p, matrix::Dense<float>* q, matrix::Dense<float>* prev_rho, matrix::Dense<float>* rho,
                array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cg_initialize_f32(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], b->get_const_values(), b->get_stride(),
                                       r->get_values(), r->get_stride(), z->get_values(), z->get_stride(), p->get_values(), p->get_stride(),
                                       q->get_values(), q->get_stride(), prev_rho->get_values(), rho->get_values(), raw_f32(stop_status)));
}

void step_1(std::shared_ptr<const HipExecutor> exec, matrix::Dense<float>* p, const matrix::Dense<float>* z, const matrix::Dense<float>* rho,
            const matrix::Dense<float>* prev_rho, const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cg_step_1_f32(GKOMI_NULL_STREAM, p->get_size()[0], p->get_size()[1], p->get_values(), p->get_stride(),
                                   z->get_const_values(), z->get_stride(), rho->get_const_values(), prev_rho->get_const_values(),
                                   raw_f32(stop_status)));
}

void step_2(std::shared_ptr<const HipExecutor> exec, matrix::Dense<float>* x, matrix::Dense<float>* r, const matrix::Dense<float>* p,
            const matrix::Dense<float>* q, const matrix::Dense<float>* beta, const matrix::Dense<float>* rho,
            const array<stopping_status>* stop_status)
{
    GKOMI_CALL(gkomi_cg_step_2_f32(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_values(), x->get_stride(), r->get_values(),
                                   r->get_stride(), p->get_const_values(), p->get_stride(), q->get_const_values(), q->get_stride(),
                                   beta->get_const_values(), rho->get_const_values(), raw_f32(stop_status)));
}

}  // namespace cg

namespace residual_norm {

void residual_norm(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<float>* tau, const matrix::Dense<float>* orig_tau,
                   float rel_residual_goal, uint8 stoppingId, bool setFinalized, array<stopping_status>* stop_status,
                   array<bool>* device_storage, bool* all_converged, bool* one_changed)
{
    uint8_t host[2] = {0, 0};
    GKOMI_CALL(gkomi_residual_norm_f32(GKOMI_NULL_STREAM, tau->get_size()[1], tau->get_const_values(), orig_tau->get_const_values(),
                                       rel_residual_goal, stoppingId, setFinalized, reinterpret_cast<uint8_t*>(stop_status->get_data()),
                                       reinterpret_cast<uint8_t*>(device_storage->get_data()), host));
    *all_converged = host[0] != 0;
    *one_changed = host[1] != 0;
}

}  // namespace residual_norm

}  // namespace hip
}  // namespace kernels
}  // namespace gko
