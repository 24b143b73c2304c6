// hip/matrix/dense_kernels.hip.cpp: the BLAS-1 kernels of the solver steps
// (core/matrix/dense_kernels.hpp; reference/matrix/dense_kernels.cpp:127-447).
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace dense {

inline void ensure(array<char>& tmp, size_type need)
{
    if (tmp.get_num_elems() < need) tmp.resize_and_reset(need);  // the caller-cached scratch (solver_boilerplate.hpp:61-66)
}

void fill(std::shared_ptr<const HipExecutor> exec, matrix::Dense<double>* x, double value)
{
    GKOMI_CALL(gkomi_dense_fill_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_values(), x->get_stride(), value));
}

void copy(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* in, matrix::Dense<double>* out)
{
    GKOMI_CALL(gkomi_dense_copy_f64(GKOMI_NULL_STREAM, in->get_size()[0], in->get_size()[1], in->get_const_values(), in->get_stride(),
                                    out->get_values(), out->get_stride()));
}

void scale(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, matrix::Dense<double>* x)
{
    GKOMI_CALL(gkomi_dense_scale_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(), alpha->get_size()[1],
                                     x->get_values(), x->get_stride()));
}

void add_scaled(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, const matrix::Dense<double>* x,
                matrix::Dense<double>* y)
{
    GKOMI_CALL(gkomi_dense_add_scaled_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(), alpha->get_size()[1],
                                          x->get_const_values(), x->get_stride(), y->get_values(), y->get_stride()));
}

void sub_scaled(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, const matrix::Dense<double>* x,
                matrix::Dense<double>* y)
{
    GKOMI_CALL(gkomi_dense_sub_scaled_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(), alpha->get_size()[1],
                                          x->get_const_values(), x->get_stride(), y->get_values(), y->get_stride()));
}

void compute_dot(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* x, const matrix::Dense<double>* y,
                 matrix::Dense<double>* result, array<char>& tmp)
{
    ensure(tmp, gkomi_dense_reduction_workspace_bytes(x->get_size()[0], x->get_size()[1]));
    GKOMI_CALL(gkomi_dense_compute_dot_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_const_values(), x->get_stride(),
                                           y->get_const_values(), y->get_stride(), result->get_values(), tmp.get_data(), tmp.get_num_elems()));
}

void compute_conj_dot(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* x, const matrix::Dense<double>* y,
                      matrix::Dense<double>* result, array<char>& tmp)
{
    compute_dot(exec, x, y, result, tmp);  // real values
}

void compute_norm2(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* x, matrix::Dense<double>* result,
                   array<char>& tmp)
{
    ensure(tmp, gkomi_dense_reduction_workspace_bytes(x->get_size()[0], x->get_size()[1]));
    GKOMI_CALL(gkomi_dense_compute_norm2_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_const_values(), x->get_stride(),
                                             result->get_values(), tmp.get_data(), tmp.get_num_elems()));
}

void inv_scale(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, matrix::Dense<double>* x)
{
    GKOMI_CALL(gkomi_dense_inv_scale_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], alpha->get_const_values(),
                                         alpha->get_size()[1], x->get_values(), x->get_stride()));
}

// the *_dispatch entries choose between a vendor BLAS and the generic reduction in the reference
// (hip/matrix/dense_kernels.hip.cpp): here they are the reduction itself
void compute_dot_dispatch(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* x, const matrix::Dense<double>* y,
                          matrix::Dense<double>* result, array<char>& tmp)
{
    compute_dot(exec, x, y, result, tmp);
}

void compute_norm2_dispatch(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* x, matrix::Dense<double>* result,
                            array<char>& tmp)
{
    compute_norm2(exec, x, result, tmp);
}

void compute_squared_norm2(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* x, matrix::Dense<double>* result,
                           array<char>& tmp)
{
    const auto need = gkomi_dense_reduction_workspace_bytes(x->get_size()[0], x->get_size()[1]);
    if (tmp.get_num_elems() < need) tmp.resize_and_reset(need);
    GKOMI_CALL(gkomi_dense_compute_squared_norm2_f64(GKOMI_NULL_STREAM, x->get_size()[0], x->get_size()[1], x->get_const_values(),
                                                     x->get_stride(), result->get_values(), tmp.get_data(), tmp.get_num_elems()));
}

void compute_sqrt(std::shared_ptr<const HipExecutor> exec, matrix::Dense<double>* data)
{
    GKOMI_CALL(gkomi_dense_compute_sqrt_f64(GKOMI_NULL_STREAM, data->get_size()[0], data->get_size()[1], data->get_values(),
                                            data->get_stride()));
}

// the halo pack of distributed::Matrix::apply (core/distributed/matrix.cpp:263-303)
void row_gather(std::shared_ptr<const HipExecutor> exec, const array<int32>* gather_indices, const matrix::Dense<double>* orig,
                matrix::Dense<double>* row_collection)
{
    GKOMI_CALL(gkomi_dense_row_gather_f64_i32(GKOMI_NULL_STREAM, static_cast<int64_t>(gather_indices->get_num_elems()),
                                              orig->get_size()[1], gather_indices->get_const_data(), orig->get_const_values(),
                                              orig->get_stride(), row_collection->get_values(), row_collection->get_stride()));
}

}  // namespace dense
}  // namespace hip
}  // namespace kernels
}  // namespace gko
