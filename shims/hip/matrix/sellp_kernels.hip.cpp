// hip/matrix/sellp_kernels.hip.cpp: sellp::spmv / advanced_spmv / compute_slice_sets
// (core/matrix/sellp_kernels.hpp).  slice_sets / slice_lengths are size_type (64-bit) arrays.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace sellp {

static_assert(sizeof(size_type) == sizeof(uint64_t), "slice sets are 64-bit");

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Sellp<double, int32>* a, const matrix::Dense<double>* b,
          matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_sellp_spmv_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_slice_size(),
                                        reinterpret_cast<const uint64_t*>(a->get_const_slice_sets()),
                                        reinterpret_cast<const uint64_t*>(a->get_const_slice_lengths()), a->get_const_col_idxs(),
                                        a->get_const_values(), b->get_const_values(), b->get_stride(), c->get_values(), c->get_stride(),
                                        nullptr, nullptr));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha, const matrix::Sellp<double, int32>* a,
                   const matrix::Dense<double>* b, const matrix::Dense<double>* beta, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_sellp_spmv_f64_i32(GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_slice_size(),
                                        reinterpret_cast<const uint64_t*>(a->get_const_slice_sets()),
                                        reinterpret_cast<const uint64_t*>(a->get_const_slice_lengths()), a->get_const_col_idxs(),
                                        a->get_const_values(), b->get_const_values(), b->get_stride(), c->get_values(), c->get_stride(),
                                        alpha->get_const_values(), beta->get_const_values()));
}

void compute_slice_sets(std::shared_ptr<const HipExecutor> exec, const array<int32>& row_ptrs, size_type slice_size,
                        size_type stride_factor, size_type* slice_sets, size_type* slice_lengths)
{
    const int64_t nrows = static_cast<int64_t>(row_ptrs.get_num_elems()) - 1;
    const int64_t nslices = (nrows + static_cast<int64_t>(slice_size) - 1) / static_cast<int64_t>(slice_size);
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(nslices + 1));
    GKOMI_CALL(gkomi_sellp_compute_slice_sets_i32(GKOMI_NULL_STREAM, row_ptrs.get_const_data(), nrows, slice_size, stride_factor,
                                                  reinterpret_cast<uint64_t*>(slice_sets), reinterpret_cast<uint64_t*>(slice_lengths),
                                                  tmp.get_data(), tmp.get_num_elems()));
}

}  // namespace sellp
}  // namespace hip
}  // namespace kernels
}  // namespace gko
