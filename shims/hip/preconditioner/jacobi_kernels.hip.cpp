// hip/preconditioner/jacobi_*_kernel.hip.cpp: jacobi::simple_apply / apply
// (core/preconditioner/jacobi_kernels.hpp:95-130; reference/preconditioner/jacobi_kernels.cpp:447-598).
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace jacobi {

// The kernels keep the blocks in the reference's block_interleaved_storage_scheme with the HIP
// stride (max_block_stride = wavefront size 64, jacobi.hpp:578-609): a scheme built with another
// max_block_stride is refused rather than misread.
inline void check_scheme(uint32 max_block_size, const preconditioner::block_interleaved_storage_scheme<int32>& scheme)
{
    int64_t ours[4] = {};
    GKOMI_CALL(gkomi_jacobi_storage_scheme(static_cast<int>(max_block_size), ours));
    if (scheme.block_offset != ours[0] || scheme.group_offset != ours[1] || scheme.group_power != static_cast<uint32>(ours[2])) {
        GKO_NOT_SUPPORTED("jacobi: only the storage scheme of max_block_stride = 64 (the HIP wavefront) is supported");
    }
}

void simple_apply(std::shared_ptr<const HipExecutor> exec, size_type num_blocks, uint32 max_block_size,
                  const preconditioner::block_interleaved_storage_scheme<int32>& storage_scheme,
                  const array<precision_reduction>& block_precisions, const array<int32>& block_pointers,
                  const array<double>& blocks, const matrix::Dense<double>* b, matrix::Dense<double>* x)
{
    check_scheme(max_block_size, storage_scheme);
    if (block_precisions.get_num_elems() > 0) {  // storage_optimization: precision_reduction is one byte
        GKOMI_CALL(gkomi_jacobi_apply_adaptive_f64_i32(
            GKOMI_NULL_STREAM, num_blocks, max_block_size, block_pointers.get_const_data(),
            reinterpret_cast<const uint8_t*>(block_precisions.get_const_data()), blocks.get_const_data(), b->get_size()[1], nullptr,
            b->get_const_values(), b->get_stride(), nullptr, x->get_values(), x->get_stride()));
    } else {
        GKOMI_CALL(gkomi_jacobi_apply_f64_i32(GKOMI_NULL_STREAM, num_blocks, max_block_size, block_pointers.get_const_data(),
                                              blocks.get_const_data(), b->get_size()[1], nullptr, b->get_const_values(), b->get_stride(),
                                              nullptr, x->get_values(), x->get_stride()));
    }
}

void apply(std::shared_ptr<const HipExecutor> exec, size_type num_blocks, uint32 max_block_size,
           const preconditioner::block_interleaved_storage_scheme<int32>& storage_scheme,
           const array<precision_reduction>& block_precisions, const array<int32>& block_pointers, const array<double>& blocks,
           const matrix::Dense<double>* alpha, const matrix::Dense<double>* b, const matrix::Dense<double>* beta,
           matrix::Dense<double>* x)
{
    check_scheme(max_block_size, storage_scheme);
    if (block_precisions.get_num_elems() > 0) {
        GKOMI_CALL(gkomi_jacobi_apply_adaptive_f64_i32(
            GKOMI_NULL_STREAM, num_blocks, max_block_size, block_pointers.get_const_data(),
            reinterpret_cast<const uint8_t*>(block_precisions.get_const_data()), blocks.get_const_data(), b->get_size()[1],
            alpha->get_const_values(), b->get_const_values(), b->get_stride(), beta->get_const_values(), x->get_values(), x->get_stride()));
    } else {
        GKOMI_CALL(gkomi_jacobi_apply_f64_i32(GKOMI_NULL_STREAM, num_blocks, max_block_size, block_pointers.get_const_data(),
                                              blocks.get_const_data(), b->get_size()[1], alpha->get_const_values(), b->get_const_values(),
                                              b->get_stride(), beta->get_const_values(), x->get_values(), x->get_stride()));
    }
}

}  // namespace jacobi
}  // namespace hip
}  // namespace kernels
}  // namespace gko
