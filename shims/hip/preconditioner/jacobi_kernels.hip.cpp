// hip/preconditioner/jacobi_*_kernel.hip.cpp: jacobi::find_blocks / generate / simple_apply / apply, the scalar
// variants, transpose_jacobi (core/preconditioner/jacobi_kernels.hpp; reference/preconditioner/jacobi_kernels.cpp:66-625).
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

void find_blocks(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* system_matrix, uint32 max_block_size,
                 size_type& num_blocks, array<int32>& block_pointers)
{
    const int64_t n = static_cast<int64_t>(system_matrix->get_size()[0]);
    array<char> tmp(exec, gkomi_jacobi_find_blocks_workspace_bytes(n));
    array<int64> device_count(exec, 1);
    int64_t count = 0;
    GKOMI_CALL(gkomi_jacobi_find_blocks_i32(GKOMI_NULL_STREAM, n, system_matrix->get_const_row_ptrs(), system_matrix->get_const_col_idxs(),
                                            static_cast<int>(max_block_size), block_pointers.get_data(),
                                            reinterpret_cast<int64_t*>(device_count.get_data()), tmp.get_data(), tmp.get_num_elems(), &count));
    num_blocks = static_cast<size_type>(count);
}

void generate(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* system_matrix, size_type num_blocks,
              uint32 max_block_size, double accuracy, const preconditioner::block_interleaved_storage_scheme<int32>& storage_scheme,
              array<double>& conditioning, array<precision_reduction>& block_precisions, const array<int32>& block_pointers,
              array<double>& blocks)
{
    check_scheme(max_block_size, storage_scheme);
    const int64_t n = static_cast<int64_t>(system_matrix->get_size()[0]);
    if (block_precisions.get_num_elems() > 0) {  // adaptive precision: the detection runs in the generating wave
        GKOMI_CALL(gkomi_jacobi_generate_adaptive_f64_i32(
            GKOMI_NULL_STREAM, n, system_matrix->get_const_row_ptrs(), system_matrix->get_const_col_idxs(), system_matrix->get_const_values(),
            static_cast<int64_t>(num_blocks), static_cast<int>(max_block_size), block_pointers.get_const_data(), accuracy,
            conditioning.get_data(), reinterpret_cast<uint8_t*>(block_precisions.get_data()), blocks.get_data()));
    } else {
        GKOMI_CALL(gkomi_jacobi_generate_f64_i32(
            GKOMI_NULL_STREAM, n, system_matrix->get_const_row_ptrs(), system_matrix->get_const_col_idxs(), system_matrix->get_const_values(),
            static_cast<int64_t>(num_blocks), static_cast<int>(max_block_size), block_pointers.get_const_data(),
            conditioning.get_num_elems() > 0 ? conditioning.get_data() : nullptr, blocks.get_data()));
    }
}

void invert_diagonal(std::shared_ptr<const HipExecutor> exec, const array<double>& diag, array<double>& inv_diag)
{
    GKOMI_CALL(gkomi_jacobi_invert_diagonal_f64(GKOMI_NULL_STREAM, static_cast<int64_t>(diag.get_num_elems()), diag.get_const_data(),
                                                inv_diag.get_data()));
}

void simple_scalar_apply(std::shared_ptr<const HipExecutor> exec, const array<double>& diag, const matrix::Dense<double>* b,
                         matrix::Dense<double>* x)
{
    GKOMI_CALL(gkomi_jacobi_scalar_apply_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], diag.get_const_data(), nullptr,
                                             b->get_const_values(), b->get_stride(), nullptr, x->get_values(), x->get_stride()));
}

void scalar_apply(std::shared_ptr<const HipExecutor> exec, const array<double>& diag, const matrix::Dense<double>* alpha,
                  const matrix::Dense<double>* b, const matrix::Dense<double>* beta, matrix::Dense<double>* x)
{
    GKOMI_CALL(gkomi_jacobi_scalar_apply_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], diag.get_const_data(),
                                             alpha->get_const_values(), b->get_const_values(), b->get_stride(), beta->get_const_values(),
                                             x->get_values(), x->get_stride()));
}

void transpose_jacobi(std::shared_ptr<const HipExecutor> exec, size_type num_blocks, uint32 max_block_size,
                      const array<precision_reduction>& block_precisions, const array<int32>& block_pointers, const array<double>& blocks,
                      const preconditioner::block_interleaved_storage_scheme<int32>& storage_scheme, array<double>& out_blocks)
{
    check_scheme(max_block_size, storage_scheme);
    GKOMI_CALL(gkomi_jacobi_transpose_f64_i32(
        GKOMI_NULL_STREAM, static_cast<int64_t>(num_blocks), static_cast<int>(max_block_size), block_pointers.get_const_data(),
        block_precisions.get_num_elems() > 0 ? reinterpret_cast<const uint8_t*>(block_precisions.get_const_data()) : nullptr,
        blocks.get_const_data(), out_blocks.get_data()));
}

// real values: the conjugate transpose is the transpose
void conj_transpose_jacobi(std::shared_ptr<const HipExecutor> exec, size_type num_blocks, uint32 max_block_size,
                           const array<precision_reduction>& block_precisions, const array<int32>& block_pointers,
                           const array<double>& blocks, const preconditioner::block_interleaved_storage_scheme<int32>& storage_scheme,
                           array<double>& out_blocks)
{
    transpose_jacobi(exec, num_blocks, max_block_size, block_precisions, block_pointers, blocks, storage_scheme, out_blocks);
}

}  // namespace jacobi
}  // namespace hip
}  // namespace kernels
}  // namespace gko
