// hip/components/format_conversion_kernels.hip.cpp: components::convert_ptrs_to_idxs / convert_idxs_to_ptrs /
// convert_ptrs_to_sizes (core/components/format_conversion_kernels.hpp), the <int32, int32> instantiations.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace components {

void convert_ptrs_to_idxs(std::shared_ptr<const HipExecutor> exec, const int32* ptrs, size_type num_blocks, int32* idxs)
{
    GKOMI_CALL(gkomi_convert_ptrs_to_idxs_i32(GKOMI_NULL_STREAM, ptrs, static_cast<int64_t>(num_blocks), idxs));
}

void convert_idxs_to_ptrs(std::shared_ptr<const HipExecutor> exec, const int32* idxs, size_type num_idxs, size_type num_blocks,
                          int32* ptrs)
{
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(static_cast<int64_t>(num_blocks) + 1));
    GKOMI_CALL(gkomi_convert_idxs_to_ptrs_i32(GKOMI_NULL_STREAM, idxs, static_cast<int64_t>(num_idxs), static_cast<int64_t>(num_blocks), ptrs,
                                              tmp.get_data(), tmp.get_num_elems()));
}

void convert_ptrs_to_sizes(std::shared_ptr<const HipExecutor> exec, const int32* ptrs, size_type num_blocks, size_type* sizes)
{
    static_assert(sizeof(size_type) == sizeof(uint64_t), "sizes are 64-bit");
    GKOMI_CALL(gkomi_convert_ptrs_to_sizes_i32(GKOMI_NULL_STREAM, ptrs, static_cast<int64_t>(num_blocks), reinterpret_cast<uint64_t*>(sizes)));
}

}  // namespace components
}  // namespace hip
}  // namespace kernels
}  // namespace gko
