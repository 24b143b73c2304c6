// hip/components/prefix_sum_kernels.hip.cpp: components::prefix_sum (core/components/prefix_sum_kernels.hpp):
// exclusive scan in place, counts[num_entries - 1] receives the total of the first num_entries - 1.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace components {

void prefix_sum(std::shared_ptr<const HipExecutor> exec, int32* counts, size_type num_entries)
{
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(static_cast<int64_t>(num_entries)));
    GKOMI_CALL(gkomi_prefix_sum_i32(GKOMI_NULL_STREAM, counts, static_cast<int64_t>(num_entries), tmp.get_data(), tmp.get_num_elems()));
}

void prefix_sum(std::shared_ptr<const HipExecutor> exec, int64* counts, size_type num_entries)
{
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(static_cast<int64_t>(num_entries)));
    GKOMI_CALL(gkomi_prefix_sum_i64(GKOMI_NULL_STREAM, reinterpret_cast<int64_t*>(counts), static_cast<int64_t>(num_entries), tmp.get_data(),
                                    tmp.get_num_elems()));
}

}  // namespace components
}  // namespace hip
}  // namespace kernels
}  // namespace gko
