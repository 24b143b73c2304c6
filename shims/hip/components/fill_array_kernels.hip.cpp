// hip/components/fill_array_kernels.hip.cpp: components::fill_array<double>
// (core/components/fill_array_kernels.hpp) = a 1 x n dense fill.  fill_seq_array and reduce_add_array
// (reduce_array_kernels.hpp) are not on the hot path and stay NotCompiled in this backend.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace components {

void fill_array(std::shared_ptr<const HipExecutor> exec, double* data, size_type num_entries, double val)
{
    GKOMI_CALL(gkomi_dense_fill_f64(GKOMI_NULL_STREAM, 1, static_cast<int64_t>(num_entries), data, static_cast<int64_t>(num_entries), val));
}

}  // namespace components
}  // namespace hip
}  // namespace kernels
}  // namespace gko
