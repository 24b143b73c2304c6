// Compile-test environment of the shims (tests/test_cpp_mirror.py): the host mirror provides the
// class and accessor names the reference's headers provide in a real tree, plus the few types of
// the reference's kernel interfaces that the mirror's public API does not need.
#pragma once
#include <ginkgo/ginkgo.hpp>
#define GKOMI_SROW_IS_OURS true  // the mirror's Csr::make_srow is gkomi_csr_make_srow_i32

namespace gko {
// include/ginkgo/core/stop/stopping_status.hpp: one byte
class stopping_status {
public:
    uint8 data_{0};
};
namespace kernels {
namespace hip {
// the mirror's Csr keeps the statistic itself (gkomi_bindings.hpp has the reference-tree version)
inline int64_t gkomi_row_hint(const matrix::Csr<double, int32>* a) { return a->get_max_row_nnz(); }
}  // namespace hip
}  // namespace kernels
namespace solver {
struct SolveStruct {
    virtual ~SolveStruct() = default;
};
enum class trisolve_algorithm { sparselib, syncfree };
}  // namespace solver
}  // namespace gko
