// Compile-test environment of the shims (tests/test_cpp_mirror.py): the host mirror provides the
// class and accessor names the reference's headers provide in a real tree, plus the few types of
// the reference's kernel interfaces that the mirror's public API does not need.
#pragma once
#include <ginkgo/ginkgo.hpp>

namespace gko {
// include/ginkgo/core/stop/stopping_status.hpp: one byte
class stopping_status {
public:
    uint8 data_{0};
};
namespace solver {
struct SolveStruct {
    virtual ~SolveStruct() = default;
};
enum class trisolve_algorithm { sparselib, syncfree };
}  // namespace solver
}  // namespace gko
