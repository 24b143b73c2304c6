// Error translation of the shims: a non-zero C-ABI return code becomes the
// gko::Error the reference's kernels raise (include/ginkgo/core/base/exception.hpp).
// In the reference tree this header is hip/base/gkomi_bindings.hip.hpp and the
// translation unit includes the reference's own headers first; the compile test of
// this repository provides the same names through the host mirror
// (shims/test/prelude_mirror.hpp), which already defines GKOMI_CALL.
#pragma once
#include <gkomi.h>

#ifndef GKOMI_CALL
#include <ginkgo/core/base/exception.hpp>
namespace gko {
namespace kernels {
namespace hip {
inline void gkomi_check(int code, const char* file, int line, const char* fn)
{
    if (code == 0) return;
    if (code > 0) throw HipError(file, line, fn, code);  // a hipError_t
    if (code == GKOMI_ENOTSUPPORTED) throw NotSupported(file, line, fn, "gkomi");
    if (code == GKOMI_ENOTIMPL) throw NotImplemented(file, line, fn);
    throw BadDimension(file, line, fn, "gkomi", 0, 0, gkomi_error_string(code));
}
}  // namespace hip
}  // namespace kernels
}  // namespace gko
#define GKOMI_CALL(expr) ::gko::kernels::hip::gkomi_check((expr), __FILE__, __LINE__, #expr)

// The row statistic the strategy objects of a reference tree keep (csr.hpp:240-277 classical, :600-705
// automatical: max_length_per_row_, filled by strategy_type::process at make_srow time): the SpMV's row-length
// hint.  -1 = the strategy keeps none (load_balance, merge_path, sparselib, gkomi_split).  The hint only ever
// selects a kernel and a sub-wave width: every kernel it can select is correct for ANY row length (the split
// kernel finishes a row longer than its hint from memory, csr_spmv.hip step 4; tests/test_csr_split_gpu.py
// "rows longer than the hint"), so a statistic that is stale because row_ptrs were edited without process()
// costs speed, never the result.  [reference-tree branch: not compiled by this repository's tests, which take
// the definition of shims/test/prelude_mirror.hpp; the reference's include/ needs its cmake-generated
// config.hpp, so it cannot be syntax-checked here either]
#include <ginkgo/core/matrix/csr.hpp>
namespace gko {
namespace kernels {
namespace hip {
inline int64_t gkomi_row_hint(const matrix::Csr<double, int32>* a)
{
    using csr = matrix::Csr<double, int32>;
    auto s = a->get_strategy();
    if (auto c = std::dynamic_pointer_cast<csr::classical>(s)) return c->get_max_length_per_row();
    if (auto c = std::dynamic_pointer_cast<csr::automatical>(s)) {
        return c->get_name() == "classical" ? static_cast<int64_t>(c->get_max_length_per_row()) : -1;
    }
    return -1;
}
}  // namespace hip
}  // namespace kernels
}  // namespace gko
#endif

// All reference kernels run on the null stream (common/cuda_hip/base/kernel_launch.hpp.inc:66).
#define GKOMI_NULL_STREAM nullptr
