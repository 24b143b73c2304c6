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
#endif

// All reference kernels run on the null stream (common/cuda_hip/base/kernel_launch.hpp.inc:66).
#define GKOMI_NULL_STREAM nullptr
