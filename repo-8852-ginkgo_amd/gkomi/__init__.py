"""Python-side plumbing for the MI355X Ginkgo hot-path backend.

The product is the HIP library (csrc/ -> lib/libgkomi.so, C ABI in
include/gkomi.h) and the C++ host mirror (include/ginkgo/).  This package only
loads the library through ctypes and wraps device buffers (torch tensors) so
that tests and bench.py can drive the C ABI.
"""
from ._lib import GkomiError, lib, parse_header, LIB_PATH, HEADER  # noqa: F401
