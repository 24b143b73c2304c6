"""ctypes binding of the C ABI in include/gkomi.h.

The prototypes are read from the header itself, so the header is the single
source of truth: a function declared there is callable as ``lib.<name>(...)``
with its ``gkomi_`` prefix dropped.  Pointer parameters accept torch tensors
(their ``data_ptr()``), numpy arrays, ints or None.

There is no fallback: if the shared library is missing this module raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
HEADER = os.path.join(REPO_ROOT, "include", "gkomi.h")
LIB_PATH = os.path.join(PKG_ROOT, "lib", "libgkomi.so")
# A/B timings against an older build of the same library (tools/ only): GKOMI_LIB=path/to/libgkomi_rNN.so;
# entry points that build lacks are left out instead of failing the load
LIB_OVERRIDE = os.environ.get("GKOMI_LIB")


class GkomiError(RuntimeError):
    """A C-ABI call returned non-zero (gko::HipError / gko::Error family)."""

    def __init__(self, fn, code, msg):
        super().__init__(f"{fn} failed with code {code}: {msg}")
        self.code = code


_SCALARS = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "uint8_t": ctypes.c_uint8,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "double": ctypes.c_double,
    "float": ctypes.c_float,
    "gkomi_stream_t": ctypes.c_void_p,
    "gkomi_apply_fn": ctypes.c_void_p,
    "gkomi_matrix_apply_fn": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """Returns {name: (restype_str, [(ctype_str, is_pointer, pname), ...])}."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(gkomi_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret = " ".join(m.group(1).split())
        name = m.group(2)
        params = []
        plist = m.group(3).strip()
        if plist and plist != "void":
            for p in plist.split(","):
                p = " ".join(p.split())
                is_ptr = "*" in p or "[" in p
                p2 = re.sub(r"\[.*?\]", "", p).replace("*", " ").replace("const", " ")
                toks = p2.split()
                params.append((toks[0], is_ptr, toks[-1]))
        protos[name] = (ret, params)
    return protos


def _as_arg(value, is_ptr, ctype):
    if not is_ptr:
        return value
    if value is None:
        return None
    if hasattr(value, "data_ptr"):
        return value.data_ptr()
    if hasattr(value, "ctypes"):
        return value.ctypes.data
    return value


class _Lib:
    def __init__(self, path=None):
        path = path or LIB_OVERRIDE or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `make -C {PKG_ROOT}` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        self._cdll = ctypes.CDLL(path)
        self.path = path
        self.protos = parse_header()
        for name, (ret, params) in self.protos.items():
            if LIB_OVERRIDE and not hasattr(self._cdll, name):
                continue
            fn = getattr(self._cdll, name)  # AttributeError if not exported
            fn.argtypes = [
                ctypes.c_void_p if is_ptr else _SCALARS[t]
                for (t, is_ptr, _) in params
            ]
            if "char" in ret and "*" in ret:
                fn.restype = ctypes.c_char_p
            elif ret.strip() == "size_t":
                fn.restype = ctypes.c_size_t
            elif ret.strip() == "int64_t":
                fn.restype = ctypes.c_int64
            else:
                fn.restype = ctypes.c_int
            setattr(self, name[len("gkomi_"):], self._wrap(name, fn, ret, params))

    def _wrap(self, name, fn, ret, params):
        checked = ret.strip() == "int"

        def call(*args):
            if len(args) != len(params):
                raise TypeError(f"{name} takes {len(params)} arguments "
                                f"({', '.join(p[2] for p in params)}), got {len(args)}")
            cargs = [_as_arg(a, p[1], p[0]) for a, p in zip(args, params)]
            rc = fn(*cargs)
            if checked and rc != 0:
                msg = self._cdll.gkomi_error_string(rc)
                raise GkomiError(name, rc, msg.decode() if msg else "?")
            return rc

        call.__name__ = name
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
