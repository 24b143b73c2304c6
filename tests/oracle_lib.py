"""Loader for the CPU oracle (oracle/libgko_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  Prototypes are parsed from the ORACLE_API definitions in oracle/*.c.
Pointer arguments take numpy arrays (C-contiguous, right dtype) or None.
"""
import ctypes
import glob
import os
import re
import subprocess

import numpy as np

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO_ROOT, "oracle")
# GKO_ORACLE_LIB: another build of the same sources (tools/sanitize_cpu.sh points it at the
# AddressSanitizer / UBSan build)
LIB = os.environ.get("GKO_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libgko_oracle.so")

_SCALARS = {
    "i64": ctypes.c_int64, "i32": ctypes.c_int32, "u8": ctypes.c_uint8,
    "int": ctypes.c_int, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
    "u64": ctypes.c_uint64, "oracle_apply_fn": ctypes.c_void_p, "float": ctypes.c_float,
}
_NP = {"i64": np.int64, "i32": np.int32, "u8": np.uint8, "double": np.float64,
       "int": np.int32, "u64": np.uint64, "float": np.float32}


def build(force=False):
    if os.environ.get("GKO_ORACLE_LIB"):
        return LIB
    srcs = glob.glob(os.path.join(ORACLE_DIR, "*.c")) + glob.glob(os.path.join(ORACLE_DIR, "*.h"))
    if force or not os.path.exists(LIB) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return LIB


def _parse():
    protos = {}
    for path in sorted(glob.glob(os.path.join(ORACLE_DIR, "*.c"))):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"ORACLE_API\s+([\w\s\*]+?)\b(\w+)\s*\(([^{};]*?)\)\s*\{", text):
            ret = " ".join(m.group(1).split())
            params = []
            plist = m.group(3).strip()
            if plist and plist != "void":
                for p in plist.split(","):
                    p = " ".join(p.split())
                    is_ptr = "*" in p
                    toks = p.replace("*", " ").replace("const", " ").split()
                    params.append((toks[0], is_ptr, toks[-1]))
            protos[m.group(2)] = (ret, params)
    return protos


class _Oracle:
    def __init__(self):
        build()
        self._cdll = ctypes.CDLL(LIB)
        self.protos = _parse()
        for name, (ret, params) in self.protos.items():
            fn = getattr(self._cdll, name)
            fn.argtypes = [ctypes.c_void_p if ptr else _SCALARS[t] for t, ptr, _ in params]
            fn.restype = None if ret == "void" else _SCALARS[ret]
            setattr(self, name, self._wrap(name, fn, params))

    @staticmethod
    def _wrap(name, fn, params):
        def call(*args):
            if len(args) != len(params):
                raise TypeError(f"{name}({', '.join(p[2] for p in params)}) got {len(args)} args")
            cargs = []
            for a, (t, ptr, pname) in zip(args, params):
                if ptr and a is not None:
                    assert isinstance(a, np.ndarray), f"{name}: {pname} must be ndarray"
                    assert a.flags["C_CONTIGUOUS"], f"{name}: {pname} not contiguous"
                    if t in _NP:
                        assert a.dtype == _NP[t], f"{name}: {pname} dtype {a.dtype} != {t}"
                    cargs.append(a.ctypes.data)
                else:
                    cargs.append(a)
            return fn(*cargs)
        call.__name__ = name
        return call


_inst = None


def load():
    global _inst
    if _inst is None:
        _inst = _Oracle()
    return _inst
