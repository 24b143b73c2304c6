"""CPU-side checks of the drop-in boundary: the C-ABI library loads and
exports every symbol include/gkomi.h declares; argument validation that
returns before any HIP call behaves like the reference's dimension checks."""
import ctypes
import os
import re

import pytest

import gkomi


def test_library_exports_every_declared_symbol():
    protos = gkomi.parse_header()
    assert len(protos) >= 20
    cdll = ctypes.CDLL(gkomi.LIB_PATH)
    missing = [n for n in protos if not hasattr(cdll, n)]
    assert not missing, f"declared in gkomi.h but not exported: {missing}"


def test_header_cites_reference_interfaces():
    text = open(gkomi.HEADER).read()
    # every component block names the reference interface it replaces
    for needle in ("core/matrix/csr_kernels.hpp", "core/matrix/dense_kernels.hpp",
                   "core/solver/cg_kernels.hpp", "core/stop/residual_norm_kernels.hpp"):
        assert needle in text


def test_no_torch_types_in_abi():
    # signatures only (comments cite C++ names of the reference)
    for name, (ret, params) in gkomi.parse_header().items():
        for ctype, _, _ in params:
            assert ctype in ("int", "int32_t", "int64_t", "uint8_t", "uint32_t", "uint64_t", "size_t", "double",
                             "float", "void", "gkomi_stream_t", "gkomi_apply_fn", "gkomi_matrix_apply_fn", "char",
                             # plain C records of the ABI itself (pointers and sizes, declared in gkomi.h)
                             "gkomi_comm", "gkomi_dist_matrix", "gkomi_dist_ctx",
                             # opaque handle (host-side analysis result, like the reference's SolveStruct)
                             "gkomi_trs_bricks",
                             # opaque handle of the column-partitioned CSR copy (sizes + offsets into the caller's plan)
                             "gkomi_csr_colpart"), (name, ctype)


def test_version_and_error_strings(gk):
    assert b"gfx950" in gk.version()
    cdll = ctypes.CDLL(gkomi.LIB_PATH)
    cdll.gkomi_error_string.restype = ctypes.c_char_p
    assert b"invalid" in cdll.gkomi_error_string(-1)
    assert b"workspace" in cdll.gkomi_error_string(-4)


def test_invalid_arguments_are_rejected_before_launch(gk):
    # negative sizes / alpha without beta: GKOMI_EINVAL, no HIP call involved
    with pytest.raises(gkomi.GkomiError) as e:
        gk.csr_spmv_f64_i32(None, -1, 1, 1, -1, None, None, None, None, 1, None, 1, None, None, 0, -1)
    assert e.value.code == -1
    with pytest.raises(gkomi.GkomiError):
        gk.csr_spmv_f64_i32(None, 1, 1, 1, -1, 8, 8, 8, 8, 1, 8, 1, 8, None, 0, -1)
    # empty output is a no-op (hip/matrix/csr_kernels.hip.cpp:291-292)
    assert gk.csr_spmv_f64_i32(None, 0, 5, 1, 0, None, None, None, None, 1, None, 1, None, None, 0, -1) == 0
    assert gk.dense_scale_f64(None, 0, 3, None, 1, None, 3) == 0
    with pytest.raises(gkomi.GkomiError):
        gk.dense_scale_f64(None, 2, 3, 8, 2, 8, 3)  # alpha must be 1 or ncols wide


def test_reduction_workspace_size(gk):
    assert gk.dense_reduction_workspace_bytes(0, 1) == 0
    assert gk.dense_reduction_workspace_bytes(10, 1) == 8
    big = gk.dense_reduction_workspace_bytes(1 << 24, 3)
    assert big == 8 * 2048 * 3


def test_product_never_references_oracle():
    """The shipped path must not import/link the oracle (only tests, smoke and
    bench's cpu_baseline leg may)."""
    root = os.path.join(os.path.dirname(gkomi.HEADER), "..", "repo-8852-ginkgo_amd")
    offenders = []
    for dirpath, _, files in os.walk(root):
        if "build" in dirpath or "/lib" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            txt = open(os.path.join(dirpath, f), errors="ignore").read()
            if re.search(r"oracle_lib|libgko_oracle|oracle/", txt):
                offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
