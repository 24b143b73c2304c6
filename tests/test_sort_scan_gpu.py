"""The library's own device-wide stable radix sort and scans (csrc/sort_scan.hip), against numpy: they replaced
the rocPRIM calls of the setup paths (device_matrix_data::sort_row_major, reference/base/device_matrix_data_kernels.cpp:
120-150; the level analysis of the triangular solves; jacobi::find_blocks; build_local_nonlocal), whose own parity
tests (bit-exact against the oracle) now run on them too."""
import numpy as np
import pytest
import torch

from gpu_util import dev, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gk():
    import gkomi
    return gkomi.lib()


def _sort(gk, keys, vals, end_bit):
    n = len(keys)
    kb = keys.dtype.itemsize
    nb = gk.diag_radix_sort_workspace_bytes(n, kb, 1 if vals is not None else 0)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda:0")
    kd = dev(keys.view(np.int64 if kb == 8 else np.int32))
    ko = torch.empty_like(kd)
    vd = dev(vals.view(np.int32)) if vals is not None else None
    vo = torch.empty_like(vd) if vals is not None else None
    gk.diag_radix_sort(torch.cuda.current_stream().cuda_stream, n, kb, kd, ko, vd, vo, end_bit, ws, nb)
    return host(ko).view(keys.dtype), (host(vo).view(np.uint32) if vals is not None else None)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 2047, 2048, 2049, 100003, 1 << 20])
@pytest.mark.parametrize("kind", ["u64_full", "u64_few_distinct", "u32_full", "u32_12bit", "u64_keys_only"])
def test_radix_sort_is_numpy_stable_sort(gk, n, kind):
    rng = np.random.default_rng(n + len(kind))
    if kind == "u64_full":
        keys, end_bit = rng.integers(0, 2**63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64), 64
    elif kind == "u64_few_distinct":     # many duplicates: stability decides the payload order
        keys, end_bit = (rng.integers(0, 7, size=n, dtype=np.uint64) << np.uint64(40)) | rng.integers(0, 3, size=n, dtype=np.uint64), 64
    elif kind == "u32_full":
        keys, end_bit = rng.integers(0, 2**32, size=n, dtype=np.uint32), 32
    elif kind == "u32_12bit":            # fewer passes than the key is wide
        keys, end_bit = rng.integers(0, 1 << 12, size=n, dtype=np.uint32), 12
    else:
        keys, end_bit = rng.integers(0, 2**62, size=n, dtype=np.uint64), 64
    vals = None if kind == "u64_keys_only" else np.arange(n, dtype=np.uint32)
    got_k, got_v = _sort(gk, keys, vals, end_bit)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got_k, keys[order])
    if vals is not None:
        assert np.array_equal(got_v, vals[order])   # the stable order, exactly


@pytest.mark.parametrize("n", [1, 5, 2048, 2049, 8199, 1 << 22])
@pytest.mark.parametrize("kind,in_place", [(0, False), (0, True), (1, False), (1, True)])
def test_scans_match_numpy(gk, n, kind, in_place):
    rng = np.random.default_rng(n + kind)
    a = rng.integers(0, 5, size=n).astype(np.int32) if kind == 0 else rng.integers(-1000, 1000, size=n).astype(np.int32)
    ad = dev(a)
    od = ad if in_place else torch.full_like(ad, -7)
    nb = (4 * ((n + 2047) // 2048) + 255) // 256 * 256
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    gk.diag_scan_i32(torch.cuda.current_stream().cuda_stream, kind, ad, od, n, ws, nb)
    want = (np.cumsum(a, dtype=np.int64) - a).astype(np.int32) if kind == 0 else np.maximum.accumulate(a)
    assert np.array_equal(host(od), want)
