"""GPU parity (through the C ABI) of the device_matrix_data kernels against the
oracle: bit-exact, including the order of duplicates (both sorts are stable)
and the left-to-right duplicate sums; plus Csr::read = sort + idxs_to_ptrs."""
import ctypes

import numpy as np
import pytest
import torch

import assembly_util as au
import matgen
from gpu_util import dev, host, stream_ptr

pytestmark = pytest.mark.gpu


def _ws(gk, n):
    nb = gk.matrix_data_workspace_bytes(n)
    return torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda:0"), nb


def gpu_sort(gk, t):
    r, c, v = dev(t[0].astype(np.int32)), dev(t[1].astype(np.int32)), dev(t[2].astype(np.float64))
    ws, nb = _ws(gk, len(t[0]))
    gk.matrix_data_sort_row_major_f64_i32(stream_ptr(), len(t[0]), r, c, v, ws, nb)
    return host(r), host(c), host(v)


def gpu_compact(gk, name, t):
    n = len(t[0])
    r, c, v = dev(t[0].astype(np.int32)), dev(t[1].astype(np.int32)), dev(t[2].astype(np.float64))
    orr = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda:0")
    oc = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda:0")
    ov = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda:0")
    ws, nb = _ws(gk, n)
    cnt = ctypes.c_int64(-1)
    getattr(gk, name)(stream_ptr(), n, r, c, v, orr, oc, ov, ws, nb, ctypes.addressof(cnt))
    k = cnt.value
    return host(orr)[:k], host(oc)[:k], host(ov)[:k]


def oracle_sort(oracle, t):
    r, c, v = t[0].astype(np.int32).copy(), t[1].astype(np.int32).copy(), t[2].astype(np.float64).copy()
    oracle.ref_matrix_data_sort_row_major(len(r), r, c, v)
    return r, c, v


def oracle_compact(fn, t):
    n = len(t[0])
    r, c, v = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
    k = fn(n, t[0].astype(np.int32), t[1].astype(np.int32), t[2].astype(np.float64), r, c, v)
    return r[:k], c[:k], v[:k]


def same(a, b):
    return all(np.array_equal(x, y) and x.tobytes() == y.tobytes() for x, y in zip(a, b))


@pytest.mark.parametrize("seed", [82754, 1, 2])
def test_reference_fixture(gk, oracle, seed):
    f = au.fixture(seed)
    assert same(gpu_sort(gk, f["host"]), f["sorted"])
    assert same(gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", f["host"]), f["nonzero"])
    assert same(gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", f["nonzero"]), f["nonzero"])
    srt = gpu_sort(gk, f["duplicate"])
    assert same(srt, oracle_sort(oracle, f["duplicate"]))  # duplicates keep their input order
    got = gpu_compact(gk, "matrix_data_sum_duplicates_f64_i32", srt)
    assert same(got, oracle_compact(oracle.ref_matrix_data_sum_duplicates, srt))
    assert same(got, au.sum_duplicates_numpy(*f["duplicate"]))
    srt = gpu_sort(gk, f["host"])
    assert same(gpu_compact(gk, "matrix_data_sum_duplicates_f64_i32", srt), f["sorted"])


def test_edge_cases(gk, oracle):
    e = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    assert len(gpu_sort(gk, e)[0]) == 0
    assert len(gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", e)[0]) == 0
    assert len(gpu_compact(gk, "matrix_data_sum_duplicates_f64_i32", e)[0]) == 0
    one = (np.array([7], np.int32), np.array([3], np.int32), np.array([-0.0]))
    assert same(gpu_sort(gk, one), one)
    got = gpu_compact(gk, "matrix_data_sum_duplicates_f64_i32", one)
    assert got[2][0] == 0.0 and not np.signbit(got[2][0])
    assert len(gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", one)[0]) == 0
    t = (np.array([0, 1, 2], np.int32), np.array([0, 1, 2], np.int32), np.array([np.nan, 0.0, np.inf]))
    got = gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", t)
    assert list(got[0]) == [0, 2] and np.isnan(got[2][0]) and np.isinf(got[2][1])
    # everything is one location; everything zero
    n = 5000
    rng = np.random.default_rng(5)
    t = (np.full(n, 4, np.int32), np.full(n, 9, np.int32), rng.standard_normal(n))
    got = gpu_compact(gk, "matrix_data_sum_duplicates_f64_i32", t)
    exp = oracle_compact(oracle.ref_matrix_data_sum_duplicates, t)
    assert same(got, exp) and len(got[0]) == 1
    z = (t[0], t[1], np.zeros(n))
    assert len(gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", z)[0]) == 0


def test_large_random_with_many_duplicates(gk, oracle):
    rng = np.random.default_rng(11)
    n = 1_000_000
    t = (rng.integers(0, 3000, n).astype(np.int32), rng.integers(0, 3000, n).astype(np.int32),
         rng.standard_normal(n))
    t[2][rng.random(n) < 0.1] = 0.0
    srt = gpu_sort(gk, t)
    assert same(srt, oracle_sort(oracle, t))
    key = srt[0].astype(np.int64) * 3000 + srt[1]
    assert np.all(np.diff(key) >= 0)
    got = gpu_compact(gk, "matrix_data_sum_duplicates_f64_i32", srt)
    assert same(got, oracle_compact(oracle.ref_matrix_data_sum_duplicates, srt))
    got = gpu_compact(gk, "matrix_data_remove_zeros_f64_i32", t)
    assert same(got, oracle_compact(oracle.ref_matrix_data_remove_zeros, t))


def test_csr_read_from_shuffled_triplets(gk, oracle):
    """Csr::read (core/matrix/csr.cpp:453-470) on the device: sort_row_major,
    then convert_idxs_to_ptrs -- gives back the CSR arrays the triplets came from."""
    n, rp, ci, v = matgen.poisson_2d_5pt(60)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    perm = np.random.default_rng(3).permutation(len(v))
    srt = gpu_sort(gk, (rows[perm], ci[perm], v[perm]))
    assert same(srt, (rows, ci, v))
    rpd = torch.zeros(n + 1, dtype=torch.int32, device="cuda:0")
    nb = gk.prefix_sum_workspace_bytes(n + 1)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device="cuda:0")
    gk.convert_idxs_to_ptrs_i32(stream_ptr(), dev(srt[0]), len(v), n, rpd, ws, nb)
    assert np.array_equal(host(rpd), rp)
