"""Oracle for the device_matrix_data kernels, checked the way the reference
checks its own (test/base/device_matrix_data_kernels.cpp:275-380: SortsRowMajor,
RemovesZeros, DoesntRemoveZerosIfThereAreNone, SumsDuplicates,
DoesntSumDuplicatesIfThereAreNone) -- the reference holds no fixed vectors for
them, so these are property checks on the same kind of seeded data."""
import numpy as np

import assembly_util as au


def _sort(oracle, t):
    r, c, v = (a.copy() for a in t)
    oracle.ref_matrix_data_sort_row_major(len(r), r, c, v)
    return r, c, v


def _compact(fn, t):
    n = len(t[0])
    r, c, v = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
    cnt = fn(n, t[0], t[1], t[2], r, c, v)
    return r[:cnt], c[:cnt], v[:cnt]


def test_sorts_row_major(oracle):
    f = au.fixture()
    got = _sort(oracle, f["host"])
    for a, b in zip(got, f["sorted"]):
        assert np.array_equal(a, b)


def test_sort_is_stable_and_handles_small_inputs(oracle):
    r = np.array([1, 0, 1, 0, 1], np.int32)
    c = np.array([2, 5, 2, 5, 0], np.int32)
    v = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    got = _sort(oracle, (r, c, v))
    assert list(got[0]) == [0, 0, 1, 1, 1] and list(got[1]) == [5, 5, 0, 2, 2]
    assert list(got[2]) == [2.0, 4.0, 5.0, 1.0, 3.0]
    for n in (0, 1):
        got = _sort(oracle, (r[:n], c[:n], v[:n]))
        assert list(got[2]) == list(v[:n])


def test_removes_zeros(oracle):
    f = au.fixture()
    got = _compact(oracle.ref_matrix_data_remove_zeros, f["host"])
    for a, b in zip(got, f["nonzero"]):
        assert np.array_equal(a, b)
    again = _compact(oracle.ref_matrix_data_remove_zeros, f["nonzero"])
    for a, b in zip(again, f["nonzero"]):
        assert np.array_equal(a, b)
    nan = (np.array([0, 1], np.int32), np.array([0, 1], np.int32), np.array([np.nan, -0.0]))
    got = _compact(oracle.ref_matrix_data_remove_zeros, nan)
    assert len(got[0]) == 1 and np.isnan(got[2][0])  # NaN is nonzero, -0.0 is zero


def test_sums_duplicates(oracle):
    f = au.fixture()
    srt = _sort(oracle, f["duplicate"])
    got = _compact(oracle.ref_matrix_data_sum_duplicates, srt)
    exp = au.sum_duplicates_numpy(*f["duplicate"])
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
    # locations = those of the de-duplicated data, sums = all values per location
    assert len(got[0]) == len(f["host"][0])
    assert abs(got[2].sum() - f["duplicate"][2].sum()) < 1e-9
    # nothing to do on unique data
    srt = _sort(oracle, f["host"])
    got = _compact(oracle.ref_matrix_data_sum_duplicates, srt)
    for a, b in zip(got, f["sorted"]):
        assert np.array_equal(a, b)


def test_sum_duplicates_starts_from_positive_zero(oracle):
    t = (np.array([3], np.int32), np.array([4], np.int32), np.array([-0.0]))
    got = _compact(oracle.ref_matrix_data_sum_duplicates, t)
    assert not np.signbit(got[2][0])  # 0.0 + -0.0 = +0.0, as in the reference loop
