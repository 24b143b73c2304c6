"""Host-side (numpy + oracle) builders of the sparse formats used by the
format tests: the conversion chain of core/matrix/csr.cpp:257-405 driven
through the ORACLE kernels.  Test infrastructure."""
import numpy as np


def oracle_to_ell(oracle, nrows, rp, ci, v, stride=None):
    k = int(oracle.ref_compute_max_row_nnz(rp, nrows))
    stride = nrows if stride is None else stride
    cols = np.full(max(stride * k, 1), -7, np.int32)
    vals = np.full(max(stride * k, 1), np.nan)
    # rows in [nrows, stride) are padding the reference never reads; mark them invalid
    cols[:] = -1
    vals[:] = 0.0
    oracle.ref_csr_convert_to_ell(nrows, rp, ci, v, k, stride, cols, vals)
    return k, stride, cols, vals


def oracle_to_sellp(oracle, nrows, rp, ci, v, slice_size=64, stride_factor=1):
    nsl = (nrows + slice_size - 1) // slice_size
    sets = np.zeros(nsl + 1, np.uint64)
    lens = np.zeros(max(nsl, 1), np.uint64)
    oracle.ref_sellp_compute_slice_sets(rp, nrows, slice_size, stride_factor, sets, lens)
    total = int(sets[nsl]) * slice_size
    cols = np.full(max(total, 1), -1, np.int32)
    vals = np.zeros(max(total, 1))
    oracle.ref_csr_convert_to_sellp(nrows, rp, ci, v, slice_size, sets, lens, cols, vals)
    return sets, lens, cols, vals


def oracle_to_hybrid(oracle, nrows, ncols, rp, ci, v, kind=4, percent=0.8, ratio=1e-4, num_columns=0,
                     ell_stride=None):
    ell_lim = int(oracle.ref_hybrid_ell_width(rp, nrows, kind, percent, ratio, num_columns))
    ell_lim = min(ell_lim, ncols)  # csr.cpp:304-307
    ell_stride = nrows if ell_stride is None else ell_stride
    crp = np.zeros(nrows + 1, np.int64)
    oracle.ref_hybrid_compute_coo_row_ptrs(rp, nrows, ell_lim, crp)
    coo_nnz = int(crp[nrows])
    ell_cols = np.full(max(ell_lim * ell_stride, 1), -1, np.int32)
    ell_vals = np.zeros(max(ell_lim * ell_stride, 1))
    coo_r = np.zeros(max(coo_nnz, 1), np.int32)
    coo_c = np.zeros(max(coo_nnz, 1), np.int32)
    coo_v = np.zeros(max(coo_nnz, 1))
    oracle.ref_csr_convert_to_hybrid(nrows, rp, ci, v, ell_lim, ell_stride, ell_cols, ell_vals, coo_r, coo_c, coo_v)
    return dict(ell_lim=ell_lim, ell_stride=ell_stride, ell_cols=ell_cols, ell_vals=ell_vals, coo_nnz=coo_nnz,
                coo_rows=coo_r, coo_cols=coo_c, coo_vals=coo_v, coo_row_ptrs=crp)
