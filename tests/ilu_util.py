"""ParIlu::generate_l_u (core/factorization/par_ilu.cpp:74-163) composed from
kernel entry points: once through the ORACLE, once through the C ABI on the
GPU.  Test infrastructure."""
import ctypes

import numpy as np


def oracle_par_ilu(oracle, n, rp, ci, v, iterations=0):
    rp = rp.copy()
    nnz = int(rp[-1])
    ncols = np.zeros(nnz + n, np.int32)
    nvals = np.zeros(nnz + n)
    new_nnz = int(oracle.ref_add_diagonal_elements(n, n, rp, ci if nnz else np.zeros(1, np.int32),
                                                   v if nnz else np.zeros(1), ncols, nvals))
    ci, v = ncols[:new_nnz].copy(), nvals[:new_nnz].copy()
    lrp, urp = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    oracle.ref_initialize_row_ptrs_l_u(n, rp, ci, lrp, urp)
    lc, lv = np.zeros(lrp[-1], np.int32), np.zeros(lrp[-1])
    uc, uv = np.zeros(urp[-1], np.int32), np.zeros(urp[-1])
    oracle.ref_initialize_l_u(n, rp, ci, v, lrp, lc, lv, urp, uc, uv)
    utrp, utc, utv = np.zeros(n + 1, np.int32), np.zeros(urp[-1], np.int32), np.zeros(urp[-1])
    oracle.ref_csr_transpose(n, n, urp, uc, uv, utrp, utc, utv)
    rows = np.zeros(max(new_nnz, 1), np.int32)
    oracle.ref_convert_ptrs_to_idxs(rp, n, rows)
    oracle.ref_par_ilu_compute_l_u_factors(iterations, new_nnz, rows, ci, v, lrp, lc, lv, utrp, utc, utv)
    urp2, uc2, uv2 = np.zeros(n + 1, np.int32), np.zeros(urp[-1], np.int32), np.zeros(urp[-1])
    oracle.ref_csr_transpose(n, n, utrp, utc, utv, urp2, uc2, uv2)
    return dict(A=(rp, ci, v), L=(lrp, lc, lv), U=(urp2, uc2, uv2))


def csr_to_dense(n, m, rp, ci, v):
    a = np.zeros((n, m))
    for r in range(n):
        for k in range(rp[r], rp[r + 1]):
            a[r, ci[k]] = v[k]
    return a


def gpu_par_ilu(gk, torch, n, rpd, cid, vd, iterations=0):
    """Same chain on the device; rpd is modified in place like the reference's
    add_diagonal_elements (pass a clone)."""
    s = torch.cuda.current_stream().cuda_stream
    dv = rpd.device
    nbytes = gk.factorization_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dv)
    missing = ctypes.c_int64(0)
    gk.factorization_count_missing_diagonal_i32(s, n, n, rpd, cid, ws, nbytes, ctypes.addressof(missing))
    nnz = int(vd.numel())
    if missing.value:
        nc = torch.zeros(nnz + missing.value, dtype=torch.int32, device=dv)
        nv = torch.zeros(nnz + missing.value, dtype=torch.float64, device=dv)
        gk.factorization_add_diagonal_elements_f64_i32(s, n, n, rpd, cid, vd, nc, nv, ws)
        cid, vd, nnz = nc, nv, nnz + missing.value
    lrp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    urp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    sb = gk.prefix_sum_workspace_bytes(n + 1)
    sws = torch.empty(max(sb, 8), dtype=torch.uint8, device=dv)
    gk.factorization_initialize_row_ptrs_l_u_i32(s, n, rpd, cid, lrp, urp, sws, sb)
    lnnz, unnz = int(lrp[n].item()), int(urp[n].item())  # copy_val_to_host, par_ilu.cpp:110-113
    lc = torch.zeros(lnnz, dtype=torch.int32, device=dv)
    lv = torch.zeros(lnnz, dtype=torch.float64, device=dv)
    uc = torch.zeros(unnz, dtype=torch.int32, device=dv)
    uv = torch.zeros(unnz, dtype=torch.float64, device=dv)
    gk.factorization_initialize_l_u_f64_i32(s, n, rpd, cid, vd, lrp, lc, lv, urp, uc, uv)
    tb = gk.csr_transpose_workspace_bytes(n)
    tws = torch.empty(tb, dtype=torch.uint8, device=dv)
    utrp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    utc, utv = torch.zeros_like(uc), torch.zeros_like(uv)
    gk.csr_transpose_f64_i32(s, n, n, unnz, urp, uc, uv, utrp, utc, utv, tws, tb)
    rows = torch.zeros(max(nnz, 1), dtype=torch.int32, device=dv)
    gk.convert_ptrs_to_idxs_i32(s, rpd, n, rows)
    gk.par_ilu_compute_l_u_factors_f64_i32(s, iterations, nnz, rows, cid, vd, lrp, lc, lv, utrp, utc, utv)
    urp2 = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    uc2, uv2 = torch.zeros_like(uc), torch.zeros_like(uv)
    gk.csr_transpose_f64_i32(s, n, n, unnz, utrp, utc, utv, urp2, uc2, uv2, tws, tb)
    return dict(A=(rpd, cid, vd), L=(lrp, lc, lv), U=(urp2, uc2, uv2), Ut=(utrp, utc, utv))


def oracle_par_ic(oracle, n, rp, ci, v):
    """ParIc::generate (core/factorization/par_ic.cpp:70-145) through the oracle:
    add_diagonal_elements, initialize_row_ptrs_l, initialize_l, init_factor,
    compute_factor, conj_transpose."""
    rp = rp.copy()
    nnz = int(rp[-1])
    ncols = np.zeros(nnz + n, np.int32)
    nvals = np.zeros(nnz + n)
    new_nnz = int(oracle.ref_add_diagonal_elements(n, n, rp, ci if nnz else np.zeros(1, np.int32),
                                                   v if nnz else np.zeros(1), ncols, nvals))
    ci, v = ncols[:new_nnz].copy(), nvals[:new_nnz].copy()
    lrp = np.zeros(n + 1, np.int32)
    oracle.ref_initialize_row_ptrs_l(n, rp, ci, lrp)
    lc, lv = np.zeros(lrp[-1], np.int32), np.zeros(lrp[-1])
    oracle.ref_initialize_l(n, rp, ci, v, lrp, lc, lv, 0)
    a_vals = lv.copy()
    oracle.ref_par_ic_init_factor(n, lrp, lc, lv)
    oracle.ref_par_ic_compute_factor(n, a_vals, lrp, lc, lv)
    ltrp, ltc, ltv = np.zeros(n + 1, np.int32), np.zeros(lrp[-1], np.int32), np.zeros(lrp[-1])
    oracle.ref_csr_transpose(n, n, lrp, lc, lv, ltrp, ltc, ltv)
    return dict(L=(lrp, lc, lv), Lt=(ltrp, ltc, ltv))


def gpu_par_ic(gk, torch, n, rpd, cid, vd, iterations=0):
    """The same chain on the device (rpd modified in place: pass a clone)."""
    s = torch.cuda.current_stream().cuda_stream
    dv = rpd.device
    nbytes = gk.factorization_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dv)
    missing = ctypes.c_int64(0)
    gk.factorization_count_missing_diagonal_i32(s, n, n, rpd, cid, ws, nbytes, ctypes.addressof(missing))
    nnz = int(vd.numel())
    if missing.value:
        nc = torch.zeros(nnz + missing.value, dtype=torch.int32, device=dv)
        nv = torch.zeros(nnz + missing.value, dtype=torch.float64, device=dv)
        gk.factorization_add_diagonal_elements_f64_i32(s, n, n, rpd, cid, vd, nc, nv, ws)
        cid, vd, nnz = nc, nv, nnz + missing.value
    lrp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    sb = gk.prefix_sum_workspace_bytes(n + 1)
    sws = torch.empty(max(sb, 8), dtype=torch.uint8, device=dv)
    gk.factorization_initialize_row_ptrs_l_i32(s, n, rpd, cid, lrp, sws, sb)
    lnnz = int(lrp[n].item())
    lc = torch.zeros(lnnz, dtype=torch.int32, device=dv)
    lv = torch.zeros(lnnz, dtype=torch.float64, device=dv)
    gk.factorization_initialize_l_f64_i32(s, n, rpd, cid, vd, lrp, lc, lv, 0)
    a_vals = lv.clone()
    rows = torch.zeros(max(lnnz, 1), dtype=torch.int32, device=dv)
    gk.convert_ptrs_to_idxs_i32(s, lrp, n, rows)
    gk.par_ic_init_factor_f64_i32(s, n, lrp, lc, lv)
    gk.par_ic_compute_factor_f64_i32(s, iterations, lnnz, rows, a_vals, lrp, lc, lv)
    tb = gk.csr_transpose_workspace_bytes(n)
    tws = torch.empty(tb, dtype=torch.uint8, device=dv)
    ltrp = torch.zeros(n + 1, dtype=torch.int32, device=dv)
    ltc, ltv = torch.zeros_like(lc), torch.zeros_like(lv)
    gk.csr_transpose_f64_i32(s, n, n, lnnz, lrp, lc, lv, ltrp, ltc, ltv, tws, tb)
    return dict(L=(lrp, lc, lv), Lt=(ltrp, ltc, ltv))
