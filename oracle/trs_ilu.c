/* ORACLE (test infrastructure).  Sparse triangular solves (ILU apply),
 * ParILU(0) and its setup kernels, csr::transpose. */
#include "oracle_common.h"

/* reference/solver/lower_trs_kernels.cpp:90-120 */
ORACLE_API void ref_lower_trs_solve(i64 n, i64 nrhs, const i32* row_ptrs,
                                    const i32* col_idxs, const double* vals,
                                    int unit_diag, const double* b, i64 b_stride,
                                    double* x, i64 x_stride)
{
    for (i64 j = 0; j < nrhs; ++j) {
        for (i64 row = 0; row < n; ++row) {
            double diag = 1.0;
            x[row * x_stride + j] = b[row * b_stride + j];
            for (i32 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
                const i64 col = col_idxs[k];
                if (col < row) x[row * x_stride + j] -= vals[k] * x[col * x_stride + j];
                if (col == row) diag = vals[k];
            }
            if (!unit_diag) x[row * x_stride + j] /= diag;
        }
    }
}

/* reference/solver/upper_trs_kernels.cpp:90-123 */
ORACLE_API void ref_upper_trs_solve(i64 n, i64 nrhs, const i32* row_ptrs,
                                    const i32* col_idxs, const double* vals,
                                    int unit_diag, const double* b, i64 b_stride,
                                    double* x, i64 x_stride)
{
    for (i64 j = 0; j < nrhs; ++j) {
        for (i64 inv_row = 0; inv_row < n; ++inv_row) {
            const i64 row = n - 1 - inv_row;
            double diag = 1.0;
            x[row * x_stride + j] = b[row * b_stride + j];
            for (i32 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
                const i64 col = col_idxs[k];
                if (col > row) x[row * x_stride + j] -= vals[k] * x[col * x_stride + j];
                if (col == row) diag = vals[k];
            }
            if (!unit_diag) x[row * x_stride + j] /= diag;
        }
    }
}

/* reference/matrix/csr_kernels.cpp:551-586 (transpose): trans_row_ptrs has
 * ncols + 1 entries */
ORACLE_API void ref_csr_transpose(i64 nrows, i64 ncols, const i32* row_ptrs,
                                  const i32* col_idxs, const double* vals,
                                  i32* t_row_ptrs, i32* t_col_idxs, double* t_vals)
{
    const i64 nnz = row_ptrs[nrows];
    for (i64 i = 0; i <= ncols; ++i) t_row_ptrs[i] = 0;
    for (i64 i = 0; i < nnz; ++i) t_row_ptrs[col_idxs[i] + 1]++;
    /* prefix_sum(trans_row_ptrs + 1, ncols) then convert_csr_to_csc advances
     * trans_row_ptrs[col + 1] while filling */
    i32 partial = 0;
    for (i64 i = 1; i <= ncols; ++i) {
        const i32 c = t_row_ptrs[i];
        t_row_ptrs[i] = partial;
        partial += c;
    }
    for (i64 row = 0; row < nrows; ++row) {
        for (i32 k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
            const i32 dest = t_row_ptrs[col_idxs[k] + 1]++;
            t_col_idxs[dest] = (i32)row;
            t_vals[dest] = vals[k];
        }
    }
}

/* reference/factorization/factorization_kernels.cpp:54-80 (count) + :84-160:
 * returns the new nnz; new arrays must hold old_nnz + min(nrows, ncols) */
ORACLE_API i64 ref_add_diagonal_elements(i64 nrows, i64 ncols, i32* row_ptrs,
                                         const i32* col_idxs, const double* vals,
                                         i32* new_cols, double* new_vals)
{
    i32 added = 0;
    for (i64 row = 0; row < nrows; ++row) {
        int handled = 0;
        const i32 start = row_ptrs[row], end = row_ptrs[row + 1];
        row_ptrs[row] = start + added;
        for (i32 old = start; old < end; ++old) {
            i32 nidx = old + added;
            const i32 col = col_idxs[old];
            if (!handled && col > row) {
                for (i32 t = old; t < end; ++t)
                    if (col_idxs[t] == row) handled = 1;
                if (!handled) {
                    new_vals[nidx] = 0.0;
                    new_cols[nidx] = (i32)row;
                    ++added;
                    nidx = old + added;
                    handled = 1;
                }
            }
            if (row >= ncols || col == row) handled = 1;
            new_vals[nidx] = vals[old];
            new_cols[nidx] = col;
        }
        if (row < ncols && !handled) {
            const i32 nidx = end + added;
            new_vals[nidx] = 0.0;
            new_cols[nidx] = (i32)row;
            ++added;
        }
    }
    row_ptrs[nrows] += added;
    return row_ptrs[nrows];
}

/* :166-192 */
ORACLE_API void ref_initialize_row_ptrs_l_u(i64 n, const i32* row_ptrs,
                                            const i32* col_idxs, i32* l_row_ptrs,
                                            i32* u_row_ptrs)
{
    i32 l_nnz = 0, u_nnz = 0;
    l_row_ptrs[0] = 0;
    u_row_ptrs[0] = 0;
    for (i64 row = 0; row < n; ++row) {
        for (i32 el = row_ptrs[row]; el < row_ptrs[row + 1]; ++el) {
            l_nnz += col_idxs[el] < row;
            u_nnz += col_idxs[el] > row;
        }
        l_nnz++;
        u_nnz++;
        l_row_ptrs[row + 1] = l_nnz;
        u_row_ptrs[row + 1] = u_nnz;
    }
}

/* :198-245 */
ORACLE_API void ref_initialize_l_u(i64 n, const i32* row_ptrs, const i32* col_idxs,
                                   const double* vals, const i32* l_row_ptrs,
                                   i32* l_cols, double* l_vals,
                                   const i32* u_row_ptrs, i32* u_cols,
                                   double* u_vals)
{
    for (i64 row = 0; row < n; ++row) {
        i64 il = l_row_ptrs[row];
        i64 iu = u_row_ptrs[row] + 1;
        double diag = 1.0;
        for (i32 el = row_ptrs[row]; el < row_ptrs[row + 1]; ++el) {
            const i32 col = col_idxs[el];
            if (col < row) {
                l_cols[il] = col;
                l_vals[il] = vals[el];
                ++il;
            } else if (col == row) {
                diag = vals[el];
            } else {
                u_cols[iu] = col;
                u_vals[iu] = vals[el];
                ++iu;
            }
        }
        l_cols[l_row_ptrs[row + 1] - 1] = (i32)row;
        u_cols[u_row_ptrs[row]] = (i32)row;
        l_vals[l_row_ptrs[row + 1] - 1] = 1.0;
        u_vals[u_row_ptrs[row]] = diag;
    }
}

/* reference/factorization/par_ilu_kernels.cpp:54-120: U is passed TRANSPOSED
 * (CSC of U = CSR of U^T); iterations == 0 means 1 */
ORACLE_API void ref_par_ilu_compute_l_u_factors(
    i64 iterations, i64 nnz, const i32* coo_rows, const i32* coo_cols,
    const double* coo_vals, const i32* l_row_ptrs, const i32* l_cols,
    double* l_vals, const i32* ut_row_ptrs, const i32* ut_cols, double* ut_vals)
{
    if (iterations == 0) iterations = 1;
    for (i64 iter = 0; iter < iterations; ++iter) {
        for (i64 el = 0; el < nnz; ++el) {
            const i32 row = coo_rows[el], col = coo_cols[el];
            i32 rl = l_row_ptrs[row], ru = ut_row_ptrs[col];
            double sum = coo_vals[el], last = 0.0;
            while (rl < l_row_ptrs[row + 1] && ru < ut_row_ptrs[col + 1]) {
                const i32 cl = l_cols[rl], cu = ut_cols[ru];
                if (cl == cu) {
                    last = l_vals[rl] * ut_vals[ru];
                    sum -= last;
                } else {
                    last = 0.0;
                }
                if (cl <= cu) ++rl;
                if (cu <= cl) ++ru;
            }
            sum += last;
            if (row > col) {
                const double w = sum / ut_vals[ut_row_ptrs[col + 1] - 1];
                if (isfinite(w)) l_vals[rl - 1] = w;
            } else {
                if (isfinite(sum)) ut_vals[ru - 1] = sum;
            }
        }
    }
}

/* ---- ParIC ------------------------------------------------
 * reference/factorization/factorization_kernels.cpp:251-318 (initialize_row_ptrs_l,
 * initialize_l) and reference/factorization/par_ic_kernels.cpp:55-124
 * (init_factor, compute_factor: ONE sequential sweep = exact IC(0) in row-major
 * order, whatever `iterations` says). */
ORACLE_API void ref_initialize_row_ptrs_l(i64 n, const i32* row_ptrs, const i32* col_idxs,
                                          i32* l_row_ptrs)
{
    i64 l_nnz = 0;
    l_row_ptrs[0] = 0;
    for (i64 row = 0; row < n; ++row) {
        for (i32 el = row_ptrs[row]; el < row_ptrs[row + 1]; ++el) l_nnz += col_idxs[el] < row;
        l_nnz++; /* the diagonal */
        l_row_ptrs[row + 1] = (i32)l_nnz;
    }
}

ORACLE_API void ref_initialize_l(i64 n, const i32* row_ptrs, const i32* col_idxs,
                                 const double* vals, const i32* l_row_ptrs, i32* l_col_idxs,
                                 double* l_vals, int diag_sqrt)
{
    for (i64 row = 0; row < n; ++row) {
        i64 cur = l_row_ptrs[row];
        double diag = 1.0;
        for (i32 el = row_ptrs[row]; el < row_ptrs[row + 1]; ++el) {
            const i32 col = col_idxs[el];
            if (col < row) {
                l_col_idxs[cur] = col;
                l_vals[cur] = vals[el];
                ++cur;
            } else if (col == row) {
                diag = vals[el];
            }
        }
        const i64 d = l_row_ptrs[row + 1] - 1;
        l_col_idxs[d] = (i32)row;
        if (diag_sqrt) {
            diag = sqrt(diag);
            if (!isfinite(diag)) diag = 1.0;
        }
        l_vals[d] = diag;
    }
}

ORACLE_API void ref_par_ic_init_factor(i64 n, const i32* l_row_ptrs, const i32* l_col_idxs,
                                       double* l_vals)
{
    for (i64 row = 0; row < n; ++row)
        for (i32 nz = l_row_ptrs[row]; nz < l_row_ptrs[row + 1]; ++nz)
            if (l_col_idxs[nz] == row) {
                const double d = sqrt(l_vals[nz]);
                l_vals[nz] = isfinite(d) ? d : 1.0;
            }
}

/* a_vals: the values of the lower triangle of A in L's pattern (the COO copy) */
ORACLE_API void ref_par_ic_compute_factor(i64 n, const double* a_vals, const i32* l_row_ptrs,
                                          const i32* l_col_idxs, double* l_vals)
{
    for (i64 row = 0; row < n; ++row) {
        for (i32 nz = l_row_ptrs[row]; nz < l_row_ptrs[row + 1]; ++nz) {
            const i32 col = l_col_idxs[nz];
            double sum = 0.0;
            i32 lb = l_row_ptrs[row], le = l_row_ptrs[row + 1];
            i32 hb = l_row_ptrs[col], he = l_row_ptrs[col + 1];
            while (lb < le && hb < he) {
                const i32 l_col = l_col_idxs[lb], lh_row = l_col_idxs[hb];
                if (l_col == lh_row && l_col < col) sum += l_vals[lb] * l_vals[hb];
                lb += (l_col <= lh_row);
                hb += (lh_row <= l_col);
            }
            double nv = a_vals[nz] - sum;
            if (row == col) {
                nv = sqrt(nv);
            } else {
                nv = nv / l_vals[l_row_ptrs[col + 1] - 1];
            }
            if (isfinite(nv)) l_vals[nz] = nv;
        }
    }
}

/* csr::sort_by_column_index / is_sorted_by_column_index
 * (reference/matrix/csr_kernels.cpp:969-1009); stable insertion sort (the
 * reference's std::sort leaves duplicate columns in unspecified order) */
ORACLE_API void ref_csr_sort_by_column_index(i64 nrows, const i32* row_ptrs, i32* col_idxs,
                                             double* vals)
{
    for (i64 row = 0; row < nrows; ++row) {
        for (i32 i = row_ptrs[row] + 1; i < row_ptrs[row + 1]; ++i) {
            const i32 c = col_idxs[i];
            const double v = vals[i];
            i32 j = i - 1;
            while (j >= row_ptrs[row] && col_idxs[j] > c) {
                col_idxs[j + 1] = col_idxs[j];
                vals[j + 1] = vals[j];
                --j;
            }
            col_idxs[j + 1] = c;
            vals[j + 1] = v;
        }
    }
}

ORACLE_API int ref_csr_is_sorted_by_column_index(i64 nrows, const i32* row_ptrs,
                                                 const i32* col_idxs)
{
    for (i64 row = 0; row < nrows; ++row)
        for (i32 k = row_ptrs[row] + 1; k < row_ptrs[row + 1]; ++k)
            if (col_idxs[k - 1] > col_idxs[k]) return 0;
    return 1;
}
