/* ORACLE (test infrastructure).  device_matrix_data kernels:
 * reference/base/device_matrix_data_kernels.cpp:84-190 (remove_zeros,
 * sum_duplicates, sort_row_major).  No fixed vectors exist for these in the
 * reference (test/base/device_matrix_data_kernels.cpp builds seeded random
 * triplets and checks properties); tests/test_oracle_assembly.py follows the
 * same recipe. */
#include "oracle_common.h"

/* :84-110: stable compaction of the entries with value != 0; returns the
 * count.  out arrays have capacity n. */
ORACLE_API i64 ref_matrix_data_remove_zeros(i64 n, const i32* rows, const i32* cols,
                                            const double* vals, i32* out_rows, i32* out_cols,
                                            double* out_vals)
{
    i64 out = 0;
    for (i64 i = 0; i < n; ++i) {
        if (vals[i] != 0.0) { /* is_nonzero */
            out_rows[out] = rows[i];
            out_cols[out] = cols[i];
            out_vals[out] = vals[i];
            ++out;
        }
    }
    return out;
}

/* :116-160: input sorted by (row, col); every run of equal (row, col) becomes
 * one entry whose value is 0 + v1 + v2 + ... in storage order */
ORACLE_API i64 ref_matrix_data_sum_duplicates(i64 n, const i32* rows, const i32* cols,
                                              const double* vals, i32* out_rows, i32* out_cols,
                                              double* out_vals)
{
    i32 row = -1, col = -1; /* invalid_index */
    i64 out = -1;
    for (i64 i = 0; i < n; ++i) {
        if (row != rows[i] || col != cols[i]) {
            row = rows[i];
            col = cols[i];
            ++out;
            out_rows[out] = row;
            out_cols[out] = col;
            out_vals[out] = 0.0;
        }
        out_vals[out] += vals[i];
    }
    return out + 1;
}

/* :166-176: sort by (row, column) (matrix_data_entry::operator<,
 * include/ginkgo/core/base/matrix_data.hpp).  The reference calls std::sort,
 * which leaves the order of entries with equal keys unspecified; this is the
 * STABLE member of that family (bottom-up merge sort), the order the GPU radix
 * sort produces as well. */
static int entry_less(i32 r1, i32 c1, i32 r2, i32 c2) { return r1 < r2 || (r1 == r2 && c1 < c2); }

ORACLE_API void ref_matrix_data_sort_row_major(i64 n, i32* rows, i32* cols, double* vals)
{
    if (n < 2) return;
    i64* idx = (i64*)malloc(sizeof(i64) * (size_t)n);
    i64* tmp = (i64*)malloc(sizeof(i64) * (size_t)n);
    for (i64 i = 0; i < n; ++i) idx[i] = i;
    for (i64 width = 1; width < n; width *= 2) {
        for (i64 lo = 0; lo < n; lo += 2 * width) {
            const i64 mid = lo + width < n ? lo + width : n;
            const i64 hi = lo + 2 * width < n ? lo + 2 * width : n;
            i64 a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only if strictly smaller: stable */
                if (entry_less(rows[idx[b]], cols[idx[b]], rows[idx[a]], cols[idx[a]])) {
                    tmp[o++] = idx[b++];
                } else {
                    tmp[o++] = idx[a++];
                }
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        i64* swap = idx;
        idx = tmp;
        tmp = swap;
    }
    i32* r2 = (i32*)malloc(sizeof(i32) * (size_t)n);
    i32* c2 = (i32*)malloc(sizeof(i32) * (size_t)n);
    double* v2 = (double*)malloc(sizeof(double) * (size_t)n);
    for (i64 i = 0; i < n; ++i) {
        r2[i] = rows[idx[i]];
        c2[i] = cols[idx[i]];
        v2[i] = vals[idx[i]];
    }
    memcpy(rows, r2, sizeof(i32) * (size_t)n);
    memcpy(cols, c2, sizeof(i32) * (size_t)n);
    memcpy(vals, v2, sizeof(double) * (size_t)n);
    free(idx);
    free(tmp);
    free(r2);
    free(c2);
    free(v2);
}
