/* ORACLE (test infrastructure).  Row-partitioned distributed matrix setup:
 * Partition builders and build_local_nonlocal.
 * reference/distributed/partition_kernels.cpp:42-160,
 * reference/distributed/matrix_kernels.cpp:49-190. */
#include "oracle_common.h"

/* partition_kernels.cpp:97-112: ranges has num_parts + 1 entries */
ORACLE_API void ref_partition_build_ranges_from_global_size(i64 num_parts, i64 global_size,
                                                            i64* ranges)
{
    const i64 per = global_size / num_parts;
    const i64 rest = global_size - num_parts * per;
    ranges[0] = 0;
    for (i64 i = 1; i < num_parts + 1; ++i) ranges[i] = ranges[i - 1] + per + ((i - 1) < rest ? 1 : 0);
}

/* :42-53 count_ranges + :72-92 build_from_mapping; returns num_ranges;
 * range_bounds needs n + 1 entries at most, part_ids n */
ORACLE_API i64 ref_partition_build_from_mapping(i64 n, const i32* mapping, i64* range_bounds,
                                                i32* part_ids)
{
    i64 range_idx = 0;
    i32 range_part = -1;
    for (i64 i = 0; i < n; ++i) {
        if (mapping[i] != range_part) {
            range_bounds[range_idx] = i;
            part_ids[range_idx] = mapping[i];
            range_idx++;
            range_part = mapping[i];
        }
    }
    range_bounds[range_idx] = n;
    return range_idx;
}

/* :55-68 build_from_contiguous */
ORACLE_API void ref_partition_build_from_contiguous(i64 num_parts, const i64* ranges,
                                                    i64* range_bounds, i32* part_ids)
{
    range_bounds[0] = 0;
    for (i64 i = 0; i < num_parts; ++i) {
        range_bounds[i + 1] = ranges[i + 1];
        part_ids[i] = (i32)i;
    }
}

/* :116-135 build_starting_indices: ranks[num_ranges], sizes[num_parts]; returns num_empty_parts */
ORACLE_API i64 ref_partition_build_starting_indices(const i64* range_bounds, const i32* part_ids,
                                                    i64 num_ranges, i64 num_parts, i32* ranks,
                                                    i32* sizes)
{
    for (i64 p = 0; p < num_parts; ++p) sizes[p] = 0;
    for (i64 r = 0; r < num_ranges; ++r) {
        const i32 part = part_ids[r];
        ranks[r] = sizes[part];
        sizes[part] += (i32)(range_bounds[r + 1] - range_bounds[r]);
    }
    i64 empty = 0;
    for (i64 p = 0; p < num_parts; ++p) empty += sizes[p] == 0;
    return empty;
}

static i64 find_range(i64 idx, const i64* bounds, i64 num_ranges, i64 hint)
{
    if (bounds[hint] <= idx && idx < bounds[hint + 1]) return hint;
    /* upper_bound(bounds + 1, bounds + num_ranges + 1, idx) - (bounds + 1) */
    i64 lo = 0, hi = num_ranges;
    while (lo < hi) {
        const i64 mid = (lo + hi) / 2;
        if (bounds[mid + 1] <= idx) lo = mid + 1; else hi = mid;
    }
    return lo;
}

typedef struct { i32 part; i64 col; } part_col;
static int cmp_part_col(const void* a, const void* b)
{
    const part_col* x = (const part_col*)a;
    const part_col* y = (const part_col*)b;
    if (x->part != y->part) return x->part < y->part ? -1 : 1;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    return 0;
}

/* matrix_kernels.cpp:49-190.  Output arrays must hold nnz entries; sizes_out =
 * {num_local, num_non_local, num_unique_non_local_cols}; recv_sizes has
 * num_parts entries; gather_idxs / non_local_to_global up to nnz. */
ORACLE_API void ref_dist_build_local_nonlocal(
    i64 nnz, const i64* rows, const i64* cols, const double* vals, const i64* row_bounds,
    const i32* row_part_ids, const i32* row_starts, i64 row_num_ranges, const i64* col_bounds,
    const i32* col_part_ids, const i32* col_starts, i64 col_num_ranges, i64 num_parts,
    i32 local_part, i32* l_rows, i32* l_cols, double* l_vals, i32* nl_rows, i32* nl_cols,
    double* nl_vals, i32* gather_idxs, i32* recv_sizes, i64* non_local_to_global, i64* sizes_out)
{
    i64 nl = 0, nn = 0, rr = 0, cr = 0;
    i64* nn_gcol = (i64*)malloc(sizeof(i64) * (size_t)(nnz + 1));
    for (i64 i = 0; i < nnz; ++i) {
        rr = find_range(rows[i], row_bounds, row_num_ranges, rr);
        if (row_part_ids[rr] != local_part) continue;
        const i32 lrow = (i32)(rows[i] - row_bounds[rr]) + row_starts[rr];
        cr = find_range(cols[i], col_bounds, col_num_ranges, cr);
        if (col_part_ids[cr] == local_part) {
            l_rows[nl] = lrow;
            l_cols[nl] = (i32)(cols[i] - col_bounds[cr]) + col_starts[cr];
            l_vals[nl] = vals[i];
            ++nl;
        } else {
            nl_rows[nn] = lrow;
            nn_gcol[nn] = cols[i];
            nl_vals[nn] = vals[i];
            ++nn;
        }
    }
    part_col* uc = (part_col*)malloc(sizeof(part_col) * (size_t)(nn + 1));
    for (i64 i = 0; i < nn; ++i) {
        uc[i].col = nn_gcol[i];
        uc[i].part = col_part_ids[find_range(nn_gcol[i], col_bounds, col_num_ranges, 0)];
    }
    qsort(uc, (size_t)nn, sizeof(part_col), cmp_part_col);
    i64 nu = 0;
    for (i64 i = 0; i < nn; ++i)
        if (i == 0 || uc[i].col != uc[i - 1].col || uc[i].part != uc[i - 1].part) uc[nu++] = uc[i];
    for (i64 i = 0; i < nu; ++i) non_local_to_global[i] = uc[i].col;
    for (i64 i = 0; i < nn; ++i) {
        part_col key;
        key.col = nn_gcol[i];
        key.part = col_part_ids[find_range(nn_gcol[i], col_bounds, col_num_ranges, 0)];
        const part_col* hit = (const part_col*)bsearch(&key, uc, (size_t)nu, sizeof(part_col), cmp_part_col);
        nl_cols[i] = (i32)(hit - uc);
    }
    for (i64 p = 0; p < num_parts; ++p) recv_sizes[p] = 0;
    for (i64 i = 0; i < nu; ++i) {
        const i64 r = find_range(uc[i].col, col_bounds, col_num_ranges, 0);
        gather_idxs[i] = (i32)(uc[i].col - col_bounds[r]) + col_starts[r];
        recv_sizes[uc[i].part]++;
    }
    sizes_out[0] = nl;
    sizes_out[1] = nn;
    sizes_out[2] = nu;
    free(nn_gcol);
    free(uc);
}

/* partition_kernels.cpp:139-155 has_ordered_parts: 1 if the part ids of consecutive ranges never decrease */
ORACLE_API i64 ref_partition_has_ordered_parts(const i32* part_ids, i64 num_ranges)
{
    for (i64 i = 1; i < num_ranges; ++i) {
        if (part_ids[i] < part_ids[i - 1]) return 0;
    }
    return 1;
}

/* reference/distributed/vector_kernels.cpp:47-96 build_local: local(local row, col) = value for the entries whose row
 * local_part owns, in input order (a later entry of the same position overwrites an earlier one); `local` is
 * row-major with `stride`, the caller has filled it (the reference's callers: with zeros) */
ORACLE_API void ref_dist_vector_build_local(i64 nnz, const i64* rows, const i64* cols, const double* vals,
                                            const i64* bounds, const i32* part_ids, const i32* starts,
                                            i64 num_ranges, i32 local_part, double* local, i64 stride)
{
    i64 hint = 0;
    for (i64 i = 0; i < nnz; ++i) {
        hint = find_range(rows[i], bounds, num_ranges, hint);
        if (part_ids[hint] != local_part) continue;
        const i64 lrow = (rows[i] - bounds[hint]) + starts[hint];
        local[lrow * stride + cols[i]] = vals[i];
    }
}
