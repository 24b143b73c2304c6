// hip/base/device_matrix_data_kernels.hip.cpp: components::{remove_zeros, sum_duplicates, sort_row_major}
// (core/base/device_matrix_data_kernels.hpp; reference/base/device_matrix_data_kernels.cpp:84-190).  The two
// compacting kernels size their own outputs in the reference (array::resize_and_reset); here through a
// count returned by the C ABI call.  soa_to_aos / aos_to_soa are host-format conversions, not on the path.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace components {

void sort_row_major(std::shared_ptr<const HipExecutor> exec, device_matrix_data<double, int32>& data)
{
    array<char> tmp(exec, gkomi_matrix_data_workspace_bytes(static_cast<int64_t>(data.get_num_elems())));
    GKOMI_CALL(gkomi_matrix_data_sort_row_major_f64_i32(GKOMI_NULL_STREAM, static_cast<int64_t>(data.get_num_elems()), data.get_row_idxs(),
                                                        data.get_col_idxs(), data.get_values(), tmp.get_data(), tmp.get_num_elems()));
}

namespace {
using compact_fn = int (*)(gkomi_stream_t, int64_t, const int32_t*, const int32_t*, const double*, int32_t*, int32_t*, double*, void*, size_t,
                           int64_t*);
inline void compact(std::shared_ptr<const HipExecutor> exec, compact_fn fn, array<double>& values, array<int32>& row_idxs,
                    array<int32>& col_idxs)
{
    const int64_t nnz = static_cast<int64_t>(values.get_num_elems());
    array<char> tmp(exec, gkomi_matrix_data_workspace_bytes(nnz));
    array<int32> out_rows(exec, static_cast<size_type>(nnz)), out_cols(exec, static_cast<size_type>(nnz));
    array<double> out_vals(exec, static_cast<size_type>(nnz));
    int64_t kept = 0;
    GKOMI_CALL(fn(GKOMI_NULL_STREAM, nnz, row_idxs.get_const_data(), col_idxs.get_const_data(), values.get_const_data(), out_rows.get_data(),
                  out_cols.get_data(), out_vals.get_data(), tmp.get_data(), tmp.get_num_elems(), &kept));
    if (kept == nnz) return;  // nothing dropped: the inputs stay as they are (reference: `if (nnz < size)`)
    row_idxs.resize_and_reset(static_cast<size_type>(kept));
    col_idxs.resize_and_reset(static_cast<size_type>(kept));
    values.resize_and_reset(static_cast<size_type>(kept));
    exec->copy(static_cast<size_type>(kept), out_rows.get_const_data(), row_idxs.get_data());
    exec->copy(static_cast<size_type>(kept), out_cols.get_const_data(), col_idxs.get_data());
    exec->copy(static_cast<size_type>(kept), out_vals.get_const_data(), values.get_data());
}
}  // namespace

void remove_zeros(std::shared_ptr<const HipExecutor> exec, array<double>& values, array<int32>& row_idxs, array<int32>& col_idxs)
{
    compact(exec, gkomi_matrix_data_remove_zeros_f64_i32, values, row_idxs, col_idxs);
}

void sum_duplicates(std::shared_ptr<const HipExecutor> exec, size_type num_rows, array<double>& values, array<int32>& row_idxs,
                    array<int32>& col_idxs)
{
    // the caller has sorted the entries (core/base/device_matrix_data.cpp:124-131)
    compact(exec, gkomi_matrix_data_sum_duplicates_f64_i32, values, row_idxs, col_idxs);
}

}  // namespace components
}  // namespace hip
}  // namespace kernels
}  // namespace gko
