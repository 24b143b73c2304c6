// hip/distributed/{matrix,partition,vector}_kernels.hip.cpp: distributed_matrix::build_local_nonlocal,
// partition::build_starting_indices, distributed_vector::build_local (core/distributed/*_kernels.hpp;
// reference/distributed/matrix_kernels.cpp:49-190, partition_kernels.cpp:114-160, vector_kernels.cpp).
// <double, int32, int64>.  A Partition's arrays live on the executor in a reference tree (partition.hpp:300-340):
// get_range_bounds() / get_part_ids() / get_range_starting_indices() are device pointers there.
#include "../gkomi_bindings.hpp"

#include <vector>

namespace gko {
namespace kernels {
namespace hip {
namespace distributed_matrix {

using part = experimental::distributed::Partition<int32, int64>;
using experimental::distributed::comm_index_type;

void build_local_nonlocal(std::shared_ptr<const HipExecutor> exec, const device_matrix_data<double, int64>& input, const part* row_partition,
                          const part* col_partition, comm_index_type local_part, array<int32>& local_row_idxs, array<int32>& local_col_idxs,
                          array<double>& local_values, array<int32>& non_local_row_idxs, array<int32>& non_local_col_idxs,
                          array<double>& non_local_values, array<int32>& local_gather_idxs, array<comm_index_type>& recv_sizes,
                          array<int64>& non_local_to_global)
{
    const int64_t nnz = static_cast<int64_t>(input.get_num_elems());
    array<char> tmp(exec, gkomi_dist_build_workspace_bytes(nnz));
    int64_t sizes[3] = {};
    GKOMI_CALL(gkomi_dist_build_local_nonlocal_sizes(
        GKOMI_NULL_STREAM, nnz, reinterpret_cast<const int64_t*>(input.get_const_row_idxs()),
        reinterpret_cast<const int64_t*>(input.get_const_col_idxs()), reinterpret_cast<const int64_t*>(row_partition->get_range_bounds()),
        row_partition->get_part_ids(), row_partition->get_range_starting_indices(), static_cast<int64_t>(row_partition->get_num_ranges()),
        reinterpret_cast<const int64_t*>(col_partition->get_range_bounds()), col_partition->get_part_ids(),
        col_partition->get_range_starting_indices(), static_cast<int64_t>(col_partition->get_num_ranges()), local_part, tmp.get_data(),
        tmp.get_num_elems(), sizes));
    local_row_idxs.resize_and_reset(static_cast<size_type>(sizes[0]));
    local_col_idxs.resize_and_reset(static_cast<size_type>(sizes[0]));
    local_values.resize_and_reset(static_cast<size_type>(sizes[0]));
    non_local_row_idxs.resize_and_reset(static_cast<size_type>(sizes[1]));
    non_local_col_idxs.resize_and_reset(static_cast<size_type>(sizes[1]));
    non_local_values.resize_and_reset(static_cast<size_type>(sizes[1]));
    local_gather_idxs.resize_and_reset(static_cast<size_type>(sizes[2]));
    non_local_to_global.resize_and_reset(static_cast<size_type>(sizes[2]));
    GKOMI_CALL(gkomi_dist_build_local_nonlocal_fill(
        GKOMI_NULL_STREAM, nnz, reinterpret_cast<const int64_t*>(input.get_const_row_idxs()),
        reinterpret_cast<const int64_t*>(input.get_const_col_idxs()), input.get_const_values(),
        reinterpret_cast<const int64_t*>(row_partition->get_range_bounds()), row_partition->get_part_ids(),
        row_partition->get_range_starting_indices(), static_cast<int64_t>(row_partition->get_num_ranges()),
        reinterpret_cast<const int64_t*>(col_partition->get_range_bounds()), col_partition->get_part_ids(),
        col_partition->get_range_starting_indices(), static_cast<int64_t>(col_partition->get_num_ranges()),
        static_cast<int64_t>(col_partition->get_num_parts()), tmp.get_const_data(), sizes[2], local_row_idxs.get_data(),
        local_col_idxs.get_data(), local_values.get_data(), non_local_row_idxs.get_data(), non_local_col_idxs.get_data(),
        non_local_values.get_data(), local_gather_idxs.get_data(), recv_sizes.get_data(),
        reinterpret_cast<int64_t*>(non_local_to_global.get_data())));
}

}  // namespace distributed_matrix

namespace partition {

// O(#ranges) metadata: the reference's device kernels sort and scan a few ranges; here on the host copy
void build_starting_indices(std::shared_ptr<const HipExecutor> exec, const int64* range_offsets, const int* range_parts,
                            size_type num_ranges, experimental::distributed::comm_index_type num_parts,
                            experimental::distributed::comm_index_type& num_empty_parts, int32* ranks, int32* sizes)
{
    auto host = exec->get_master();
    std::vector<int64_t> offsets(num_ranges + 1);
    std::vector<int32_t> parts(num_ranges), starts(num_ranges), part_sizes(static_cast<size_type>(num_parts));
    host->copy_from(exec.get(), num_ranges + 1, reinterpret_cast<const int64_t*>(range_offsets), offsets.data());
    host->copy_from(exec.get(), num_ranges, range_parts, parts.data());
    int64_t empty = 0;
    GKOMI_CALL(gkomi_partition_build_starting_indices(offsets.data(), parts.data(), static_cast<int64_t>(num_ranges), num_parts, starts.data(),
                                                      part_sizes.data(), &empty));
    num_empty_parts = static_cast<experimental::distributed::comm_index_type>(empty);
    exec->copy_from(host.get(), num_ranges, starts.data(), ranks);
    exec->copy_from(host.get(), static_cast<size_type>(num_parts), part_sizes.data(), sizes);
}

// core/distributed/partition_kernels.hpp GKO_DECLARE_PARTITION_IS_ORDERED (reference/distributed/partition_kernels.cpp:139-155)
void has_ordered_parts(std::shared_ptr<const HipExecutor> exec, const experimental::distributed::Partition<int32, int64>* partition, bool* result)
{
    const auto num_ranges = partition->get_num_ranges();
    std::vector<int32_t> parts(num_ranges);
    exec->get_master()->copy_from(exec.get(), num_ranges, partition->get_part_ids(), parts.data());
    int64_t ordered = 0;
    GKOMI_CALL(gkomi_partition_has_ordered_parts(parts.data(), static_cast<int64_t>(num_ranges), &ordered));
    *result = ordered != 0;
}

}  // namespace partition

namespace distributed_vector {

// core/distributed/vector_kernels.hpp GKO_DECLARE_DISTRIBUTED_VECTOR_BUILD_LOCAL (reference/distributed/vector_kernels.cpp:47-96)
void build_local(std::shared_ptr<const HipExecutor> exec, const device_matrix_data<double, int64>& input,
                 const experimental::distributed::Partition<int32, int64>* partition, experimental::distributed::comm_index_type local_part,
                 matrix::Dense<double>* local_mtx)
{
    GKOMI_CALL(gkomi_dist_vector_build_local_f64(
        GKOMI_NULL_STREAM, static_cast<int64_t>(input.get_num_elems()), reinterpret_cast<const int64_t*>(input.get_const_row_idxs()),
        reinterpret_cast<const int64_t*>(input.get_const_col_idxs()), input.get_const_values(),
        reinterpret_cast<const int64_t*>(partition->get_range_bounds()), partition->get_part_ids(), partition->get_range_starting_indices(),
        static_cast<int64_t>(partition->get_num_ranges()), local_part, local_mtx->get_values(), static_cast<int64_t>(local_mtx->get_stride())));
}

}  // namespace distributed_vector
}  // namespace hip
}  // namespace kernels
}  // namespace gko
