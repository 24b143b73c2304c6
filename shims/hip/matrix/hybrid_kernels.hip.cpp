// hip/matrix/hybrid_kernels.hip.cpp: hybrid::compute_coo_row_ptrs (core/matrix/hybrid_kernels.hpp).
// Hybrid::apply itself is composed in core/ from ell::spmv and coo::spmv2 (core/matrix/hybrid.cpp:133-159), so
// it needs no kernel of its own at this boundary; gkomi_hybrid_spmv_f64_i32 is that composition for C callers.
// compute_coo_row_ptrs takes the row lengths as size_type in the reference; the C ABI's kernel reads the int32
// row pointers of the CSR source (core/matrix/csr.cpp:438-470 is its only caller), so this binding turns the lengths
// back into pointers (setup path, O(rows) on the host) and serves matrices whose pointers fit 32 bits.
#include "../gkomi_bindings.hpp"

#include <vector>

namespace gko {
namespace kernels {
namespace hip {
namespace hybrid {

void compute_coo_row_ptrs(std::shared_ptr<const HipExecutor> exec, const array<size_type>& row_nnz, size_type ell_lim,
                          int64* coo_row_ptrs)
{
    // row lengths -> int32 row pointers on the host (setup path, O(rows)), then the device kernel
    const size_type n = row_nnz.get_num_elems();
    array<size_type> host_nnz(exec->get_master(), row_nnz);
    std::vector<int32> ptrs(n + 1, 0);
    for (size_type i = 0; i < n; ++i) {
        const size_type next = static_cast<size_type>(ptrs[i]) + host_nnz.get_const_data()[i];
        if (next > 0x7fffffffu) GKO_NOT_SUPPORTED("hybrid::compute_coo_row_ptrs: more than 2^31 - 1 nonzeros");
        ptrs[i + 1] = static_cast<int32>(next);
    }
    array<int32> dev_ptrs(exec, ptrs.begin(), ptrs.end());
    array<char> tmp(exec, gkomi_prefix_sum_workspace_bytes(static_cast<int64_t>(n) + 1));
    GKOMI_CALL(gkomi_hybrid_compute_coo_row_ptrs_i32(GKOMI_NULL_STREAM, dev_ptrs.get_const_data(), static_cast<int64_t>(n),
                                                     static_cast<int64_t>(ell_lim), reinterpret_cast<int64_t*>(coo_row_ptrs), tmp.get_data(),
                                                     tmp.get_num_elems()));
}

}  // namespace hybrid
}  // namespace hip
}  // namespace kernels
}  // namespace gko
