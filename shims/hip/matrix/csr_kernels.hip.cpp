// hip/matrix/csr_kernels.hip.cpp of a reference tree that binds libgkomi.so:
// csr::spmv / csr::advanced_spmv (core/matrix/csr_kernels.hpp:58-75).
#include "../gkomi_bindings.hpp"

#ifndef GKOMI_SROW_IS_OURS
#define GKOMI_SROW_IS_OURS false  // the host mirror's Csr::make_srow always builds ours (shims/test/prelude_mirror.hpp)
#endif

namespace gko {
namespace kernels {
namespace hip {
namespace csr {

// strategy objects keep their API (csr.hpp:170-705); their name selects the kernel.  "gkomi_split" is the
// strategy object of INTEGRATION.md (a user-defined strategy_type, the reference's own extension point): its
// process() fills Csr::srow_ with gkomi_csr_make_srow_i32's tile start rows and keeps the longest row.
inline int strategy_code(const matrix::Csr<double, int32>* a)
{
    const auto name = a->get_strategy()->get_name();
    return name == "classical"      ? GKOMI_CSR_VECTOR
           : name == "load_balance" ? GKOMI_CSR_BALANCED
           : name == "merge_path"   ? GKOMI_CSR_STREAM
                                    : GKOMI_CSR_AUTO;  // automatical, sparselib, cusparse, gkomi_split
}

// Csr::srow_ holds OUR tile start rows only when OUR strategy filled it: load_balance / automatical fill it with
// their own warp bounds (csr.hpp:421-470), which the split kernel must never be handed
inline const int32* split_srow(const matrix::Csr<double, int32>* a)
{
    const bool ours = a->get_strategy()->get_name() == "gkomi_split" || GKOMI_SROW_IS_OURS;
    return ours && a->get_num_srow_elements() ? a->get_const_srow() : nullptr;
}

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int32>* a,
          const matrix::Dense<double>* b, matrix::Dense<double>* c)
{
    // the row statistic of the strategy object is the SpMV's hint, like Csr::automatical uses it
    // (csr.hpp:526-705); srow only when it is ours (split_srow above)
    GKOMI_CALL(gkomi_csr_spmv_srow_f64_i32(
        GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
        a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
        b->get_stride(), c->get_values(), c->get_stride(), nullptr, nullptr, strategy_code(a),
        gkomi_row_hint(a), split_srow(a), gkomi_csr_srow_tile_for(static_cast<int64_t>(a->get_num_stored_elements()))));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha,
                   const matrix::Csr<double, int32>* a, const matrix::Dense<double>* b,
                   const matrix::Dense<double>* beta, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_csr_spmv_srow_f64_i32(
        GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
        a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
        b->get_stride(), c->get_values(), c->get_stride(), alpha->get_const_values(), beta->get_const_values(),
        strategy_code(a), gkomi_row_hint(a), split_srow(a), gkomi_csr_srow_tile_for(static_cast<int64_t>(a->get_num_stored_elements()))));
}

// ---- <double, int64> (GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE, include/ginkgo/core/base/types.hpp:544-560):
// the instantiation of matrices with more than 2^31 nonzeros.  The automatic strategy of gkomi_csr_spmv_srow_f64_i64
// serves every strategy name (split kernel with our srow, row-cut stream kernel otherwise: bit-exact for any rows).
inline const int64* split_srow(const matrix::Csr<double, int64>* a)
{
    const bool ours = a->get_strategy()->get_name() == "gkomi_split" || GKOMI_SROW_IS_OURS;
    return ours && a->get_num_srow_elements() ? a->get_const_srow() : nullptr;
}

void spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Csr<double, int64>* a,
          const matrix::Dense<double>* b, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_csr_spmv_srow_f64_i64(
        GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
        a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
        b->get_stride(), c->get_values(), c->get_stride(), nullptr, nullptr, GKOMI_CSR_AUTO, -1, split_srow(a),
        gkomi_csr_srow_tile_for(static_cast<int64_t>(a->get_num_stored_elements()))));
}

void advanced_spmv(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* alpha,
                   const matrix::Csr<double, int64>* a, const matrix::Dense<double>* b,
                   const matrix::Dense<double>* beta, matrix::Dense<double>* c)
{
    GKOMI_CALL(gkomi_csr_spmv_srow_f64_i64(
        GKOMI_NULL_STREAM, a->get_size()[0], a->get_size()[1], b->get_size()[1], a->get_num_stored_elements(),
        a->get_const_row_ptrs(), a->get_const_col_idxs(), a->get_const_values(), b->get_const_values(),
        b->get_stride(), c->get_values(), c->get_stride(), alpha->get_const_values(), beta->get_const_values(),
        GKOMI_CSR_AUTO, -1, split_srow(a), gkomi_csr_srow_tile_for(static_cast<int64_t>(a->get_num_stored_elements()))));
}

}  // namespace csr
}  // namespace hip
}  // namespace kernels
}  // namespace gko
