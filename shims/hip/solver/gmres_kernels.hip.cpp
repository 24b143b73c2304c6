// hip/solver/common_gmres_kernels.hip.cpp + gmres_kernels.hip.cpp: common_gmres::{initialize, hessenberg_qr,
// solve_krylov} and gmres::{restart, multi_axpy} (core/solver/common_gmres_kernels.hpp, gmres_kernels.hpp).
// The Dense arguments are contiguous workspace matrices (stride == #columns) in Gmres::apply_dense_impl
// (core/solver/gmres.cpp:139-372); a padded one is refused rather than misread.
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace {
inline void contiguous(const matrix::Dense<double>* m, const char* what)
{
    if (m->get_stride() != m->get_size()[1]) GKO_NOT_SUPPORTED(what);
}
inline uint8_t* raw(stopping_status* s) { return reinterpret_cast<uint8_t*>(s); }
inline const uint8_t* raw(const stopping_status* s) { return reinterpret_cast<const uint8_t*>(s); }
static_assert(sizeof(size_type) == sizeof(uint64_t), "final_iter_nums are 64-bit");
}  // namespace

namespace common_gmres {

void initialize(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* b, matrix::Dense<double>* residual,
                matrix::Dense<double>* givens_sin, matrix::Dense<double>* givens_cos, stopping_status* stop_status)
{
    contiguous(givens_sin, "common_gmres::initialize: padded givens_sin");
    contiguous(givens_cos, "common_gmres::initialize: padded givens_cos");
    GKOMI_CALL(gkomi_gmres_initialize_f64(GKOMI_NULL_STREAM, b->get_size()[0], b->get_size()[1], givens_sin->get_size()[0],
                                          b->get_const_values(), b->get_stride(), residual->get_values(), residual->get_stride(),
                                          givens_sin->get_values(), givens_cos->get_values(), raw(stop_status)));
}

void hessenberg_qr(std::shared_ptr<const HipExecutor> exec, matrix::Dense<double>* givens_sin, matrix::Dense<double>* givens_cos,
                   matrix::Dense<double>* residual_norm, matrix::Dense<double>* residual_norm_collection,
                   matrix::Dense<double>* hessenberg_iter, size_type iter, size_type* final_iter_nums,
                   const stopping_status* stop_status)
{
    contiguous(givens_sin, "common_gmres::hessenberg_qr: padded givens_sin");
    contiguous(residual_norm_collection, "common_gmres::hessenberg_qr: padded residual_norm_collection");
    GKOMI_CALL(gkomi_gmres_hessenberg_qr_f64(GKOMI_NULL_STREAM, residual_norm->get_size()[1], givens_sin->get_values(),
                                             givens_cos->get_values(), residual_norm->get_values(), residual_norm_collection->get_values(),
                                             hessenberg_iter->get_values(), hessenberg_iter->get_stride(), static_cast<int64_t>(iter),
                                             reinterpret_cast<uint64_t*>(final_iter_nums), raw(stop_status)));
}

void solve_krylov(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* residual_norm_collection,
                  const matrix::Dense<double>* hessenberg, matrix::Dense<double>* y, const size_type* final_iter_nums,
                  const stopping_status* stop_status)
{
    contiguous(residual_norm_collection, "common_gmres::solve_krylov: padded residual_norm_collection");
    contiguous(y, "common_gmres::solve_krylov: padded y");
    GKOMI_CALL(gkomi_gmres_solve_krylov_f64(GKOMI_NULL_STREAM, residual_norm_collection->get_size()[1],
                                            residual_norm_collection->get_const_values(), hessenberg->get_const_values(),
                                            hessenberg->get_stride(), y->get_values(), reinterpret_cast<const uint64_t*>(final_iter_nums),
                                            raw(stop_status)));
}

}  // namespace common_gmres

namespace gmres {

void restart(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* residual, const matrix::Dense<double>* residual_norm,
             matrix::Dense<double>* residual_norm_collection, matrix::Dense<double>* krylov_bases, size_type* final_iter_nums)
{
    contiguous(residual_norm_collection, "gmres::restart: padded residual_norm_collection");
    GKOMI_CALL(gkomi_gmres_restart_f64(GKOMI_NULL_STREAM, residual->get_size()[0], residual->get_size()[1], residual->get_const_values(),
                                       residual->get_stride(), residual_norm->get_const_values(), residual_norm_collection->get_values(),
                                       krylov_bases->get_values(), krylov_bases->get_stride(), reinterpret_cast<uint64_t*>(final_iter_nums)));
}

void multi_axpy(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* krylov_bases, const matrix::Dense<double>* y,
                matrix::Dense<double>* before_preconditioner, const size_type* final_iter_nums, stopping_status* stop_status)
{
    contiguous(y, "gmres::multi_axpy: padded y");
    GKOMI_CALL(gkomi_gmres_multi_axpy_f64(GKOMI_NULL_STREAM, before_preconditioner->get_size()[0], before_preconditioner->get_size()[1],
                                          krylov_bases->get_const_values(), krylov_bases->get_stride(), y->get_const_values(),
                                          before_preconditioner->get_values(), before_preconditioner->get_stride(),
                                          reinterpret_cast<const uint64_t*>(final_iter_nums), raw(stop_status)));
}

}  // namespace gmres
}  // namespace hip
}  // namespace kernels
}  // namespace gko
