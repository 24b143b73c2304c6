// hip/stop/residual_norm_kernels.hip.cpp: residual_norm / implicit_residual_norm
// (core/stop/residual_norm_kernels.hpp; reference/stop/residual_norm_kernels.cpp:57-126).
#include "../gkomi_bindings.hpp"

namespace gko {
namespace kernels {
namespace hip {
namespace residual_norm {

void residual_norm(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* tau, const matrix::Dense<double>* orig_tau,
                   double rel_residual_goal, uint8 stoppingId, bool setFinalized, array<stopping_status>* stop_status,
                   array<bool>* device_storage, bool* all_converged, bool* one_changed)
{
    uint8_t host[2] = {0, 0};  // the two blocking 1-byte copies of hip/stop/residual_norm_kernels.hip.cpp:119-120, in one
    GKOMI_CALL(gkomi_residual_norm_f64(GKOMI_NULL_STREAM, tau->get_size()[1], tau->get_const_values(), orig_tau->get_const_values(),
                                       rel_residual_goal, stoppingId, setFinalized, reinterpret_cast<uint8_t*>(stop_status->get_data()),
                                       reinterpret_cast<uint8_t*>(device_storage->get_data()), host));
    *all_converged = host[0] != 0;
    *one_changed = host[1] != 0;
}

}  // namespace residual_norm

namespace implicit_residual_norm {

void implicit_residual_norm(std::shared_ptr<const HipExecutor> exec, const matrix::Dense<double>* tau,
                            const matrix::Dense<double>* orig_tau, double rel_residual_goal, uint8 stoppingId, bool setFinalized,
                            array<stopping_status>* stop_status, array<bool>* device_storage, bool* all_converged, bool* one_changed)
{
    uint8_t host[2] = {0, 0};
    GKOMI_CALL(gkomi_implicit_residual_norm_f64(GKOMI_NULL_STREAM, tau->get_size()[1], tau->get_const_values(), orig_tau->get_const_values(),
                                                rel_residual_goal, stoppingId, setFinalized, reinterpret_cast<uint8_t*>(stop_status->get_data()),
                                                reinterpret_cast<uint8_t*>(device_storage->get_data()), host));
    *all_converged = host[0] != 0;
    *one_changed = host[1] != 0;
}

}  // namespace implicit_residual_norm
}  // namespace hip
}  // namespace kernels
}  // namespace gko
