// Stopping-criterion kernels for gfx950.  Replaces
// gko::kernels::hip::residual_norm::residual_norm,
// implicit_residual_norm::implicit_residual_norm and
// set_all_statuses::set_all_statuses (core/stop/residual_norm_kernels.hpp,
// core/stop/criterion_kernels.hpp); semantics =
// reference/stop/residual_norm_kernels.cpp:57-126,
// reference/stop/criterion_kernels.cpp:50-60.
//
// nrhs is tiny (1..64 typically): one 256-thread block does the compare and
// the two reductions (all_converged = AND, one_changed = OR) through LDS.
#include "common.hpp"

namespace gkomi {
namespace {

constexpr int block = 256;

template <bool Implicit>
__global__ __launch_bounds__(block) void residual_norm_kernel(
    int64_t nrhs, const double* __restrict__ tau,
    const double* __restrict__ orig_tau, double goal, uint8_t id,
    bool set_finalized, uint8_t* __restrict__ stop_status,
    uint8_t* __restrict__ flags)
{
    __shared__ int s_all, s_one;
    if (threadIdx.x == 0) {
        s_all = 1;
        s_one = 0;
    }
    __syncthreads();
    int all = 1, one = 0;
    for (int64_t i = threadIdx.x; i < nrhs; i += block) {
        uint8_t st = stop_status[i];
        const double t = Implicit ? sqrt(fabs(tau[i])) : tau[i];
        if (t < goal * orig_tau[i]) {
            // stopping_status::converge (stopping_status.hpp:83-92)
            if (!status_has_stopped(st)) {
                st |= GKOMI_STATUS_CONVERGED | (id & GKOMI_STATUS_ID_MASK);
                if (set_finalized) st |= GKOMI_STATUS_FINALIZED;
                stop_status[i] = st;
            }
            one = 1;
        }
        if (!status_has_stopped(st)) all = 0;
    }
    if (!all) atomicAnd(&s_all, 0);
    if (one) atomicOr(&s_one, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        flags[0] = static_cast<uint8_t>(s_all);
        flags[1] = static_cast<uint8_t>(s_one);
    }
}

__global__ __launch_bounds__(block) void set_all_statuses_kernel(
    int64_t nrhs, uint8_t id, bool set_finalized,
    uint8_t* __restrict__ stop_status)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(block) + threadIdx.x;
         i < nrhs; i += static_cast<int64_t>(gridDim.x) * block) {
        uint8_t st = stop_status[i];
        // stopping_status::stop (stopping_status.hpp:66-75)
        if (!status_has_stopped(st)) {
            st |= (id & GKOMI_STATUS_ID_MASK);
            if (set_finalized) st |= GKOMI_STATUS_FINALIZED;
            stop_status[i] = st;
        }
    }
}

template <bool Implicit>
int launch_residual_norm(gkomi_stream_t s, int64_t nrhs, const double* tau,
                         const double* orig_tau, double goal, uint8_t id,
                         int set_finalized, uint8_t* stop_status,
                         uint8_t* device_flags, uint8_t* host_flags)
{
    if (nrhs < 0 || device_flags == nullptr) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(s);
    hipLaunchKernelGGL(residual_norm_kernel<Implicit>, dim3(1), dim3(block), 0,
                       stream, nrhs, tau, orig_tau, goal, id,
                       set_finalized != 0, stop_status, device_flags);
    int err = check_launch();
    if (err) return err;
    if (host_flags != nullptr) {
        err = static_cast<int>(hipMemcpyAsync(host_flags, device_flags, 2,
                                              hipMemcpyDeviceToHost, stream));
        if (err) return err;
        err = static_cast<int>(hipStreamSynchronize(stream));
    }
    return err;
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int gkomi_residual_norm_f64(gkomi_stream_t s, int64_t nrhs,
                                       const double* tau,
                                       const double* orig_tau,
                                       double rel_residual_goal,
                                       uint8_t stopping_id, int set_finalized,
                                       uint8_t* stop_status,
                                       uint8_t* device_flags,
                                       uint8_t* host_flags)
{
    return launch_residual_norm<false>(s, nrhs, tau, orig_tau,
                                       rel_residual_goal, stopping_id,
                                       set_finalized, stop_status,
                                       device_flags, host_flags);
}

extern "C" int gkomi_implicit_residual_norm_f64(
    gkomi_stream_t s, int64_t nrhs, const double* tau, const double* orig_tau,
    double rel_residual_goal, uint8_t stopping_id, int set_finalized,
    uint8_t* stop_status, uint8_t* device_flags, uint8_t* host_flags)
{
    return launch_residual_norm<true>(s, nrhs, tau, orig_tau,
                                      rel_residual_goal, stopping_id,
                                      set_finalized, stop_status, device_flags,
                                      host_flags);
}

extern "C" int gkomi_set_all_statuses(gkomi_stream_t s, int64_t nrhs,
                                      uint8_t stopping_id, int set_finalized,
                                      uint8_t* stop_status)
{
    if (nrhs < 0) return GKOMI_EINVAL;
    if (nrhs == 0) return GKOMI_SUCCESS;
    hipLaunchKernelGGL(set_all_statuses_kernel, dim3(grid_for(nrhs, block)),
                       dim3(block), 0, to_stream(s), nrhs, stopping_id,
                       set_finalized != 0, stop_status);
    return check_launch();
}
