// Shared device/host helpers for the gfx950 kernels.  Wave = 64 lanes,
// hard-coded (cdna_hip_programming.md §1).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gkomi.h"

namespace gkomi {

constexpr int wave_size = 64;
// memory-bound elementwise kernels: cap the grid and grid-stride
// (cdna_hip_programming.md Guideline 11: 256 CUs x 8 blocks)
constexpr int max_stream_blocks = 2048;

inline hipStream_t to_stream(gkomi_stream_t s)
{
    return reinterpret_cast<hipStream_t>(s);
}

inline int check_launch()
{
    return static_cast<int>(hipGetLastError());
}

inline int64_t ceildiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline int grid_for(int64_t work_items, int block, int64_t cap = max_stream_blocks)
{
    int64_t g = ceildiv(work_items, block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

// stopping_status helpers (include/ginkgo/core/stop/stopping_status.hpp)
__device__ __forceinline__ bool status_has_stopped(uint8_t s)
{
    return (s & GKOMI_STATUS_ID_MASK) != 0;
}

// The same question for a status byte every lane of the workgroup asks about, answered through the SCALAR cache: the
// aligned word around the byte is a uniform, read-only load the compiler turns into s_load_dword (gfx950 has no scalar
// byte load, so status[0] itself goes down the vector-memory pipe and queues behind the streaming loads of the other
// workgroups on the compute unit -- about 1 us in front of every workgroup of a bandwidth-bound launch,
// profiles/r03_p3_cg_kernels.md).  Status arrays are device allocations, so the word around a byte is inside them.
__device__ __forceinline__ bool status_has_stopped_uniform(const uint8_t* __restrict__ status)
{
    // (constant address space = same addresses as global, not written while the kernel runs: what makes the load
    // eligible for the scalar unit after the detour through an integer)
    using scalar_word = const uint32_t __attribute__((address_space(4)));
    const uintptr_t at = reinterpret_cast<uintptr_t>(status);
    const uint32_t word = *reinterpret_cast<scalar_word*>(at & ~uintptr_t{3});
    return status_has_stopped(static_cast<uint8_t>(word >> (8 * (at & 3))));
}

// full-wave sum via DPP/shuffles; every lane gets the total.  The order of
// the tree is fixed, so results are run-to-run reproducible.
__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, 64);
    }
    return v;
}

template <int Width>
__device__ __forceinline__ double subwave_reduce_sum(double v)
{
#pragma unroll
    for (int off = Width / 2; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, 64);
    }
    return v;
}

// Block-wide sum for blocks of Block threads (multiple of 64); result valid
// in thread 0.  `smem` must hold Block/64 doubles.
template <int Block>
__device__ __forceinline__ double block_reduce_sum(double v, double* smem)
{
    constexpr int nwaves = Block / wave_size;
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < nwaves; ++w) total += smem[w];
    }
    return total;
}

}  // namespace gkomi
