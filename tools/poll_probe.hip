// How long until a polling wave SEES a value another workgroup stored -- as a function of what the poll
// looks like?  Workgroup A stores a counter write-through (sc1) into ONE of the words workgroup B polls; B polls
// `loads` gathers of 64 lanes each, every lane its own cache line (stride bytes apart), until the counter
// arrives, then answers through a second word that A polls with one lane.  Round trip / 2 - the simple
// direction = what a pump of csrc/trs_bricks.hip pays.  Optional background: `noise` other workgroups polling
// their own words the same way.
// build: hipcc -O3 --offload-arch=gfx950 tools/poll_probe.hip -o tools/bin/poll_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long ld(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int Loads>
__global__ __launch_bounds__(64) void probe(unsigned long long* words, long long stride_words, int rounds, int noise_polls,
                                            long long* out)
{
    const int lane = threadIdx.x;
    unsigned long long* mine = words + static_cast<long long>(blockIdx.x) * (64 * Loads + 64) * stride_words;
    if (blockIdx.x >= 2) {  // background: poll my own words for a while
        unsigned long long acc = 0;
        for (int i = 0; i < noise_polls; ++i) {
#pragma unroll
            for (int k = 0; k < Loads; ++k) acc += ld(mine + (64 * k + lane) * stride_words);
            __builtin_amdgcn_s_sleep(2);
        }
        if (acc == 12345) out[3] = 1;
        return;
    }
    unsigned long long* gather = words + (64 * Loads + 64) * stride_words;  // block 1's words: what B polls
    unsigned long long* answer = words;                                     // block 0's word 0: what A polls
    const int hot = (64 * (Loads - 1) + 37) * static_cast<int>(stride_words);  // the word that carries the counter
    long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (blockIdx.x == 0) {
            if (lane == 0) st(gather + hot, r);
            unsigned long long a = 0;
            long long spins = 0;
            while (true) {
                a = lane == 0 ? ld(answer) : r;
                if (__all(a == static_cast<unsigned long long>(r)) || ++spins > (1 << 20)) break;
            }
        } else {
            long long spins = 0;
            while (true) {
                unsigned long long v[Loads];
#pragma unroll
                for (int k = 0; k < Loads; ++k) v[k] = ld(gather + (64 * k + lane) * stride_words);
                bool seen = false;
#pragma unroll
                for (int k = 0; k < Loads; ++k) seen |= v[k] == static_cast<unsigned long long>(r);
                if (__any(seen) || ++spins > (1 << 20)) break;
            }
            if (lane == 0) st(answer, r);
        }
    }
    long long t1 = wall_clock64();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
}

int main()
{
    unsigned long long* words; long long* out;
    const size_t bytes = 2048ull * (64 * 4 + 64) * 256;  // room for 2048 workgroups at 256-byte stride
    hipMalloc(&words, bytes); hipMalloc(&out, 64);
    const int rounds = 2000;
    for (int noise : {0, 254, 1022}) {
        for (long long stride_bytes : {8ll, 128ll, 256ll}) {
            long long h[4];
#define RUN(L)                                                                                                       \
    hipMemset(words, 0, bytes);                                                                                      \
    hipLaunchKernelGGL(probe<L>, dim3(2 + noise), dim3(64), 0, 0, words, stride_bytes / 8, rounds, 40000, out);      \
    hipDeviceSynchronize();                                                                                          \
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);                                                             \
    printf("noise %4d workgroups, stride %3lld B, %d gather(s) of 64 lanes: %6.3f us per round trip\n", noise, stride_bytes, L, \
           double(h[0]) * 10.0 / 1e3 / rounds);
            RUN(1) RUN(2) RUN(4)
        }
    }
    return 0;
}
