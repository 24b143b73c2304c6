// Experiment (not shipped): one-way latency of a value hand-off between two
// workgroups -- the hop that bounds the sync-free triangular solve.  Workgroup A
// and workgroup B bounce a counter: each waits until the other's word equals
// its round number, then stores its own.  Variants: agent-scope (sc1) relaxed
// atomics as trs.hip uses, workgroup-scope loads (L2-coherent inside one XCD),
// and the pair placed on the same or on different XCDs (workgroups are dealt
// round-robin to the 8 XCDs: block b runs on XCD b % 8).
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/handoff_experiment tools/handoff_experiment.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr long long max_spin = 1ll << 18;  // bounded: ~0.1 s worst case per variant

// LoadScope: 0 = "sc0" (bypass the CU's L1, served by this XCD's L2),
// 1 = "sc1" (agent scope), 2 = "sc0 sc1" (system scope)
template <int LoadScope>
__device__ __forceinline__ double poll_load(const double* p)
{
    double v;
    if (LoadScope == 0) {
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if (LoadScope == 1) {
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    }
    return v;
}

template <int LoadScope>
__global__ void pingpong(double* words, int partner_block, int rounds, int sleep, int* overrun,
                         int* xcc_ids)
{
    // only blocks 0 and partner_block play; the others exit at once
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
    if (me < 0 || threadIdx.x != 0) return;
    // XCC_ID hardware register (gfx942+): which XCD this wave runs on
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_ids[me] = xcc & 0xf;
    double* mine = words + 32 * me;  // separate 256-B lines
    double* theirs = words + 32 * (1 - me);
    for (int r = 1; r <= rounds; ++r) {
        if (me == 1 || r > 1) {
            // wait for the partner's round: B waits for A's r, A waits for B's r - 1
            const double want = me == 1 ? r : r - 1;
            long long spins = 0;
            while (poll_load<LoadScope>(theirs) != want) {
                if (sleep == 1) __builtin_amdgcn_s_sleep(1);
                if (sleep == 4) __builtin_amdgcn_s_sleep(4);
                if (++spins > max_spin) {
                    *overrun = 1;
                    return;
                }
            }
        }
        __hip_atomic_store(mine, static_cast<double>(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int LoadScope>
void run(const char* name, double* words, int partner, int sleep, int* d_over, int* d_xcc)
{
    const int rounds = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    int xcc[2] = {-1, -1}, over = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(words, 0, 64 * sizeof(double)));
        CHECK(hipMemset(d_over, 0, sizeof(int)));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(pingpong<LoadScope>, dim3(partner + 1), dim3(64), 0, 0, words, partner, rounds,
                           sleep, d_over, d_xcc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(&over, d_over, sizeof(int), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(xcc, d_xcc, 2 * sizeof(int), hipMemcpyDeviceToHost));
        if (over) break;
    }
    // one round = two hand-offs (A -> B, B -> A)
    printf("%-46s partner block %2d (XCD %d / %d) sleep %d: %7.3f us per hand-off%s\n", name, partner, xcc[0],
           xcc[1], sleep, best * 1000.f / rounds / 2, over ? "  OVERRUN (never saw the partner's store)" : "");
}

int main()
{
    double* words;
    int *d_over, *d_xcc;
    CHECK(hipMalloc(&words, 64 * sizeof(double)));
    CHECK(hipMalloc(&d_over, sizeof(int)));
    CHECK(hipMalloc(&d_xcc, 2 * sizeof(int)));
    for (int sleep : {0, 1, 4}) {
        run<1>("sc1 (agent) load, other XCD", words, 1, sleep, d_over, d_xcc);
        run<1>("sc1 (agent) load, same XCD", words, 8, sleep, d_over, d_xcc);
        run<0>("sc0 (this XCD's L2) load, same XCD", words, 8, sleep, d_over, d_xcc);
        run<2>("sc0 sc1 (system) load, other XCD", words, 1, sleep, d_over, d_xcc);
    }
    // expected to hang without the spin bound: another XCD's L2 never sees the store
    run<0>("sc0 (this XCD's L2) load, other XCD", words, 1, 1, d_over, d_xcc);
    return 0;
}
