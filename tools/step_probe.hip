// What does one level of the brick triangular solve (csrc/trs_bricks.hip, pipelined compute wave) cost,
// piece by piece?  One wave, synthetic data in LDS, the loop built up in stages; shader ticks per iteration.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/step_probe.hip -o tools/bin/step_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int REC = 56;  // record bytes (K = 2)

template <int Stage>
__global__ __launch_bounds__(128) void probe(double* out, long long* ticks, int n, unsigned long long* xg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int counter;
    const int lane = threadIdx.x;
    if (lane >= 64) {  // the second wave of the workgroup waits at the barrier, like the pump without inflow
        __syncthreads();
        __syncthreads();
        return;
    }
    // cells [0, 64): x; records behind them
    char* rec = reinterpret_cast<char*>(lds + 128);
    lds[lane] = 1.0 + lane * 1e-3;
    lds[64 + lane] = 0.0;
    for (int k = 0; k < 4; ++k) {
        char* r = rec + (lane + 64 * k) * REC;
        *reinterpret_cast<double*>(r) = 2.0;      // d
        *reinterpret_cast<double*>(r + 8) = 0.5;  // 1 / d
        *reinterpret_cast<double*>(r + 16) = 1e-3;
        *reinterpret_cast<double*>(r + 24) = 2e-3;
        *reinterpret_cast<int*>(r + 32) = 8 * ((lane + 1) & 63);
        *reinterpret_cast<int*>(r + 36) = 8 * ((lane + 63) & 63);
        *reinterpret_cast<int*>(r + 40) = 8 * lane;
        *reinterpret_cast<int*>(r + 44) = 8 * lane;
    }
    if (lane == 0) counter = 1 << 30;
    __syncthreads();
    const char* lxb = reinterpret_cast<const char*>(lds);
    double d = 2.0, r = 0.5, v0 = 1e-3, v1 = 2e-3;
    int ca0 = 8 * ((lane + 1) & 63), ca1 = 8 * ((lane + 63) & 63), xa = 8 * lane, off = 8 * lane;
    int begin = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        int ready = 1 << 30;
        if (Stage >= 3) ready = __hip_atomic_load(&counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double x0 = *reinterpret_cast<const double*>(lxb + ca0);
        const double x1 = *reinterpret_cast<const double*>(lxb + ca1);
        double sum = *reinterpret_cast<const double*>(lxb + xa);
        if (Stage >= 3 && __builtin_expect(ready < i, 0)) {
            while (ready < i) {
                __builtin_amdgcn_s_sleep(1);
                ready = __hip_atomic_load(&counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        double nd = d, nr = r, nv0 = v0, nv1 = v1;
        int nca0 = ca0, nca1 = ca1, nxa = xa, noff = off;
        if (Stage >= 2) {  // the next step's record
            const int cnt = 64;
            const char* p = rec + __umul24(lane < cnt ? ((begin + lane) & 255) : 0, REC);
            begin += cnt;
            nd = *reinterpret_cast<const double*>(p);
            nr = *reinterpret_cast<const double*>(p + 8);
            nv0 = *reinterpret_cast<const double*>(p + 16);
            nv1 = *reinterpret_cast<const double*>(p + 24);
            nca0 = *reinterpret_cast<const int*>(p + 32);
            nca1 = *reinterpret_cast<const int*>(p + 36);
            nxa = *reinterpret_cast<const int*>(p + 40);
            noff = *reinterpret_cast<const int*>(p + 44);
        }
        if (Stage == 6) {  // VALU compare + branch on a register value
            if (__builtin_expect(ca0 > (1 << 20), 0)) lds[100] = 1.0;
        }
        if (Stage == 7) {  // the same, uniform value through an SGPR
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(ca0) > (1 << 20), 0)) lds[100] = 1.0;
        }
        int counter_value = 0;
        if (Stage == 8 || Stage == 9) counter_value = __hip_atomic_load(&counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sum -= v0 * x0;
        sum -= v1 * x1;
        const double q = sum * r;
        const double rem = __builtin_fma(-d, q, sum);
        double xr = __builtin_fma(rem, r, q);
        if (Stage >= 5) {
            const bool plain = __builtin_fabs(sum) >= 0x1p-383 && __builtin_fabs(sum) < 0x1p+383 && xr == xr;
            if (__builtin_expect(__any(!plain), 0)) {
                if (!plain) xr = sum / d;
            }
        }
        if (Stage == 8) begin += counter_value & 1;  // counter read, no branch
        if (Stage == 9) {                            // counter read, uniform scalar compare
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(counter_value) < i, 0)) lds[100] = 1.0;
        }
        if (Stage == 10) {  // integer fast-box test: 3 VALU + one branch
            const unsigned e = (static_cast<unsigned>(__double2hiint(sum)) >> 20) & 0x7ffu;
            if (__builtin_expect(__any(e - 640u >= static_cast<unsigned>(xa + 766)), 0)) xr = sum / d;
        }
        if (off >= 0) {
            *reinterpret_cast<double*>(const_cast<char*>(lxb) + xa) = xr + 1.0;
            if (Stage >= 4) {
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(xg) + static_cast<unsigned>(off)),
                                   static_cast<unsigned long long>(__double_as_longlong(xr)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        d = nd; r = nr; v0 = nv0; v1 = nv1; ca0 = nca0; ca1 = nca1; xa = nxa; off = noff;
    }
    long long t1 = __builtin_readcyclecounter();
    out[lane] = lds[lane];
    if (lane == 0) ticks[Stage] = t1 - t0;
    __syncthreads();
    __syncthreads();
}

int main()
{
    double* out; long long* ticks; unsigned long long* xg;
    hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 16 * 8); hipMalloc(&xg, 64 * 8);
    const int n = 4000;
    const size_t lds = 8 * 128 + 256 * REC;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<1>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<2>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<3>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<4>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<5>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<6>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<7>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<8>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<9>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipLaunchKernelGGL(probe<10>, dim3(1), dim3(128), lds, 0, out, ticks, n, xg);
        hipDeviceSynchronize();
    }
    long long h[16];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[11] = {"", "chain only: x reads, 2 mul-sub, division tail, LDS write", "+ next record (4 LDS reads)",
                             "+ inflow counter read and test", "+ write-through store of the row", "+ fast-box test",
                             "stage 2 + VALU compare and branch (not taken)", "stage 2 + readfirstlane, scalar compare and branch",
                             "stage 2 + counter read, no branch", "stage 2 + counter read, scalar compare and branch",
                             "stage 2 + integer fast-box test"};
    for (int k = 1; k <= 10; ++k) printf("stage %d %-60s %7.1f shader ticks per level\n", k, names[k], double(h[k]) / n);
    return 0;
}
