// Latencies a single wave sees on an otherwise idle MI355X (what bounds a level of the brick
// triangular solve, csrc/trs_bricks.hip): dependent fp64 FMA, LDS write -> read turnaround,
// ds_bpermute, and the shader clock itself (s_memtime ticks per 10 ns of s_memrealtime).
// build: hipcc -O3 --offload-arch=gfx950 tools/latency_probe.hip -o tools/bin/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(double* out, long long* ticks, int n, int waves_busy)
{
    __shared__ double cell[64];
    const int lane = threadIdx.x;
    if (blockIdx.x > 0) {  // optional load on the other CUs: keeps the clock up?
        double a = lane;
        for (int i = 0; i < n * 8; ++i) a = __builtin_fma(a, 1.0000001, 0.5);
        if (a == 42.0) out[63] = a;
        return;
    }
    double a = 1.0 + lane;
    long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < n; ++i) a = __builtin_fma(a, 1.0000001, 0.5);
    long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    cell[lane] = a;
    for (int i = 0; i < n; ++i) {  // write -> read of the neighbour's value
        const double v = cell[(lane + 1) & 63];
        cell[lane] = v + 1.0;
    }
    long long t2 = __builtin_readcyclecounter(), r2 = wall_clock64();
    double b = a;
    for (int i = 0; i < n; ++i) b = __shfl(b, (lane + 1) & 63, 64) + 1.0;
    long long t3 = __builtin_readcyclecounter(), r3 = wall_clock64();
    double c = b;
    for (int i = 0; i < n; ++i) c = c / (1.0 + 1e-9 * c);  // the compiler's f64 division, dependent
    long long t4 = __builtin_readcyclecounter(), r4 = wall_clock64();
    out[lane] = a + b + c + cell[lane];
    if (lane == 0) {
        ticks[0] = t1 - t0; ticks[1] = r1 - r0;
        ticks[2] = t2 - t1; ticks[3] = r2 - r1;
        ticks[4] = t3 - t2; ticks[5] = r3 - r2;
        ticks[6] = t4 - t3; ticks[7] = r4 - r3;
    }
}

int main()
{
    double* out; long long* ticks;
    hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8 * 8);
    const int n = 20000;
    for (int busy : {1, 256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(busy), dim3(64), 0, 0, out, ticks, n, busy);
            hipDeviceSynchronize();
        }
        long long h[8];
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        const char* names[4] = {"dependent fp64 fma", "LDS write -> read + add", "ds_bpermute + add", "f64 division + fma"};
        printf("workgroups %d:\n", busy);
        for (int k = 0; k < 4; ++k) {
            printf("  %-26s %7.1f shader ticks, %7.1f ns per iteration (s_memtime %.0f MHz)\n", names[k],
                   double(h[2 * k]) / n, double(h[2 * k + 1]) * 10.0 / n, double(h[2 * k]) / (double(h[2 * k + 1]) * 10.0) * 1e3);
        }
    }
    return 0;
}
