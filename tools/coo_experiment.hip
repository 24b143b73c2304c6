// Experiment (not shipped): where does the COO spmv2 time go?  Variants of the
// tile kernel of csrc/formats.hip on the 1000^2 5-pt matrix:
//   0 = one fp64 atomic per row segment, issued from the scanning thread
//   1 = plain (racy) read-modify-write instead of the atomic: timing only
//   2 = no update of c at all: the streaming ceiling of the kernel
//   3 = segment sums compacted in LDS, then consecutive lanes issue the
//       atomics of consecutive segments (coalesced atomic instructions)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/coo_experiment tools/coo_experiment.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

constexpr int block = 256;
constexpr int items = 6;
constexpr int tile = block * items;

template <int Mode>
__global__ __launch_bounds__(block) void coo_kernel(int64_t nnz, const int32_t* __restrict__ row_idxs,
                                                    const int32_t* __restrict__ col_idxs,
                                                    const double* __restrict__ vals,
                                                    const double* __restrict__ b, double* __restrict__ c)
{
    __shared__ __attribute__((aligned(16))) double prod[tile];
    __shared__ __attribute__((aligned(8))) int32_t rowid[tile];
    __shared__ int wave_count[block / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * tile;
    const int count = static_cast<int>(min(static_cast<int64_t>(tile), nnz - base));
    const int tid = threadIdx.x;
    constexpr int pairs = items / 2;
    double2 v[pairs];
    int2 r[pairs], cc[pairs];
#pragma unroll
    for (int u = 0; u < pairs; ++u) {
        const int e = 2 * (tid + u * block);
        v[u] = make_double2(0.0, 0.0);
        r[u] = make_int2(0, 0);
        cc[u] = make_int2(0, 0);
        if (e + 1 < count) {
            v[u] = *reinterpret_cast<const double2*>(vals + base + e);
            r[u] = *reinterpret_cast<const int2*>(row_idxs + base + e);
            cc[u] = *reinterpret_cast<const int2*>(col_idxs + base + e);
        } else if (e < count) {
            v[u].x = vals[base + e];
            r[u].x = row_idxs[base + e];
            cc[u].x = col_idxs[base + e];
        }
    }
    double2 x[pairs];
#pragma unroll
    for (int u = 0; u < pairs; ++u) {
        x[u].x = b[cc[u].x];
        x[u].y = b[cc[u].y];
    }
#pragma unroll
    for (int u = 0; u < pairs; ++u) {
        const int e = 2 * (tid + u * block);
        double2 pr;
        pr.x = v[u].x * x[u].x;
        pr.y = v[u].y * x[u].y;
        *reinterpret_cast<double2*>(prod + e) = pr;
        *reinterpret_cast<int2*>(rowid + e) = r[u];
    }
    __syncthreads();
    const int first = tid * items;
    if (Mode != 3) {
#pragma unroll
        for (int u = 0; u < items; ++u) {
            const int e = first + u;
            if (e < count) {
                const int row = rowid[e];
                if (e == 0 || rowid[e - 1] != row) {
                    double sum = prod[e];
                    int k = e + 1;
                    while (k < count && rowid[k] == row) {
                        sum += prod[k];
                        ++k;
                    }
                    if (Mode == 0) unsafeAtomicAdd(c + row, sum);
                    if (Mode == 1) c[row] += sum;
                    if (Mode == 2 && sum == 1.2345e-300) c[row] = sum;
                }
            }
        }
    } else {
        // heads in this thread's run, sums in registers
        double sums[items];
        int rows[items];
        int nheads = 0;
#pragma unroll
        for (int u = 0; u < items; ++u) {
            const int e = first + u;
            rows[u] = -1;
            sums[u] = 0.0;
            if (e < count) {
                const int row = rowid[e];
                if (e == 0 || rowid[e - 1] != row) {
                    double sum = prod[e];
                    int k = e + 1;
                    while (k < count && rowid[k] == row) {
                        sum += prod[k];
                        ++k;
                    }
                    rows[u] = row;
                    sums[u] = sum;
                    ++nheads;
                }
            }
        }
        // exclusive scan of nheads over the workgroup
        const int lane = tid & 63;
        int incl = nheads;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_count[tid >> 6] = incl;
        __syncthreads();  // also: everyone is done reading prod / rowid
        int offset = incl - nheads;
        int total = 0;
#pragma unroll
        for (int w = 0; w < block / 64; ++w) {
            const int wc = wave_count[w];
            if (w < (tid >> 6)) offset += wc;
            total += wc;
        }
#pragma unroll
        for (int u = 0; u < items; ++u) {
            if (rows[u] >= 0) {
                prod[offset] = sums[u];
                rowid[offset] = rows[u];
                ++offset;
            }
        }
        __syncthreads();
        for (int i = tid; i < total; i += block) {
            unsafeAtomicAdd(c + rowid[i], prod[i]);
        }
    }
}

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

template <int Mode>
float run(int reps, int64_t nnz, const int32_t* rows, const int32_t* cols, const double* vals,
          const double* b, double* c, int64_t n)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int nblocks = static_cast<int>((nnz + tile - 1) / tile);
    for (int i = 0; i < 5; ++i) {
        CHECK(hipMemsetAsync(c, 0, n * 8, 0));
        hipLaunchKernelGGL(coo_kernel<Mode>, dim3(nblocks), dim3(block), 0, 0, nnz, rows, cols, vals, b, c);
    }
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(coo_kernel<Mode>, dim3(nblocks), dim3(block), 0, 0, nnz, rows, cols, vals, b, c);
    }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main()
{
    const int m = 1000;
    const int64_t n = static_cast<int64_t>(m) * m;
    std::vector<int32_t> rows, cols;
    std::vector<double> vals;
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < m; ++j) {
            const int32_t row = i * m + j;
            if (i > 0) { rows.push_back(row); cols.push_back(row - m); vals.push_back(-1.0); }
            if (j > 0) { rows.push_back(row); cols.push_back(row - 1); vals.push_back(-1.0); }
            rows.push_back(row); cols.push_back(row); vals.push_back(4.0);
            if (j < m - 1) { rows.push_back(row); cols.push_back(row + 1); vals.push_back(-1.0); }
            if (i < m - 1) { rows.push_back(row); cols.push_back(row + m); vals.push_back(-1.0); }
        }
    }
    const int64_t nnz = static_cast<int64_t>(vals.size());
    std::vector<double> b(n), ref(n, 0.0), out(n);
    for (int64_t i = 0; i < n; ++i) b[i] = sin(0.001 * i) + 1.5;
    for (int64_t k = 0; k < nnz; ++k) ref[rows[k]] += vals[k] * b[cols[k]];
    int32_t *d_rows, *d_cols;
    double *d_vals, *d_b, *d_c;
    CHECK(hipMalloc(&d_rows, nnz * 4));
    CHECK(hipMalloc(&d_cols, nnz * 4));
    CHECK(hipMalloc(&d_vals, nnz * 8));
    CHECK(hipMalloc(&d_b, n * 8));
    CHECK(hipMalloc(&d_c, n * 8));
    CHECK(hipMemcpy(d_rows, rows.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_cols, cols.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_vals, vals.data(), nnz * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_b, b.data(), n * 8, hipMemcpyHostToDevice));
    const int reps = 200;
    for (int round = 0; round < 2; ++round) {
        printf("mode 0 (atomic per segment)      %7.2f us\n", run<0>(reps, nnz, d_rows, d_cols, d_vals, d_b, d_c, n));
        printf("mode 1 (plain rmw, racy)         %7.2f us\n", run<1>(reps, nnz, d_rows, d_cols, d_vals, d_b, d_c, n));
        printf("mode 2 (no update)               %7.2f us\n", run<2>(reps, nnz, d_rows, d_cols, d_vals, d_b, d_c, n));
        printf("mode 3 (compacted atomics)       %7.2f us\n", run<3>(reps, nnz, d_rows, d_cols, d_vals, d_b, d_c, n));
    }
    // correctness of mode 3
    CHECK(hipMemset(d_c, 0, n * 8));
    const int nblocks = static_cast<int>((nnz + tile - 1) / tile);
    hipLaunchKernelGGL(coo_kernel<3>, dim3(nblocks), dim3(block), 0, 0, nnz, d_rows, d_cols, d_vals, d_b, d_c);
    CHECK(hipMemcpy(out.data(), d_c, n * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int64_t i = 0; i < n; ++i) err = fmax(err, fabs(out[i] - ref[i]));
    printf("mode 3 max abs error %g\n", err);
    return 0;
}
