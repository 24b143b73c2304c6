// Ceiling probe (not shipped): a pure streaming kernel that moves the same
// bytes as CSR SpMV on the 5-pt matrix (reads 12 B/nnz + 12 B/row, writes
// 8 B/row) with fully coalesced 16-B loads and nothing else.  Gives the
// practical HBM/Infinity-Cache ceiling for a ~15 us launch on this box.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void stream_like_spmv(
    int64_t nnz, int64_t nrows, const double2* __restrict__ vals,
    const int4* __restrict__ cols, const double2* __restrict__ x,
    const int4* __restrict__ rp, double2* __restrict__ y)
{
    const int64_t tid = blockIdx.x * 256ll + threadIdx.x;
    const int64_t nth = gridDim.x * 256ll;
    double acc = 0.0;
    int iacc = 0;
    for (int64_t i = tid; i < nnz / 2; i += nth) {
        double2 v = vals[i];
        acc += v.x + v.y;
    }
    for (int64_t i = tid; i < nnz / 4; i += nth) {
        int4 c = cols[i];
        iacc += c.x + c.y + c.z + c.w;
    }
    for (int64_t i = tid; i < nrows / 4; i += nth) {
        int4 c = rp[i];
        iacc += c.x + c.y + c.z + c.w;
    }
    for (int64_t i = tid; i < nrows / 2; i += nth) {
        double2 xv = x[i];
        y[i] = make_double2(xv.x + acc, xv.y + iacc);
    }
}

extern "C" int membench_launch(void* stream, int blocks, int64_t nnz, int64_t nrows,
                               const void* vals, const void* cols, const void* x,
                               const void* rp, void* y)
{
    hipLaunchKernelGGL(stream_like_spmv, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nnz,
                       nrows, (const double2*)vals, (const int4*)cols, (const double2*)x,
                       (const int4*)rp, (double2*)y);
    return (int)hipGetLastError();
}
