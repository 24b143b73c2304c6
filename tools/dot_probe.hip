// Diagnostic build (not shipped): the product's nonzero-split SpMV kernel with
// the dot epilogue, compiled once per value of GKOMI_DOT_PROBE so that the parts
// of the epilogue can be priced one by one on the same box (tools/dot_probe.py).
#include "../repo-8852-ginkgo_amd/csrc/csr_spmv.hip"

extern "C" int probe_launch(void* stream, int dot, int nrows, int nnz, const int32_t* row_ptrs,
                            const int32_t* col_idxs, const double* vals, const double* b, double* c,
                            const int32_t* srow, int over, double* partial, const uint8_t* status)
{
    using namespace gkomi;
    constexpr int Block = 256, Tile = 3072;
    const int ntiles = nnz / Tile + 1;
    const int per = 16;
    dim3 grid(static_cast<unsigned>(ceildiv(ntiles, num_xcd * per) * num_xcd * per), 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dot) {
        hipLaunchKernelGGL((csr_split_kernel<int32_t, Block, Tile, split_max_over, false, true, true, true, true>), grid,
                           dim3(Block), 0, s, nrows, nnz, row_ptrs, col_idxs, vals, b, int64_t{1}, c, int64_t{1},
                           nullptr, nullptr, srow, ntiles, per, over, partial, status, nullptr, nullptr);
    } else {
        hipLaunchKernelGGL((csr_split_kernel<int32_t, Block, Tile, split_max_over, false, true, false, true, true>), grid,
                           dim3(Block), 0, s, nrows, nnz, row_ptrs, col_idxs, vals, b, int64_t{1}, c, int64_t{1},
                           nullptr, nullptr, srow, ntiles, per, over, nullptr, nullptr, nullptr, nullptr);
    }
    return static_cast<int>(hipGetLastError());
}
