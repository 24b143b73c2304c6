// Diagnostic build (not shipped): the product's nonzero-split SpMV kernel
// compiled with GKOMI_TIMELINE, which makes lane 0 of every workgroup write
// phase stamps (s_memrealtime, 100 MHz, comparable across CUs) into a buffer of
// its own: 0 entry, 1 products in LDS (all of this wave's loads and gathers
// done), 2 behind the barrier, 3 row sums stored; 5 XCC id, 6 HW_ID.
// tools/spmv_timeline.py turns them into a launch timeline: when workgroups
// start, how long each phase takes, how many are resident.
#define GKOMI_TIMELINE 1
#include "../repo-8852-ginkgo_amd/csrc/csr_spmv.hip"

extern "C" int timeline_launch(void* stream, int nt, int swizzle, int nrows, int nnz, const int32_t* row_ptrs,
                               const int32_t* col_idxs, const double* vals, const double* b, double* c,
                               const int32_t* srow, int over, unsigned long long* stamps)
{
    using namespace gkomi;
    constexpr int Block = 256, Tile = 1536;
    const int ntiles = nnz / Tile + 1;
    const int per = static_cast<int>(ceildiv(ntiles, num_xcd));
    dim3 grid(swizzle ? per * num_xcd : ntiles, 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TL(SWZ, NT)                                                                                   \
    hipLaunchKernelGGL((csr_split_kernel<int32_t, Block, Tile, split_max_over, false, SWZ, false, NT, true>), \
                       grid, dim3(Block), 0, s, nrows, nnz, row_ptrs, col_idxs, vals, b, int64_t{1}, c, \
                       int64_t{1}, nullptr, nullptr, srow, ntiles, per, over, nullptr, nullptr, nullptr, \
                       nullptr, stamps)
    if (swizzle) {
        if (nt) TL(true, true); else TL(true, false);
    } else {
        if (nt) TL(false, true); else TL(false, false);
    }
#undef TL
    return static_cast<int>(hipGetLastError());
}
