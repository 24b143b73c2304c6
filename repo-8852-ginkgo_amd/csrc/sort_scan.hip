// Device-wide stable radix sort and scans for the setup paths (device_matrix_data::sort_row_major, the level
// analysis of the triangular solves, Jacobi's block detection, the distributed build_local_nonlocal) -- integer
// work on index arrays, written for wave64 instead of taken from a vendor library (rocPRIM calls
// on a path north_star wants hand-written).
//
// Radix sort: least significant digit first, 8 bits per pass, three launches per pass:
//   histogram  a workgroup counts the digits of its tile of 2048 keys            -> hist[digit][workgroup]
//   scan       exclusive sum over that table in (digit, workgroup) order          -> first output position of
//              every (digit, workgroup) pair
//   scatter    the workgroup reads its tile again, STRIPED (item = round * 256 + thread, so the items of a round
//              are consecutive), and ranks every item among the earlier items of its digit: inside its wave by
//              eight ballots (the lanes that agree with it on every bit of the digit, below it), across the waves
//              of the round and across earlier rounds by LDS counters -- equal keys keep their order: stable.
// Passes over bits that no key has set are skipped by the caller's `end_bit`.  Payloads are 32-bit (positions).
// Scans: a workgroup scans tiles of 2048 items, one workgroup scans the tile totals, the tiles add their offset;
// the operation (sum, maximum) and the flavour (exclusive, inclusive) are template parameters.
#include "sort_scan.hpp"

namespace gkomi {
namespace {

constexpr int sort_block = 256;
constexpr int sort_items = 8;
constexpr int sort_tile = sort_block * sort_items;
constexpr int radix_bits = 8;
constexpr int radix = 1 << radix_bits;

template <typename K>
__global__ __launch_bounds__(sort_block) void radix_histogram_kernel(const K* __restrict__ keys, int64_t n, int shift,
                                                                     int32_t nblocks, int32_t* __restrict__ hist)
{
    __shared__ int32_t counts[radix];
    counts[threadIdx.x] = 0;  // radix == sort_block
    __syncthreads();
    const int64_t base = static_cast<int64_t>(blockIdx.x) * sort_tile;
#pragma unroll
    for (int r = 0; r < sort_items; ++r) {
        const int64_t i = base + r * sort_block + threadIdx.x;
        if (i < n) atomicAdd(&counts[static_cast<int>((keys[i] >> shift) & (radix - 1))], 1);
    }
    __syncthreads();
    hist[static_cast<int64_t>(threadIdx.x) * nblocks + blockIdx.x] = counts[threadIdx.x];
}

template <typename K, bool Pairs>
__global__ __launch_bounds__(sort_block) void radix_scatter_kernel(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                                   int32_t nblocks, const int32_t* __restrict__ first)
{
    constexpr int waves = sort_block / 64;
    __shared__ int32_t run[radix];            // items of every digit in the rounds before this one
    __shared__ int32_t cnt[waves][radix];     // ... in every wave of this round
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    run[tid] = first[static_cast<int64_t>(tid) * nblocks + blockIdx.x];  // starts at the pair's first output position
    const int64_t base = static_cast<int64_t>(blockIdx.x) * sort_tile;
    for (int r = 0; r < sort_items; ++r) {
#pragma unroll
        for (int w = 0; w < waves; ++w) cnt[w][tid] = 0;
        __syncthreads();
        const int64_t i = base + r * sort_block + tid;
        const bool have = i < n;
        const K key = have ? keys_in[i] : K{0};
        const int digit = static_cast<int>((key >> shift) & (radix - 1));
        // the lanes of my wave that hold the same digit
        unsigned long long peers = __ballot(have);
#pragma unroll
        for (int b = 0; b < radix_bits; ++b) {
            const unsigned long long set = __ballot((digit >> b) & 1);
            peers &= ((digit >> b) & 1) ? set : ~set;
        }
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        const int rank_in_wave = __popcll(peers & below);
        if (have && rank_in_wave == 0) cnt[wave][digit] = __popcll(peers);  // the first of its peers speaks for all
        __syncthreads();
        if (have) {
            int pos = run[digit] + rank_in_wave;
#pragma unroll
            for (int w = 0; w < waves; ++w) pos += w < wave ? cnt[w][digit] : 0;
            keys_out[pos] = key;
            if (Pairs) vals_out[pos] = vals_in[i];
        }
        __syncthreads();
        int add = 0;
#pragma unroll
        for (int w = 0; w < waves; ++w) add += cnt[w][tid];
        run[tid] += add;
        __syncthreads();
    }
}

// ---- scans ---------------------------------------------------------------------------------------------------
constexpr int scan_block = 256;
constexpr int scan_items = 8;
constexpr int scan_tile = scan_block * scan_items;

struct op_sum {
    __device__ static int32_t identity() { return 0; }
    __device__ static int32_t apply(int32_t a, int32_t b) { return a + b; }
};
struct op_max {
    __device__ static int32_t identity() { return INT32_MIN; }
    __device__ static int32_t apply(int32_t a, int32_t b) { return a > b ? a : b; }
};

// scans the workgroup's tile held blocked in v (thread t: items t*8 .. t*8+7); returns the tile's total
template <typename Op, bool Inclusive>
__device__ __forceinline__ int32_t block_scan(int32_t (&v)[scan_items], int32_t* smem)
{
    int32_t local = Op::identity();
#pragma unroll
    for (int i = 0; i < scan_items; ++i) {
        const int32_t t = v[i];
        if (Inclusive) {
            local = Op::apply(local, t);
            v[i] = local;
        } else {
            v[i] = local;
            local = Op::apply(local, t);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl = Op::apply(up, incl);
    }
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    int32_t before = Op::identity(), total = Op::identity();
#pragma unroll
    for (int w = 0; w < scan_block / 64; ++w) {
        if (w < wave) before = Op::apply(before, smem[w]);
        total = Op::apply(total, smem[w]);
    }
    // what precedes this thread: the waves before it, then the lanes before it in its wave
    const int32_t lanes_before = __shfl_up(incl, 1, 64);
    const int32_t thread_before = lane == 0 ? before : Op::apply(before, lanes_before);
#pragma unroll
    for (int i = 0; i < scan_items; ++i) v[i] = Op::apply(thread_before, v[i]);
    __syncthreads();
    return total;
}

template <typename Op, bool Inclusive>
// (in == out allowed, and what every caller in the tree passes: no __restrict__ on the two)
__global__ __launch_bounds__(scan_block) void scan_tiles_op_kernel(const int32_t* in, int32_t* out,
                                                                  int64_t n, int32_t* __restrict__ tile_totals)
{
    __shared__ int32_t smem[scan_block / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * scan_tile + threadIdx.x * scan_items;
    int32_t v[scan_items];
#pragma unroll
    for (int i = 0; i < scan_items; ++i) v[i] = base + i < n ? in[base + i] : Op::identity();
    const int32_t total = block_scan<Op, Inclusive>(v, smem);
#pragma unroll
    for (int i = 0; i < scan_items; ++i) {
        if (base + i < n) out[base + i] = v[i];
    }
    if (threadIdx.x == 0) tile_totals[blockIdx.x] = total;
}

// one workgroup: EXCLUSIVE scan of the tile totals (what precedes every tile), in chunks with a carry
template <typename Op>
__global__ __launch_bounds__(scan_block) void scan_totals_op_kernel(int32_t* __restrict__ totals, int64_t ntiles)
{
    __shared__ int32_t smem[scan_block / 64];
    int32_t carry = Op::identity();
    for (int64_t chunk = 0; chunk < ntiles; chunk += scan_tile) {
        const int64_t base = chunk + threadIdx.x * scan_items;
        int32_t v[scan_items];
#pragma unroll
        for (int i = 0; i < scan_items; ++i) v[i] = base + i < ntiles ? totals[base + i] : Op::identity();
        const int32_t total = block_scan<Op, false>(v, smem);
#pragma unroll
        for (int i = 0; i < scan_items; ++i) {
            if (base + i < ntiles) totals[base + i] = Op::apply(carry, v[i]);
        }
        carry = Op::apply(carry, total);
    }
}

template <typename Op>
__global__ __launch_bounds__(scan_block) void scan_add_offsets_op_kernel(int32_t* __restrict__ data, int64_t n,
                                                                        const int32_t* __restrict__ tile_offsets)
{
    const int32_t off = tile_offsets[blockIdx.x];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * scan_tile + threadIdx.x * scan_items;
#pragma unroll
    for (int i = 0; i < scan_items; ++i) {
        if (base + i < n) data[base + i] = Op::apply(off, data[base + i]);
    }
}

template <typename Op, bool Inclusive>
int scan_impl(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes)
{
    if (n < 0) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    const int64_t ntiles = ceildiv(n, scan_tile);
    if (ws == nullptr || ws_bytes < scan_workspace_bytes(n)) return GKOMI_EWORKSPACE;
    int32_t* totals = static_cast<int32_t*>(ws);
    hipLaunchKernelGGL((scan_tiles_op_kernel<Op, Inclusive>), dim3(static_cast<unsigned>(ntiles)), dim3(scan_block), 0, s, in, out,
                       n, totals);
    if (ntiles > 1) {
        hipLaunchKernelGGL(scan_totals_op_kernel<Op>, dim3(1), dim3(scan_block), 0, s, totals, ntiles);
        hipLaunchKernelGGL(scan_add_offsets_op_kernel<Op>, dim3(static_cast<unsigned>(ntiles)), dim3(scan_block), 0, s, out, n,
                           totals);
    }
    return check_launch();
}

template <typename K>
int radix_sort_impl(hipStream_t s, int64_t n, const K* keys_in, K* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                    int end_bit, void* ws, size_t ws_bytes)
{
    if (n < 0 || end_bit < 0 || end_bit > static_cast<int>(8 * sizeof(K))) return GKOMI_EINVAL;
    if (n == 0) return GKOMI_SUCCESS;
    if (n > INT32_MAX - sort_tile) return GKOMI_ENOTSUPPORTED;
    const bool pairs = vals_in != nullptr;
    if (ws == nullptr || ws_bytes < radix_sort_workspace_bytes(n, sizeof(K), pairs)) return GKOMI_EWORKSPACE;
    const int32_t nblocks = static_cast<int32_t>(ceildiv(n, sort_tile));
    char* w = static_cast<char*>(ws);
    auto take = [&](size_t bytes) {
        char* at = w;
        w += (bytes + 255) / 256 * 256;
        return at;
    };
    K* keys_alt = reinterpret_cast<K*>(take(sizeof(K) * n));
    uint32_t* vals_alt = pairs ? reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * n)) : nullptr;
    const int64_t hist_entries = static_cast<int64_t>(radix) * nblocks;
    int32_t* hist = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * hist_entries));
    void* scan_ws = take(scan_workspace_bytes(hist_entries));
    const int passes = (end_bit + radix_bits - 1) / radix_bits;
    if (passes == 0) {
        int err = static_cast<int>(hipMemcpyAsync(keys_out, keys_in, sizeof(K) * n, hipMemcpyDeviceToDevice, s));
        if (!err && pairs) err = static_cast<int>(hipMemcpyAsync(vals_out, vals_in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, s));
        return err;
    }
    const K* src_k = keys_in;
    const uint32_t* src_v = vals_in;
    for (int p = 0; p < passes; ++p) {
        // the last pass lands in the caller's output, the ones before alternate
        const bool to_out = (passes - 1 - p) % 2 == 0;
        K* dst_k = to_out ? keys_out : keys_alt;
        uint32_t* dst_v = to_out ? vals_out : vals_alt;
        const int shift = p * radix_bits;
        hipLaunchKernelGGL(radix_histogram_kernel<K>, dim3(nblocks), dim3(sort_block), 0, s, src_k, n, shift, nblocks, hist);
        const int err = scan_impl<op_sum, false>(s, hist, hist, hist_entries, scan_ws, scan_workspace_bytes(hist_entries));
        if (err) return err;
        if (pairs) {
            hipLaunchKernelGGL((radix_scatter_kernel<K, true>), dim3(nblocks), dim3(sort_block), 0, s, src_k, dst_k, src_v, dst_v, n,
                               shift, nblocks, hist);
        } else {
            hipLaunchKernelGGL((radix_scatter_kernel<K, false>), dim3(nblocks), dim3(sort_block), 0, s, src_k, dst_k, src_v, dst_v, n,
                               shift, nblocks, hist);
        }
        src_k = dst_k;
        src_v = dst_v;
    }
    return check_launch();
}

}  // namespace

size_t scan_workspace_bytes(int64_t n)
{
    const int64_t ntiles = ceildiv(n > 0 ? n : 1, scan_tile);
    return (sizeof(int32_t) * static_cast<size_t>(ntiles) + 255) / 256 * 256;
}

size_t radix_sort_workspace_bytes(int64_t n, size_t key_bytes, bool pairs)
{
    const size_t m = static_cast<size_t>(n > 0 ? n : 1);
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const int64_t hist_entries = static_cast<int64_t>(radix) * ceildiv(static_cast<int64_t>(m), sort_tile);
    return up(key_bytes * m) + (pairs ? up(4 * m) : 0) + up(sizeof(int32_t) * hist_entries) + scan_workspace_bytes(hist_entries) + 256;
}

int radix_sort_u64(hipStream_t s, int64_t n, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                   uint32_t* vals_out, int end_bit, void* ws, size_t ws_bytes)
{
    return radix_sort_impl<uint64_t>(s, n, keys_in, keys_out, vals_in, vals_out, end_bit, ws, ws_bytes);
}

int radix_sort_u32(hipStream_t s, int64_t n, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                   uint32_t* vals_out, int end_bit, void* ws, size_t ws_bytes)
{
    return radix_sort_impl<uint32_t>(s, n, keys_in, keys_out, vals_in, vals_out, end_bit, ws, ws_bytes);
}

int exclusive_sum_i32(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes)
{
    return scan_impl<op_sum, false>(s, in, out, n, ws, ws_bytes);
}

int inclusive_max_i32(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes)
{
    return scan_impl<op_max, true>(s, in, out, n, ws, ws_bytes);
}

}  // namespace gkomi

// test entry points (tests/test_sort_scan_gpu.py): the sort and the scans against numpy
extern "C" size_t gkomi_diag_radix_sort_workspace_bytes(int64_t n, int key_bytes, int pairs)
{
    return gkomi::radix_sort_workspace_bytes(n, static_cast<size_t>(key_bytes), pairs != 0);
}

extern "C" int gkomi_diag_radix_sort(gkomi_stream_t s, int64_t n, int key_bytes, const void* keys_in, void* keys_out,
                                     const uint32_t* vals_in, uint32_t* vals_out, int end_bit, void* workspace,
                                     size_t workspace_bytes)
{
    if (key_bytes == 8) {
        return gkomi::radix_sort_u64(gkomi::to_stream(s), n, static_cast<const uint64_t*>(keys_in), static_cast<uint64_t*>(keys_out),
                                     vals_in, vals_out, end_bit, workspace, workspace_bytes);
    }
    if (key_bytes == 4) {
        return gkomi::radix_sort_u32(gkomi::to_stream(s), n, static_cast<const uint32_t*>(keys_in), static_cast<uint32_t*>(keys_out),
                                     vals_in, vals_out, end_bit, workspace, workspace_bytes);
    }
    return GKOMI_EINVAL;
}

extern "C" int gkomi_diag_scan_i32(gkomi_stream_t s, int kind, const int32_t* in, int32_t* out, int64_t n, void* workspace,
                                   size_t workspace_bytes)
{
    if (kind == 0) return gkomi::exclusive_sum_i32(gkomi::to_stream(s), in, out, n, workspace, workspace_bytes);
    if (kind == 1) return gkomi::inclusive_max_i32(gkomi::to_stream(s), in, out, n, workspace, workspace_bytes);
    return GKOMI_EINVAL;
}
