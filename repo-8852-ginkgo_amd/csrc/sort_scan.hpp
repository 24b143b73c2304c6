// Hand-written device-wide stable radix sort and scans of the setup paths (sort_scan.hip).  Internal to the library.
#pragma once
#include "common.hpp"

namespace gkomi {

// scratch of a scan over n 32-bit items (one total per tile of 2048)
size_t scan_workspace_bytes(int64_t n);
// out[i] = in[0] + ... + in[i - 1]; in == out allowed
int exclusive_sum_i32(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes);
// out[i] = max(in[0], ..., in[i]); in == out allowed
int inclusive_max_i32(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes);

// scratch of a sort of n keys of key_bytes (4 or 8) bytes, with or without 32-bit payloads
size_t radix_sort_workspace_bytes(int64_t n, size_t key_bytes, bool pairs);
// Stable ascending sort by the key bits [0, end_bit) (least significant digit first, 8 bits per pass; bits at and
// above end_bit must be equal in all keys or irrelevant to the order wanted).  vals_in == nullptr: keys only.
// The inputs are not modified; outputs must not alias inputs.
int radix_sort_u64(hipStream_t s, int64_t n, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                   uint32_t* vals_out, int end_bit, void* ws, size_t ws_bytes);
int radix_sort_u32(hipStream_t s, int64_t n, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                   uint32_t* vals_out, int end_bit, void* ws, size_t ws_bytes);

}  // namespace gkomi
