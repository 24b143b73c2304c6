// Cross-file host entry points inside the library (not part of the C ABI).
#pragma once
#include "common.hpp"

namespace gkomi {

int csr_spmv_dot_launch(hipStream_t stream, int nrows, int64_t nnz,
                        const int32_t* row_ptrs, const int32_t* col_idxs,
                        const double* vals, const double* p, double* q,
                        double* partial, const uint8_t* stop_status,
                        bool swizzle);
int csr_spmv_dot_num_partials(int nrows);
bool csr_auto_swizzle(int64_t nrows, int64_t nnz);

}  // namespace gkomi
