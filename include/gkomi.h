/*
 * gkomi.h -- C ABI of the MI355X (gfx950) kernel backend for the Ginkgo
 * SpMV + Krylov-solver hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one free
 * function `gko::kernels::hip::<component>::<fn>(std::shared_ptr<const
 * HipExecutor>, ...)` of the reference (declared once for all backends in
 * core/<area>/<x>_kernels.hpp, enumerated in
 * core/device_hooks/common_kernels.inc.cpp:182-845).  The C++ shims that
 * re-create the reference signatures on top of these symbols are in
 * repo-8852-ginkgo_amd/include/ginkgo/ and shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; all data pointers are DEVICE pointers
 *    unless a parameter is named host_*;
 *  - values are fp64 (`_f64`), indices int32 (`_i32`); dense matrices are
 *    row-major with an explicit stride (gko::matrix::Dense layout);
 *  - scalars alpha/beta/rho... live in device memory (1x1 or 1xnrhs Dense in
 *    the reference);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream, the
 *    reference's choice, common/cuda_hip/base/kernel_launch.hpp.inc:66);
 *  - every function returns 0 on success, a positive hipError_t value when
 *    the HIP runtime failed, or a negative GKOMI_E* code;
 *  - nothing here allocates or synchronizes unless documented;
 *    workspace is caller-provided (reference: array<char>& tmp).
 */
#ifndef GKOMI_H_
#define GKOMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gkomi_stream_t;

enum {
    GKOMI_SUCCESS = 0,
    GKOMI_EINVAL = -1,       /* bad argument (gko::BadDimension / ValueMismatch) */
    GKOMI_ENOTSUPPORTED = -2, /* gko::NotSupported */
    GKOMI_ENOTIMPL = -3,      /* gko::NotImplemented */
    GKOMI_EWORKSPACE = -4,    /* workspace too small */
    GKOMI_ECOMM = -5,         /* a collective failed (gko::MpiError's role) */
    GKOMI_ETRS_OVERRUN = -6   /* a triangular solve hit its spin bound: x holds NaNs */
};

/* stopping_status bit layout (include/ginkgo/core/stop/stopping_status.hpp:144-147) */
#define GKOMI_STATUS_CONVERGED 0x80u
#define GKOMI_STATUS_FINALIZED 0x40u
#define GKOMI_STATUS_ID_MASK 0x3fu

/* ---- library / device ------------------------------------------------- */

/* version_info::get_hip_version analogue (core/device_hooks/hip_hooks.cpp:45) */
const char* gkomi_version(void);
/* HipExecutor::get_num_devices (hip/base/executor.hip.cpp) */
int gkomi_get_num_devices(int* count);
/* HipExecutor::set_gpu_property: CUs, wave size, LDS bytes, XCDs.
 * out[0]=num CUs, out[1]=wavefront size, out[2]=LDS bytes/CU, out[3]=L2 bytes */
int gkomi_device_properties(int device, int64_t out[4]);
/* HipExecutor::synchronize */
int gkomi_synchronize(gkomi_stream_t stream);
/* readable name of the last HIP error code returned (Hip*Error::get_error) */
const char* gkomi_error_string(int code);

/* HipExecutor memory interface (hip/base/executor.hip.cpp: raw_alloc, raw_free,
 * raw_copy_to x4, core/device_hooks/hip_hooks.cpp:66-112) so that a host
 * program needs no HIP headers.  kind: 0 host->device, 1 device->host,
 * 2 device->device; copies are synchronous like the reference's hipMemcpy. */
int gkomi_set_device(int device);
int gkomi_raw_alloc(size_t num_bytes, void** out_ptr);
int gkomi_raw_free(void* ptr);
int gkomi_raw_copy(void* dst, const void* src, size_t num_bytes, int kind);

/* Profiler ranges around an operation: what HipExecutor::run does with the logger events
 * operation_launched / operation_completed (include/ginkgo/core/base/executor.hpp:1153-1158),
 * as roctx ranges (rocprofv3 --marker-trace).  roctx is opened at run time; no-ops without it. */
int64_t gkomi_roctx_available(void);
int gkomi_roctx_push(const char* name);
int gkomi_roctx_pop(void);

/* ---- CSR SpMV (core/matrix/csr_kernels.hpp:58-75) ----------------------- */

/* Kernel selection, the role of Csr::strategy_type / srow
 * (include/ginkgo/core/matrix/csr.hpp:170-705). */
enum {
    GKOMI_CSR_AUTO = 0,    /* "automatical": pick from row statistics        */
    GKOMI_CSR_STREAM = 1,  /* row-block streaming through LDS, bit-exact      */
    GKOMI_CSR_VECTOR = 2,  /* "classical": one sub-wave per row               */
    GKOMI_CSR_BALANCED = 3, /* "load_balance": nnz-split + segmented reduction */
    GKOMI_CSR_SPLIT = 4    /* nnz-split streaming through LDS over srow, bit-exact
                              (gkomi_csr_spmv_srow_f64_i32 only)               */
};
/* OR-ed into the strategy word: the caller's working set between two applies of
 * this matrix exceeds the 256 MiB Infinity Cache (a Krylov basis, other
 * matrices, ...), so the matrix arrays will come from HBM anyway: the automatic
 * strategy then streams them with nontemporal loads (no cache allocation),
 * which is what it does by itself for matrices larger than the cache.  Pure
 * speed (cold 18.6 -> 16.6 us on the 1M-row 5-pt matrix), never results. */
#define GKOMI_CSR_STREAMING (1 << 24)
/* OR-ed into the strategy word: the gathers of b range over more than one XCD's
 * 4 MiB L2 keeps (what gkomi_csr_analyse_gather_i32 finds for uniformly random or
 * randomly permuted column patterns on > 400 k columns).  Half the gathers then
 * miss L2 and each miss moves a 128-B line over the fabric for 8 useful bytes:
 * the fabric, not HBM, bounds the SpMV (profiles/r04_gather_pmc.md).  With this
 * flag the load-balanced kernel (GKOMI_CSR_BALANCED, or the automatic strategy on
 * a matrix whose long rows select it anyway) runs one pass per window of 4 MiB of
 * b: the matrix is streamed once per window, more of b comes from L2 (power-law
 * class 140 -> 126 us, uniformly random columns 172 -> 144 us).  The partial sums
 * of the passes add up in c: tolerance parity like the reference's own load_balance
 * kernel, not bit for bit -- which is why matrices of short rows keep their
 * bit-exact kernels under the automatic strategy, flag or not. */
#define GKOMI_CSR_COLBLOCK (1 << 25)

/* csr::spmv  c = A b   and   csr::advanced_spmv  c = alpha A b + beta c.
 * alpha == NULL && beta == NULL selects the simple form (c is never read).
 * nnz = Csr::get_num_stored_elements() (known on the host in the reference;
 * pass -1 if unknown: the kernels that need it are then not selected).
 * reference/matrix/csr_kernels.cpp:75-128. */
int gkomi_csr_spmv_f64_i32(gkomi_stream_t stream, int64_t nrows, int64_t ncols,
                           int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
                           const int32_t* col_idxs, const double* vals,
                           const double* b, int64_t b_stride, double* c,
                           int64_t c_stride, const double* alpha,
                           const double* beta, int strategy,
                           int64_t max_row_nnz_hint);

/* Csr::make_srow (include/ginkgo/core/matrix/csr.hpp:1139-1157; load_balance::process
 * :395-459 fills srow with the row each wavefront's share of the nonzeros starts
 * in).  Ours: srow[t] = first row whose row_ptrs entry is >= t * tile, for
 * t = 0 .. nnz / tile + 1 -- the rows that START in every tile of `tile`
 * nonzeros -- and, in one more entry behind them, the most rows that start in any
 * one tile (beyond 2048 the kernel hands the rows behind a tile's first 512 out
 * by row index: a matrix with long runs of empty rows -- the non-local block of a
 * distributed matrix, a selection matrix -- does not leave 10^5 rows to one
 * workgroup).  Built once per matrix (whenever row_ptrs changes) into a caller-owned
 * device int32 array of gkomi_csr_srow_entries(nnz, tile) entries -- the kernels
 * read ALL of them, an srow must come from gkomi_csr_make_srow_*;
 * gkomi_csr_srow_tile_for(nnz) is the tile the kernels are tuned for at that size
 * (1536 while the matrix can be Infinity-Cache resident, 2048 / 3072 for matrices
 * that stream from HBM: 300 vs 309 us on the 256^3 7-point matrix;
 * gkomi_csr_srow_tile() = the former).  Two small launches, no synchronisation. */
int64_t gkomi_csr_srow_tile(void);
int64_t gkomi_csr_srow_tile_for(int64_t nnz);
int64_t gkomi_csr_srow_entries(int64_t nnz, int64_t tile);
int gkomi_csr_make_srow_i32(gkomi_stream_t stream, int64_t nrows, int64_t nnz,
                            const int32_t* row_ptrs, int64_t tile, int32_t* srow,
                            int64_t nsrow);

/* csr::spmv / advanced_spmv of a matrix that carries its srow (the reference's
 * Csr always does: csr.hpp:1265-1266, passed to the kernels by
 * hip/matrix/csr_kernels.hip.cpp:293-309).  Same contract as
 * gkomi_csr_spmv_f64_i32; with srow the automatic strategy cuts the work by
 * nonzeros (GKOMI_CSR_SPLIT) for matrices of short rows, whose streaming loads
 * then do not wait for row_ptrs.  srow == NULL: identical to
 * gkomi_csr_spmv_f64_i32.  max_row_nnz_hint stays advisory: a longer row is
 * still summed correctly, only slowly. */
int gkomi_csr_spmv_srow_f64_i32(gkomi_stream_t stream, int64_t nrows, int64_t ncols,
                                int64_t nrhs, int64_t nnz, const int32_t* row_ptrs,
                                const int32_t* col_idxs, const double* vals,
                                const double* b, int64_t b_stride, double* c,
                                int64_t c_stride, const double* alpha,
                                const double* beta, int strategy,
                                int64_t max_row_nnz_hint, const int32_t* srow,
                                int64_t srow_tile);

/* The <double, int64> instantiation of csr::spmv / advanced_spmv, Csr::make_srow and the
 * row statistic (GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE,
 * include/ginkgo/core/base/types.hpp:544-560): the one a 288 GB GPU needs beyond
 * <double, int32> (nnz > 2^31).  Same contracts as the _i32 entries; vals and
 * col_idxs 16-byte aligned (GKOMI_ENOTSUPPORTED otherwise); the automatic strategy
 * runs the nonzero-split kernel when the matrix carries its srow and the row-cut
 * stream kernel otherwise -- both bit-identical to the reference for any row lengths;
 * "classical" / "load_balance" requests are served by the stream kernel. */
int gkomi_csr_spmv_f64_i64(gkomi_stream_t stream, int64_t nrows, int64_t ncols,
                           int64_t nrhs, int64_t nnz, const int64_t* row_ptrs,
                           const int64_t* col_idxs, const double* vals,
                           const double* b, int64_t b_stride, double* c,
                           int64_t c_stride, const double* alpha,
                           const double* beta, int strategy,
                           int64_t max_row_nnz_hint);
int gkomi_csr_make_srow_i64(gkomi_stream_t stream, int64_t nrows, int64_t nnz,
                            const int64_t* row_ptrs, int64_t tile, int64_t* srow,
                            int64_t nsrow);
int gkomi_csr_spmv_srow_f64_i64(gkomi_stream_t stream, int64_t nrows, int64_t ncols,
                                int64_t nrhs, int64_t nnz, const int64_t* row_ptrs,
                                const int64_t* col_idxs, const double* vals,
                                const double* b, int64_t b_stride, double* c,
                                int64_t c_stride, const double* alpha,
                                const double* beta, int strategy,
                                int64_t max_row_nnz_hint, const int64_t* srow,
                                int64_t srow_tile);
/* result: device int64[1] */
int gkomi_csr_max_row_nnz_i64(gkomi_stream_t stream, int64_t nrows,
                              const int64_t* row_ptrs, int64_t* result);

/* One-time analysis of the column pattern (setup, like Csr::make_srow; blocking:
 * one small launch + an 16-byte copy): the mean over (a sample of) 1536-nonzero
 * tiles of the 64-KiB pages of b the tile's gathers touch, x the page size -- the
 * bytes of b a tile's gathers range over (*host_footprint_bytes, may be NULL; a
 * banded tile with a few far columns touches a few pages, a tile of uniformly
 * random columns all of them) -- and the strategy
 * flags the caller should OR into the strategy word of this matrix's applies
 * (*host_flags: GKOMI_CSR_COLBLOCK or 0).  scratch: 2 device doubles.
 * Role of the row statistics the reference's strategy objects collect in
 * process() (include/ginkgo/core/matrix/csr.hpp:600-705). */
int gkomi_csr_analyse_gather_i32(gkomi_stream_t stream, int64_t ncols, int64_t nnz,
                                 const int32_t* col_idxs, double* scratch,
                                 int* host_flags, int64_t* host_footprint_bytes);

/* ---- <float, int32>: the single-precision instantiation of the core of the path ------
 * GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE (include/ginkgo/core/base/types.hpp:544-560)
 * instantiates every kernel for float too.  Here: csr::spmv / advanced_spmv
 * (core/matrix/csr_kernels.hpp:58-75; reference/matrix/csr_kernels.cpp:75-128), the dense
 * BLAS-1 kernels (reference/matrix/dense_kernels.cpp:127-364), the CG kernels
 * (reference/solver/cg_kernels.cpp:53-123), stop::residual_norm
 * (reference/stop/residual_norm_kernels.cpp:57-83) and a Cg driver on them
 * (core/solver/cg.cpp:107-193: the reference's kernel sequence, Identity preconditioner,
 * Combined(Iteration, ResidualNorm), one right-hand side).  Same contracts as the _f64
 * entries of the same names; csrc/f32.hip.  csr::spmv and the elementwise kernels are
 * bit-identical to the reference executor's float instantiation (every intermediate a
 * float, no contraction), reductions are two-stage and reproducible (tolerance parity).
 * Everything else of the path stays <double, *>. */
int gkomi_csr_spmv_f32_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs,
                           int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs,
                           const float* vals, const float* b, int64_t b_stride, float* c,
                           int64_t c_stride, const float* alpha, const float* beta);
int gkomi_dense_fill_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, float* x,
                         int64_t stride, float value);
int gkomi_dense_copy_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols, const float* in,
                         int64_t in_stride, float* out, int64_t out_stride);
int gkomi_dense_scale_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                          const float* alpha, int64_t alpha_ncols, float* x,
                          int64_t stride);
int gkomi_dense_inv_scale_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                              const float* alpha, int64_t alpha_ncols, float* x,
                              int64_t stride);
int gkomi_dense_add_scaled_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                               const float* alpha, int64_t alpha_ncols, const float* x,
                               int64_t x_stride, float* y, int64_t y_stride);
int gkomi_dense_sub_scaled_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                               const float* alpha, int64_t alpha_ncols, const float* x,
                               int64_t x_stride, float* y, int64_t y_stride);
size_t gkomi_dense_reduction_workspace_bytes_f32(int64_t nrows, int64_t ncols);
int gkomi_dense_compute_dot_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                                const float* x, int64_t x_stride, const float* y,
                                int64_t y_stride, float* result, void* workspace,
                                size_t workspace_bytes);
int gkomi_dense_compute_norm2_f32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                                  const float* x, int64_t x_stride, float* result,
                                  void* workspace, size_t workspace_bytes);
int gkomi_cg_initialize_f32(gkomi_stream_t s, int64_t nrows, int64_t nrhs, const float* b,
                            int64_t b_stride, float* r, int64_t r_stride, float* z,
                            int64_t z_stride, float* p, int64_t p_stride, float* q,
                            int64_t q_stride, float* prev_rho, float* rho,
                            uint8_t* stop_status);
int gkomi_cg_step_1_f32(gkomi_stream_t s, int64_t nrows, int64_t nrhs, float* p,
                        int64_t p_stride, const float* z, int64_t z_stride,
                        const float* rho, const float* prev_rho,
                        const uint8_t* stop_status);
int gkomi_cg_step_2_f32(gkomi_stream_t s, int64_t nrows, int64_t nrhs, float* x,
                        int64_t x_stride, float* r, int64_t r_stride, const float* p,
                        int64_t p_stride, const float* q, int64_t q_stride,
                        const float* beta, const float* rho, const uint8_t* stop_status);
int gkomi_residual_norm_f32(gkomi_stream_t s, int64_t nrhs, const float* tau,
                            const float* orig_tau, float rel_residual_goal,
                            uint8_t stopping_id, int set_finalized, uint8_t* stop_status,
                            uint8_t* device_flags, uint8_t* host_flags);
size_t gkomi_cg_workspace_bytes_f32(int64_t n);
/* host_info[4] = { iterations, converged, ||r||, baseline norm }; baseline 0 rhs_norm,
 * 1 initial_resnorm, 2 absolute */
int gkomi_cg_solve_f32(gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs,
                       const int32_t* col_idxs, const float* vals, const float* b,
                       float* x, int64_t max_iters, float reduction, int baseline,
                       void* workspace, size_t workspace_bytes, double* host_info);

/* ---- column-partitioned copy: an analysis-based CSR strategy for scattered columns --
 * Role: the reference's `sparselib` strategy (hipSPARSE csrmv behind an analysis,
 * hip/matrix/csr_kernels.hip.cpp:293-330) -- a second representation built once per
 * matrix that makes the SpMV faster; opt-in.  The matrix is stored once more as a CSR of
 * nb * nrows VIRTUAL rows (virtual row k * nrows + r = row r's nonzeros in column block
 * k of nb, in their order), so that the workgroups resident at any moment gather from
 * ONE <= 2 MiB slice of b (it stays in every XCD's L2); the library's CSR kernels run on
 * the virtual matrix and a small kernel adds each row's nb partial sums in block order
 * (csrc/csr_colpart.hip; uniform random 16 per row on 1 M columns 172 -> 89 us,
 * power-law rows 126 -> 100 us).  Tolerance parity like load_balance (the groups of a
 * row are added in another association), one right-hand side.
 *   blocks_for  nb for a matrix of this shape, 0 = cannot pay (b within one L2, fewer
 *               than 4 nonzeros per row, slices beyond 6 MiB); slices of ~2 MiB but no
 *               more blocks than leave 1.25 nonzeros per (row, block) group; whether it DOES pay, create's
 *               timed analysis decides
 *   create      blocking set-up into `plan` (device memory, gkomi_csr_colpart_plan_bytes
 *               bytes, 16-B aligned, owned by the caller while the handle lives);
 *               nb in {2, 4, 8}, or 0: the analysis builds blocks_for's count and half
 *               of it, times a few applies of each and of the matrix's own automatic
 *               kernel, and keeps the faster copy -- or none (GKOMI_ENOTSUPPORTED, *out =
 *               NULL) when it does not beat that kernel by 10 % (plan_bytes with nb = 0
 *               is the room of the largest)
 *   refresh     the matrix's VALUES changed (same pattern): gathers them again -- the
 *               copy knows nothing of writes through Csr::get_values()
 *   spmv        c = A b (alpha = beta = NULL) or c = alpha A b + beta c; the partial sums
 *               live in the plan: a handle applies on one stream at a time
 *   info        out[5] = { nb, virtual rows, longest virtual row, srow tile, strategy word
 *               of the kernel the analysis kept for the virtual matrix } */
typedef struct gkomi_csr_colpart gkomi_csr_colpart;
int64_t gkomi_csr_colpart_blocks_for(int64_t nrows, int64_t ncols, int64_t nnz);
size_t gkomi_csr_colpart_plan_bytes(int64_t nrows, int64_t nnz, int64_t nb);
int gkomi_csr_colpart_create_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                                     int64_t nnz, const int32_t* row_ptrs,
                                     const int32_t* col_idxs, const double* vals,
                                     int64_t nb, void* plan, size_t plan_bytes,
                                     gkomi_csr_colpart** out);
int gkomi_csr_colpart_refresh_f64(gkomi_stream_t s, gkomi_csr_colpart* h,
                                  const double* vals);
int gkomi_csr_colpart_spmv_f64(gkomi_stream_t s, const gkomi_csr_colpart* h,
                               const double* b, int64_t b_stride, double* c,
                               int64_t c_stride, const double* alpha,
                               const double* beta);
int gkomi_csr_colpart_info(const gkomi_csr_colpart* h, int64_t* out);
void gkomi_csr_colpart_destroy(gkomi_csr_colpart* h);
/* the copy as the system matrix of the *_solve_op_f64 drivers (a gkomi_matrix_apply_fn,
 * declared below; ctx = the handle): a solve holds its matrix const, the one place where
 * the copy cannot go stale */
int gkomi_csr_colpart_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                                      const double* alpha, const double* b,
                                      int64_t b_stride, const double* beta, double* c,
                                      int64_t c_stride);

/* ell::compute_max_row_nnz analogue on a CSR row_ptrs array
 * (reference/matrix/ell_kernels.cpp:159-170 / csr strategy statistics).
 * result: device int32[1]. */
int gkomi_csr_max_row_nnz_i32(gkomi_stream_t stream, int64_t nrows,
                              const int32_t* row_ptrs, int32_t* result);

/* ---- dense BLAS-1 (core/matrix/dense_kernels.hpp) ----------------------- */
/* x is nrows x ncols row-major with stride; alpha is 1 x alpha_ncols on the
 * device with alpha_ncols in {1, ncols} (reference/matrix/dense_kernels.cpp:157-246). */
int gkomi_dense_fill_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                         double* x, int64_t stride, double value);
int gkomi_dense_copy_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                         const double* in, int64_t in_stride, double* out,
                         int64_t out_stride);
int gkomi_dense_scale_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                          const double* alpha, int64_t alpha_ncols, double* x,
                          int64_t stride);
int gkomi_dense_inv_scale_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                              const double* alpha, int64_t alpha_ncols,
                              double* x, int64_t stride);
int gkomi_dense_add_scaled_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                               const double* alpha, int64_t alpha_ncols,
                               const double* x, int64_t x_stride, double* y,
                               int64_t y_stride);
int gkomi_dense_sub_scaled_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                               const double* alpha, int64_t alpha_ncols,
                               const double* x, int64_t x_stride, double* y,
                               int64_t y_stride);
/* bytes of scratch the reductions below need for an nrows x ncols operand
 * (the reference's caller-cached array<char>& tmp). */
size_t gkomi_dense_reduction_workspace_bytes(int64_t nrows, int64_t ncols);
/* compute_dot / compute_conj_dot (identical for real values): result[j] = sum_i x[i,j] y[i,j] */
int gkomi_dense_compute_dot_f64(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                                const double* x, int64_t x_stride,
                                const double* y, int64_t y_stride,
                                double* result, void* workspace,
                                size_t workspace_bytes);
/* compute_norm2: result[j] = sqrt(sum_i x[i,j]^2) */
int gkomi_dense_compute_norm2_f64(gkomi_stream_t s, int64_t nrows,
                                  int64_t ncols, const double* x,
                                  int64_t x_stride, double* result,
                                  void* workspace, size_t workspace_bytes);
/* compute_squared_norm2 */
int gkomi_dense_compute_squared_norm2_f64(gkomi_stream_t s, int64_t nrows,
                                          int64_t ncols, const double* x,
                                          int64_t x_stride, double* result,
                                          void* workspace,
                                          size_t workspace_bytes);
/* compute_norm1: result[j] = sum_i |x[i,j]| */
int gkomi_dense_compute_norm1_f64(gkomi_stream_t s, int64_t nrows,
                                  int64_t ncols, const double* x,
                                  int64_t x_stride, double* result,
                                  void* workspace, size_t workspace_bytes);
/* compute_sqrt in place on an nrows x ncols matrix */
int gkomi_dense_compute_sqrt_f64(gkomi_stream_t s, int64_t nrows,
                                 int64_t ncols, double* x, int64_t stride);
/* row_gather: out[i,:] = in[rows[i],:]  (dense::row_gather, used to pack halos) */
int gkomi_dense_row_gather_f64_i32(gkomi_stream_t s, int64_t nout,
                                   int64_t ncols, const int32_t* rows,
                                   const double* in, int64_t in_stride,
                                   double* out, int64_t out_stride);

/* ---- CG step kernels (core/solver/cg_kernels.hpp:54-80) ----------------- */
/* stop_status: device uint8[nrhs] (array<stopping_status>) */
int gkomi_cg_initialize_f64(gkomi_stream_t s, int64_t nrows, int64_t nrhs,
                            const double* b, int64_t b_stride, double* r,
                            int64_t r_stride, double* z, int64_t z_stride,
                            double* p, int64_t p_stride, double* q,
                            int64_t q_stride, double* prev_rho, double* rho,
                            uint8_t* stop_status);
int gkomi_cg_step_1_f64(gkomi_stream_t s, int64_t nrows, int64_t nrhs,
                        double* p, int64_t p_stride, const double* z,
                        int64_t z_stride, const double* rho,
                        const double* prev_rho, const uint8_t* stop_status);
int gkomi_cg_step_2_f64(gkomi_stream_t s, int64_t nrows, int64_t nrhs,
                        double* x, int64_t x_stride, double* r,
                        int64_t r_stride, const double* p, int64_t p_stride,
                        const double* q, int64_t q_stride, const double* beta,
                        const double* rho, const uint8_t* stop_status);

/* ---- stopping criteria (core/stop/residual_norm_kernels.hpp,
 *      core/stop/criterion_kernels.hpp) ------------------------------------ */
/* residual_norm: for each rhs i: tau[i] < goal*orig_tau[i] -> converge(id, finalized).
 * device_flags: device uint8[2] = {all_converged, one_changed} (the reference's
 * array<bool> device_storage); the two values are also copied to the two host
 * bytes when host_flags != NULL (blocking, like
 * hip/stop/residual_norm_kernels.hip.cpp:119-120). */
int gkomi_residual_norm_f64(gkomi_stream_t s, int64_t nrhs, const double* tau,
                            const double* orig_tau, double rel_residual_goal,
                            uint8_t stopping_id, int set_finalized,
                            uint8_t* stop_status, uint8_t* device_flags,
                            uint8_t* host_flags);
/* implicit_residual_norm: sqrt(|tau|) < goal*orig_tau */
int gkomi_implicit_residual_norm_f64(gkomi_stream_t s, int64_t nrhs,
                                     const double* tau, const double* orig_tau,
                                     double rel_residual_goal,
                                     uint8_t stopping_id, int set_finalized,
                                     uint8_t* stop_status,
                                     uint8_t* device_flags,
                                     uint8_t* host_flags);
/* set_all_statuses: stop(id, finalized) on every entry */
int gkomi_set_all_statuses(gkomi_stream_t s, int64_t nrhs, uint8_t stopping_id,
                           int set_finalized, uint8_t* stop_status);

/* ---- ELL / SELL-P / COO / Hybrid SpMV (core/matrix/{ell,sellp,coo}_kernels.hpp,
 *      core/matrix/hybrid.cpp:133-159) -------------------------------------- */
/* alpha == beta == NULL: c = A b; otherwise c = alpha A b + beta c.
 * ELL is column-major: entry (row, i) at row + i*stride, padding col == -1
 * (reference/matrix/ell_kernels.cpp:57-157). */
int gkomi_ell_spmv_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                           int64_t nrhs, int64_t num_stored_per_row,
                           int64_t stride, const int32_t* col_idxs,
                           const double* vals, const double* b,
                           int64_t b_stride, double* c, int64_t c_stride,
                           const double* alpha, const double* beta);
/* SELL-P: entry (row r of slice s, i) at (slice_sets[s]+i)*slice_size + r;
 * slice_sets/slice_lengths are size_type = 64-bit (sellp.hpp:382,
 * reference/matrix/sellp_kernels.cpp:57-131). */
int gkomi_sellp_spmv_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                             int64_t nrhs, int64_t slice_size,
                             const uint64_t* slice_sets,
                             const uint64_t* slice_lengths,
                             const int32_t* col_idxs, const double* vals,
                             const double* b, int64_t b_stride, double* c,
                             int64_t c_stride, const double* alpha,
                             const double* beta);
/* coo::spmv / advanced_spmv = fill|scale + spmv2 (reference/matrix/coo_kernels.cpp:63-88) */
int gkomi_coo_spmv_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                           int64_t nrhs, int64_t nnz, const int32_t* row_idxs,
                           const int32_t* col_idxs, const double* vals,
                           const double* b, int64_t b_stride, double* c,
                           int64_t c_stride, const double* alpha,
                           const double* beta);
/* coo::spmv2 / advanced_spmv2: c += [alpha] A b (:92-131); alpha may be NULL */
int gkomi_coo_spmv2_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                            int64_t nrhs, int64_t nnz, const int32_t* row_idxs,
                            const int32_t* col_idxs, const double* vals,
                            const double* b, int64_t b_stride, double* c,
                            int64_t c_stride, const double* alpha);
/* The same four kernels for a COO matrix whose row_idxs are non-decreasing
 * (what Coo::read and every conversion to Coo produce; the reference's HIP
 * kernel, common/cuda_hip/matrix/coo_kernels.hpp.inc:57-222, is tuned for this
 * order too) -- no atomics, no fill / scale launch, bit-wise reproducible:
 *   1 <= max_row_nnz_hint <= 64 (the caller vouches that no row has more
 *     nonzeros): one launch; each tile also reads the 64 nonzeros in front of
 *     it and stores every row that ends inside it, summed from its first
 *     nonzero in the reference's order (c = A b is bit-identical to
 *     reference/matrix/coo_kernels.cpp:63-71 for every row);
 *   any other hint (-1: unknown): rows inside one tile get a plain store, rows
 *     cut by a tile boundary are finished by a second small launch from the
 *     partial sums in `workspace`.
 * gkomi_coo_analyse_rows_i32 (blocking) supplies both preconditions:
 * *host_sorted = 1 when row_idxs is non-decreasing, *host_max_row_nnz = the
 * longest run of equal row indices, capped at 65 (may be NULL).  Unsorted input
 * must use the entries above.  A hint that was too small is recorded in the
 * workspace (sticky) and read back by gkomi_coo_sorted_check (blocking; the
 * result of such a call is wrong).
 * workspace: gkomi_coo_sorted_workspace_bytes(nnz, nrhs) bytes; the carry
 * slots need not be kept between calls.  GKOMI_ENOTSUPPORTED when vals is not
 * 16-B or the index arrays are not 8-B aligned (the entries above take any
 * alignment). */
size_t gkomi_coo_sorted_workspace_bytes(int64_t nnz, int64_t nrhs);
int gkomi_coo_analyse_rows_i32(gkomi_stream_t s, int64_t nnz,
                               const int32_t* row_idxs, void* workspace,
                               size_t workspace_bytes, int* host_sorted,
                               int64_t* host_max_row_nnz);
int gkomi_coo_sorted_check(gkomi_stream_t s, const void* workspace,
                           int* host_flag);
int gkomi_coo_spmv_sorted_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
    const int32_t* row_idxs, const int32_t* col_idxs, const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha, const double* beta, int64_t max_row_nnz_hint,
    void* workspace, size_t workspace_bytes);
int gkomi_coo_spmv2_sorted_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int64_t nrhs, int64_t nnz,
    const int32_t* row_idxs, const int32_t* col_idxs, const double* vals,
    const double* b, int64_t b_stride, double* c, int64_t c_stride,
    const double* alpha, int64_t max_row_nnz_hint, void* workspace,
    size_t workspace_bytes);
/* Hybrid::apply_impl: ELL apply then COO apply2 */
int gkomi_hybrid_spmv_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                              int64_t nrhs, int64_t ell_num_stored_per_row,
                              int64_t ell_stride, const int32_t* ell_col_idxs,
                              const double* ell_vals, int64_t coo_nnz,
                              const int32_t* coo_row_idxs,
                              const int32_t* coo_col_idxs,
                              const double* coo_vals, const double* b,
                              int64_t b_stride, double* c, int64_t c_stride,
                              const double* alpha, const double* beta);

/* ---- index components and format conversions (bit-exact) ---------------- */
/* components::prefix_sum: exclusive, in place
 * (reference/components/prefix_sum_kernels.cpp:43-53) */
size_t gkomi_prefix_sum_workspace_bytes(int64_t n);
int gkomi_prefix_sum_i32(gkomi_stream_t s, int32_t* counts, int64_t n,
                         void* workspace, size_t workspace_bytes);
int gkomi_prefix_sum_i64(gkomi_stream_t s, int64_t* counts, int64_t n,
                         void* workspace, size_t workspace_bytes);
/* components::convert_ptrs_to_idxs / convert_idxs_to_ptrs / convert_ptrs_to_sizes
 * (reference/components/format_conversion_kernels.cpp:50-92); idxs_to_ptrs
 * needs prefix-sum scratch for num_blocks + 1 entries */
int gkomi_convert_ptrs_to_idxs_i32(gkomi_stream_t s, const int32_t* ptrs,
                                   int64_t num_blocks, int32_t* idxs);
int gkomi_convert_idxs_to_ptrs_i32(gkomi_stream_t s, const int32_t* idxs,
                                   int64_t num_idxs, int64_t num_blocks,
                                   int32_t* ptrs, void* workspace,
                                   size_t workspace_bytes);
int gkomi_convert_ptrs_to_sizes_i32(gkomi_stream_t s, const int32_t* ptrs,
                                    int64_t num_blocks, uint64_t* sizes);
/* the same three for int64 index arrays (reference/components/format_conversion_kernels.cpp:50-95,
 * instantiated for both index types) */
int gkomi_convert_ptrs_to_idxs_i64(gkomi_stream_t s, const int64_t* ptrs,
                                   int64_t num_blocks, int64_t* idxs);
int gkomi_convert_idxs_to_ptrs_i64(gkomi_stream_t s, const int64_t* idxs,
                                   int64_t num_idxs, int64_t num_blocks,
                                   int64_t* ptrs, void* workspace,
                                   size_t workspace_bytes);
int gkomi_convert_ptrs_to_sizes_i64(gkomi_stream_t s, const int64_t* ptrs,
                                    int64_t num_blocks, uint64_t* sizes);
/* csr::convert_to_ell (reference/matrix/csr_kernels.cpp:431-459) */
int gkomi_csr_convert_to_ell_f64_i32(gkomi_stream_t s, int64_t nrows,
                                     const int32_t* row_ptrs,
                                     const int32_t* col_idxs,
                                     const double* vals,
                                     int64_t num_stored_per_row,
                                     int64_t stride, int32_t* ell_col_idxs,
                                     double* ell_vals);
/* sellp::compute_slice_sets (reference/matrix/sellp_kernels.cpp:134-159):
 * slice_lengths[num_slices], slice_sets[num_slices + 1] */
int gkomi_sellp_compute_slice_sets_i32(gkomi_stream_t s,
                                       const int32_t* row_ptrs, int64_t nrows,
                                       int64_t slice_size,
                                       int64_t stride_factor,
                                       uint64_t* slice_sets,
                                       uint64_t* slice_lengths,
                                       void* workspace, size_t workspace_bytes);
/* csr::convert_to_sellp (reference/matrix/csr_kernels.cpp:385-425) */
int gkomi_csr_convert_to_sellp_f64_i32(gkomi_stream_t s, int64_t nrows,
                                       const int32_t* row_ptrs,
                                       const int32_t* col_idxs,
                                       const double* vals, int64_t slice_size,
                                       const uint64_t* slice_sets,
                                       const uint64_t* slice_lengths,
                                       int32_t* out_col_idxs, double* out_vals);
/* hybrid::compute_coo_row_ptrs (reference/matrix/hybrid_kernels.cpp:60-71): nrows + 1 entries */
int gkomi_hybrid_compute_coo_row_ptrs_i32(gkomi_stream_t s,
                                          const int32_t* row_ptrs,
                                          int64_t nrows, int64_t ell_lim,
                                          int64_t* coo_row_ptrs,
                                          void* workspace,
                                          size_t workspace_bytes);
/* csr::convert_to_hybrid (reference/matrix/csr_kernels.cpp:768-812) */
int gkomi_csr_convert_to_hybrid_f64_i32(
    gkomi_stream_t s, int64_t nrows, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, const int64_t* coo_row_ptrs,
    int64_t ell_lim, int64_t ell_stride, int32_t* ell_col_idxs,
    double* ell_vals, int32_t* coo_row_idxs, int32_t* coo_col_idxs,
    double* coo_vals);
/* Hybrid strategy_type::compute_ell_num_stored_elements_per_row
 * (include/ginkgo/core/matrix/hybrid.hpp:206-370), host side like the
 * reference (blocking copy of row_ptrs).  kind: 0 column_limit(num_columns),
 * 1 imbalance_limit(percent), 2 imbalance_bounded_limit(percent, ratio),
 * 3 minimal_storage_limit, 4 automatic. */
int gkomi_hybrid_ell_width_i32(gkomi_stream_t s, const int32_t* row_ptrs,
                               int64_t nrows, int kind, double percent,
                               double ratio, int64_t num_columns,
                               int64_t* host_result);

/* ---- block-Jacobi preconditioner (core/preconditioner/jacobi_kernels.hpp:50-190)
 * fp64 block storage (precision_reduction(0,0)) or adaptive precision (the
 * *_adaptive entry points); blocks use the reference's
 * block_interleaved_storage_scheme with max_block_stride = 64, the HIP
 * wavefront size (include/ginkgo/core/preconditioner/jacobi.hpp:62-167,
 * 578-609).  1 <= max_block_size <= 32. ---------------------------------- */
/* out = {block_offset, group_offset, group_power, stride} */
int gkomi_jacobi_storage_scheme(int max_block_size, int64_t out[4]);
/* storage_scheme.compute_storage_space(num_blocks), in values */
size_t gkomi_jacobi_storage_elements(int max_block_size, int64_t num_blocks);
/* jacobi::find_blocks (reference/preconditioner/jacobi_kernels.cpp:66-151):
 * block_ptrs has nrows + 1 entries; the count goes to num_blocks_device
 * (device int64) and, if host_num_blocks != NULL, to the host (blocking).
 * workspace: gkomi_jacobi_find_blocks_workspace_bytes(nrows). */
size_t gkomi_jacobi_find_blocks_workspace_bytes(int64_t nrows);
int gkomi_jacobi_find_blocks_i32(gkomi_stream_t s, int64_t nrows,
                                 const int32_t* row_ptrs,
                                 const int32_t* col_idxs, int max_block_size,
                                 int32_t* block_ptrs,
                                 int64_t* num_blocks_device, void* workspace,
                                 size_t workspace_bytes,
                                 int64_t* host_num_blocks);
/* jacobi::generate (:339-441): inverts every diagonal block (Gauss-Jordan,
 * implicit row pivoting) into `blocks`; conditioning (may be NULL) receives
 * ||D||_inf * ||D^-1||_inf per block as the reference computes it. */
int gkomi_jacobi_generate_f64_i32(gkomi_stream_t s, int64_t nrows,
                                  const int32_t* row_ptrs,
                                  const int32_t* col_idxs, const double* vals,
                                  int64_t num_blocks, int max_block_size,
                                  const int32_t* block_ptrs,
                                  double* conditioning, double* blocks);
/* jacobi::simple_apply (alpha == beta == NULL) / jacobi::apply (:505-561) */
int gkomi_jacobi_apply_f64_i32(gkomi_stream_t s, int64_t num_blocks,
                               int max_block_size, const int32_t* block_ptrs,
                               const double* blocks, int64_t nrhs,
                               const double* alpha, const double* b,
                               int64_t b_stride, const double* beta, double* x,
                               int64_t x_stride);
/* The same two kernels with the reference's ADAPTIVE block storage precision
 * (storage_optimization, include/ginkgo/core/preconditioner/jacobi.hpp:262-330;
 * core/preconditioner/jacobi_utils.hpp:46-201).  block_precisions[num_blocks]
 * holds gko::precision_reduction bytes ((preserving << 4) | nonpreserving;
 * 0xff = autodetect): on input the request per block, on output the precision
 * used, common to all blocks of a storage group.  Autodetection picks per
 * group the smallest of {double (0,0), float (0,1), half (0,2),
 * truncated<double,2> (1,0), truncated<float,2> (1,1), truncated<double,4>
 * (2,0)} with cond * eps < accuracy that passes the reference's feasibility
 * checks (jacobi_kernels.cpp:311-336).  conditioning is required.  half uses
 * the reference executor's conversion (extended_float.hpp:357-399:
 * truncation, flush below 2^-14). */
int gkomi_jacobi_generate_adaptive_f64_i32(
    gkomi_stream_t s, int64_t nrows, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int64_t num_blocks,
    int max_block_size, const int32_t* block_ptrs, double accuracy,
    double* conditioning, uint8_t* block_precisions, double* blocks);
int gkomi_jacobi_apply_adaptive_f64_i32(
    gkomi_stream_t s, int64_t num_blocks, int max_block_size,
    const int32_t* block_ptrs, const uint8_t* block_precisions,
    const double* blocks, int64_t nrhs, const double* alpha, const double* b,
    int64_t b_stride, const double* beta, double* x, int64_t x_stride);
/* jacobi::transpose_jacobi (= conj_transpose_jacobi for real values;
 * reference/preconditioner/jacobi_kernels.cpp:629-694, Jacobi::transpose,
 * core/preconditioner/jacobi.cpp): every stored block transposed in its storage
 * precision (block_precisions may be NULL = all fp64); out_blocks has the size
 * of blocks and must not alias it.  The transposed preconditioner BiCG needs. */
int gkomi_jacobi_transpose_f64_i32(gkomi_stream_t s, int64_t num_blocks,
                                   int max_block_size, const int32_t* block_ptrs,
                                   const uint8_t* block_precisions,
                                   const double* blocks, double* out_blocks);
/* scalar Jacobi (max_block_size == 1): csr::extract_diagonal
 * (reference/matrix/csr_kernels.cpp:1016-1034), jacobi::invert_diagonal
 * (:608-620), simple_scalar_apply / scalar_apply (:565-594) */
int gkomi_csr_extract_diagonal_f64_i32(gkomi_stream_t s, int64_t nrows,
                                       const int32_t* row_ptrs,
                                       const int32_t* col_idxs,
                                       const double* vals, double* diag);
int gkomi_jacobi_invert_diagonal_f64(gkomi_stream_t s, int64_t n,
                                     const double* diag, double* inv_diag);
int gkomi_jacobi_scalar_apply_f64(gkomi_stream_t s, int64_t nrows,
                                  int64_t nrhs, const double* inv_diag,
                                  const double* alpha, const double* b,
                                  int64_t b_stride, const double* beta,
                                  double* x, int64_t x_stride);

/* ---- sparse triangular solves = ILU apply (core/solver/{lower,upper}_trs_kernels.hpp;
 *      composition include/ginkgo/core/preconditioner/ilu.hpp:265-305) ------ */
/* x = L^-1 b / x = U^-1 b for a CSR matrix whose other triangle (if stored) is
 * ignored; unit_diag as solver::LowerTrs/UpperTrs::parameters (triangular.hpp:117-132).
 * x and b must not alias.  workspace: gkomi_trs_workspace_bytes() bytes of
 * ZEROED device memory (the reference's SolveStruct; generate() needs no analysis
 * for this variant).  reference/solver/lower_trs_kernels.cpp:90-120, upper_trs_kernels.cpp:90-123 */
size_t gkomi_trs_workspace_bytes(void);
int gkomi_lower_trs_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                  const int32_t* row_ptrs,
                                  const int32_t* col_idxs, const double* vals,
                                  int unit_diag, const double* b,
                                  int64_t b_stride, double* x, int64_t x_stride,
                                  void* workspace, size_t workspace_bytes);
int gkomi_upper_trs_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                  const int32_t* row_ptrs,
                                  const int32_t* col_idxs, const double* vals,
                                  int unit_diag, const double* b,
                                  int64_t b_stride, double* x, int64_t x_stride,
                                  void* workspace, size_t workspace_bytes);
/* *host_flag != 0 if a solve on this workspace hit its spin bound since the workspace
 * was zeroed by its owner (sticky: a later solve does not clear it; the role of the
 * nan_produced guard, cuda/solver/common_trs_kernels.cuh:444-449).  The solver drivers
 * of this library ask once per solve and return GKOMI_ETRS_OVERRUN for an Ilu
 * preconditioner (gkomi_ilu_apply_cb) whose solves gave up. */
int gkomi_trs_check_overrun(gkomi_stream_t s, const void* workspace,
                            int* host_flag);

/* ---- triangular solves with an analysis phase = LowerTrs/UpperTrs::generate ----
 * The reference analyses the factor once at generate() (hipsparseXcsrsv2_analysis into
 * its SolveStruct, hip/solver/common_trs_kernels.hip.hpp:61-253; sync-free CUDA variant
 * cuda/solver/common_trs_kernels.cuh:374-455) and solves with the result at every
 * apply.  Ours, in the same two steps:
 *   symbolic  dependency level of every row, rows sorted by level, 64-row slices:
 *             blocking (returns the sizes the caller needs to allocate the plan);
 *             host_out = { nslices, entries (SELL slots), nlevels, max_deps (longest
 *             dependency list of a row; -1 for the solve = unknown) }
 *   numeric   the factor once more in level order, dependencies only, column-major
 *             inside each slice, diagonal apart -> `plan` (device memory of
 *             gkomi_trs_plan_bytes(nslices, entries) bytes); re-run when the values
 *             of the factor change (same sparsity: same symbolic workspace)
 *   solve     x = L^-1 b / U^-1 b from the plan alone (row_ptrs/col_idxs/vals are not
 *             read again); one wave per slice, rows that become ready together,
 *             bit-identical to reference/solver/{lower,upper}_trs_kernels.cpp:90-123.
 *             x and b must not alias.
 * A solve that hits its spin bound leaves NaNs in x and raises a STICKY flag in the
 * plan (gkomi_trs_plan_check_overrun; cleared only by a new numeric phase). */
size_t gkomi_trs_symbolic_workspace_bytes(int64_t n);
int gkomi_trs_analyse_symbolic_i32(gkomi_stream_t s, int64_t n,
                                   const int32_t* row_ptrs,
                                   const int32_t* col_idxs, int lower,
                                   void* workspace, size_t workspace_bytes,
                                   int64_t* host_out);
size_t gkomi_trs_plan_bytes(int64_t nslices, int64_t entries);
int gkomi_trs_analyse_numeric_f64_i32(gkomi_stream_t s, int64_t n,
                                      const int32_t* row_ptrs,
                                      const int32_t* col_idxs,
                                      const double* vals, int lower,
                                      const void* symbolic_workspace,
                                      int64_t nslices, int64_t entries,
                                      int64_t nlevels, void* plan,
                                      size_t plan_bytes);
int gkomi_trs_solve_plan_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                             void* plan, int64_t nslices, int64_t entries,
                             int64_t max_deps, int unit_diag, const double* b,
                             int64_t b_stride, double* x, int64_t x_stride);
int gkomi_trs_plan_check_overrun(gkomi_stream_t s, const void* plan,
                                 int* host_flag);
/* The rule `generate` follows when it chooses between the level plan (1) and the
 * analysis-free kernel (0) for a factor whose symbolic analysis reported nlevels and
 * max_deps (out[2], out[3] of gkomi_trs_analyse_symbolic_i32): wide levels, or any
 * factor of at most 4096 rows with at most 8 dependencies per row -- those are
 * solved by ONE workgroup with x in LDS and a barrier per level (the reference's
 * ani4 factors: 170 -> ~30 us per solve) -- take the plan. */
int64_t gkomi_trs_use_plan(int64_t n, int64_t nlevels, int64_t max_deps);
/* ... and whether a box-grid factor whose brick analysis found levels_estimate levels and
 * coarse_levels brick-to-brick hand-offs on its longest path should take the brick plan (1)
 * or the level plan (0): the cost model of profiles/r02_trs_bricks.md, with the
 * single-workgroup solve's price per level for factors of at most 4096 rows. */
int64_t gkomi_trs_prefer_bricks(int64_t n, int64_t levels_estimate, int64_t coarse_levels);

/* ---- the brick plan: a second analysis for factors of grid problems -------------
 * Same place in the reference (LowerTrs/UpperTrs::generate, common_trs_kernels.hip.hpp:61-253),
 * same numerical contract (reference/solver/{lower,upper}_trs_kernels.cpp:90-123, bit-identical).
 * The level plan above pays one memory hand-off per dependency level; factors of stencil
 * matrices have hundreds to thousands of levels.  This analysis recovers the box grid from the
 * factor's dependency offsets (a divisor chain 1 | nx | nx ny ...), cuts the rows into bricks of
 * about `brick_rows` rows (<= 0: chosen: 8 x 8 x 27 / 45 x 45) and lets ONE workgroup solve a brick out of LDS, a level
 * inside a brick costing an LDS round trip (~0.15 us) instead of a hand-off through memory.
 * mode 2 (= 0, default): PIPELINED -- a brick starts at once, a second wave pumps its inflow from
 * memory into LDS while the one compute wave runs; x, pre-filled with a NaN sentinel, is its own ready
 * flag.  A brick trails its neighbour by a brick edge and the critical path is about the levels of the
 * factor.  `threads` is 64.  (A solve with x aliasing b, or with n * x_stride * 8 beyond 31 bits,
 * takes the kernel of mode 1 on the same plan.)  mode 1: a brick starts when the bricks it depends
 * on have FINISHED (`threads` = 64 / 128 / 256 compute threads, 0 = from the widest level; x may alias b;
 * 2 - 2.6 x the levels of the factor on the critical path).
 * The geometry is a guess that is never trusted: the brick graph is built from the actual
 * entries and must be acyclic, rows may have at most 8 dependencies, a brick with its inflow
 * must fit LDS -- else GKOMI_ENOTSUPPORTED (*out = NULL) and the caller keeps the level plan.
 *   create   blocking, host-side symbolic analysis (downloads row_ptrs / col_idxs); the handle
 *            owns host memory only -> gkomi_trs_bricks_destroy
 *   info     out[8] = { bricks, brick levels, steps, steps on the critical path of mode 1, LDS
 *            bytes of the largest brick, dependency slots per row, threads, mode }
 *   numeric  fills `plan` (device memory, gkomi_trs_bricks_plan_bytes(h) bytes) from the
 *            factor's values; again whenever the values change.  Blocking.
 *   solve    x = L^-1 b / U^-1 b from the plan alone; a handle solves on one stream at a time.
 * A solve whose bounded waits run out (GKOMI_TRS_MAX_POLLS polls, default 2^22) writes NaNs, and raises a
 * STICKY flag in the plan (gkomi_trs_bricks_check_overrun; cleared by the numeric phase).  The analysis
 * (gkomi_trs_bricks_create_i32) runs on the device -- the pattern never leaves HBM; one workgroup per
 * brick classifies the dependencies of its rows and relaxes their levels in LDS, the host only orders the
 * few hundred bricks -- and blocks until it is done; GKOMI_TRS_ANALYSIS=host selects round 2's analysis on
 * host threads (the one gkomi_trs_bricks_create_host_i32 runs, up to 8 threads, GKOMI_ANALYSIS_THREADS);
 * both produce the same arrays. */
typedef struct gkomi_trs_bricks gkomi_trs_bricks;
int gkomi_trs_bricks_create_i32(gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
                                const int32_t* col_idxs, int lower, int64_t brick_rows,
                                int threads, int mode, gkomi_trs_bricks** out);
/* gkomi_trs_bricks_levels_estimate: the dependency levels of the factor if the box geometry
 * the analysis guessed holds (sum of (extent - 1) + 1) -- for cost models only, so that a
 * caller that takes the brick plan does not need the level analysis just to count levels. */
int64_t gkomi_trs_bricks_levels_estimate(const gkomi_trs_bricks* h);
/* the same analysis from HOST copies of row_ptrs / col_idxs (no device needed), and read-only
 * views of the handle's host arrays for inspection: which = 0 perm (plan position -> row),
 * 1 brick_row_begin, 2 brick_step_ptr, 3 step_begin (top bit: the step opens a level),
 * 4 brick_ext_begin, 5 ext_col, 6 pred_ptr, 7 pred_idx, 8 row -> brick rank, 9 row -> LDS index,
 * 10 plan position -> index of its first inflow entry */
int gkomi_trs_bricks_create_host_i32(int64_t n, const int32_t* host_row_ptrs,
                                     const int32_t* host_col_idxs, int lower,
                                     int64_t brick_rows, int threads, int mode,
                                     gkomi_trs_bricks** out);
int gkomi_trs_bricks_host_array(const gkomi_trs_bricks* h, int which, const int32_t** data,
                                int64_t* count);
void gkomi_trs_bricks_destroy(gkomi_trs_bricks* h);
size_t gkomi_trs_bricks_plan_bytes(const gkomi_trs_bricks* h);
int gkomi_trs_bricks_info(const gkomi_trs_bricks* h, int64_t* out);
int gkomi_trs_bricks_numeric_f64_i32(gkomi_stream_t s, gkomi_trs_bricks* h,
                                     const int32_t* row_ptrs, const int32_t* col_idxs,
                                     const double* vals, void* plan, size_t plan_bytes);
int gkomi_trs_bricks_solve_f64(gkomi_stream_t s, gkomi_trs_bricks* h, void* plan,
                               int64_t nrhs, int unit_diag, const double* b,
                               int64_t b_stride, double* x, int64_t x_stride);
int gkomi_trs_bricks_check_overrun(gkomi_stream_t s, const void* plan, int* host_flag);

/* ---- ParILU(0) (core/factorization/par_ilu.cpp:74-163) -------------------- */
size_t gkomi_factorization_workspace_bytes(int64_t nrows);
/* factorization::add_diagonal_elements in two phases because the caller owns
 * the allocation (reference/factorization/factorization_kernels.cpp:54-160):
 * count (blocking copy of the count to *host_missing), then fill arrays of
 * nnz + missing entries and shift row_ptrs in place.  Same workspace for both. */
int gkomi_factorization_count_missing_diagonal_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, const int32_t* row_ptrs,
    const int32_t* col_idxs, void* workspace, size_t workspace_bytes,
    int64_t* host_missing);
int gkomi_factorization_add_diagonal_elements_f64_i32(
    gkomi_stream_t s, int64_t nrows, int64_t ncols, int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int32_t* new_col_idxs,
    double* new_vals, const void* workspace);
/* factorization::initialize_row_ptrs_l_u (:166-192); workspace = prefix-sum scratch for n + 1 */
int gkomi_factorization_initialize_row_ptrs_l_u_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
    const int32_t* col_idxs, int32_t* l_row_ptrs, int32_t* u_row_ptrs,
    void* workspace, size_t workspace_bytes);
/* factorization::initialize_l_u (:198-245) */
int gkomi_factorization_initialize_l_u_f64_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, const int32_t* l_row_ptrs,
    int32_t* l_col_idxs, double* l_vals, const int32_t* u_row_ptrs,
    int32_t* u_col_idxs, double* u_vals);
/* par_ilu_factorization::compute_l_u_factors
 * (reference/factorization/par_ilu_kernels.cpp:54-120): `iterations`
 * asynchronous sweeps over the COO entries of A; U is passed transposed (CSC of
 * U = CSR of U^T).  iterations == 0 ("Auto") = 10 sweeps, the reference's HIP choice. */
int gkomi_par_ilu_compute_l_u_factors_f64_i32(
    gkomi_stream_t s, int64_t iterations, int64_t nnz,
    const int32_t* coo_row_idxs, const int32_t* coo_col_idxs,
    const double* coo_vals, const int32_t* l_row_ptrs,
    const int32_t* l_col_idxs, double* l_vals, const int32_t* ut_row_ptrs,
    const int32_t* ut_col_idxs, double* ut_vals);
/* ParIC: factorization::initialize_row_ptrs_l /
 * initialize_l (reference/factorization/factorization_kernels.cpp:251-318;
 * diag_sqrt != 0 stores sqrt of the diagonal, 1 if not finite) and
 * par_ic_factorization::{init_factor, compute_factor}
 * (reference/factorization/par_ic_kernels.cpp:55-124).  compute_factor gets the
 * row index of every stored entry of L (convert_ptrs_to_idxs of l_row_ptrs) and
 * a copy of the lower-triangle values of A in L's pattern (the COO copy of
 * core/factorization/par_ic.cpp:121-131); `iterations` asynchronous sweeps,
 * 0 = 10.  The Ic preconditioner is gkomi_ilu_apply_cb with U = L^T
 * (gkomi_csr_transpose_f64_i32). */
int gkomi_factorization_initialize_row_ptrs_l_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
    const int32_t* col_idxs, int32_t* l_row_ptrs, void* workspace,
    size_t workspace_bytes);
int gkomi_factorization_initialize_l_f64_i32(
    gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, const int32_t* l_row_ptrs,
    int32_t* l_col_idxs, double* l_vals, int diag_sqrt);
int gkomi_par_ic_init_factor_f64_i32(gkomi_stream_t s, int64_t n,
                                     const int32_t* l_row_ptrs,
                                     const int32_t* l_col_idxs, double* l_vals);
int gkomi_par_ic_compute_factor_f64_i32(gkomi_stream_t s, int64_t iterations,
                                        int64_t l_nnz,
                                        const int32_t* l_row_idxs,
                                        const double* a_lower_vals,
                                        const int32_t* l_row_ptrs,
                                        const int32_t* l_col_idxs,
                                        double* l_vals);
/* ---- matrix assembly: device_matrix_data ---------- */
/* components::{sort_row_major, sum_duplicates, remove_zeros}
 * (core/base/device_matrix_data_kernels.hpp;
 * reference/base/device_matrix_data_kernels.cpp:84-190) on SoA triplets in
 * device memory.  sort_row_major is in place and STABLE (the reference's
 * std::sort leaves the order of duplicate (row, col) entries unspecified).
 * remove_zeros / sum_duplicates write into caller arrays of capacity nnz and
 * return the new count through host_nnz (blocking), the decision the
 * reference takes with array::resize_and_reset; sum_duplicates expects
 * sorted input and adds each run left to right starting from 0 (bit-exact).
 * Csr::read is then gkomi_convert_idxs_to_ptrs_i32 on the sorted row indices
 * (core/matrix/csr.cpp:453-470).  nnz < 2^31 - 1. */
size_t gkomi_matrix_data_workspace_bytes(int64_t nnz);
int gkomi_matrix_data_sort_row_major_f64_i32(gkomi_stream_t s, int64_t nnz,
                                             int32_t* row_idxs,
                                             int32_t* col_idxs, double* values,
                                             void* workspace,
                                             size_t workspace_bytes);
int gkomi_matrix_data_remove_zeros_f64_i32(
    gkomi_stream_t s, int64_t nnz, const int32_t* row_idxs,
    const int32_t* col_idxs, const double* values, int32_t* out_row_idxs,
    int32_t* out_col_idxs, double* out_values, void* workspace,
    size_t workspace_bytes, int64_t* host_nnz);
int gkomi_matrix_data_sum_duplicates_f64_i32(
    gkomi_stream_t s, int64_t nnz, const int32_t* row_idxs,
    const int32_t* col_idxs, const double* values, int32_t* out_row_idxs,
    int32_t* out_col_idxs, double* out_values, void* workspace,
    size_t workspace_bytes, int64_t* host_nnz);

/* csr::sort_by_column_index / is_sorted_by_column_index
 * (reference/matrix/csr_kernels.cpp:969-1009), the first step of the
 * factorizations unless skip_sorting is set.  The sort is stable.
 * is_sorted needs 4 bytes of workspace and blocks. */
int gkomi_csr_sort_by_column_index_f64_i32(gkomi_stream_t s, int64_t nrows,
                                           const int32_t* row_ptrs,
                                           int32_t* col_idxs, double* vals);
int gkomi_csr_is_sorted_by_column_index_i32(gkomi_stream_t s, int64_t nrows,
                                            const int32_t* row_ptrs,
                                            const int32_t* col_idxs,
                                            void* workspace,
                                            size_t workspace_bytes,
                                            int* host_is_sorted);
/* csr::transpose (reference/matrix/csr_kernels.cpp:551-586) */
size_t gkomi_csr_transpose_workspace_bytes(int64_t ncols);
int gkomi_csr_transpose_f64_i32(gkomi_stream_t s, int64_t nrows, int64_t ncols,
                                int64_t nnz, const int32_t* row_ptrs,
                                const int32_t* col_idxs, const double* vals,
                                int32_t* t_row_ptrs, int32_t* t_col_idxs,
                                double* t_vals, void* workspace,
                                size_t workspace_bytes);

/* ---- CG solver driver (core/solver/cg.cpp:107-193) ----------------------- */
/* Cg::apply_dense_impl for a CSR system matrix, an optional preconditioner
 * and the criteria Combined(Iteration(max_iters) [id 1], ResidualNorm(
 * reduction_factor, baseline) [id 2]) (core/stop/combined.cpp:40,
 * core/stop/residual_norm.cpp:119-228).  b, x: n x nrhs row-major, stride nrhs.
 *
 * precond: z = M r callback (NULL = Identity, which copies r to z like
 * matrix::Identity::apply); it must enqueue on `s` and not synchronize.
 * baseline: 0 rhs_norm, 1 initial_resnorm, 2 absolute
 *           (include/ginkgo/core/stop/residual_norm.hpp mode).
 * mode 0: the reference's kernel sequence, criterion checked on the host
 *         every iteration (one blocking 2-byte copy per iteration, like
 *         hip/stop/residual_norm_kernels.hip.cpp:119-120); any nrhs.
 * mode 1: fused single-rhs path -- 3 launches per iteration, scalars stay on
 *         the device, the device stops updating at the iteration the
 *         criterion fires (same iteration count and iterates as mode 0 up to
 *         reduction order) and the host polls the status every `check_every`
 *         iterations.  It moves 16 B per lane through x and the workspace
 *         vectors: when x or workspace is not 16-B aligned (hipMalloc gives
 *         256 B) the call runs mode 0 instead; CSR arrays may have any
 *         alignment (aligned ones get the SpMV + dot-product epilogue).
 * host_info (may be NULL): [0] iterations, [1] converged (1) / iteration
 * limit (0), then per rhs j: [2+2j] final ||r||_2 (recurrence residual),
 * [3+2j] baseline norm.  Needs 2 + 2 nrhs doubles.
 * Blocks until the solve is complete. */
typedef int (*gkomi_apply_fn)(void* ctx, gkomi_stream_t s, const double* in,
                              double* out);
size_t gkomi_cg_workspace_bytes(int64_t n, int64_t nrhs);
/* Diagnostics: how many solves of this process were finished by the
 * single-launch ("persistent") CG -- mode 1, Identity preconditioner, aligned
 * CSR with rows of at most 7 nonzeros (max_row_nnz_hint), 64 * #CU <= n <= about
 * 1 M rows: vectors and matrix then live in the register files for the whole
 * solve and the workgroups meet three times per iteration instead of the
 * kernel boundaries (16 vs 31 us per iteration on the 1M-row Poisson matrix).
 * Everything else -- and a solve whose workgroups could not all be resident --
 * runs the three-launch iteration.  GKOMI_CG_PERSISTENT=0 disables it. */
/* Diagnostics / test entries of the library's own device-wide stable radix sort and scans
 * (csrc/sort_scan.hip: what device_matrix_data::sort_row_major, the level analysis of the
 * triangular solves, Jacobi's block detection and build_local_nonlocal run -- no vendor
 * library).  Sort: ascending by key bits [0, end_bit), stable, keys of key_bytes = 4 or 8
 * bytes, optional 32-bit payloads (vals_in == NULL: keys only), outputs must not alias
 * inputs.  Scan kind 0: exclusive sum, 1: inclusive maximum, of int32 (in == out allowed);
 * its workspace: 4 bytes per 2048 items, rounded up to 256. */
size_t gkomi_diag_radix_sort_workspace_bytes(int64_t n, int key_bytes, int pairs);
int gkomi_diag_radix_sort(gkomi_stream_t s, int64_t n, int key_bytes, const void* keys_in,
                          void* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                          int end_bit, void* workspace, size_t workspace_bytes);
int gkomi_diag_scan_i32(gkomi_stream_t s, int kind, const int32_t* in, int32_t* out,
                        int64_t n, void* workspace, size_t workspace_bytes);
/* Diagnostics: a pure streaming kernel over the arrays of a CSR SpMV -- reads
 * vals, col_idxs, row_ptrs and b, writes c (c is OVERWRITTEN with meaningless
 * values), 16 B per lane, `blocks` workgroups of 256 threads, grid-stride --
 * i.e. exactly the 12 nnz + 4 (nrows + 1) + 8 nrows + 8 nrows bytes of
 * SURVEY 8(d) with no gather and no dependent access.  bench.py times it next to
 * the SpMV as the practical ceiling of the box for that byte mix.  All arrays
 * 16-byte aligned. */
int gkomi_diag_stream_csr_bytes(gkomi_stream_t s, int blocks, int64_t nrows,
                                int64_t nnz, const int32_t* row_ptrs,
                                const int32_t* col_idxs, const double* vals,
                                const double* b, double* c);
/* Diagnostics / benchmark support: the 7-point Poisson matrix of a g^3 grid
 * (row = (i g + j) g + k, ascending columns, 6 / -1: BASELINE config 5's matrix,
 * tests/matgen.py poisson_3d_7pt) written on the device, row_ptrs in closed form.
 * Arrays of g^3 + 1 and 7 g^3 - 6 g^2 entries.  The _i64 form builds matrices of
 * more than 2^31 nonzeros (700^3) in place. */
int gkomi_diag_poisson3d_7pt_f64_i32(gkomi_stream_t s, int64_t g, int32_t* row_ptrs,
                                     int32_t* col_idxs, double* vals);
int gkomi_diag_poisson3d_7pt_f64_i64(gkomi_stream_t s, int64_t g, int64_t* row_ptrs,
                                     int64_t* col_idxs, double* vals);
/* Diagnostics: automatic CSR applies of this process whose matrix the library's residency
 * tracker found evicted from the Infinity Cache (more than 256 MiB of other CSR applies since
 * its last one) and that therefore read the matrix with nontemporal loads. */
int64_t gkomi_diag_csr_evicted_applies(void);
int64_t gkomi_cg_persistent_solves(void);
/* Process-wide switch of the single-launch CG (what GKOMI_CG_PERSISTENT sets at
 * start-up): 0 = off (every solve runs the three-launch iteration), 1 = on. */
int gkomi_cg_persistent_enable(int mode);
/* Diagnostics: how many GMRES solves of this process had a meeting of the
 * single-launch Arnoldi step time out and were finished, from the x of the last
 * completed restart, by the launch-per-vector kernels.  GKOMI_MEET_MAX_POLLS
 * (test hook) shortens the wait of the CG and GMRES meetings. */
int64_t gkomi_gmres_meeting_fallbacks(void);
int gkomi_cg_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
                           int64_t nnz, const int32_t* row_ptrs,
                           const int32_t* col_idxs, const double* vals,
                           int spmv_strategy, int64_t max_row_nnz_hint,
                           gkomi_apply_fn precond, void* precond_ctx,
                           const double* b, double* x, int64_t max_iters,
                           double reduction_factor, int baseline, int mode,
                           int check_every, void* workspace,
                           size_t workspace_bytes, double* host_info);

/* ---- row-partitioned distributed matrix (core/distributed/{partition,matrix}_kernels.hpp) */
/* ---- BiCGSTAB / FCG / CGS ---------------------------
 * Step kernels of core/solver/{bicgstab,fcg,cgs}_kernels.hpp with the
 * semantics of reference/solver/{bicgstab_kernels.cpp:57-232,
 * fcg_kernels.cpp:55-145, cgs_kernels.cpp:55-185}: vectors n x nrhs row-major
 * with a stride, the per-column scalars (1 x nrhs) in device memory,
 * stop_status one byte per column.  Elementwise bit-exact.  The drivers are
 * {Bicgstab,Fcg,Cgs}::apply_dense_impl (core/solver/bicgstab.cpp:107-234,
 * fcg.cpp:104-196, cgs.cpp:107-205) for a CSR matrix, an optional
 * preconditioner and Combined(Iteration(max_iters), ResidualNorm(reduction,
 * baseline)), evaluated on the device at every point where the reference
 * evaluates it (the statuses stop the columns at once); the host looks at the
 * outcome every `check_every` evaluations (>= 1), the iterations launched
 * meanwhile change nothing, and the reported iteration count is the one the
 * device recorded.  Other arguments and host_info as for
 * gkomi_gmres_solve_f64_i32, workspace gkomi_krylov_workspace_bytes(n, nrhs). */
int gkomi_bicgstab_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
    const double* b, int64_t b_stride, double* r, int64_t r_stride, double*
    rr, int64_t rr_stride, double* y, int64_t y_stride, double* s_vec, int64_t
    s_stride, double* t, int64_t t_stride, double* z, int64_t z_stride,
    double* v, int64_t v_stride, double* p, int64_t p_stride, double*
    prev_rho, double* rho, double* alpha, double* beta, double* gamma, double*
    omega, uint8_t* stop_status);
int gkomi_bicgstab_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* r, int64_t r_stride, double* p, int64_t p_stride, const double* v,
    int64_t v_stride, const double* rho, const double* prev_rho, const double*
    alpha, const double* omega, const uint8_t* stop_status);
int gkomi_bicgstab_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* r, int64_t r_stride, double* s_vec, int64_t s_stride, const
    double* v, int64_t v_stride, const double* rho, double* alpha, const
    double* beta, const uint8_t* stop_status);
int gkomi_bicgstab_step_3_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
    double* x, int64_t x_stride, double* r, int64_t r_stride, const double*
    s_vec, int64_t s_stride, const double* t, int64_t t_stride, const double*
    y, int64_t y_stride, const double* z, int64_t z_stride, const double*
    alpha, const double* beta, const double* gamma, double* omega, const
    uint8_t* stop_status);
int gkomi_bicgstab_finalize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
    double* x, int64_t x_stride, const double* y, int64_t y_stride, const
    double* alpha, uint8_t* stop_status);
int gkomi_fcg_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* b, int64_t b_stride, double* r, int64_t r_stride, double* z,
    int64_t z_stride, double* p, int64_t p_stride, double* q, int64_t
    q_stride, double* t, int64_t t_stride, double* prev_rho, double* rho,
    double* rho_t, uint8_t* stop_status);
int gkomi_fcg_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* p,
    int64_t p_stride, const double* z, int64_t z_stride, const double* rho_t,
    const double* prev_rho, const uint8_t* stop_status);
int gkomi_fcg_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double* x,
    int64_t x_stride, double* r, int64_t r_stride, double* t, int64_t
    t_stride, const double* p, int64_t p_stride, const double* q, int64_t
    q_stride, const double* beta, const double* rho, const uint8_t*
    stop_status);
int gkomi_cgs_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* b, int64_t b_stride, double* r, int64_t r_stride, double* r_tld,
    int64_t r_tld_stride, double* p, int64_t p_stride, double* q, int64_t
    q_stride, double* u, int64_t u_stride, double* u_hat, int64_t
    u_hat_stride, double* v_hat, int64_t v_hat_stride, double* t, int64_t
    t_stride, double* alpha, double* beta, double* gamma, double* prev_rho,
    double* rho, uint8_t* stop_status);
int gkomi_cgs_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* r, int64_t r_stride, double* u, int64_t u_stride, double* p,
    int64_t p_stride, const double* q, int64_t q_stride, double* beta, const
    double* rho, const double* prev_rho, const uint8_t* stop_status);
int gkomi_cgs_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* u, int64_t u_stride, const double* v_hat, int64_t v_hat_stride,
    double* q, int64_t q_stride, double* t, int64_t t_stride, double* alpha,
    const double* rho, const double* gamma, const uint8_t* stop_status);
int gkomi_cgs_step_3_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* t, int64_t t_stride, const double* u_hat, int64_t u_hat_stride,
    double* r, int64_t r_stride, double* x, int64_t x_stride, const double*
    alpha, const uint8_t* stop_status);
size_t gkomi_krylov_workspace_bytes(int64_t n, int64_t nrhs);
int gkomi_bicgstab_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
    int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs, const
    double* vals, int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn
    precond, void* precond_ctx, const double* b, double* x, int64_t max_iters,
    double reduction_factor, int baseline, int64_t check_every, void* workspace, size_t
    workspace_bytes, double* host_info);
int gkomi_fcg_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t
    nnz, const int32_t* row_ptrs, const int32_t* col_idxs, const double* vals,
    int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond, void*
    precond_ctx, const double* b, double* x, int64_t max_iters, double
    reduction_factor, int baseline, int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
int gkomi_cgs_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t
    nnz, const int32_t* row_ptrs, const int32_t* col_idxs, const double* vals,
    int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn precond, void*
    precond_ctx, const double* b, double* x, int64_t max_iters, double
    reduction_factor, int baseline, int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);

/* BiCG (reference/solver/bicg_kernels.cpp:55-145; driver core/solver/bicg.cpp:
 * 117-232: t_* = csr::transpose of the system matrix, precond_t = the
 * transposed preconditioner, both NULL = Identity) and IR (ir_kernels.cpp:
 * 48-56; driver core/solver/ir.cpp:186-277 with the caller's x as initial
 * guess: x += relaxation_factor * inner(b - A x), inner == NULL = Richardson).
 * Other arguments as for gkomi_bicgstab_solve_f64_i32. */
int gkomi_bicg_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, const
    double* b, int64_t b_stride, double* r, int64_t r_stride, double* z,
    int64_t z_stride, double* p, int64_t p_stride, double* q, int64_t
    q_stride, double* prev_rho, double* rho, double* r2, int64_t r2_stride,
    double* z2, int64_t z2_stride, double* p2, int64_t p2_stride, double* q2,
    int64_t q2_stride, uint8_t* stop_status);
int gkomi_bicg_step_1_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double*
    p, int64_t p_stride, const double* z, int64_t z_stride, double* p2,
    int64_t p2_stride, const double* z2, int64_t z2_stride, const double* rho,
    const double* prev_rho, const uint8_t* stop_status);
int gkomi_bicg_step_2_f64(gkomi_stream_t s, int64_t n, int64_t nrhs, double*
    x, int64_t x_stride, double* r, int64_t r_stride, double* r2, int64_t
    r2_stride, const double* p, int64_t p_stride, const double* q, int64_t
    q_stride, const double* q2, int64_t q2_stride, const double* beta, const
    double* rho, const uint8_t* stop_status);
int gkomi_ir_initialize(gkomi_stream_t s, int64_t nrhs, uint8_t* stop_status);
int gkomi_bicg_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
    int64_t nnz, const int32_t* row_ptrs, const int32_t* col_idxs, const
    double* vals, const int32_t* t_row_ptrs, const int32_t* t_col_idxs, const
    double* t_vals, int spmv_strategy, int64_t max_row_nnz_hint,
    gkomi_apply_fn precond, void* precond_ctx, gkomi_apply_fn precond_t, void*
    precond_t_ctx, const double* b, double* x, int64_t max_iters, double
    reduction_factor, int baseline, int64_t check_every, void* workspace,
    size_t workspace_bytes, double* host_info);
int gkomi_ir_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t
    nnz, const int32_t* row_ptrs, const int32_t* col_idxs, const double* vals,
    int spmv_strategy, int64_t max_row_nnz_hint, gkomi_apply_fn inner, void*
    inner_ctx, double relaxation_factor, const double* b, double* x, int64_t
    max_iters, double reduction_factor, int baseline, void* workspace, size_t
    workspace_bytes, double* host_info);

/* Partition metadata on HOST arrays (O(#ranges); core/distributed/matrix.cpp
 * consumes it on the host for the communication plan):
 * reference/distributed/partition_kernels.cpp:42-135.
 * ranges: num_parts + 1; range_bounds: num_ranges + 1; part_ids, starting
 * indices: num_ranges; part_sizes: num_parts. */
int gkomi_partition_build_ranges_from_global_size(int64_t num_parts,
                                                  int64_t global_size,
                                                  int64_t* host_ranges);
int gkomi_partition_build_from_contiguous(int64_t num_parts,
                                          const int64_t* host_ranges,
                                          int64_t* host_range_bounds,
                                          int32_t* host_part_ids);
int gkomi_partition_build_from_mapping(int64_t n, const int32_t* host_mapping,
                                       int64_t* host_range_bounds,
                                       int32_t* host_part_ids,
                                       int64_t* host_num_ranges);
int gkomi_partition_build_starting_indices(const int64_t* host_range_bounds,
                                           const int32_t* host_part_ids,
                                           int64_t num_ranges,
                                           int64_t num_parts,
                                           int32_t* host_starting_indices,
                                           int32_t* host_part_sizes,
                                           int64_t* host_num_empty_parts);
/* partition::has_ordered_parts (reference/distributed/partition_kernels.cpp:139-155):
 * *host_result = 1 if the part ids of consecutive ranges never decrease, else 0.
 * Partition::has_connected_parts is num_parts - num_empty_parts == num_ranges
 * (core/distributed/partition.cpp:120-124; num_empty_parts from
 * gkomi_partition_build_starting_indices). */
int gkomi_partition_has_ordered_parts(const int32_t* host_part_ids,
                                      int64_t num_ranges, int64_t* host_result);
/* distributed_vector::build_local (reference/distributed/vector_kernels.cpp:47-96, what
 * Vector::read_distributed runs, core/distributed/vector.cpp:120-170): device COO
 * input with 64-bit global indices; local(local row, col) = value for every entry
 * whose row local_part owns; the rest of `local` (row-major, local_stride) is left
 * as the caller set it.  Partition arrays: device copies of the host metadata. */
int gkomi_dist_vector_build_local_f64(gkomi_stream_t s, int64_t nnz,
                                      const int64_t* rows, const int64_t* cols,
                                      const double* vals, const int64_t* range_bounds,
                                      const int32_t* part_ids, const int32_t* starts,
                                      int64_t num_ranges, int32_t local_part,
                                      double* local, int64_t local_stride);
/* distributed_matrix::build_local_nonlocal
 * (reference/distributed/matrix_kernels.cpp:49-190) on device-resident COO
 * input with 64-bit global indices; partition arrays are device copies of the
 * metadata above.  Two phases because the caller allocates the outputs:
 * sizes -> host_sizes = {num_local, num_non_local, num_unique_non_local_cols}
 * (blocking), then fill with the SAME workspace.  Global columns < 2^40. */
size_t gkomi_dist_build_workspace_bytes(int64_t nnz);
int gkomi_dist_build_local_nonlocal_sizes(
    gkomi_stream_t s, int64_t nnz, const int64_t* rows, const int64_t* cols,
    const int64_t* row_range_bounds, const int32_t* row_part_ids,
    const int32_t* row_starts, int64_t row_num_ranges,
    const int64_t* col_range_bounds, const int32_t* col_part_ids,
    const int32_t* col_starts, int64_t col_num_ranges, int32_t local_part,
    void* workspace, size_t workspace_bytes, int64_t host_sizes[3]);
int gkomi_dist_build_local_nonlocal_fill(
    gkomi_stream_t s, int64_t nnz, const int64_t* rows, const int64_t* cols,
    const double* vals, const int64_t* row_range_bounds,
    const int32_t* row_part_ids, const int32_t* row_starts,
    int64_t row_num_ranges, const int64_t* col_range_bounds,
    const int32_t* col_part_ids, const int32_t* col_starts,
    int64_t col_num_ranges, int64_t num_parts, const void* workspace,
    int64_t num_unique, int32_t* local_row_idxs, int32_t* local_col_idxs,
    double* local_vals, int32_t* non_local_row_idxs,
    int32_t* non_local_col_idxs, double* non_local_vals, int32_t* gather_idxs,
    int32_t* recv_sizes, int64_t* non_local_to_global);

/* ---- communicator (the role of gko::experimental::mpi::communicator for
 *      core/distributed/{matrix,vector}.cpp) -------------------------------------
 * The distributed drivers below only see this record: a context pointer and the
 * two collectives the path needs, so any transport can stand behind it.
 *   allreduce_sum_f64  in place on `count` doubles in device memory, ordered on
 *                      stream s; every rank ends with the same bits
 *                      (vector.cpp:317-409: MPI_Allreduce of the local results)
 *   alltoallv          rank p gets send_counts[p] elements from send_offsets[p] on
 *                      and delivers recv_counts[p] at recv_offsets[p]; counts and
 *                      offsets are HOST arrays of `size` entries, in elements of
 *                      elem_bytes bytes; device buffers; ordered on stream s
 *                      (matrix.cpp:198-224, 263-303: all_to_all_v)
 * gkomi_comm_rccl_*: the RCCL transport, one rank per GPU of a node (xGMI).  RCCL is
 * opened at run time; GKOMI_ENOTSUPPORTED if it cannot be.  Bootstrap like NCCL:
 * rank 0 draws a unique id (gkomi_comm_unique_id_bytes() bytes), every rank gets
 * it over whatever channel the launcher has and calls create (collective). */
typedef struct gkomi_comm {
    void* self;
    int rank;
    int size;
    int (*allreduce_sum_f64)(void* self, gkomi_stream_t s, double* buf, int64_t count);
    int (*alltoallv)(void* self, gkomi_stream_t s, const void* send,
                     const int64_t* send_counts, const int64_t* send_offsets,
                     void* recv, const int64_t* recv_counts,
                     const int64_t* recv_offsets, int elem_bytes);
} gkomi_comm;
int64_t gkomi_comm_unique_id_bytes(void);
int64_t gkomi_comm_rccl_available(void);
int gkomi_comm_rccl_unique_id(void* id_out);
int gkomi_comm_rccl_create(const void* id_in, int rank, int size, gkomi_comm* out);
int gkomi_comm_rccl_destroy(gkomi_comm* comm);
/* ncclCommCount / ncclCommUserRank of an RCCL communicator made by
 * gkomi_comm_rccl_create: what RCCL itself reports (bench.py prints it). */
int gkomi_comm_rccl_query(const gkomi_comm* comm, int* count, int* user_rank);
/* thin callers of the two function pointers (for bindings that cannot call through
 * a struct member) */
int gkomi_comm_allreduce_sum_f64(const gkomi_comm* comm, gkomi_stream_t s,
                                 double* buf, int64_t count);
int gkomi_comm_alltoallv(const gkomi_comm* comm, gkomi_stream_t s, const void* send,
                         const int64_t* send_counts, const int64_t* send_offsets,
                         void* recv, const int64_t* recv_counts,
                         const int64_t* recv_offsets, int elem_bytes);

/* ---- distributed::Matrix of one rank (core/distributed/matrix.hpp: local_mtx_,
 *      non_local_mtx_, gather_idxs_, send/recv sizes and offsets, one rhs) -------- */
typedef struct gkomi_dist_matrix {
    int64_t n_local;            /* rows owned = columns of the local block */
    int64_t n_halo;             /* columns of the non-local block = x entries received */
    int64_t l_nnz;              /* local block, CSR */
    const int32_t* l_row_ptrs;
    const int32_t* l_col_idxs;
    const double* l_vals;
    int64_t l_max_row_nnz;      /* hint, -1 unknown */
    const int32_t* l_srow;      /* optional (gkomi_csr_make_srow_i32), else NULL */
    int64_t l_srow_tile;
    int64_t nl_rows;            /* non-local block, reduced to its non-empty rows */
    int64_t nl_nnz;             /*   (gkomi_dist_nonlocal_rows_i32)               */
    const int32_t* nl_row_idxs; /* nl_rows local row indices, ascending */
    const int32_t* nl_row_ptrs; /* nl_rows + 1 */
    const int32_t* nl_col_idxs; /* into the received halo, 0 .. n_halo-1 */
    const double* nl_vals;
    int64_t send_total;
    const int32_t* gather_idxs; /* device: local rows to send, grouped by receiver */
    const int64_t* send_counts; /* host arrays, comm size entries each */
    const int64_t* send_offsets;
    const int64_t* recv_counts;
    const int64_t* recv_offsets;
    double* send_buf;           /* device, send_total doubles */
    double* recv_buf;           /* device, n_halo doubles */
} gkomi_dist_matrix;
/* side stream + events of the overlapped halo exchange (one per rank / thread) */
typedef struct gkomi_dist_ctx gkomi_dist_ctx;
int gkomi_dist_ctx_create(gkomi_dist_ctx** out);
int gkomi_dist_ctx_destroy(gkomi_dist_ctx* ctx);
/* rows of the non-local block that have entries: row_idxs_out[k] ascending and
 * compact_ptrs_out[k], k = 0 .. count (both sized n_local + 1); blocking (*host_count) */
size_t gkomi_dist_nonlocal_rows_workspace_bytes(int64_t n_local);
int gkomi_dist_nonlocal_rows_i32(gkomi_stream_t s, int64_t n_local,
                                 const int32_t* row_ptrs, int32_t* row_idxs_out,
                                 int32_t* compact_ptrs_out, void* workspace,
                                 size_t workspace_bytes, int64_t* host_count);
/* Matrix::apply_impl (matrix.cpp:307-335), one right-hand side: x = A b on the local
 * rows; the halo exchange overlaps the local SpMV.  Collective: every rank calls it. */
int gkomi_dist_matrix_apply_f64(gkomi_stream_t s, const gkomi_comm* comm,
                                gkomi_dist_ctx* ctx, const gkomi_dist_matrix* A,
                                const double* b, double* x);
/* Cg::apply_dense_impl on distributed vectors (core/solver/cg.cpp:107-193), fused:
 * three local kernels + two all-reduces (rho and tau^2 together, beta) + one halo
 * exchange per iteration, scalars device-resident, host poll every check_every
 * iterations.  precond: rank-local gkomi_apply_fn or NULL.  baseline / host_info as
 * gkomi_cg_solve_f64_i32 (host_info: 4 doubles).  Collective; blocks until done. */
size_t gkomi_dist_cg_workspace_bytes(int64_t n_local, int64_t nl_rows);
int gkomi_dist_cg_solve_f64(gkomi_stream_t s, const gkomi_comm* comm,
                            gkomi_dist_ctx* ctx, const gkomi_dist_matrix* A,
                            gkomi_apply_fn precond, void* precond_ctx,
                            const double* b, double* x, int64_t max_iters,
                            double reduction_factor, int baseline, int check_every,
                            void* workspace, size_t workspace_bytes,
                            double* host_info);

/* ---- preconditioners as solver callbacks ---------------------------------
 * Ready-made gkomi_apply_fn implementations and their context records, the
 * generated state of preconditioner::Jacobi / preconditioner::Ilu.  All
 * pointers are device pointers; the records themselves live on the host. */
/* A system matrix in any format as a callback for the *_solve_op_f64 drivers:
 * c = A b (alpha == beta == NULL) or c = alpha A b + beta c, vectors n x nrhs
 * row-major.  The records below + gkomi_*_matrix_apply_cb wrap the SpMV entry
 * points of this header; the C++ mirror passes any gko::LinOp this way. */
typedef int (*gkomi_matrix_apply_fn)(void* ctx, gkomi_stream_t s, int64_t nrhs,
                                     const double* alpha, const double* b,
                                     int64_t b_stride, const double* beta,
                                     double* c, int64_t c_stride);
typedef struct gkomi_csr_ctx {
    int64_t nrows, ncols, nnz;
    const int32_t* row_ptrs;
    const int32_t* col_idxs;
    const double* vals;
    int64_t strategy;          /* strategy word of gkomi_csr_spmv_f64_i32 */
    int64_t max_row_nnz_hint;
    /* Csr::srow_ (gkomi_csr_make_srow_i32) and its tile, or NULL / 0.  With it the
     * solver drivers run the nonzero-split kernel -- the one gkomi_csr_spmv_srow_f64_i32
     * runs -- also for the SpMV + dot-product launches of their fused iterations. */
    const int32_t* srow;
    int64_t srow_tile;
} gkomi_csr_ctx;
/* Csr<double, int64> (gkomi_csr_spmv_srow_f64_i64) */
typedef struct gkomi_csr64_ctx {
    int64_t nrows, ncols, nnz;
    const int64_t* row_ptrs;
    const int64_t* col_idxs;
    const double* vals;
    int64_t strategy;
    int64_t max_row_nnz_hint;
    const int64_t* srow;
    int64_t srow_tile;
} gkomi_csr64_ctx;
typedef struct gkomi_ell_ctx {
    int64_t nrows, ncols, num_stored_per_row, stride;
    const int32_t* col_idxs;
    const double* vals;
} gkomi_ell_ctx;
typedef struct gkomi_sellp_ctx {
    int64_t nrows, ncols, slice_size;
    const uint64_t* slice_sets;
    const uint64_t* slice_lengths;
    const int32_t* col_idxs;
    const double* vals;
} gkomi_sellp_ctx;
typedef struct gkomi_coo_ctx {
    int64_t nrows, ncols, nnz;
    const int32_t* row_idxs;
    const int32_t* col_idxs;
    const double* vals;
} gkomi_coo_ctx;
typedef struct gkomi_hybrid_ctx {
    int64_t nrows, ncols, ell_num_stored_per_row, ell_stride;
    const int32_t* ell_col_idxs;
    const double* ell_vals;
    int64_t coo_nnz;
    const int32_t* coo_row_idxs;
    const int32_t* coo_col_idxs;
    const double* coo_vals;
} gkomi_hybrid_ctx;
int gkomi_csr_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                              const double* alpha, const double* b,
                              int64_t b_stride, const double* beta, double* c,
                              int64_t c_stride);
int gkomi_csr64_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                                const double* alpha, const double* b,
                                int64_t b_stride, const double* beta, double* c,
                                int64_t c_stride);
int gkomi_ell_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                              const double* alpha, const double* b,
                              int64_t b_stride, const double* beta, double* c,
                              int64_t c_stride);
int gkomi_sellp_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                                const double* alpha, const double* b,
                                int64_t b_stride, const double* beta,
                                double* c, int64_t c_stride);
int gkomi_coo_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                              const double* alpha, const double* b,
                              int64_t b_stride, const double* beta, double* c,
                              int64_t c_stride);
int gkomi_hybrid_matrix_apply_cb(void* ctx, gkomi_stream_t s, int64_t nrhs,
                                 const double* alpha, const double* b,
                                 int64_t b_stride, const double* beta,
                                 double* c, int64_t c_stride);
/* The solver drivers with the system matrix behind a callback (config 4 of
 * BASELINE.json runs GMRES on ELL / SELL-P).  Same loops, same arguments as the
 * CSR entry points; gkomi_cg_solve_op_f64 runs the reference kernel sequence,
 * gkomi_cg_solve_fused_op_f64 (one right-hand side) the fused loop of
 * gkomi_cg_solve_f64_i32's mode 1.  Every *_fused_op_* driver recognises
 * gkomi_{csr,ell,sellp}_matrix_apply_cb + its record and runs that format's
 * SpMV with the dot-product epilogue (3 launches per CG iteration; iterates
 * bit-identical across the three formats); any other callback is followed by
 * a separate partials kernel. */
int gkomi_cg_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                          gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                          gkomi_apply_fn precond, void* precond_ctx,
                          const double* b, double* x, int64_t max_iters,
                          double reduction_factor, int baseline,
                          void* workspace, size_t workspace_bytes,
                          double* host_info);
int gkomi_cg_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x,
    int64_t max_iters, double reduction_factor, int baseline,
    int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
int gkomi_gmres_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                             gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                             gkomi_apply_fn precond, void* precond_ctx,
                             const double* b, double* x, int64_t krylov_dim,
                             int64_t max_iters, double reduction_factor,
                             int baseline, void* workspace,
                             size_t workspace_bytes, double* host_info);
int gkomi_bicgstab_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                                gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                                gkomi_apply_fn precond, void* precond_ctx,
                                const double* b, double* x, int64_t max_iters,
                                double reduction_factor, int baseline,
                                int64_t check_every, void* workspace,
                                size_t workspace_bytes, double* host_info);
/* Fused single right-hand-side BiCGSTAB: the recurrences and check points of
 * core/solver/bicgstab.cpp:107-234 in 6 launches per iteration instead of 25
 * (dot partials in the SpMV / step epilogues, criterion on the device).  Same
 * arguments and host_info as gkomi_bicgstab_solve_f64_i32 with nrhs = 1;
 * workspace gkomi_krylov_workspace_bytes(n, 1).  Iterates agree with the
 * reference sequence to rounding (the dots are summed in a different fixed
 * order); they do not depend on check_every. */
int gkomi_bicgstab_solve_fused_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy,
    int64_t max_row_nnz_hint, gkomi_apply_fn precond, void* precond_ctx,
    const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
int gkomi_bicgstab_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x,
    int64_t max_iters, double reduction_factor, int baseline,
    int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
/* Fused single right-hand-side CGS (core/solver/cgs.cpp:107-205 in 5 launches
 * per iteration; conventions of gkomi_bicgstab_solve_fused_f64_i32). */
int gkomi_cgs_solve_fused_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy,
    int64_t max_row_nnz_hint, gkomi_apply_fn precond, void* precond_ctx,
    const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
int gkomi_cgs_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x,
    int64_t max_iters, double reduction_factor, int baseline,
    int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
/* Fused single right-hand-side FCG (core/solver/fcg.cpp:104-196 in 3 launches
 * per iteration; see gkomi_bicgstab_solve_fused_f64_i32 for the conventions). */
int gkomi_fcg_solve_fused_f64_i32(
    gkomi_stream_t s, int64_t n, int64_t nnz, const int32_t* row_ptrs,
    const int32_t* col_idxs, const double* vals, int spmv_strategy,
    int64_t max_row_nnz_hint, gkomi_apply_fn precond, void* precond_ctx,
    const double* b, double* x, int64_t max_iters, double reduction_factor,
    int baseline, int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
int gkomi_fcg_solve_fused_op_f64(
    gkomi_stream_t s, int64_t n, gkomi_matrix_apply_fn matrix, void* matrix_ctx,
    gkomi_apply_fn precond, void* precond_ctx, const double* b, double* x,
    int64_t max_iters, double reduction_factor, int baseline,
    int64_t check_every, void* workspace, size_t workspace_bytes,
    double* host_info);
int gkomi_fcg_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                           gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                           gkomi_apply_fn precond, void* precond_ctx,
                           const double* b, double* x, int64_t max_iters,
                           double reduction_factor, int baseline,
                           int64_t check_every, void* workspace,
                           size_t workspace_bytes, double* host_info);
int gkomi_cgs_solve_op_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                           gkomi_matrix_apply_fn matrix, void* matrix_ctx,
                           gkomi_apply_fn precond, void* precond_ctx,
                           const double* b, double* x, int64_t max_iters,
                           double reduction_factor, int baseline,
                           int64_t check_every, void* workspace,
                           size_t workspace_bytes, double* host_info);
typedef struct gkomi_jacobi_ctx {
    int64_t n;               /* rows */
    int64_t nrhs;            /* columns of the vectors (stride == nrhs) */
    int64_t num_blocks;
    int32_t max_block_size;  /* 1: scalar Jacobi, blocks = inverted diagonal */
    int32_t pad_;
    const int32_t* block_ptrs;
    const double* blocks;
    const uint8_t* block_precisions; /* NULL: fp64 storage; else adaptive (see above) */
} gkomi_jacobi_ctx;
typedef struct gkomi_ilu_ctx {
    int64_t n;
    int64_t nrhs;
    const int32_t* l_row_ptrs;
    const int32_t* l_col_idxs;
    const double* l_vals;
    const int32_t* u_row_ptrs;
    const int32_t* u_col_idxs;
    const double* u_vals;
    double* intermediate;    /* n x nrhs scratch (Ilu's cached intermediate) */
    void* trs_workspace;
    size_t trs_workspace_bytes;
    int32_t l_unit_diag;     /* ParIlu stores the unit diagonal of L explicitly: 0 */
    int32_t pad_;            /* set to 0; gkomi_ilu_apply_cb keeps its chain state here (1: the
                              * intermediate vector is armed for the next lower brick solve) */
    /* analysed factors (gkomi_trs_analyse_*): when non-NULL the level-scheduled
     * solve replaces the analysis-free one for that factor */
    void* l_plan;
    int64_t l_nslices;
    int64_t l_entries;
    int64_t l_max_deps;
    void* u_plan;
    int64_t u_nslices;
    int64_t u_entries;
    int64_t u_max_deps;
    /* factors with a brick plan (gkomi_trs_bricks_*; handle + its device plan after the numeric
     * phase): when non-NULL the brick solve is used for that factor, before the two above */
    struct gkomi_trs_bricks* l_bricks;
    void* l_bricks_plan;
    struct gkomi_trs_bricks* u_bricks;
    void* u_bricks_plan;
} gkomi_ilu_ctx;
int gkomi_jacobi_apply_cb(void* ctx, gkomi_stream_t s, const double* in,
                          double* out);
int gkomi_ilu_apply_cb(void* ctx, gkomi_stream_t s, const double* in,
                       double* out);

/* ---- GMRES (core/solver/common_gmres_kernels.hpp, core/solver/gmres_kernels.hpp,
 *      driver core/solver/gmres.cpp:139-372) ------------------------------ */
/* givens_sin/cos: krylov_dim x nrhs, residual_norm_collection: (krylov_dim+1) x nrhs,
 * y: krylov_dim x nrhs (all row stride nrhs); hessenberg: (krylov_dim+1) x
 * (krylov_dim*nrhs), entry (i, j*nrhs + k); krylov_bases: ((krylov_dim+1)*n) x nrhs;
 * final_iter_nums: size_type (64-bit) per rhs.
 * reference/solver/common_gmres_kernels.cpp:140-217, gmres_kernels.cpp:55-100 */
int gkomi_gmres_initialize_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                               int64_t krylov_dim, const double* b,
                               int64_t b_stride, double* residual,
                               int64_t r_stride, double* givens_sin,
                               double* givens_cos, uint8_t* stop_status);
int gkomi_gmres_restart_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                            const double* residual, int64_t r_stride,
                            const double* residual_norm,
                            double* residual_norm_collection,
                            double* krylov_bases, int64_t kb_stride,
                            uint64_t* final_iter_nums);
/* hessenberg_iter = column block `iter` of the Hessenberg matrix (entry (row, rhs)
 * at row*h_stride + rhs), as Gmres passes its create_submatrix view */
int gkomi_gmres_hessenberg_qr_f64(gkomi_stream_t s, int64_t nrhs,
                                  double* givens_sin, double* givens_cos,
                                  double* residual_norm,
                                  double* residual_norm_collection,
                                  double* hessenberg_iter, int64_t h_stride,
                                  int64_t iter, uint64_t* final_iter_nums,
                                  const uint8_t* stop_status);
int gkomi_gmres_solve_krylov_f64(gkomi_stream_t s, int64_t nrhs,
                                 const double* residual_norm_collection,
                                 const double* hessenberg, int64_t h_stride,
                                 double* y, const uint64_t* final_iter_nums,
                                 const uint8_t* stop_status);
int gkomi_gmres_multi_axpy_f64(gkomi_stream_t s, int64_t n, int64_t nrhs,
                               const double* krylov_bases, int64_t kb_stride,
                               const double* y, double* before_preconditioner,
                               int64_t bp_stride,
                               const uint64_t* final_iter_nums,
                               uint8_t* stop_status);
/* Gmres::apply_dense_impl for a CSR matrix, optional (right) preconditioner and
 * Combined(Iteration(max_iters) [id 1], ResidualNorm(reduction, baseline) [id 2]);
 * the criterion is fed the implicit residual norm, checked on the host every
 * iteration.  host_info as for gkomi_cg_solve_f64_i32.  Blocks until done. */
size_t gkomi_gmres_workspace_bytes(int64_t n, int64_t nrhs, int64_t krylov_dim);
int gkomi_gmres_solve_f64_i32(gkomi_stream_t s, int64_t n, int64_t nrhs,
                              int64_t nnz, const int32_t* row_ptrs,
                              const int32_t* col_idxs, const double* vals,
                              int spmv_strategy, int64_t max_row_nnz_hint,
                              gkomi_apply_fn precond, void* precond_ctx,
                              const double* b, double* x, int64_t krylov_dim,
                              int64_t max_iters, double reduction_factor,
                              int baseline, void* workspace,
                              size_t workspace_bytes, double* host_info);

#ifdef __cplusplus
}
#endif

#endif /* GKOMI_H_ */
