// Sparse triangular solves of STENCIL-SHAPED factors: the brick plan, a second kind of
// LowerTrs / UpperTrs::generate analysis (hip/solver/common_trs_kernels.hip.hpp:61-253 runs
// hipsparseXcsrsv2_analysis there).  Numerical contract = reference/solver/lower_trs_kernels.cpp:90-120,
// upper_trs_kernels.cpp:90-123: per row the subtractions run in storage order, one division.
//
// Why a second plan: the level plan of trs_levels.hip pays one MEMORY hand-off per dependency
// level (~1.7 us: publish, fabric round trip, poll cadence, pass), and the factors of grid
// problems have hundreds to thousands of levels (7-point 108^3: 322, 5-point 1000^2: 1999).
// Here the rows are cut into BRICKS -- boxes of the grid the matrix was assembled on -- and a
// brick is solved by ONE workgroup with the brick's part of x in LDS:
//   * inside a brick a level costs an LDS round trip + the division (+ an s_barrier when the
//     workgroup has more than one wave): ~0.15-0.2 us, nothing on the fabric;
//   * a brick starts when the bricks it depends on have FINISHED (one flag per brick, release /
//     acquire at agent scope), so a memory hand-off is paid once per brick on the critical
//     path, not once per level: 19 instead of 322 on the 108^3 factor with 16^3 bricks;
//   * everything a level needs is in LDS by then: the right-hand side of the brick's rows and
//     the values of x the brick needs from other bricks are gathered when the brick starts
//     (all loads in flight together), the stored column indices are LDS indices, and the
//     factor's entries are prefetched one step ahead -- the loop has no memory wait and no
//     store (x leaves LDS in one sweep at the end).
// The grid is not given, it is recovered: the distinct |row - col| of the factor's dependencies
// must form a divisor chain 1 | s1 | s2 ... (lexicographic numbering of a box grid: 1, nx,
// nx ny); rows get mixed-radix coordinates from it.  That is a guess about the numbering, never
// trusted: the brick dependency graph is built from the actual entries and must be acyclic, every
// row must have at most 8 dependencies, and a brick with its inflow must fit LDS -- otherwise the
// analysis answers GKOMI_ENOTSUPPORTED and the caller keeps the level plan.  Any factor that
// passes is solved bit-identically to the reference, whatever its values.
//
// Deadlock freedom: bricks are handed out by an atomic ticket in topological order of the
// brick graph, so the bricks a workgroup waits for belong to workgroups that have started; waits
// are bounded and a brick that gives up poisons its rows (NaN), marks itself finished (no
// cascade of timeouts) and raises a STICKY flag.
#include "internal.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <new>
#include <system_error>
#include <memory>
#include <thread>
#include <vector>

struct gkomi_trs_bricks {
    int64_t n = 0;
    int lower = 1;
    int width = 0;    // dependency slots per row (ELL width of the plan)
    int threads = 64; // compute threads of a workgroup of the solve
    int mode = 2;     // 1: a brick starts when its predecessors have finished; 2: pipelined (inflow pump)
    int64_t nbricks = 0, nsteps = 0, coarse_levels = 0, critical_steps = 0, lds_bytes_max = 0;
    int64_t max_brick_steps = 0, nlevels_fine = 0;
    int64_t levels_estimate = 0;  // levels of the factor if the guessed box geometry holds: sum (extent - 1) + 1 (cost models only)
    int lexicographic = 1;        // 0: the grid coordinates were recovered from the dependency graph (any monotone numbering)
    int geometry_failed_width = 0;  // device analysis: the offsets are no divisor chain; widest dependency list (host recovery next)
    std::vector<int32_t> perm;             // plan position -> row
    std::vector<int32_t> inv_local;        // row -> LDS index inside its brick
    std::vector<int32_t> row_rank;         // row -> rank of its brick (topological order)
    std::vector<int32_t> ext_row_off;      // plan position -> index of its first inflow entry
    std::vector<int32_t> brick_row_begin;  // nbricks + 1
    std::vector<int32_t> brick_step_ptr;   // nbricks + 1
    std::vector<int32_t> step_begin;       // nsteps + 1; top bit: the step opens a level
    std::vector<int32_t> brick_ext_begin;  // nbricks + 1
    std::vector<int32_t> ext_col;          // row whose x an inflow entry needs
    std::vector<int32_t> pred_ptr, pred_idx;
    std::vector<int64_t> image_off;        // pipelined solve: byte offset of a brick's LDS image, nbricks + 1
    // sizes of the index arrays (what the plan layout needs), valid for both analyses
    int64_t n_step_begin = 0, n_ext_col = 0, n_pred_idx = 0, image_bytes = 0;
    // analysis on the device (gkomi_trs_bricks_create_i32): the index arrays stay in device memory owned by
    // the handle -- the numeric phase copies them into the plan device to device -- and come to the host
    // only when gkomi_trs_bricks_host_array asks for them
    char* dev_index = nullptr;
    size_t dev_perm = 0, dev_row_rank = 0, dev_inv_local = 0, dev_ext_row_off = 0, dev_ext_col = 0,
           dev_brick_row_begin = 0, dev_brick_step_ptr = 0, dev_step_begin = 0, dev_brick_ext_begin = 0, dev_bytes = 0;
    bool host_arrays_valid = true;
    ~gkomi_trs_bricks()
    {
        if (dev_index != nullptr) (void)hipFree(dev_index);
    }
    uint32_t epoch = 0;
    const void* uploaded_to = nullptr;
    uint64_t token = 0;  // written into the plan with the index arrays: a later numeric phase skips the upload only
                         // if the plan still carries it (the caller may have reallocated the same address)
};

namespace gkomi {
namespace {

constexpr int max_width = 8;
constexpr int max_offsets = 16;
constexpr int max_dims = 6;
constexpr int max_preds = 64;
constexpr int32_t pad_col = INT32_MIN;
constexpr int32_t level_bit = INT32_MIN;  // top bit of a step_begin entry
constexpr long long default_max_polls = 1ll << 22;
constexpr size_t max_lds_bytes = 144 * 1024;  // of 160 KiB per CU

struct brick_header {
    int64_t n;
    int64_t nbricks;
    unsigned int ticket;
    unsigned int finished;
    unsigned int overrun;  // sticky: zeroed by the numeric phase, never by a solve
    int32_t lower;
    uint64_t token;  // of the handle whose index arrays this plan holds
};
static_assert(sizeof(brick_header) <= 256, "the plan header has 256 bytes");

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct brick_layout {
    size_t perm, diag, rdiag, cols, vals, brick_row_begin, brick_step_ptr, step_begin, brick_ext_begin, ext_col,
        pred_ptr, pred_idx, done, row_rank, inv_local, ext_row_off, stamps, image_off, image, total;
};

brick_layout make_layout(const gkomi_trs_bricks& h)
{
    brick_layout l{};
    const size_t n = static_cast<size_t>(h.n > 0 ? h.n : 1);
    const size_t nb = static_cast<size_t>(h.nbricks);
    auto ints = [](size_t count) { return align_up(sizeof(int32_t) * (count > 0 ? count : 1), 256); };
    size_t off = 256;
    l.perm = off; off += ints(n);
    l.diag = off; off += align_up(sizeof(double) * n, 256);
    l.rdiag = off; off += align_up(sizeof(double) * n, 256);
    l.cols = off; off += ints(n * h.width);
    l.vals = off; off += align_up(sizeof(double) * n * h.width, 256);
    l.brick_row_begin = off; off += ints(nb + 1);
    l.brick_step_ptr = off; off += ints(nb + 1);
    l.step_begin = off; off += ints(static_cast<size_t>(h.n_step_begin));
    l.brick_ext_begin = off; off += ints(nb + 1);
    l.ext_col = off; off += ints(static_cast<size_t>(h.n_ext_col));
    l.pred_ptr = off; off += ints(nb + 1);
    l.pred_idx = off; off += ints(static_cast<size_t>(h.n_pred_idx));
    l.done = off; off += ints(nb);
    l.row_rank = off; off += ints(n);
    l.inv_local = off; off += ints(n);
    l.ext_row_off = off; off += ints(n);
    l.stamps = off; off += std::max<size_t>(8 * 1024, 64 * nb);  // tools only: shader-clock stamps
    l.image_off = off; off += align_up(sizeof(int64_t) * (nb + 1), 256);
    l.image = off; off += align_up(static_cast<size_t>(h.image_bytes), 256);
    l.total = off;
    return l;
}

// bytes of a row's record in the pipelined solve: diag, 1 / diag, K values, K + 3 ints, rounded up to
// 8 mod 16 -- 64 lanes read the same field of 64 consecutive records, and a stride of 4 m + 2 dwords
// spreads a 16-lane pass of 8-byte reads over all 32 LDS banks (a 64-byte record: 16-way conflicts)
__host__ __device__ constexpr int brick_record_bytes(int width)
{
    const int need = 28 + 12 * width;
    return need % 16 <= 8 ? need / 16 * 16 + 8 : need / 16 * 16 + 24;
}

// the part of a brick's LDS that does not depend on the right-hand side -- records (+ the spare one),
// {rows, inflow needed} per step (+ 4 beyond the end), the rows the inflow comes from -- is stored
// ready-made in the plan (its "image") and only copied when the brick starts
__host__ __device__ constexpr int64_t brick_image_bytes(int64_t rows, int64_t inflow, int64_t steps, int width)
{
    return ((rows + 1) * brick_record_bytes(width) + 8 * (steps + 4) + 4 * inflow + 15) / 16 * 16;
}

// LDS of a brick with R rows, E inflow values, S steps and K dependency slots per row:
//   x[R + E] | 0.0 | diag[R] | 1 / diag[R] | vals[K R] | cols[K R] | step bounds[S + 4] | rows[R] | inflow rows[E]
// | inflow needed by step[S + 4]  (the last five: int)
__host__ __device__ inline size_t brick_lds_bytes(int64_t rows, int64_t inflow, int64_t steps, int width)
{
    // + one spare entry per array: the row a lane without work points at (pipelined solve, which keeps
    // one record per row: diag, 1 / diag, values, LDS addresses of the dependencies and of the row's
    // own cell, row index -- (24 + 12 K) bytes rounded up to 8 -- and {rows, inflow needed} per step)
    const int64_t r1 = rows + 1;
    const size_t arrays = sizeof(double) * static_cast<size_t>(rows + inflow + 2 + 2 * r1 + width * r1) +
                          sizeof(int32_t) * static_cast<size_t>(width * r1 + 2 * (steps + 4) + r1 + inflow);
    const size_t records = (sizeof(double) * static_cast<size_t>(rows + inflow + 2) + 15) / 16 * 16 +
                           static_cast<size_t>(brick_image_bytes(rows, inflow, steps, width));
    return arrays > records ? arrays : records;
}

inline bool is_dep(bool lower, int64_t col, int64_t row) { return lower ? col < row : col > row; }

// ---------------------------------------------------------------- analysis (host) ----------

// f(part, lo, hi) on `parts` contiguous pieces of [0, n), piece boundaries at multiples of `grain`
template <typename F>
void in_parallel(int64_t n, int64_t grain, int parts, F f)
{
    grain = std::max<int64_t>(grain, 1);
    const int64_t units = ceildiv(n, grain);
    parts = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(parts, units)));
    std::vector<std::thread> pool;
    pool.reserve(static_cast<size_t>(parts));
    for (int p = 0; p < parts; ++p) {
        const int64_t lo = std::min(n, units * p / parts * grain), hi = std::min(n, units * (p + 1) / parts * grain);
        if (p + 1 == parts) {
            f(p, lo, hi);  // the caller's thread takes the last piece
        } else {
            try {
                pool.emplace_back(f, p, lo, hi);
            } catch (const std::system_error&) {
                f(p, lo, hi);  // no more threads to be had: this piece on the caller's thread, too
            }
        }
    }
    for (auto& t : pool) t.join();
}

inline int analysis_threads()
{
    const char* v = getenv("GKOMI_ANALYSIS_THREADS");
    if (v != nullptr && v[0] != 0) return std::max(1, atoi(v));
    const unsigned hw = std::thread::hardware_concurrency();
    return static_cast<int>(std::min(8u, std::max(1u, hw)));
}

// ---- grid coordinates from the dependency graph --------------------------------------------------------------
// The divisor-chain test below recognises a box grid only when it is numbered lexicographically.  The factor of a grid
// problem numbered any other way that keeps "smaller index = earlier" along every axis -- patches / tiles (the locality a
// FEM or a cache-blocked ordering has), space-filling curves -- has the same dependency graph: every row depends on at most
// one neighbour per dimension, one step back.  Here the coordinates are read off that graph: walked in dependency order, a
// row sits at the componentwise maximum of its neighbours' coordinates (interior rows, rows on coordinate planes), or one
// step further along its single neighbour's axis (rows on the coordinate axes; the children of the origin take the axes
// in the order they come).  Like the divisor chain this is a guess that is never trusted: every dependency must be exactly
// one step back along one axis, no two rows may share a cell, the cells must nearly fill their bounding box -- and the brick
// graph is still built from the actual entries afterwards.  Any failure = this factor is not for the brick plan.
bool recover_grid_coordinates(int64_t n, bool lower, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci, int width,
                              std::vector<int32_t>* coord, int64_t* extent, int* dims_out)
{
    const int dims = width;  // one dependency per dimension at most
    if (dims < 1 || dims > 3 || n < 2) return false;
    for (int k = 0; k < dims; ++k) coord[k].assign(static_cast<size_t>(n), 0);
    std::vector<int32_t> level(static_cast<size_t>(n), 0);
    int origin_children = 0;
    int64_t origins = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t row = lower ? i : n - 1 - i;  // dependency order
        int32_t parents[3];
        int np = 0;
        for (int32_t k = rp[row]; k < rp[row + 1]; ++k) {
            const int64_t col = ci[k];
            if (!is_dep(lower, col, row) || col < 0 || col >= n) continue;
            if (np == dims) return false;
            parents[np++] = static_cast<int32_t>(col);
        }
        if (np == 0) {
            if (++origins > 1) return false;  // one corner only
            continue;
        }
        int32_t m[3] = {0, 0, 0};
        int32_t lvl = 0;
        for (int j = 0; j < np; ++j) {
            lvl = std::max(lvl, level[parents[j]] + 1);
            for (int k = 0; k < dims; ++k) m[k] = std::max(m[k], coord[k][parents[j]]);
        }
        const int64_t sum = static_cast<int64_t>(m[0]) + m[1] + m[2];
        if (sum == lvl - 1 && np == 1) {  // on a coordinate axis: one step further along it
            int nonzero = 0, axis = -1;
            for (int k = 0; k < dims; ++k) {
                if (coord[k][parents[0]] != 0) {
                    ++nonzero;
                    axis = k;
                }
            }
            if (nonzero == 0) {
                if (origin_children >= dims) return false;
                axis = origin_children++;
            } else if (nonzero != 1) {
                return false;
            }
            ++m[axis];
        } else if (sum != lvl) {
            return false;
        }
        for (int j = 0; j < np; ++j) {  // every neighbour exactly one step back along one axis
            int steps = 0;
            for (int k = 0; k < dims; ++k) {
                const int32_t d = m[k] - coord[k][parents[j]];
                if (d < 0 || d > 1) return false;
                steps += d;
            }
            if (steps != 1) return false;
        }
        level[row] = lvl;
        for (int k = 0; k < dims; ++k) coord[k][row] = m[k];
    }
    if (origins != 1) return false;
    int64_t cells = 1;
    for (int k = 0; k < dims; ++k) {
        int32_t mx = 0;
        for (int64_t row = 0; row < n; ++row) mx = std::max(mx, coord[k][row]);
        extent[k] = static_cast<int64_t>(mx) + 1;
        cells *= extent[k];
        if (cells > 2 * n + 1024) return false;  // the rows must nearly fill their box
    }
    std::vector<char> taken(static_cast<size_t>(cells), 0);
    for (int64_t row = 0; row < n; ++row) {
        int64_t cell = 0, mul = 1;
        for (int k = 0; k < dims; ++k) {
            cell += coord[k][row] * mul;
            mul *= extent[k];
        }
        if (taken[cell]) return false;
        taken[cell] = 1;
    }
    *dims_out = dims;
    return true;
}

// ---- bricks of a THIN factor: consecutive pieces of the level order ------------------------------------------------
// A factor without a grid in it can still have many more levels than a level holds rows: chains, narrow bands, the factors
// of small unstructured meshes (the reference's ani4: 183 levels of 17 rows).  The level plan pays a memory hand-off per
// level there, and the bricks' LDS cadence (0.165 us per level) needs only that consecutive levels live in ONE workgroup:
// rows sorted by (level, row) and cut every brick_rows rows give bricks whose graph is acyclic by construction (a
// dependency runs from a lower level to a strictly higher one, so never from a later piece to an earlier one).  A piece
// holds brick_rows / (rows per level) levels: worth it when that is many -- `thin` = at most 64 rows per level on average.
// level[row] (longest path), order[i] = rows by (level, row); returns the number of levels
int64_t level_order(int64_t n, bool lower, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci,
                    std::vector<int32_t>& order)
{
    std::vector<int32_t> level(static_cast<size_t>(n), 0);
    int32_t nlevels = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t row = lower ? i : n - 1 - i;
        int32_t lvl = 0;
        for (int32_t k = rp[row]; k < rp[row + 1]; ++k) {
            const int64_t col = ci[k];
            if (is_dep(lower, col, row) && col >= 0 && col < n) lvl = std::max(lvl, level[col] + 1);
        }
        level[row] = lvl;
        nlevels = std::max(nlevels, lvl + 1);
    }
    std::vector<int32_t> start(static_cast<size_t>(nlevels) + 1, 0);
    for (int64_t row = 0; row < n; ++row) ++start[level[row] + 1];
    for (int32_t l = 0; l < nlevels; ++l) start[l + 1] += start[l];
    order.assign(static_cast<size_t>(n), 0);
    for (int64_t row = 0; row < n; ++row) order[start[level[row]]++] = static_cast<int32_t>(row);
    return nlevels;
}

// the whole symbolic analysis; GKOMI_ENOTSUPPORTED = this factor is not for the brick plan
int analyse(gkomi_trs_bricks& h, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci,
            int64_t brick_rows, int threads, int mode)
{
    h.mode = mode == 1 ? 1 : 2;
    if (h.mode == 2) threads = 64;  // one compute wave (+ the pump)
    const bool trace = getenv("GKOMI_ANALYSIS_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  analysis: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int64_t n = h.n;
    const bool lower = h.lower != 0;
    // 1. the distinct dependency offsets and the longest dependency list
    const int nthreads = analysis_threads();
    int64_t offs[max_offsets];
    int noffs = 0;
    int width = 0;
    bool too_many = false;  // more distinct offsets than a lexicographic box numbering has (a space-filling curve has many)
    {
        struct partial {
            int64_t offs[max_offsets];
            int noffs = 0, width = 0;
            bool too_many = false;
        };
        std::vector<partial> part(static_cast<size_t>(nthreads));
        in_parallel(n, 4096, nthreads, [&](int t, int64_t lo, int64_t hi) {
            partial& me = part[t];
            for (int64_t row = lo; row < hi; ++row) {  // (the widest row counts also when the offsets are too many)
                int deps = 0;
                for (int32_t k = rp[row]; k < rp[row + 1]; ++k) {
                    const int64_t col = ci[k];
                    if (!is_dep(lower, col, row) || col < 0 || col >= n) continue;
                    ++deps;
                    if (me.too_many) continue;
                    const int64_t d = lower ? row - col : col - row;
                    int j = 0;
                    while (j < me.noffs && me.offs[j] != d) ++j;
                    if (j == me.noffs) {
                        if (me.noffs == max_offsets) {
                            me.too_many = true;
                        } else {
                            me.offs[me.noffs++] = d;
                        }
                    }
                }
                me.width = std::max(me.width, deps);
            }
        });
        for (const partial& me : part) {
            too_many = too_many || me.too_many;
            width = std::max(width, me.width);
            for (int q = 0; q < me.noffs && !too_many; ++q) {
                int j = 0;
                while (j < noffs && offs[j] != me.offs[q]) ++j;
                if (j == noffs) {
                    if (noffs == max_offsets) {
                        too_many = true;
                        break;
                    }
                    offs[noffs++] = me.offs[q];
                }
            }
        }
    }
    mark("offsets");
    if ((noffs == 0 && !too_many) || width > max_width) return GKOMI_ENOTSUPPORTED;
    std::sort(offs, offs + noffs);
    // 2. strides of a lexicographic box numbering: a divisor chain ...
    int64_t stride[max_dims];
    int64_t extent[max_dims];
    int dims = 0;
    bool lexicographic = !too_many;
    if (lexicographic) {
        stride[dims++] = 1;
        for (int j = 0; j < noffs && lexicographic; ++j) {
            if (offs[j] == stride[dims - 1]) continue;
            if (offs[j] % stride[dims - 1] != 0 || dims == max_dims) {
                lexicographic = false;
            } else {
                stride[dims++] = offs[j];
            }
        }
    }
    // ... or any other numbering of a box grid whose graph gives the coordinates away (recover_grid_coordinates)
    std::vector<int32_t> coord[3];
    std::vector<int32_t> band_order;  // thin factors: rows by (level, row)
    bool banded = false;
    if (lexicographic) {
        for (int k = 0; k + 1 < dims; ++k) extent[k] = stride[k + 1] / stride[k];
        extent[dims - 1] = ceildiv(n, stride[dims - 1]);
    } else {
        static const bool recover = [] {
            const char* e = getenv("GKOMI_TRS_RECOVER_GRID");  // =0: lexicographic numberings only (rounds 2-3)
            return e == nullptr || e[0] != '0';
        }();
        if (!recover) return GKOMI_ENOTSUPPORTED;
        if (width <= 3 && recover_grid_coordinates(n, lower, rp, ci, width, coord, extent, &dims)) {
            mark("grid coordinates from the graph");
        } else {
            // ... or no grid at all, but a thin factor: pieces of the level order (level_order above)
            const int64_t nlevels = level_order(n, lower, rp, ci, band_order);
            mark("level order");
            if (nlevels < 2 || n > 64 * nlevels) return GKOMI_ENOTSUPPORTED;
            banded = true;
            dims = 1;
            extent[0] = nlevels;
        }
    }
    h.lexicographic = lexicographic ? 1 : (banded ? 2 : 0);
    int active = 0;  // dimensions that are more than one point wide
    for (int k = 0; k < dims; ++k) active += extent[k] > 1;
    h.levels_estimate = 1;
    for (int k = 0; k < dims; ++k) h.levels_estimate += extent[k] - 1;
    // pipelined: the levels of a brick should fit the one compute wave, and a brick about fill a CU's LDS
    // (8 x 8 x 27, 45^2: fewer bricks = fewer hand-offs); else large bricks
    if (brick_rows <= 0) brick_rows = h.mode == 2 ? (active >= 3 ? 1728 : 2025) : 4096;
    if (banded) {
        // pieces of the level order: every piece is a hand-off on the only path there is, so as few as LDS allows -- start
        // from what the records of a piece take (no inflow counted yet) and come down in steps of a fifth, not by halves
        const int k = width <= 2 ? 2 : width <= 3 ? 3 : width <= 4 ? 4 : 8;
        const int64_t fit = static_cast<int64_t>(max_lds_bytes - 4096) / (8 * (3 + k) + 4 * (k + 1) + 8);
        brick_rows = std::min(brick_rows, std::max<int64_t>(fit, 8));
    }
    // 3. brick edges: about brick_rows rows per brick, near-cubic, an even split of every extent;
    //    shrunk until a brick with its inflow fits LDS
    for (int attempt = 0; attempt < (banded ? 16 : 8);
         ++attempt, brick_rows = std::max<int64_t>(banded ? brick_rows * 4 / 5 : brick_rows / 2, 8)) {
        int64_t edge[max_dims], nbk[max_dims];
        if (banded) {
            edge[0] = 1;  // (unused: the brick map below is a cut of band_order)
            nbk[0] = ceildiv(n, brick_rows);
        } else if (h.mode == 2 && active >= 3) {
            // pipelined, three or more dimensions: a level of a box is at most the product of all its edges but
            // the longest -- keep that within the 64 lanes of the compute wave (8 x 8, 4 x 4 x 4) so that every
            // level is ONE step, and spend the rows on the last dimension (measured on the 108^3 factor:
            // 8 x 8 x 27 bricks 167 us, 12^3 181, 10^3 186)
            const int64_t cross = active == 3 ? 8 : 4;
            int last = dims - 1;
            while (extent[last] <= 1) --last;
            int64_t product = 1;
            for (int k = 0; k < dims; ++k) {
                if (k == last) continue;
                edge[k] = std::min<int64_t>(extent[k], extent[k] > 1 ? cross : 1);
                product *= edge[k];
            }
            edge[last] = std::max<int64_t>(1, std::min<int64_t>(extent[last], brick_rows / product));
            for (int k = 0; k < dims; ++k) {
                nbk[k] = ceildiv(extent[k], edge[k]);
                if (k == last) edge[k] = ceildiv(extent[k], nbk[k]);  // an even split of the long edge
                nbk[k] = ceildiv(extent[k], edge[k]);
            }
        } else {
            int order[max_dims];
            for (int k = 0; k < dims; ++k) order[k] = k;
            std::sort(order, order + dims, [&](int a, int b) { return extent[a] < extent[b]; });
            double remaining = static_cast<double>(std::max<int64_t>(brick_rows, 1));
            for (int q = 0; q < dims; ++q) {
                const int k = order[q];
                const double want = std::pow(remaining, 1.0 / (dims - q));
                int64_t e = std::max<int64_t>(1, static_cast<int64_t>(std::llround(want)));
                e = std::min(e, extent[k]);
                nbk[k] = ceildiv(extent[k], e);
                edge[k] = ceildiv(extent[k], nbk[k]);
                nbk[k] = ceildiv(extent[k], edge[k]);
                remaining = std::max(1.0, remaining / static_cast<double>(edge[k]));
            }
        }
        if (const char* forced = banded ? nullptr : getenv("GKOMI_TRS_BRICK_EDGES")) {  // tuning: "e0,e1,e2" (tools/trs_bricks_probe.py edges)
            int k = 0;
            for (const char* q = forced; *q != 0 && k < dims; ++k) {
                edge[k] = std::max<int64_t>(1, std::min<int64_t>(extent[k], atoll(q)));
                nbk[k] = ceildiv(extent[k], edge[k]);
                while (*q != 0 && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        int64_t nbricks = 1;
        for (int k = 0; k < dims; ++k) {
            nbricks *= nbk[k];
            if (nbricks > (1 << 24)) return GKOMI_ENOTSUPPORTED;
        }
        // a LAYER = the bricks with one coordinate along the last dimension = a contiguous range of
        // rows: bricks never straddle layers, so layers are analysed side by side.  Recovered coordinates: any
        // row can be anywhere -- one layer, the passes below that walk rows in dependency order run on one thread.
        const int64_t layer_rows = lexicographic ? edge[dims - 1] * stride[dims - 1] : n;
        std::vector<int32_t> brick(static_cast<size_t>(n));
        if (banded) {
            in_parallel(n, 4096, nthreads, [&](int, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) brick[band_order[i]] = static_cast<int32_t>(i / brick_rows);
            });
        } else if (!lexicographic) {
            int64_t bmul[max_dims];
            for (int k = 0, m = 1; k < dims; ++k) {
                bmul[k] = m;
                m *= static_cast<int>(nbk[k]);
            }
            in_parallel(n, 4096, nthreads, [&](int, int64_t lo, int64_t hi) {
                for (int64_t row = lo; row < hi; ++row) {
                    int64_t id = 0;
                    for (int k = 0; k < dims; ++k) id += (coord[k][row] / edge[k]) * bmul[k];
                    brick[row] = static_cast<int32_t>(id);
                }
            });
        } else {
        in_parallel(n, layer_rows, nthreads, [&](int, int64_t lo, int64_t hi) {
            // coordinates of `lo` by division, then counted up row by row (dimension 0 has stride 1)
            int64_t c[max_dims], within[max_dims], mul[max_dims], id = 0;
            for (int k = 0, m = 1; k < dims; ++k) {
                c[k] = k + 1 < dims ? (lo / stride[k]) % extent[k] : lo / stride[k];
                within[k] = c[k] % edge[k];
                mul[k] = m;
                id += (c[k] / edge[k]) * m;
                m *= static_cast<int>(nbk[k]);
            }
            for (int64_t row = lo; row < hi; ++row) {
                brick[row] = static_cast<int32_t>(id);
                for (int k = 0; k < dims; ++k) {  // row + 1
                    if (++c[k] < extent[k] || k + 1 == dims) {
                        if (++within[k] == edge[k]) {
                            within[k] = 0;
                            id += mul[k];
                        }
                        break;
                    }
                    id -= ((c[k] - 1) / edge[k]) * mul[k];  // back to the first brick along k, carry on
                    c[k] = 0;
                    within[k] = 0;
                }
            }
        });
        }
        mark("brick of every row");
        // 4. the brick graph from the actual entries (never from the guessed geometry), and
        // 6. the level of a row inside its brick (dependencies on other bricks do not count: their
        //    values are inflow), rows and inflow entries per brick
        std::vector<int32_t> npred(static_cast<size_t>(nbricks), 0);
        std::vector<int32_t> preds(static_cast<size_t>(nbricks) * max_preds);
        std::vector<int32_t> fine(static_cast<size_t>(n), 0);
        std::vector<int32_t> nfine(static_cast<size_t>(nbricks), 0), brick_rows_count(static_cast<size_t>(nbricks), 0),
            brick_ext(static_cast<size_t>(nbricks), 0);
        std::vector<char> failed(static_cast<size_t>(nthreads), 0);
        std::vector<uint8_t> row_ext(static_cast<size_t>(n), 0);  // inflow entries of a row (<= max_width)
        in_parallel(n, layer_rows, nthreads, [&](int t, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t row = lower ? i : hi - 1 - (i - lo);  // in dependency order
                const int32_t mine = brick[row];
                int32_t lvl = 0;
                for (int32_t k = rp[row]; k < rp[row + 1]; ++k) {
                    const int64_t col = ci[k];
                    if (!is_dep(lower, col, row) || col < 0 || col >= n) continue;
                    const int32_t other = brick[col];
                    if (other == mine) {
                        lvl = std::max(lvl, fine[col] + 1);
                        continue;
                    }
                    ++brick_ext[mine];
                    ++row_ext[row];
                    int32_t* list = preds.data() + static_cast<size_t>(mine) * max_preds;
                    int j = 0;
                    while (j < npred[mine] && list[j] != other) ++j;
                    if (j == npred[mine]) {
                        if (npred[mine] == max_preds) {
                            failed[t] = 1;
                            return;
                        }
                        list[npred[mine]++] = other;
                    }
                }
                fine[row] = lvl;
                nfine[mine] = std::max(nfine[mine], lvl + 1);
                ++brick_rows_count[mine];
            }
        });
        for (char f : failed) {
            if (f) return GKOMI_ENOTSUPPORTED;
        }
        mark("brick graph + levels in bricks");
        // coarse levels by longest path (Kahn); a cycle = the guessed geometry is wrong
        std::vector<int32_t> succ_ptr(static_cast<size_t>(nbricks) + 1, 0);
        for (int64_t b = 0; b < nbricks; ++b) {
            for (int j = 0; j < npred[b]; ++j) ++succ_ptr[preds[b * max_preds + j] + 1];
        }
        for (int64_t b = 0; b < nbricks; ++b) succ_ptr[b + 1] += succ_ptr[b];
        std::vector<int32_t> succ(static_cast<size_t>(succ_ptr[nbricks]));
        {
            std::vector<int32_t> cursor(succ_ptr.begin(), succ_ptr.end() - 1);
            for (int64_t b = 0; b < nbricks; ++b) {
                for (int j = 0; j < npred[b]; ++j) succ[cursor[preds[b * max_preds + j]]++] = static_cast<int32_t>(b);
            }
        }
        std::vector<int32_t> coarse(static_cast<size_t>(nbricks), 0), waiting(npred);
        std::vector<int32_t> topo;
        topo.reserve(static_cast<size_t>(nbricks));
        for (int64_t b = 0; b < nbricks; ++b) {
            if (waiting[b] == 0) topo.push_back(static_cast<int32_t>(b));
        }
        for (size_t q = 0; q < topo.size(); ++q) {
            const int32_t b = topo[q];
            for (int32_t j = succ_ptr[b]; j < succ_ptr[b + 1]; ++j) {
                const int32_t t = succ[j];
                coarse[t] = std::max(coarse[t], coarse[b] + 1);
                if (--waiting[t] == 0) topo.push_back(t);
            }
        }
        if (static_cast<int64_t>(topo.size()) != nbricks) return GKOMI_ENOTSUPPORTED;
        // 5. rank = position of a brick in (coarse level, id) order; empty bricks stay (zero rows)
        int32_t ncoarse = 0;
        for (int64_t b = 0; b < nbricks; ++b) ncoarse = std::max(ncoarse, coarse[b] + 1);
        std::vector<int32_t> rank(static_cast<size_t>(nbricks));
        {
            std::vector<int32_t> count(static_cast<size_t>(ncoarse) + 1, 0);
            for (int64_t b = 0; b < nbricks; ++b) ++count[coarse[b] + 1];
            for (int32_t l = 0; l < ncoarse; ++l) count[l + 1] += count[l];
            for (int64_t b = 0; b < nbricks; ++b) rank[b] = count[coarse[b]]++;
        }
        // LDS of the largest brick: x + inflow, and the brick's part of the factor (see the solve)
        const int slots = width <= 2 ? 2 : width <= 3 ? 3 : width <= 4 ? 4 : 8;  // the widths the solve is built for
        int64_t max_lds = 0;
        for (int64_t b = 0; b < nbricks; ++b) {
            const int64_t r = brick_rows_count[b];
            // a level of w rows is ceil(w / threads) steps: steps <= levels + rows / threads (64: the fewest threads)
            max_lds = std::max<int64_t>(max_lds, static_cast<int64_t>(brick_lds_bytes(r, brick_ext[b], nfine[b] + r / 64 + 1, slots)));
        }
        int64_t max_rows = 0;
        for (int64_t b = 0; b < nbricks; ++b) max_rows = std::max<int64_t>(max_rows, brick_rows_count[b]);
        // (the pipelined solve gathers a brick's right-hand side 16 rows per lane of its 128)
        if (static_cast<size_t>(max_lds) + 256 > max_lds_bytes || (h.mode == 2 && max_rows > 2048)) {
            if (brick_rows <= 8) return GKOMI_ENOTSUPPORTED;
            continue;  // smaller bricks
        }
        mark("brick levels, sizes");
        // 7. plan order: bricks by rank, rows of a brick by level, rows of a level by row index
        h.nbricks = nbricks;
        h.coarse_levels = ncoarse;
        h.width = slots;
        h.lds_bytes_max = max_lds;
        h.brick_row_begin.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.brick_ext_begin.assign(static_cast<size_t>(nbricks) + 1, 0);
        std::vector<int32_t> level_off(static_cast<size_t>(nbricks) + 1, 0);  // by rank
        std::vector<int32_t> by_rank(static_cast<size_t>(nbricks));
        for (int64_t b = 0; b < nbricks; ++b) by_rank[rank[b]] = static_cast<int32_t>(b);
        for (int64_t r = 0; r < nbricks; ++r) {
            const int32_t b = by_rank[r];
            h.brick_row_begin[r + 1] = h.brick_row_begin[r] + brick_rows_count[b];
            h.brick_ext_begin[r + 1] = h.brick_ext_begin[r] + brick_ext[b];
            level_off[r + 1] = level_off[r] + nfine[b];
        }
        // (a layer's rows only touch the counters of that layer's bricks: layers side by side, no atomics)
        std::vector<int32_t> level_pos(static_cast<size_t>(level_off[nbricks]) + 1, 0);
        in_parallel(n, layer_rows, nthreads, [&](int, int64_t lo, int64_t hi) {
            for (int64_t row = lo; row < hi; ++row) ++level_pos[level_off[rank[brick[row]]] + fine[row] + 1];
        });
        for (size_t j = 0; j + 1 < level_pos.size(); ++j) level_pos[j + 1] += level_pos[j];
        // level_pos[level_off[r] + l] = first plan position of level l of the brick with rank r
        int max_level_rows = 0;
        for (size_t j = 0; j + 1 < level_pos.size(); ++j) {
            max_level_rows = std::max(max_level_rows, level_pos[j + 1] - level_pos[j]);
        }
        if (threads <= 0) threads = max_level_rows <= 64 ? 64 : max_level_rows <= 160 ? 128 : 256;
        h.threads = threads;
        h.nlevels_fine = level_off[nbricks];
        // steps: a level in chunks of `threads` rows
        h.brick_step_ptr.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.step_begin.clear();
        for (int64_t r = 0; r < nbricks; ++r) {
            for (int32_t j = level_off[r]; j < level_off[r + 1]; ++j) {
                for (int32_t p = level_pos[j]; p < level_pos[j + 1]; p += threads) {
                    h.step_begin.push_back(p == level_pos[j] ? (p | level_bit) : p);
                }
            }
            h.brick_step_ptr[r + 1] = static_cast<int32_t>(h.step_begin.size());
        }
        h.nsteps = static_cast<int64_t>(h.step_begin.size());
        h.step_begin.push_back(static_cast<int32_t>(n) | level_bit);
        h.perm.assign(static_cast<size_t>(n), 0);
        h.inv_local.assign(static_cast<size_t>(n), 0);
        h.row_rank.assign(static_cast<size_t>(n), 0);
        {
            // rows of a level in row order: every layer walks its own rows in order and owns its bricks' cursors
            std::vector<int32_t> cursor(level_pos.begin(), level_pos.end() - 1);
            in_parallel(n, layer_rows, nthreads, [&](int, int64_t lo, int64_t hi) {
                for (int64_t row = lo; row < hi; ++row) {
                    const int32_t r = rank[brick[row]];
                    const int32_t p = cursor[level_off[r] + fine[row]]++;
                    h.perm[p] = static_cast<int32_t>(row);
                    h.inv_local[row] = p - h.brick_row_begin[r];
                    h.row_rank[row] = r;
                }
            });
        }
        mark("plan order, steps");
        // inflow lists in plan order, entries of a row in storage order
        h.ext_row_off.assign(static_cast<size_t>(n), 0);
        {
            // exclusive scan of the rows' inflow counts in plan order: per-piece sums, then offsets
            const int pieces = nthreads;
            std::vector<int64_t> piece_sum(static_cast<size_t>(pieces) + 1, 0);
            std::vector<int64_t> piece_lo(static_cast<size_t>(pieces) + 1, n);
            in_parallel(n, 4096, pieces, [&](int t, int64_t lo, int64_t hi) {
                int64_t sum = 0;
                for (int64_t p = lo; p < hi; ++p) sum += row_ext[h.perm[p]];
                piece_sum[t + 1] = sum;
                piece_lo[t] = lo;
            });
            for (int t = 0; t < pieces; ++t) piece_sum[t + 1] += piece_sum[t];
            if (piece_sum[pieces] > INT32_MAX) return GKOMI_ENOTSUPPORTED;
            in_parallel(n, 4096, pieces, [&](int t, int64_t lo, int64_t hi) {
                int32_t running = static_cast<int32_t>(piece_sum[t]);
                for (int64_t p = lo; p < hi; ++p) {
                    h.ext_row_off[p] = running;
                    running += row_ext[h.perm[p]];
                }
            });
            h.ext_col.assign(static_cast<size_t>(piece_sum[pieces]), 0);
        }
        in_parallel(n, 4096, nthreads, [&](int, int64_t lo, int64_t hi) {
            for (int64_t p = lo; p < hi; ++p) {
                const int64_t row = h.perm[p];
                if (row_ext[row] == 0) continue;
                int32_t at = h.ext_row_off[p];
                for (int32_t k = rp[row]; k < rp[row + 1]; ++k) {
                    const int64_t col = ci[k];
                    if (!is_dep(lower, col, row) || col < 0 || col >= n) continue;
                    if (brick[col] != brick[row]) h.ext_col[at++] = static_cast<int32_t>(col);
                }
            }
        });
        mark("inflow lists");
        // the bricks a brick waits for, as ranks; critical path in steps
        h.pred_ptr.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.pred_idx.clear();
        std::vector<int64_t> path(static_cast<size_t>(nbricks), 0);
        h.critical_steps = 0;
        h.max_brick_steps = 0;
        for (int64_t r = 0; r < nbricks; ++r) {
            const int32_t b = by_rank[r];
            int64_t before = 0;
            for (int j = 0; j < npred[b]; ++j) {
                const int32_t pr = rank[preds[static_cast<size_t>(b) * max_preds + j]];
                h.pred_idx.push_back(pr);
                before = std::max(before, path[pr]);  // pr < r: already final
            }
            h.pred_ptr[r + 1] = static_cast<int32_t>(h.pred_idx.size());
            const int64_t steps = h.brick_step_ptr[r + 1] - h.brick_step_ptr[r];
            path[r] = before + steps;
            h.critical_steps = std::max(h.critical_steps, path[r]);
            h.max_brick_steps = std::max(h.max_brick_steps, steps);
        }
        mark("predecessor lists");
        h.image_off.assign(h.mode == 2 ? static_cast<size_t>(nbricks) + 1 : 0, 0);
        for (int64_t r = 0; r < nbricks && h.mode == 2; ++r) {
            h.image_off[r + 1] = h.image_off[r] + brick_image_bytes(h.brick_row_begin[r + 1] - h.brick_row_begin[r],
                                                                    h.brick_ext_begin[r + 1] - h.brick_ext_begin[r],
                                                                    h.brick_step_ptr[r + 1] - h.brick_step_ptr[r], slots);
        }
        h.n_step_begin = static_cast<int64_t>(h.step_begin.size());
        h.n_ext_col = static_cast<int64_t>(h.ext_col.size());
        h.n_pred_idx = static_cast<int64_t>(h.pred_idx.size());
        h.image_bytes = h.image_off.empty() ? 0 : h.image_off.back();
        return GKOMI_SUCCESS;
    }
    return GKOMI_ENOTSUPPORTED;
}

// ---------------------------------------------------------------- analysis (device) --------
// The same symbolic analysis as `analyse` above, arrays in HBM from start to end (round 2's ran on host
// threads after copying the pattern out: 17 ms per 108^3 factor + the copies both ways).  Integer work on
// the pattern, one workgroup per brick:
//   A  ana_offsets_kernel   the distinct dependency offsets |row - col| (<= 16) and the longest dependency
//                           list; the host turns them into strides / extents / brick edges (a handful of
//                           numbers) -- the one decision that needs no data
//   D  ana_brick_kernel     a workgroup enumerates ITS brick's rows from the brick's coordinates, classifies
//                           every dependency (same brick: LDS index; other brick: inflow + predecessor list),
//                           relaxes the levels inside the brick in LDS, and leaves per brick: rows, inflow
//                           entries, levels, steps, predecessors; per row: level and inflow count
//   --  host: longest-path levels of the brick graph (Kahn, a few hundred nodes), ranks, prefix sums of the
//       per-brick sizes, the LDS check
//   E  ana_place_kernel     a workgroup places its brick's rows in plan order -- by level, rows of a level
//                           in row order (a stable counting pass in LDS) -- and writes perm, inv_local,
//                           row_rank, the step table and the rows' inflow counts in plan order
//   F  prefix sum of the inflow counts (gkomi_prefix_sum_i32), G  ana_inflow_kernel fills the inflow lists
// Every array is identical to the host analysis' (tests/test_trs_bricks_gpu.py compares them all).
struct ana_geometry {
    int dims;
    int n;
    int lower;
    int stride[max_dims], extent[max_dims], edge[max_dims], nbk[max_dims], mul[max_dims];
};

__device__ __forceinline__ bool ana_is_dep(const ana_geometry& g, int col, int row)
{
    return (g.lower ? col < row : col > row) && col >= 0 && col < g.n;
}

// brick id of a row and, on request, its index among the cells of that brick (dimension 0 fastest)
__device__ __forceinline__ int ana_brick_of(const ana_geometry& g, int row, int* cell)
{
    int id = 0, local = 0, radix = 1;
#pragma unroll
    for (int k = 0; k < max_dims; ++k) {
        if (k >= g.dims) break;
        const int c = k + 1 < g.dims ? (row / g.stride[k]) % g.extent[k] : row / g.stride[k];
        const int bc = c / g.edge[k];
        id += bc * g.mul[k];
        const int lo = bc * g.edge[k];
        const int len = min(g.edge[k], g.extent[k] - lo);
        local += (c - lo) * radix;
        radix *= len;
    }
    if (cell != nullptr) *cell = local;
    return id;
}

constexpr unsigned long long ana_empty = ~0ull;

__global__ __launch_bounds__(256) void ana_offsets_kernel(int n, int lower, const int32_t* __restrict__ rp,
                                                          const int32_t* __restrict__ ci, unsigned long long* table,
                                                          int* width, int* too_many)
{
    // the workgroup collects its offsets in LDS (a handful of distinct values: after the first few rows every
    // lookup is a hit in the lane's own snapshot) and merges them into the global table once, at its end --
    // half a million lanes compare-and-swapping the same three global words took 1.6 ms of a 2.6 ms analysis
    __shared__ unsigned long long local[max_offsets];
    __shared__ int s_too_many;
    if (threadIdx.x < max_offsets) local[threadIdx.x] = ana_empty;
    if (threadIdx.x == 0) s_too_many = 0;
    __syncthreads();
    unsigned long long seen[max_offsets];
#pragma unroll
    for (int j = 0; j < max_offsets; ++j) seen[j] = ana_empty;
    int widest = 0;
    for (int row = blockIdx.x * 256 + threadIdx.x; row < n; row += gridDim.x * 256) {
        int deps = 0;
        for (int k = rp[row]; k < rp[row + 1]; ++k) {
            const int col = ci[k];
            if (!((lower ? col < row : col > row) && col >= 0 && col < n)) continue;
            ++deps;
            const unsigned long long d = static_cast<unsigned long long>(lower ? row - col : col - row);
            bool found = false;
#pragma unroll
            for (int j = 0; j < max_offsets; ++j) found = found || seen[j] == d;
            if (found) continue;
            for (int j = 0; j < max_offsets && !found; ++j) {
                unsigned long long cur = local[j];
                if (cur == ana_empty) {
                    const unsigned long long old = atomicCAS(&local[j], ana_empty, d);
                    cur = old == ana_empty ? d : old;
                }
                seen[j] = cur;
                found = cur == d;
            }
            if (!found) s_too_many = 1;
        }
        widest = max(widest, deps);
    }
    widest = max(widest, __shfl_xor(widest, 32, 64));
    for (int off = 16; off > 0; off >>= 1) widest = max(widest, __shfl_xor(widest, off, 64));
    if ((threadIdx.x & 63) == 0 && widest > 0) atomicMax(width, widest);
    __syncthreads();
    if (threadIdx.x < max_offsets && local[threadIdx.x] != ana_empty) {
        const unsigned long long d = local[threadIdx.x];
        bool found = false;
        for (int j = 0; j < max_offsets && !found; ++j) {
            unsigned long long cur = __hip_atomic_load(table + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == ana_empty) {
                const unsigned long long old = atomicCAS(table + j, ana_empty, d);
                cur = old == ana_empty ? d : old;
            }
            found = cur == d;
        }
        if (!found) atomicExch(too_many, 1);
    }
    if (threadIdx.x == 0 && s_too_many) atomicExch(too_many, 1);
}

// per brick, in this order: rows, inflow entries, levels, steps with 64 / 128 / 256 threads, widest level,
// predecessors, failed
constexpr int ana_rows = 0, ana_ext = 1, ana_nfine = 2, ana_steps64 = 3, ana_steps128 = 4, ana_steps256 = 5,
              ana_widest = 6, ana_npred = 7, ana_failed = 8, ana_stats = 9;
constexpr int ana_max_cells = 2048;
constexpr int ana_block = 256;

// coordinates of brick b: first cell and lengths along every dimension; returns the number of cells
__device__ __forceinline__ int ana_brick_box(const ana_geometry& g, int b, int* lo, int* len)
{
    int cells = 1;
#pragma unroll
    for (int k = 0; k < max_dims; ++k) {
        if (k >= g.dims) break;
        const int bc = (b / g.mul[k]) % g.nbk[k];
        lo[k] = bc * g.edge[k];
        len[k] = min(g.edge[k], g.extent[k] - lo[k]);
        cells *= len[k];
    }
    return cells;
}

__device__ __forceinline__ int ana_row_of_cell(const ana_geometry& g, const int* lo, const int* len, int cell)
{
    int row = 0;
#pragma unroll
    for (int k = 0; k < max_dims; ++k) {
        if (k >= g.dims) break;
        const int c = cell % len[k];
        cell /= len[k];
        row += (lo[k] + c) * g.stride[k];
    }
    return row;  // may be >= n in the last, incomplete plane
}

__global__ __launch_bounds__(ana_block) void ana_brick_kernel(ana_geometry g, const int32_t* __restrict__ rp,
                                                              const int32_t* __restrict__ ci, int32_t* __restrict__ fine_out,
                                                              uint8_t* __restrict__ row_ext, int32_t* __restrict__ stats,
                                                              int32_t* __restrict__ preds)
{
    __shared__ short fine[ana_max_cells];
    __shared__ short dep_cell[ana_max_cells * max_width];  // in-brick dependencies as cells, -1 = none
    __shared__ int hist[ana_max_cells + 1];
    __shared__ int pred_list[max_preds];
    __shared__ int s_rows, s_ext, s_failed, s_changed;
    const int b = blockIdx.x, tid = threadIdx.x;
    int lo[max_dims], len[max_dims];
    const int cells = ana_brick_box(g, b, lo, len);
    if (tid == 0) {
        s_rows = 0; s_ext = 0; s_failed = cells > ana_max_cells ? 1 : 0;
    }
    for (int j = tid; j < max_preds; j += ana_block) pred_list[j] = -1;
    __syncthreads();
    if (cells > ana_max_cells) {
        if (tid == 0) stats[b * ana_stats + ana_failed] = 1;
        return;
    }
    // 1. classify every dependency of every row of the brick
    int my_rows = 0, my_ext = 0;
    for (int cell = tid; cell < cells; cell += ana_block) {
        fine[cell] = 0;
        for (int e = 0; e < max_width; ++e) dep_cell[cell * max_width + e] = -1;
        const int row = ana_row_of_cell(g, lo, len, cell);
        if (row >= g.n) {
            fine[cell] = -1;  // no such row
            continue;
        }
        ++my_rows;
        int inside = 0, outside = 0;
        for (int k = rp[row]; k < rp[row + 1]; ++k) {
            const int col = ci[k];
            if (!ana_is_dep(g, col, row)) continue;
            int other_cell;
            const int other = ana_brick_of(g, col, &other_cell);
            if (other == b) {
                if (inside < max_width) dep_cell[cell * max_width + inside] = static_cast<short>(other_cell);
                ++inside;
                continue;
            }
            ++outside;
            bool known = false;
            for (int j = 0; j < max_preds && !known; ++j) {
                int cur = pred_list[j];
                if (cur == -1) {
                    const int old = atomicCAS(&pred_list[j], -1, other);
                    cur = old == -1 ? other : old;
                }
                known = cur == other;
            }
            if (!known) atomicExch(&s_failed, 1);
        }
        if (inside + outside > max_width) atomicExch(&s_failed, 1);
        row_ext[row] = static_cast<uint8_t>(outside);
        my_ext += outside;
    }
    atomicAdd(&s_rows, my_rows);
    atomicAdd(&s_ext, my_ext);
    __syncthreads();
    // 2. levels inside the brick: fine(row) = 1 + max over its in-brick dependencies, by relaxation (values only
    //    grow; a triangular pattern has no cycles, so this ends after at most `levels` rounds)
    for (int round = 0; round <= cells; ++round) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        int changed = 0;
        for (int cell = tid; cell < cells; cell += ana_block) {
            if (fine[cell] < 0) continue;
            int lvl = 0;
            for (int e = 0; e < max_width; ++e) {
                const int d = dep_cell[cell * max_width + e];
                if (d >= 0) lvl = max(lvl, fine[d] + 1);
            }
            if (lvl != fine[cell]) {
                fine[cell] = static_cast<short>(lvl);
                changed = 1;
            }
        }
        if (changed) s_changed = 1;
        __syncthreads();
        const int again = s_changed;
        __syncthreads();
        if (!again) break;
    }
    // 3. what the host needs of this brick
    for (int l = tid; l <= cells; l += ana_block) hist[l] = 0;
    __syncthreads();
    for (int cell = tid; cell < cells; cell += ana_block) {
        if (fine[cell] < 0) continue;
        atomicAdd(&hist[fine[cell]], 1);
        fine_out[ana_row_of_cell(g, lo, len, cell)] = fine[cell];
    }
    __syncthreads();
    if (tid == 0) {
        int nfine = 0, s64 = 0, s128 = 0, s256 = 0, widest = 0, npred = 0;
        for (int l = 0; l < cells && hist[l] > 0; ++l) {
            ++nfine;
            s64 += (hist[l] + 63) / 64;
            s128 += (hist[l] + 127) / 128;
            s256 += (hist[l] + 255) / 256;
            widest = max(widest, hist[l]);
        }
        for (int j = 0; j < max_preds; ++j) {
            if (pred_list[j] >= 0) ++npred;
        }
        int32_t* st = stats + b * ana_stats;
        st[ana_rows] = s_rows; st[ana_ext] = s_ext; st[ana_nfine] = nfine; st[ana_steps64] = s64;
        st[ana_steps128] = s128; st[ana_steps256] = s256; st[ana_widest] = widest; st[ana_npred] = npred;
        st[ana_failed] = s_failed;
    }
    // the host analysis lists a brick's predecessors in the order its rows meet them; the LDS list fills in
    // arrival order -- sorted by id here, and there (after the fact) too: only the SET matters downstream
    for (int j = tid; j < max_preds; j += ana_block) preds[b * max_preds + j] = pred_list[j];
}

// per brick (by id): its rank, first plan position, first step, first level slot -- from the host
struct ana_place_args {
    const int32_t* rank;             // brick id -> rank
    const int32_t* brick_row_begin;  // by rank
    const int32_t* brick_step_ptr;   // by rank
};

__global__ __launch_bounds__(ana_block) void ana_place_kernel(ana_geometry g, int threads, ana_place_args a,
                                                              const int32_t* __restrict__ fine_in,
                                                              const uint8_t* __restrict__ row_ext,
                                                              int32_t* __restrict__ perm, int32_t* __restrict__ inv_local,
                                                              int32_t* __restrict__ row_rank, int32_t* __restrict__ step_begin,
                                                              int32_t* __restrict__ ext_in_plan_order)
{
    __shared__ short fine[ana_max_cells];
    __shared__ int base[ana_max_cells + 1];   // next free plan position of every level
    __shared__ int count[ana_max_cells + 1];  // rows of every level, then its first step
    __shared__ short chunk_level[ana_block];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r = a.rank[b];
    const int p0 = a.brick_row_begin[r];
    int lo[max_dims], len[max_dims];
    const int cells = ana_brick_box(g, b, lo, len);
    for (int l = tid; l <= cells; l += ana_block) count[l] = 0;
    __syncthreads();
    for (int cell = tid; cell < cells; cell += ana_block) {
        const int row = ana_row_of_cell(g, lo, len, cell);
        const int lvl = row < g.n ? fine_in[row] : -1;
        fine[cell] = static_cast<short>(lvl);
        if (lvl >= 0) atomicAdd(&count[lvl], 1);
    }
    __syncthreads();
    if (tid == 0) {
        // first plan position and first step of every level; the step table (a level in chunks of `threads`)
        int pos = p0, step = a.brick_step_ptr[r];
        for (int l = 0; l < cells && count[l] > 0; ++l) {
            base[l] = pos;
            for (int q = 0; q < count[l]; q += threads) step_begin[step++] = q == 0 ? ((pos + q) | level_bit) : (pos + q);
            pos += count[l];
        }
    }
    __syncthreads();
    // rows of a level in row order: cells in ascending order, a chunk of ana_block at a time; inside a chunk
    // a row's place is the number of earlier cells of its level
    for (int c0 = 0; c0 < cells; c0 += ana_block) {
        const int cell = c0 + tid;
        const int lvl = cell < cells ? fine[cell] : -1;
        chunk_level[tid] = static_cast<short>(lvl);
        __syncthreads();
        int before = 0, after = 0;
        if (lvl >= 0) {
            for (int u = 0; u < ana_block; ++u) {
                const int same = chunk_level[u] == lvl;
                before += same & (u < tid);
                after += same & (u > tid);
            }
        }
        int p = -1;
        if (lvl >= 0) p = base[lvl] + before;
        __syncthreads();
        if (lvl >= 0) {
            if (after == 0) base[lvl] = p + 1;  // the last of its level in this chunk moves the level on
            const int row = ana_row_of_cell(g, lo, len, cell);
            perm[p] = row;
            inv_local[row] = p - p0;
            row_rank[row] = r;
            ext_in_plan_order[p] = row_ext[row];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void ana_inflow_kernel(ana_geometry g, const int32_t* __restrict__ rp,
                                                         const int32_t* __restrict__ ci, const int32_t* __restrict__ perm,
                                                         const int32_t* __restrict__ ext_row_off, int32_t* __restrict__ ext_col)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= g.n) return;
    const int row = perm[p];
    const int mine = ana_brick_of(g, row, nullptr);
    int at = ext_row_off[p];
    for (int k = rp[row]; k < rp[row + 1]; ++k) {
        const int col = ci[k];
        if (!ana_is_dep(g, col, row)) continue;
        if (ana_brick_of(g, col, nullptr) != mine) ext_col[at++] = col;
    }
}

struct device_buffer {
    void* p = nullptr;
    ~device_buffer()
    {
        if (p != nullptr) (void)hipFree(p);
    }
    int alloc(size_t bytes) { return static_cast<int>(hipMalloc(&p, bytes > 0 ? bytes : 8)); }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

// edges of the bricks for a box of `dims` dimensions (the rule of the host analysis, step 3)
void pick_edges(const gkomi_trs_bricks& h, int dims, int active, const int64_t* extent, int64_t brick_rows, int64_t* edge,
                int64_t* nbk)
{
    if (h.mode == 2 && active >= 3) {
        const int64_t cross = active == 3 ? 8 : 4;
        int last = dims - 1;
        while (extent[last] <= 1) --last;
        int64_t product = 1;
        for (int k = 0; k < dims; ++k) {
            if (k == last) continue;
            edge[k] = std::min<int64_t>(extent[k], extent[k] > 1 ? cross : 1);
            product *= edge[k];
        }
        edge[last] = std::max<int64_t>(1, std::min<int64_t>(extent[last], brick_rows / product));
        for (int k = 0; k < dims; ++k) {
            nbk[k] = ceildiv(extent[k], edge[k]);
            if (k == last) edge[k] = ceildiv(extent[k], nbk[k]);
            nbk[k] = ceildiv(extent[k], edge[k]);
        }
    } else {
        int order[max_dims];
        for (int k = 0; k < dims; ++k) order[k] = k;
        std::sort(order, order + dims, [&](int a, int b) { return extent[a] < extent[b]; });
        double remaining = static_cast<double>(std::max<int64_t>(brick_rows, 1));
        for (int q = 0; q < dims; ++q) {
            const int k = order[q];
            const double want = std::pow(remaining, 1.0 / (dims - q));
            int64_t e = std::max<int64_t>(1, static_cast<int64_t>(std::llround(want)));
            e = std::min(e, extent[k]);
            nbk[k] = ceildiv(extent[k], e);
            edge[k] = ceildiv(extent[k], nbk[k]);
            nbk[k] = ceildiv(extent[k], edge[k]);
            remaining = std::max(1.0, remaining / static_cast<double>(edge[k]));
        }
    }
    if (const char* forced = getenv("GKOMI_TRS_BRICK_EDGES")) {
        int k = 0;
        for (const char* q = forced; *q != 0 && k < dims; ++k) {
            edge[k] = std::max<int64_t>(1, std::min<int64_t>(extent[k], atoll(q)));
            nbk[k] = ceildiv(extent[k], edge[k]);
            while (*q != 0 && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
}

#define ANA_TRY(expr)                              \
    do {                                           \
        const int e_ = static_cast<int>(expr);     \
        if (e_) return e_;                         \
    } while (0)

int analyse_device(gkomi_trs_bricks& h, hipStream_t stream, const int32_t* rp, const int32_t* ci, int64_t brick_rows,
                   int threads, int mode)
{
    h.mode = mode == 1 ? 1 : 2;
    if (h.mode == 2) threads = 64;
    const int64_t n = h.n;
    const bool lower = h.lower != 0;
    // A. offsets and width
    device_buffer small;
    ANA_TRY(small.alloc(sizeof(unsigned long long) * max_offsets + 2 * sizeof(int)));
    struct {
        unsigned long long table[max_offsets];
        int width, too_many;
    } head;
    for (unsigned long long& t : head.table) t = ana_empty;
    head.width = 0;
    head.too_many = 0;
    ANA_TRY(hipMemcpyAsync(small.p, &head, sizeof(head), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(ana_offsets_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, static_cast<int>(n), lower ? 1 : 0, rp,
                       ci, small.as<unsigned long long>(), reinterpret_cast<int*>(small.as<unsigned long long>() + max_offsets),
                       reinterpret_cast<int*>(small.as<unsigned long long>() + max_offsets) + 1);
    ANA_TRY(check_launch());
    ANA_TRY(hipMemcpyAsync(&head, small.p, sizeof(head), hipMemcpyDeviceToHost, stream));
    ANA_TRY(hipStreamSynchronize(stream));
    const int width = head.width;
    // (not a lexicographic box numbering: the caller may still find the grid in the dependency graph, on the host)
    if (head.too_many) {
        h.geometry_failed_width = width;
        return GKOMI_ENOTSUPPORTED;
    }
    int64_t offs[max_offsets];
    int noffs = 0;
    for (unsigned long long t : head.table) {
        if (t != ana_empty) offs[noffs++] = static_cast<int64_t>(t);
    }
    if (noffs == 0 || width > max_width) return GKOMI_ENOTSUPPORTED;
    std::sort(offs, offs + noffs);
    // strides of a lexicographic box numbering: a divisor chain (step 2 of the host analysis)
    int64_t stride[max_dims];
    int dims = 0;
    stride[dims++] = 1;
    for (int j = 0; j < noffs; ++j) {
        if (offs[j] == stride[dims - 1]) continue;
        if (offs[j] % stride[dims - 1] != 0 || dims == max_dims) {
            h.geometry_failed_width = width;
            return GKOMI_ENOTSUPPORTED;
        }
        stride[dims++] = offs[j];
    }
    int64_t extent[max_dims];
    for (int k = 0; k + 1 < dims; ++k) extent[k] = stride[k + 1] / stride[k];
    extent[dims - 1] = ceildiv(n, stride[dims - 1]);
    int active = 0;
    for (int k = 0; k < dims; ++k) active += extent[k] > 1;
    h.levels_estimate = 1;
    for (int k = 0; k < dims; ++k) h.levels_estimate += extent[k] - 1;
    if (brick_rows <= 0) brick_rows = h.mode == 2 ? (active >= 3 ? 1728 : 2025) : 4096;
    const int slots = width <= 2 ? 2 : width <= 3 ? 3 : width <= 4 ? 4 : 8;

    device_buffer fine, row_ext;
    ANA_TRY(fine.alloc(sizeof(int32_t) * n));
    ANA_TRY(row_ext.alloc(static_cast<size_t>(n)));
    for (int attempt = 0; attempt < 8; ++attempt, brick_rows = std::max<int64_t>(brick_rows / 2, 8)) {
        int64_t edge[max_dims], nbk[max_dims];
        pick_edges(h, dims, active, extent, brick_rows, edge, nbk);
        int64_t nbricks = 1, cells = 1;
        for (int k = 0; k < dims; ++k) {
            nbricks *= nbk[k];
            cells *= edge[k];
            if (nbricks > (1 << 24)) return GKOMI_ENOTSUPPORTED;
        }
        if (cells > ana_max_cells) {
            if (brick_rows <= 8) return GKOMI_ENOTSUPPORTED;
            continue;  // smaller bricks (the host analysis would find max_rows > 2048 or too much LDS)
        }
        ana_geometry g{};
        g.dims = dims;
        g.n = static_cast<int>(n);
        g.lower = lower ? 1 : 0;
        for (int k = 0, m = 1; k < dims; ++k) {
            g.stride[k] = static_cast<int>(stride[k]);
            g.extent[k] = static_cast<int>(extent[k]);
            g.edge[k] = static_cast<int>(edge[k]);
            g.nbk[k] = static_cast<int>(nbk[k]);
            g.mul[k] = m;
            m *= static_cast<int>(nbk[k]);
        }
        // D. one workgroup per brick
        device_buffer stats, preds;
        ANA_TRY(stats.alloc(sizeof(int32_t) * ana_stats * nbricks));
        ANA_TRY(preds.alloc(sizeof(int32_t) * max_preds * nbricks));
        hipLaunchKernelGGL(ana_brick_kernel, dim3(static_cast<unsigned>(nbricks)), dim3(ana_block), 0, stream, g, rp, ci,
                           fine.as<int32_t>(), row_ext.as<uint8_t>(), stats.as<int32_t>(), preds.as<int32_t>());
        ANA_TRY(check_launch());
        std::vector<int32_t> hstats(static_cast<size_t>(ana_stats * nbricks)), hpreds(static_cast<size_t>(max_preds * nbricks));
        ANA_TRY(hipMemcpyAsync(hstats.data(), stats.p, sizeof(int32_t) * hstats.size(), hipMemcpyDeviceToHost, stream));
        ANA_TRY(hipMemcpyAsync(hpreds.data(), preds.p, sizeof(int32_t) * hpreds.size(), hipMemcpyDeviceToHost, stream));
        ANA_TRY(hipStreamSynchronize(stream));
        std::vector<int32_t> npred(static_cast<size_t>(nbricks), 0);
        for (int64_t b = 0; b < nbricks; ++b) {
            if (hstats[b * ana_stats + ana_failed]) return GKOMI_ENOTSUPPORTED;
            // the set of predecessors, in ascending id order
            int32_t* list = hpreds.data() + b * max_preds;
            int cnt = 0;
            for (int j = 0; j < max_preds; ++j) {
                if (list[j] >= 0) list[cnt++] = list[j];
            }
            std::sort(list, list + cnt);
            npred[b] = cnt;
        }
        // coarse levels by longest path (Kahn); a cycle = the guessed geometry is wrong
        std::vector<int32_t> succ_ptr(static_cast<size_t>(nbricks) + 1, 0);
        for (int64_t b = 0; b < nbricks; ++b) {
            for (int j = 0; j < npred[b]; ++j) ++succ_ptr[hpreds[b * max_preds + j] + 1];
        }
        for (int64_t b = 0; b < nbricks; ++b) succ_ptr[b + 1] += succ_ptr[b];
        std::vector<int32_t> succ(static_cast<size_t>(succ_ptr[nbricks]));
        {
            std::vector<int32_t> cursor(succ_ptr.begin(), succ_ptr.end() - 1);
            for (int64_t b = 0; b < nbricks; ++b) {
                for (int j = 0; j < npred[b]; ++j) succ[cursor[hpreds[b * max_preds + j]]++] = static_cast<int32_t>(b);
            }
        }
        std::vector<int32_t> coarse(static_cast<size_t>(nbricks), 0), waiting(npred);
        std::vector<int32_t> topo;
        topo.reserve(static_cast<size_t>(nbricks));
        for (int64_t b = 0; b < nbricks; ++b) {
            if (waiting[b] == 0) topo.push_back(static_cast<int32_t>(b));
        }
        for (size_t q = 0; q < topo.size(); ++q) {
            const int32_t b = topo[q];
            for (int32_t j = succ_ptr[b]; j < succ_ptr[b + 1]; ++j) {
                const int32_t t = succ[j];
                coarse[t] = std::max(coarse[t], coarse[b] + 1);
                if (--waiting[t] == 0) topo.push_back(t);
            }
        }
        if (static_cast<int64_t>(topo.size()) != nbricks) return GKOMI_ENOTSUPPORTED;
        int32_t ncoarse = 0;
        for (int64_t b = 0; b < nbricks; ++b) ncoarse = std::max(ncoarse, coarse[b] + 1);
        std::vector<int32_t> rank(static_cast<size_t>(nbricks));
        {
            std::vector<int32_t> count(static_cast<size_t>(ncoarse) + 1, 0);
            for (int64_t b = 0; b < nbricks; ++b) ++count[coarse[b] + 1];
            for (int32_t l = 0; l < ncoarse; ++l) count[l + 1] += count[l];
            for (int64_t b = 0; b < nbricks; ++b) rank[b] = count[coarse[b]]++;
        }
        int64_t max_lds = 0, max_rows = 0;
        int max_level_rows = 0;
        for (int64_t b = 0; b < nbricks; ++b) {
            const int32_t* st = hstats.data() + b * ana_stats;
            const int64_t r = st[ana_rows];
            max_lds = std::max<int64_t>(max_lds, static_cast<int64_t>(brick_lds_bytes(r, st[ana_ext], st[ana_nfine] + r / 64 + 1, slots)));
            max_rows = std::max<int64_t>(max_rows, r);
            max_level_rows = std::max(max_level_rows, st[ana_widest]);
        }
        if (static_cast<size_t>(max_lds) + 256 > max_lds_bytes || (h.mode == 2 && max_rows > 2048)) {
            if (brick_rows <= 8) return GKOMI_ENOTSUPPORTED;
            continue;
        }
        if (threads <= 0) threads = max_level_rows <= 64 ? 64 : max_level_rows <= 160 ? 128 : 256;
        const int steps_of = threads == 64 ? ana_steps64 : threads == 128 ? ana_steps128 : ana_steps256;
        h.nbricks = nbricks;
        h.coarse_levels = ncoarse;
        h.width = slots;
        h.lds_bytes_max = max_lds;
        h.threads = threads;
        std::vector<int32_t> by_rank(static_cast<size_t>(nbricks));
        for (int64_t b = 0; b < nbricks; ++b) by_rank[rank[b]] = static_cast<int32_t>(b);
        h.brick_row_begin.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.brick_ext_begin.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.brick_step_ptr.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.nlevels_fine = 0;
        int64_t total_ext = 0;
        for (int64_t r = 0; r < nbricks; ++r) {
            const int32_t* st = hstats.data() + by_rank[r] * ana_stats;
            h.brick_row_begin[r + 1] = h.brick_row_begin[r] + st[ana_rows];
            total_ext += st[ana_ext];
            if (total_ext > INT32_MAX) return GKOMI_ENOTSUPPORTED;
            h.brick_ext_begin[r + 1] = static_cast<int32_t>(total_ext);
            h.brick_step_ptr[r + 1] = h.brick_step_ptr[r] + st[steps_of];
            h.nlevels_fine += st[ana_nfine];
        }
        h.nsteps = h.brick_step_ptr[nbricks];
        // predecessors as ranks; critical path in steps
        h.pred_ptr.assign(static_cast<size_t>(nbricks) + 1, 0);
        h.pred_idx.clear();
        std::vector<int64_t> path(static_cast<size_t>(nbricks), 0);
        h.critical_steps = 0;
        h.max_brick_steps = 0;
        for (int64_t r = 0; r < nbricks; ++r) {
            const int32_t b = by_rank[r];
            int64_t before = 0;
            for (int j = 0; j < npred[b]; ++j) {
                const int32_t pr = rank[hpreds[static_cast<size_t>(b) * max_preds + j]];
                h.pred_idx.push_back(pr);
                before = std::max(before, path[pr]);
            }
            h.pred_ptr[r + 1] = static_cast<int32_t>(h.pred_idx.size());
            const int64_t steps = h.brick_step_ptr[r + 1] - h.brick_step_ptr[r];
            path[r] = before + steps;
            h.critical_steps = std::max(h.critical_steps, path[r]);
            h.max_brick_steps = std::max(h.max_brick_steps, steps);
        }
        h.image_off.assign(h.mode == 2 ? static_cast<size_t>(nbricks) + 1 : 0, 0);
        for (int64_t r = 0; r < nbricks && h.mode == 2; ++r) {
            h.image_off[r + 1] = h.image_off[r] + brick_image_bytes(h.brick_row_begin[r + 1] - h.brick_row_begin[r],
                                                                    h.brick_ext_begin[r + 1] - h.brick_ext_begin[r],
                                                                    h.brick_step_ptr[r + 1] - h.brick_step_ptr[r], slots);
        }
        h.n_step_begin = h.nsteps + 1;
        h.n_ext_col = total_ext;
        h.n_pred_idx = static_cast<int64_t>(h.pred_idx.size());
        h.image_bytes = h.image_off.empty() ? 0 : h.image_off.back();
        // the index arrays, in device memory owned by the handle
        auto ints = [](size_t count) { return align_up(sizeof(int32_t) * (count > 0 ? count : 1), 256); };
        size_t off = 0;
        h.dev_perm = off; off += ints(n);
        h.dev_row_rank = off; off += ints(n);
        h.dev_inv_local = off; off += ints(n);
        h.dev_ext_row_off = off; off += ints(n);
        h.dev_ext_col = off; off += ints(static_cast<size_t>(total_ext));
        h.dev_brick_row_begin = off; off += ints(nbricks + 1);
        h.dev_brick_step_ptr = off; off += ints(nbricks + 1);
        h.dev_step_begin = off; off += ints(static_cast<size_t>(h.n_step_begin));
        h.dev_brick_ext_begin = off; off += ints(nbricks + 1);
        h.dev_bytes = off;
        void* index = nullptr;
        ANA_TRY(hipMalloc(&index, off));
        h.dev_index = static_cast<char*>(index);
        char* d = h.dev_index;
        device_buffer drank, scan_ws;
        ANA_TRY(drank.alloc(sizeof(int32_t) * nbricks));
        ANA_TRY(hipMemcpyAsync(drank.p, rank.data(), sizeof(int32_t) * nbricks, hipMemcpyHostToDevice, stream));
        ANA_TRY(hipMemcpyAsync(d + h.dev_brick_row_begin, h.brick_row_begin.data(), sizeof(int32_t) * (nbricks + 1), hipMemcpyHostToDevice, stream));
        ANA_TRY(hipMemcpyAsync(d + h.dev_brick_step_ptr, h.brick_step_ptr.data(), sizeof(int32_t) * (nbricks + 1), hipMemcpyHostToDevice, stream));
        ANA_TRY(hipMemcpyAsync(d + h.dev_brick_ext_begin, h.brick_ext_begin.data(), sizeof(int32_t) * (nbricks + 1), hipMemcpyHostToDevice, stream));
        const int32_t last_step = static_cast<int32_t>(n) | level_bit;
        ANA_TRY(hipMemcpyAsync(d + h.dev_step_begin + sizeof(int32_t) * h.nsteps, &last_step, sizeof(int32_t), hipMemcpyHostToDevice, stream));
        // E. plan order
        const ana_place_args pa{drank.as<int32_t>(), reinterpret_cast<const int32_t*>(d + h.dev_brick_row_begin),
                                reinterpret_cast<const int32_t*>(d + h.dev_brick_step_ptr)};
        hipLaunchKernelGGL(ana_place_kernel, dim3(static_cast<unsigned>(nbricks)), dim3(ana_block), 0, stream, g, threads, pa,
                           fine.as<int32_t>(), row_ext.as<uint8_t>(), reinterpret_cast<int32_t*>(d + h.dev_perm),
                           reinterpret_cast<int32_t*>(d + h.dev_inv_local), reinterpret_cast<int32_t*>(d + h.dev_row_rank),
                           reinterpret_cast<int32_t*>(d + h.dev_step_begin), reinterpret_cast<int32_t*>(d + h.dev_ext_row_off));
        ANA_TRY(check_launch());
        // F. first inflow entry of every plan position, G. the inflow lists
        const size_t ws_bytes = gkomi_prefix_sum_workspace_bytes(n);
        ANA_TRY(scan_ws.alloc(ws_bytes));
        ANA_TRY(gkomi_prefix_sum_i32(reinterpret_cast<gkomi_stream_t>(stream), reinterpret_cast<int32_t*>(d + h.dev_ext_row_off), n,
                                     scan_ws.p, ws_bytes));
        hipLaunchKernelGGL(ana_inflow_kernel, dim3(static_cast<unsigned>(ceildiv(n, 256))), dim3(256), 0, stream, g, rp, ci,
                           reinterpret_cast<const int32_t*>(d + h.dev_perm), reinterpret_cast<const int32_t*>(d + h.dev_ext_row_off),
                           reinterpret_cast<int32_t*>(d + h.dev_ext_col));
        ANA_TRY(check_launch());
        ANA_TRY(hipStreamSynchronize(stream));  // the scratch buffers go out of scope here
        h.host_arrays_valid = false;
        return GKOMI_SUCCESS;
    }
    return GKOMI_ENOTSUPPORTED;
}

// the device-analysed index arrays on the host, when somebody asks (gkomi_trs_bricks_host_array)
int fetch_host_arrays(gkomi_trs_bricks& h)
{
    if (h.host_arrays_valid || h.dev_index == nullptr) return GKOMI_SUCCESS;
    auto get = [&](std::vector<int32_t>& v, size_t off, int64_t count) {
        v.resize(static_cast<size_t>(count));
        return count > 0 ? static_cast<int>(hipMemcpy(v.data(), h.dev_index + off, sizeof(int32_t) * count, hipMemcpyDeviceToHost)) : 0;
    };
    ANA_TRY(get(h.perm, h.dev_perm, h.n));
    ANA_TRY(get(h.row_rank, h.dev_row_rank, h.n));
    ANA_TRY(get(h.inv_local, h.dev_inv_local, h.n));
    ANA_TRY(get(h.ext_row_off, h.dev_ext_row_off, h.n));
    ANA_TRY(get(h.ext_col, h.dev_ext_col, h.n_ext_col));
    ANA_TRY(get(h.step_begin, h.dev_step_begin, h.n_step_begin));
    h.host_arrays_valid = true;
    return GKOMI_SUCCESS;
}
#undef ANA_TRY

// ---------------------------------------------------------------- numeric phase (device) ----

// ---- x / d without the division's latency -------------------------------------------------
// The compiler's f64 division is v_div_scale (x2), v_rcp, two Newton steps on the reciprocal,
// q = n r, e = fma(-d, q, n), v_div_fmas = fma(e, r, q), v_div_fixup: 12 dependent instructions on
// the critical path of every level.  The reciprocal part depends on d alone -- as long as
// v_div_scale leaves both operands unscaled, which it does when their exponents are moderate --
// so the numeric phase runs exactly that part once per row, and the solve finishes with the
// last three operations: the same instructions on the same operands, hence the same bits as
// sum / d.  Outside the box (|d| or |sum| not in [2^-383, 2^383), zero, NaN, infinity, subnormal)
// the solve divides.  tests/test_trs_bricks_gpu.py pins it against the oracle's division over
// the full exponent range.
constexpr int safe_exponent_lo = 1023 - 383, safe_exponent_hi = 1023 + 383;  // biased, half-open

__device__ __forceinline__ bool exponent_is_safe(double v)
{
    const int e = static_cast<int>((__double_as_longlong(v) >> 52) & 0x7ff);
    return e >= safe_exponent_lo && e < safe_exponent_hi;
}

// r as the division computes it for an unscaled denominator; NaN = "divide"
__device__ __forceinline__ double refined_reciprocal(double d)
{
    if (!exponent_is_safe(d)) return __longlong_as_double(0x7ff8000000000000ll);
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

__device__ __forceinline__ double divide_by_row_diagonal(double sum, double d, double r)
{
    // the quotient first, the question whether it may be used beside it (not in front of it)
    const double q = sum * r;
    const double e = __builtin_fma(-d, q, sum);
    double result = __builtin_fma(e, r, q);
    if (!(exponent_is_safe(sum) && r == r)) result = sum / d;
    return result;
}

// the factor once more in plan order: dependencies only, as LDS indices of the brick (own rows:
// 0 .. R-1, inflow: R ...), ELL over the whole plan (slot e of position p at e * n + p), the
// diagonal (last stored occurrence, like the reference's loop) apart
__global__ __launch_bounds__(256) void trs_brick_fill_kernel(
    int32_t n, int width, bool lower, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, const int32_t* __restrict__ perm, const int32_t* __restrict__ row_rank,
    const int32_t* __restrict__ inv_local, const int32_t* __restrict__ ext_row_off,
    const int32_t* __restrict__ brick_row_begin, const int32_t* __restrict__ brick_ext_begin,
    double* __restrict__ diag, double* __restrict__ rdiag, int32_t* __restrict__ cols,
    double* __restrict__ pvals)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int row = perm[p];
    const int rank = row_rank[row];
    const int rows_here = brick_row_begin[rank + 1] - brick_row_begin[rank];
    int inflow = ext_row_off[p] - brick_ext_begin[rank];
    double d = 1.0;
    int e = 0;
    for (int k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k) {
        const int col = col_idxs[k];
        if (col == row) d = vals[k];
        if ((lower ? col < row : col > row) && col >= 0 && col < n) {
            int c;
            if (row_rank[col] == rank) {
                c = inv_local[col];
            } else {
                c = rows_here + inflow;
                ++inflow;
            }
            if (e < width) {
                cols[static_cast<int64_t>(e) * n + p] = c;
                pvals[static_cast<int64_t>(e) * n + p] = vals[k];
            }
            ++e;
        }
    }
    diag[p] = d;
    rdiag[p] = refined_reciprocal(d);
    for (; e < width; ++e) {
        cols[static_cast<int64_t>(e) * n + p] = pad_col;
        pvals[static_cast<int64_t>(e) * n + p] = 0.0;
    }
}

// ---------------------------------------------------------------- solve ---------------------

// LDS-only barrier: the prefetch loads of the next step stay in flight (__syncthreads waits
// for vmcnt(0)); a one-wave workgroup needs the ordering only
template <int T>
__device__ __forceinline__ void lds_barrier()
{
    if (T > 64) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int T, int K, bool Unit>
__global__ __launch_bounds__(T) void trs_brick_solve_kernel(
    brick_header* hdr, const int32_t* __restrict__ perm, const double* __restrict__ diag,
    const double* __restrict__ rdiag, const int32_t* __restrict__ cols, const double* __restrict__ pvals,
    const int32_t* __restrict__ brick_row_begin, const int32_t* __restrict__ brick_step_ptr,
    const int32_t* __restrict__ step_begin, const int32_t* __restrict__ brick_ext_begin,
    const int32_t* __restrict__ ext_col, const int32_t* __restrict__ pred_ptr,
    const int32_t* __restrict__ pred_idx, unsigned int* done, int32_t n, const double* b,
    int64_t b_stride, double* x, int64_t x_stride, unsigned int epoch, long long max_polls)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned int s_ticket;
    __shared__ int s_gave_up;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_ticket = atomicAdd(&hdr->ticket, 1u);
        s_gave_up = 0;
    }
    __syncthreads();
    const int bk = static_cast<int>(s_ticket);
    const int r0 = brick_row_begin[bk], rows = brick_row_begin[bk + 1] - r0;
    const int e0 = brick_ext_begin[bk], inflow = brick_ext_begin[bk + 1] - e0;
    const int s0 = brick_step_ptr[bk], nsteps = brick_step_ptr[bk + 1] - s0;
    double* lx = lds;                    // right-hand side, then the solution, then the inflow
    const int zero_cell = rows + inflow; // lx[zero_cell] = 0.0: what an empty dependency slot points at
    double* ld = lx + zero_cell + 1;     // diagonal
    double* lr = ld + rows;              // its reciprocal as the division would compute it
    double* lv = lr + rows;              // values, slot-major
    int32_t* lc = reinterpret_cast<int32_t*>(lv + static_cast<size_t>(K) * rows);  // LDS indices, slot-major
    int32_t* ls = lc + static_cast<size_t>(K) * rows;                              // step bounds
    int32_t* lrow = ls + nsteps + 4;                                               // my rows
    int32_t* lext = lrow + rows;                                                   // rows my inflow comes from
    // A. nothing here depends on other bricks, and the brick usually still waits for them: the
    //    right-hand side of my rows and my part of the factor into LDS, every load in flight
    for (int i = tid; i < rows; i += T) {
        const int row = perm[r0 + i];
        lrow[i] = row;
        lx[i] = b[row * b_stride];
        if (!Unit) {
            ld[i] = diag[r0 + i];
            lr[i] = rdiag[r0 + i];
        }
#pragma unroll
        for (int e = 0; e < K; ++e) {
            // an empty slot: value 0.0 (as stored) times the zero cell, sum - (+0.0) = sum bit for bit
            const int c = cols[static_cast<int64_t>(e) * n + r0 + i];
            lv[e * rows + i] = pvals[static_cast<int64_t>(e) * n + r0 + i];
            lc[e * rows + i] = c == pad_col ? zero_cell : c;
        }
    }
    if (tid == 0) lx[zero_cell] = 0.0;
    for (int i = tid; i < nsteps + 4; i += T) ls[i] = step_begin[s0 + min(i, nsteps)];  // read up to 3 steps ahead
    for (int j = tid; j < inflow; j += T) lext[j] = ext_col[e0 + j];
    // B. the bricks I depend on have finished?  (they hold smaller tickets: they have started)
    {
        bool gave_up = false;
        for (int j = pred_ptr[bk] + tid; j < pred_ptr[bk + 1]; j += T) {
            const unsigned int* flag = done + pred_idx[j];
            long long polls = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                if (++polls > max_polls) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (gave_up) s_gave_up = 1;
        // no acquire fence (it would invalidate the XCD's L2): the only data of other bricks read
        // below is x, with agent-scope loads that go to memory, issued after the flags were seen
    }
    __syncthreads();
    const bool poisoned = s_gave_up != 0;
    if (!poisoned) {
        // C. what my rows need from other bricks, all loads in flight together
        for (int j = tid; j < inflow; j += T) {
            lx[rows + j] = __longlong_as_double(static_cast<long long>(__hip_atomic_load(
                reinterpret_cast<const unsigned long long*>(x) + static_cast<int64_t>(lext[j]) * x_stride,
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        }
        __syncthreads();
        // D. level by level, LDS only.  A step = at most T rows of one level.  What a step costs is
        //    the chain barrier -> x of the dependencies -> subtractions -> division -> write, so
        //    everything else is kept off it: the factor's entries of the NEXT step are requested
        //    right behind this step's x reads (LDS answers in order) and the step bounds two steps
        //    ahead, all of it in registers by the time it is needed.
        struct step_data {
            int i;  // my row of the brick, -1 = none
            bool opens_level;
            double d, r;
            int c[K];
            double v[K];
        };
        auto fetch = [&](int s, int first, int next_first, step_data& sd) {
            sd.i = -1;
            sd.opens_level = false;
            if (s >= nsteps) return;
            const int begin = (first & ~level_bit) - r0;
            const int end = min((next_first & ~level_bit) - r0, begin + T);
            sd.opens_level = first < 0;
            const int i = begin + tid;
            if (i < end) {
                sd.i = i;
                if (!Unit) {
                    sd.d = ld[i];
                    sd.r = lr[i];
                }
#pragma unroll
                for (int e = 0; e < K; ++e) {
                    sd.c[e] = lc[e * rows + i];
                    sd.v[e] = lv[e * rows + i];
                }
            }
        };
        // one step: `cur` is complete; `nxt` and the bound three steps on are requested on the way
        auto step = [&](int s, const step_data& cur, step_data& nxt, int w1, int w2, int& w3) {
            if (cur.opens_level) lds_barrier<T>();  // the level before is in LDS
            double xd[K];
            double sum = 0.0;
            if (cur.i >= 0) {
#pragma unroll
                for (int e = 0; e < K; ++e) xd[e] = lx[cur.c[e]];
                sum = lx[cur.i];
            }
            w3 = ls[s + 3];
            fetch(s + 1, w1, w2, nxt);
            if (cur.i >= 0) {
#pragma unroll
                for (int e = 0; e < K; ++e) sum -= cur.v[e] * xd[e];
                lx[cur.i] = Unit ? sum : divide_by_row_diagonal(sum, cur.d, cur.r);
            }
        };
        step_data even, odd;
        int w1 = ls[1], w2 = ls[2];  // bounds of the steps s + 1, s + 2
        fetch(0, ls[0], w1, even);
        for (int s = 0; s < nsteps; s += 2) {
            int w3, w4;
            step(s, even, odd, w1, w2, w3);      // requests the bound of step s + 3
            step(s + 1, odd, even, w2, w3, w4);  // ... s + 4
            w1 = w3;
            w2 = w4;
        }
    }
    __syncthreads();
    // E. the brick leaves LDS in one sweep (write-through: other XCDs read it from memory)
    for (int i = tid; i < rows; i += T) {
        const double xr = poisoned ? __longlong_as_double(0x7ff8dead0badbeefll) : lx[i];
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(x) + lrow[i] * x_stride,
                           static_cast<unsigned long long>(__double_as_longlong(xr)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // every wave has waited for its (write-through) stores: x is in memory
    if (tid == 0) {
        // a release, although x went out write-through and every wave has counted its stores down: the
        // count says the XCD's L2 has them, not that they have reached memory, and a relaxed flag
        // overtook them once in ~10^2 full-size solves (stale x in a brick of another XCD)
        __hip_atomic_store(done + bk, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (poisoned) atomicExch(&hdr->overrun, 1u);
        // the last brick to finish re-arms the plan for the next solve (all tickets are taken by then)
        const unsigned int before = atomicAdd(&hdr->finished, 1u);
        if (before + 1 == static_cast<unsigned int>(hdr->nbricks)) {
            hdr->finished = 0;
            hdr->ticket = 0;
        }
    }
}


// The image of every brick (numeric phase of the pipelined plan): what the solve copies into LDS.
template <int K>
__global__ __launch_bounds__(128) void trs_brick_image_kernel(
    int32_t n, const int32_t* __restrict__ perm, const double* __restrict__ diag, const double* __restrict__ rdiag,
    const int32_t* __restrict__ cols, const double* __restrict__ pvals, const int32_t* __restrict__ brick_row_begin,
    const int32_t* __restrict__ brick_step_ptr, const int32_t* __restrict__ step_begin,
    const int32_t* __restrict__ brick_ext_begin, const int32_t* __restrict__ ext_col,
    const int32_t* __restrict__ ext_row_off, char* __restrict__ image, const int64_t* __restrict__ image_off)
{
    constexpr int T = 64;
    constexpr int rec_bytes = brick_record_bytes(K);
    constexpr int off_v = 16, off_ca = 16 + 8 * K, off_xa = 16 + 12 * K, off_row = 20 + 12 * K, off_span = 24 + 12 * K;
    // the division's fast box as an unsigned test on the biased exponent e of the numerator:
    // e - box_first < span.  span = box_span for a row whose diagonal is in the box, 0 for one whose
    // diagonal is not (always the careful path), everything for the spare record (never)
    constexpr unsigned int box_span = safe_exponent_hi - safe_exponent_lo;
    const int bk = blockIdx.x, tid = threadIdx.x;
    const int r0 = brick_row_begin[bk], rows = brick_row_begin[bk + 1] - r0;
    const int e0 = brick_ext_begin[bk], inflow = brick_ext_begin[bk + 1] - e0;
    const int s0 = brick_step_ptr[bk], nsteps = brick_step_ptr[bk + 1] - s0;
    const int zero_cell = rows + inflow, scratch_cell = zero_cell + 1;
    char* lrec = image + image_off[bk];
    int2* lstep = reinterpret_cast<int2*>(lrec + static_cast<size_t>(rows + 1) * rec_bytes);
    int32_t* lext = reinterpret_cast<int32_t*>(lstep + nsteps + 4);
    for (int i = tid; i <= rows; i += 128) {
        const bool real = i < rows;
        const int row = real ? perm[r0 + i] : -1;
        char* rec = lrec + i * rec_bytes;
        const double rd = real ? rdiag[r0 + i] : 1.0;
        *reinterpret_cast<double*>(rec) = real ? diag[r0 + i] : 1.0;
        *reinterpret_cast<double*>(rec + 8) = rd;
#pragma unroll
        for (int e = 0; e < K; ++e) {
            // an empty slot: value 0.0 (as stored) times the zero cell, sum - (+0.0) = sum bit for bit
            const int c = real ? cols[static_cast<int64_t>(e) * n + r0 + i] : pad_col;
            *reinterpret_cast<double*>(rec + off_v + 8 * e) = real ? pvals[static_cast<int64_t>(e) * n + r0 + i] : 0.0;
            *reinterpret_cast<int32_t*>(rec + off_ca + 4 * e) = 8 * (c == pad_col ? zero_cell : c);
        }
        *reinterpret_cast<int32_t*>(rec + off_xa) = 8 * (real ? i : scratch_cell);
        // the row's byte offset in a contiguous x (the solve multiplies by its stride; it is launched
        // only if that fits 31 bits); the spare record's is out of any buffer's range: the store is dropped
        *reinterpret_cast<uint32_t*>(rec + off_row) = real ? static_cast<uint32_t>(row) * 8u : 0xffffffffu;
        *reinterpret_cast<uint32_t*>(rec + off_span) = !real ? 0xffffffffu : rd == rd ? box_span : 0u;
    }
    for (int i = tid; i < nsteps + 4; i += 128) {
        // steps beyond the last are empty; rows of the steps up to and including s end at `end`:
        // their inflow entries come first in the list
        int count = 0, need = inflow;
        if (i < nsteps) {
            const int begin = step_begin[s0 + i] & ~level_bit;
            const int end = min(step_begin[s0 + i + 1] & ~level_bit, begin + T);
            count = end - begin;
            if (end < r0 + rows) need = ext_row_off[end] - e0;
        }
        lstep[i] = make_int2(count, need);
    }
    for (int j = tid; j < inflow; j += 128) lext[j] = ext_col[e0 + j];
}

// ---- pipelined variant ----------------------------------------------------------------------
// Waiting for whole bricks makes the critical path (bricks on the longest chain) x (levels of a
// brick): 2-2.6 times the levels of the factor.  Here a brick starts at once and its inflow
// arrives WHILE it runs: x is pre-filled with a sentinel NaN and doubles as its own ready flag
// (as in the level plan), every row is written through to memory the moment it is complete, and
// a second wave of the workgroup -- the pump -- walks the brick's inflow list (it is sorted by
// the step that needs it), polls x in memory, drops the values into LDS and advances a counter.
// The compute wave (ONE wave: no barrier in the loop) reads the counter and its dependencies in
// one LDS round trip per step and repeats the step only if the counter is short.  A brick then
// trails its neighbour by one brick edge, not by a whole brick: critical path ~ levels of the
// factor + bricks on the chain x memory latency.  x and b must not alias (x carries the flags).
constexpr unsigned long long sentinel_bits = 0x7ff8dead0badbeefull;
constexpr unsigned long long poison_bits = 0x7ff8dead0badf00dull;  // a NaN that is not the sentinel

__global__ __launch_bounds__(256) void trs_brick_prepare_kernel(int64_t n, double* __restrict__ x, int64_t x_stride)
{
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
        reinterpret_cast<unsigned long long*>(x)[i * x_stride] = sentinel_bits;
    }
}

template <int K, bool Unit, bool Stamps = false>
__global__ __launch_bounds__(128) void trs_brick_pipelined_kernel(
    brick_header* hdr, const int32_t* __restrict__ perm, const double* __restrict__ diag,
    const double* __restrict__ rdiag, const int32_t* __restrict__ cols, const double* __restrict__ pvals,
    const int32_t* __restrict__ brick_row_begin, const int32_t* __restrict__ brick_step_ptr,
    const int32_t* __restrict__ step_begin, const int32_t* __restrict__ brick_ext_begin,
    const int32_t* __restrict__ ext_col, const int32_t* __restrict__ ext_row_off, int32_t n,
    const double* __restrict__ b, int64_t b_stride, double* x, int64_t x_stride, long long max_polls,
    int nap_max, const char* __restrict__ image, const int64_t* __restrict__ image_off,
    long long* __restrict__ stamps = nullptr, int stamp_brick = 0, double* arm = nullptr, double* rearm_b = nullptr)
{
    constexpr int T = 64;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned int s_ticket;
    __shared__ int s_inflow_ready;  // inflow entries [0, s_inflow_ready) are in LDS; -1: the pump gave up
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_ticket = atomicAdd(&hdr->ticket, 1u);
        s_inflow_ready = 0;
    }
    __syncthreads();
    const int bk = static_cast<int>(s_ticket);
    if (Stamps && stamp_brick < 0 && tid == 0) stamps[8 * bk] = wall_clock64();  // started
    const int r0 = brick_row_begin[bk], rows = brick_row_begin[bk + 1] - r0;
    const int e0 = brick_ext_begin[bk], inflow = brick_ext_begin[bk + 1] - e0;
    const int s0 = brick_step_ptr[bk], nsteps = brick_step_ptr[bk + 1] - s0;
    // LDS: x cells [rows | inflow | 0.0 | scratch] (doubles), then the brick's image (see
    // trs_brick_image_kernel): one RECORD per row (+ a spare one, the row of a lane without work:
    // values 0.0 on the zero cell, diagonal 1, result into the scratch cell, a store offset out of
    // range), {rows, inflow needed} per step, the rows my inflow comes from.
    // A record holds everything a step needs of its row behind ONE address computation, the LDS
    // byte addresses of its dependencies' cells ready-made: the compute wave issues every
    // instruction itself, 4+ cycles each.
    constexpr int rec_bytes = brick_record_bytes(K);
    constexpr int off_v = 16, off_ca = 16 + 8 * K, off_xa = 16 + 12 * K, off_row = 20 + 12 * K, off_span = 24 + 12 * K;
    constexpr unsigned int box_first = safe_exponent_lo;
    double* lx = lds;
    const int zero_cell = rows + inflow, scratch_cell = zero_cell + 1;
    char* lrec = reinterpret_cast<char*>(lds) + (sizeof(double) * (rows + inflow + 2) + 15) / 16 * 16;
    int2* lstep = reinterpret_cast<int2*>(lrec + static_cast<size_t>(rows + 1) * rec_bytes);
    int32_t* lext = reinterpret_cast<int32_t*>(lstep + nsteps + 4);
    // A. the image: a straight copy, global memory -> LDS without a register in between
    //    (global_load_lds_dwordx4: 1 KiB per wave and instruction, all of a brick's ~40 per wave in
    //    flight together).  Gathering the brick from the plan's arrays here took 15 us per brick --
    //    dependent loads, one row per lane and round trip -- and copying it through registers, eight
    //    loads in flight per lane, 18 us: brick START-UP, not the levels, was the pace of the solve.
    {
        // the right-hand side is a gather behind the permutation (two round trips): its first half goes
        // out before the image, its second half while the image is on its way (loads return in order)
        constexpr int gather_max = 16;  // rows <= 128 * 16: every brick that fits LDS
        int my_row[gather_max];
#pragma unroll
        for (int u = 0; u < gather_max; ++u) {
            const int i = tid + 128 * u;
            my_row[u] = i < rows ? perm[r0 + i] : 0;
        }
        const char* src = image + image_off[bk];
        const int bytes = static_cast<int>(image_off[bk + 1] - image_off[bk]);  // a multiple of 16
        const int lane16 = (tid & 63) * 16;
        for (int o = (tid >> 6) * 1024; o < bytes; o += 2048) {
            if (o + lane16 < bytes) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + o + lane16),
                    (__attribute__((address_space(3))) void*)(lrec + o), 16, 0, 0);
            }
        }
        double my_b[gather_max];
#pragma unroll
        for (int u = 0; u < gather_max; ++u) my_b[u] = tid + 128 * u < rows ? b[my_row[u] * b_stride] : 0.0;
#pragma unroll
        for (int u = 0; u < gather_max; ++u) {
            if (tid + 128 * u < rows) lx[tid + 128 * u] = my_b[u];
        }
        if (tid == 0) {
            lx[zero_cell] = 0.0;
            lx[scratch_cell] = 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the image has landed
        // Chained solves (Ilu = L^-1 then U^-1, precond.hip): instead of a launch that pre-fills the NEXT solve's
        // output with the sentinel, every brick arms its own rows of it here -- `arm` (the lower solve arms the
        // upper solve's output) -- and, having read its right-hand side, re-arms those rows for the solve that
        // will write them next -- `rearm_b` (the upper solve re-arms the intermediate vector for the next apply).
        // Behind the wait above: the loads of b have returned before their cells are overwritten.
        if (arm != nullptr || rearm_b != nullptr) {
#pragma unroll
            for (int u = 0; u < gather_max; ++u) {
                if (tid + 128 * u < rows) {
                    if (arm != nullptr) reinterpret_cast<unsigned long long*>(arm)[my_row[u] * x_stride] = sentinel_bits;
                    if (rearm_b != nullptr) reinterpret_cast<unsigned long long*>(rearm_b)[my_row[u] * b_stride] = sentinel_bits;
                }
            }
        }
    }
    __syncthreads();
    if (x_stride != 1) {  // the image holds row * 8: the store offsets of a strided x
        for (int i = tid; i < rows; i += 128) {
            uint32_t* off = reinterpret_cast<uint32_t*>(lrec + i * rec_bytes + off_row);
            *off = static_cast<uint32_t>(*off * x_stride);
        }
        __syncthreads();
    }
    if (tid >= T) {
        // ---- the pump: a window of W x 64 consecutive inflow entries, W per lane, all polled together
        //      (a round trip to memory is ~1 us and a 10^3 brick has 300 entries: 64 per round trip made
        //      the pump, not the arithmetic, the brick's pace).  The counter is the window's READY
        //      PREFIX -- an entry that is late does not hold back the ones a step needs first -- and the
        //      window slides by 64 as soon as its first 64 are in.
        constexpr int W = 4;
        const int lane = tid - T;
        const unsigned long long* xb = reinterpret_cast<const unsigned long long*>(x);
        auto poll = [&](int j) {
            return j < inflow ? __hip_atomic_load(xb + static_cast<int64_t>(lext[j]) * x_stride, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT)
                              : 0ull;  // behind the list: "ready"
        };
        bool gave_up = false;
        int base = 0, published = 0;  // entries [0, base + published) are in LDS
        long long polls = 0;
        // until the brick's first entry is in, only the first 16 entries of the window are polled: most resident
        // bricks are far behind the front, and hundreds of pumps gathering 256 lines each raise the hand-off
        // latency of the few that matter (tools/poll_probe.hip: 1.3 -> 3 us round trip)
        constexpr int cold_lanes = 16;  // 1 .. 64 measured: +-2 % (profiles/r02_trs_bricks.log)
        unsigned long long v[W];
        v[0] = lane < cold_lanes ? poll(lane) : (lane < inflow ? sentinel_bits : 0ull);
#pragma unroll
        for (int k = 1; k < W; ++k) v[k] = T * k + lane < inflow ? sentinel_bits : 0ull;
        bool hot = false;
        int nap = 1;
        while (base + published < inflow) {
            int prefix = W * T;
#pragma unroll
            for (int k = W - 1; k >= 0; --k) {
                const unsigned long long late = __ballot(v[k] == sentinel_bits);
                if (late != 0ull) prefix = T * k + __builtin_ctzll(late);
            }
            if (prefix > published) {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const int w = T * k + lane;  // my entry of this quarter, counted from `base`
                    if (w >= published && w < prefix && base + w < inflow) {
                        lx[rows + base + w] = __longlong_as_double(static_cast<long long>(v[k]));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the values before the counter
                if (lane == 0) {
                    __hip_atomic_store(&s_inflow_ready, min(base + prefix, inflow), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (Stamps && stamp_brick < 0 && base == 0 && published == 0) stamps[8 * bk + 4] = wall_clock64();  // first inflow published
                }
                published = prefix;
                nap = 1;
                hot = true;
            }
            if (base + published >= inflow) break;
            if (published >= T) {  // slide: every quarter moves down, the last one takes new entries
#pragma unroll
                for (int k = 0; k + 1 < W; ++k) v[k] = v[k + 1];
                base += T;
                published -= T;
                v[W - 1] = poll(base + T * (W - 1) + lane);
                continue;
            }
            if (++polls > max_polls) {
                gave_up = true;
                break;
            }
            for (int k = 0; k < nap; ++k) __builtin_amdgcn_s_sleep(2);
            nap = min(nap + 1, nap_max);  // a brick far behind the front backs off (8: to ~0.5 us)
#pragma unroll
            for (int k = 0; k < W; ++k) {
                if (v[k] == sentinel_bits && (hot || (k == 0 && lane < cold_lanes))) v[k] = poll(base + T * k + lane);
            }
        }
        if (gave_up && lane == 0) {
            __hip_atomic_store(&s_inflow_ready, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            atomicExch(&hdr->overrun, 1u);
        }
    } else {
        // ---- the compute wave ----
        // What a step costs is the chain result -> dependants -> result: an LDS turnaround (~60
        // cycles), the subtractions and the division's tail (~9 cycles per dependent f64
        // instruction) -- plus every instruction the wave issues at all (4+ cycles each) and
        // every TAKEN branch (~20 cycles; tools/latency_probe.hip, profiles/r02_trs_bricks.md).  So
        // the loop is straight-line code over ready-made addresses: a lane without a row works on
        // the spare record, only its two stores are masked, and the rare cases (inflow not in yet,
        // an operand outside the division's fast box, a NaN that looks like the sentinel) sit
        // behind wave-uniform branches that are not taken.
        struct step_data {
            double d, r;
            double v[K];
            int ca[K];          // LDS byte addresses of the dependencies' cells
            int xa;             // ... of my own cell (the spare record: the scratch cell)
            unsigned int off;   // byte offset of my row in x (the spare record: out of range)
            unsigned int span;  // see box_span
            int need;           // inflow entries this step needs
        };
        const char* lxb = reinterpret_cast<const char*>(lx);
        int begin = 0;  // first row of the step to fetch next
        auto fetch = [&](int2 st, step_data& sd) {
            const char* rec = lrec + __umul24(tid < st.x ? begin + tid : rows, rec_bytes);
            begin += st.x;
            sd.d = *reinterpret_cast<const double*>(rec);
            sd.r = *reinterpret_cast<const double*>(rec + 8);
#pragma unroll
            for (int e = 0; e < K; ++e) {
                sd.v[e] = *reinterpret_cast<const double*>(rec + off_v + 8 * e);
                sd.ca[e] = *reinterpret_cast<const int32_t*>(rec + off_ca + 4 * e);
            }
            sd.xa = *reinterpret_cast<const int32_t*>(rec + off_xa);
            sd.off = *reinterpret_cast<const uint32_t*>(rec + off_row);
            sd.span = *reinterpret_cast<const uint32_t*>(rec + off_span);
            sd.need = st.y;
        };
        bool poisoned = false;
        unsigned int healthy = ~0u;  // 0 once the brick has given up: every step takes the careful path
        // x as a buffer: a store behind its end is dropped, so the lanes without a row need no branch
        const __amdgpu_buffer_rsrc_t xbuf = __builtin_amdgcn_make_buffer_rsrc(
            x, 0, static_cast<int>(((static_cast<int64_t>(n) - 1) * x_stride + 1) * 8), 0x00020000);
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        // A step, measured piece by piece (tools/step_probe.hip, shader clocks): the chain itself -- LDS
        // turnaround, K multiply-subtracts, the division's tail -- 140; the next step's record +55; and
        // EVERY conditional branch, taken or not, +35, a v_cmp_f64 + s_or chain more.  So: the step
        // computes on whatever the inflow cells hold, then asks ONE question -- is the inflow counter
        // short, is an operand outside the division's fast box (integer test on the exponent), has
        // the brick given up -- and only the (not taken) branch behind it redoes the step carefully.
        auto step = [&](int s, const step_data& cur, step_data& nxt, int2 st1, int2& st2) {
            double xd[K];
            // counter first, then the values: LDS serves a wave in order, so values read behind a
            // sufficient counter are the pump's
            int ready = __hip_atomic_load(&s_inflow_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int e = 0; e < K; ++e) xd[e] = *reinterpret_cast<const double*>(lxb + cur.ca[e]);
            const double rhs = *reinterpret_cast<const double*>(lxb + cur.xa);
            st2 = lstep[s + 2];
            fetch(st1, nxt);
            double sum = rhs;
#pragma unroll
            for (int e = 0; e < K; ++e) sum -= cur.v[e] * xd[e];
            double xr = sum;
            if (!Unit) {
                // the division's tail on the prepared reciprocal (see divide_by_row_diagonal)
                const double q = sum * cur.r;
                const double rem = __builtin_fma(-cur.d, q, sum);
                xr = __builtin_fma(rem, cur.r, q);
            }
            // all ones if the inflow is in and the brick has not given up, else 0 (masks, not && / ?: --
            // those come back as branches)
            // (the quotient is pinned in front of the question: left alone, the compiler sinks it behind the
            // branch, into an else of its own with a second branch)
            asm volatile("" : "+v"(xr));
            const unsigned int fine = static_cast<unsigned int>((cur.need - 1 - __builtin_amdgcn_readfirstlane(ready)) >> 31) & healthy;
            const unsigned int exponent = (static_cast<unsigned int>(__double2hiint(sum)) >> 20) & 0x7ffu;
            if (__builtin_expect(__any(exponent - box_first >= (cur.span & fine)), 0)) {
                // carefully: wait for the inflow, read again, divide where the fast box does not hold
                long long spins = 0;
                while (ready < cur.need && !poisoned) {
                    if (ready < 0 || ++spins > max_polls) {
                        poisoned = true;
                        healthy = 0u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    ready = __hip_atomic_load(&s_inflow_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                sum = rhs;
#pragma unroll
                for (int e = 0; e < K; ++e) sum -= cur.v[e] * *reinterpret_cast<const double*>(lxb + cur.ca[e]);
                xr = Unit ? sum : divide_by_row_diagonal(sum, cur.d, cur.r);
                if (poisoned || __double_as_longlong(xr) == static_cast<long long>(sentinel_bits)) {
                    xr = __longlong_as_double(static_cast<long long>(poison_bits));  // a result must not look unfinished
                }
            }
            *reinterpret_cast<double*>(const_cast<char*>(lxb) + cur.xa) = xr;
            u32x2 out;
            out.x = static_cast<unsigned int>(__double2loint(xr));
            out.y = static_cast<unsigned int>(__double2hiint(xr));
            __builtin_amdgcn_raw_buffer_store_b64(out, xbuf, cur.off, 0, 16);  // 16 = sc1: write-through, agent scope
        };
        step_data even, odd;
        int2 sa = lstep[1];  // {rows, inflow needed} of the step after the current one
        fetch(lstep[0], even);
        if (Stamps && stamp_brick < 0 && tid == 0) stamps[8 * bk + 1] = wall_clock64();  // in LDS
        for (int s = 0; s < nsteps; s += 2) {  // an odd count runs one empty step
            int2 sb, sc;
            if (Stamps && bk == stamp_brick && tid == 0 && s < 1022) stamps[1 + s / 2] = __builtin_readcyclecounter();
            step(s, even, odd, sa, sb);
            if (Stamps && stamp_brick < 0 && tid == 0 && s == 0) stamps[8 * bk + 2] = wall_clock64();  // first step done
            step(s + 1, odd, even, sb, sc);
            if (Stamps && stamp_brick < 0 && tid == 0 && s == 8) stamps[8 * bk + 5] = wall_clock64();  // steps 0..9 done
            sa = sc;
        }
        if (Stamps && stamp_brick < 0 && tid == 0) stamps[8 * bk + 3] = wall_clock64();  // last step done
        // the compute wave may run out of patience before the pump does (it spins faster): poison in x
        // must never leave without the flag
        if (poisoned && tid == 0) atomicExch(&hdr->overrun, 1u);
    }
    if (Stamps && bk == stamp_brick && tid == 0) stamps[0] = nsteps;
    __syncthreads();
    if (tid == 0) {
        const unsigned int before = atomicAdd(&hdr->finished, 1u);
        if (before + 1 == static_cast<unsigned int>(hdr->nbricks)) {
            hdr->finished = 0;
            hdr->ticket = 0;
        }
    }
}

template <typename T>
int upload(hipStream_t stream, char* plan, size_t off, const std::vector<T>& v)
{
    if (v.empty()) return 0;
    return static_cast<int>(hipMemcpyAsync(plan + off, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, stream));
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

namespace {

int create_from_host(int64_t n, std::vector<int32_t>& rp, std::vector<int32_t>& ci, int lower, int64_t brick_rows,
                     int threads, int mode, gkomi_trs_bricks** out)
{
    const int64_t nnz = rp[n];
    for (int64_t i = 0; i < n; ++i) {
        if (rp[i + 1] < rp[i]) return GKOMI_EINVAL;
    }
    if (nnz < 0 || rp[0] != 0 || static_cast<int64_t>(ci.size()) < nnz) return GKOMI_EINVAL;
    gkomi_trs_bricks* h = new (std::nothrow) gkomi_trs_bricks;
    if (h == nullptr) return GKOMI_EINVAL;
    h->n = n;
    h->lower = lower ? 1 : 0;
    const int err = analyse(*h, rp, ci, brick_rows, threads, mode);
    if (err != GKOMI_SUCCESS) {
        delete h;
        return err;
    }
    *out = h;
    return GKOMI_SUCCESS;
}

bool create_args_ok(int64_t n, int threads, int mode, gkomi_trs_bricks** out, int* err)
{
    if (out == nullptr) {
        *err = GKOMI_EINVAL;
        return false;
    }
    *out = nullptr;
    if (n < 0 || (threads != 0 && threads != 64 && threads != 128 && threads != 256) || mode < 0 || mode > 2) {
        *err = GKOMI_EINVAL;
        return false;
    }
    if (n < 2 || n > INT32_MAX - 1024) {
        *err = GKOMI_ENOTSUPPORTED;
        return false;
    }
    return true;
}

}  // namespace

extern "C" int gkomi_trs_bricks_create_i32(gkomi_stream_t s, int64_t n, const int32_t* row_ptrs,
                                           const int32_t* col_idxs, int lower, int64_t brick_rows, int threads,
                                           int mode, gkomi_trs_bricks** out)
{
    int err = 0;
    if (!create_args_ok(n, threads, mode, out, &err)) return err;
    if (row_ptrs == nullptr) return GKOMI_EINVAL;
    hipStream_t stream = to_stream(s);
    static const bool on_host = [] {
        const char* e = getenv("GKOMI_TRS_ANALYSIS");  // =host: round 2's analysis on host threads
        return e != nullptr && e[0] == 'h';
    }();
    if (!on_host) {
        // the analysis on the device: the pattern never leaves HBM
        int32_t nnz32 = 0;
        err = static_cast<int>(hipMemcpyAsync(&nnz32, row_ptrs + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        if (!err) err = static_cast<int>(hipStreamSynchronize(stream));
        if (err) return err;
        if (nnz32 < 0 || (nnz32 > 0 && col_idxs == nullptr)) return GKOMI_EINVAL;
        gkomi_trs_bricks* h = new (std::nothrow) gkomi_trs_bricks;
        if (h == nullptr) return GKOMI_EINVAL;
        h->n = n;
        h->lower = lower ? 1 : 0;
        err = analyse_device(*h, stream, row_ptrs, col_idxs, brick_rows, threads, mode);
        if (err == GKOMI_SUCCESS) {
            *out = h;
            return GKOMI_SUCCESS;
        }
        // Not a lexicographic numbering, but rows of at most three dependencies: it may be a box grid numbered in
        // patches / along a curve -- the host analysis reads the coordinates off the dependency graph
        // (recover_grid_coordinates; the pattern comes to the host for it: set-up, tens of milliseconds per million rows)
        const int failed_width = err == GKOMI_ENOTSUPPORTED ? h->geometry_failed_width : 0;
        delete h;
        if (failed_width < 1 || failed_width > max_width) return err;
        if (failed_width > 3) {
            // no grid to recover (more than one dependency per dimension): the host analysis is worth the copy of the
            // pattern only for a THIN factor (pieces of the level order as bricks) -- ask the level analysis how many
            // levels there are before anything leaves the device
            const size_t bytes = gkomi_trs_symbolic_workspace_bytes(n);
            void* ws = nullptr;
            if (hipMalloc(&ws, bytes) != hipSuccess) {
                (void)hipGetLastError();
                return GKOMI_ENOTSUPPORTED;
            }
            int64_t sym[4] = {};
            const int serr = gkomi_trs_analyse_symbolic_i32(s, n, row_ptrs, col_idxs, lower, ws, bytes, sym);
            (void)hipFree(ws);
            if (serr != GKOMI_SUCCESS || sym[2] < 2 || n > 64 * sym[2]) return GKOMI_ENOTSUPPORTED;
        }
    }
    std::vector<int32_t> rp(static_cast<size_t>(n) + 1);
    err = static_cast<int>(hipMemcpyAsync(rp.data(), row_ptrs, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    if (err) return err;
    const int64_t nnz = rp[n];
    if (nnz < 0) return GKOMI_EINVAL;
    std::vector<int32_t> ci(static_cast<size_t>(nnz > 0 ? nnz : 1));
    if (nnz > 0) {
        if (col_idxs == nullptr) return GKOMI_EINVAL;
        err = static_cast<int>(hipMemcpyAsync(ci.data(), col_idxs, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, stream));
        if (err) return err;
        err = static_cast<int>(hipStreamSynchronize(stream));
        if (err) return err;
    }
    return create_from_host(n, rp, ci, lower, brick_rows, threads, mode, out);
}

extern "C" int64_t gkomi_trs_bricks_levels_estimate(const gkomi_trs_bricks* h) { return h != nullptr ? h->levels_estimate : 0; }

extern "C" int gkomi_trs_bricks_create_host_i32(int64_t n, const int32_t* host_row_ptrs,
                                                const int32_t* host_col_idxs, int lower, int64_t brick_rows,
                                                int threads, int mode, gkomi_trs_bricks** out)
{
    int err = 0;
    if (!create_args_ok(n, threads, mode, out, &err)) return err;
    if (host_row_ptrs == nullptr) return GKOMI_EINVAL;
    std::vector<int32_t> rp(host_row_ptrs, host_row_ptrs + n + 1);
    const int64_t nnz = rp[n];
    if (nnz < 0 || (nnz > 0 && host_col_idxs == nullptr)) return GKOMI_EINVAL;
    std::vector<int32_t> ci(host_col_idxs, host_col_idxs + (nnz > 0 ? nnz : 0));
    return create_from_host(n, rp, ci, lower, brick_rows, threads, mode, out);
}

extern "C" int gkomi_trs_bricks_host_array(const gkomi_trs_bricks* h, int which, const int32_t** data,
                                           int64_t* count)
{
    if (h == nullptr || data == nullptr || count == nullptr) return GKOMI_EINVAL;
    if (!h->host_arrays_valid) {  // analysed on the device: the arrays come to the host on first request
        const int err = fetch_host_arrays(*const_cast<gkomi_trs_bricks*>(h));
        if (err) return err;
    }
    const std::vector<int32_t>* v = nullptr;
    switch (which) {
    case 0: v = &h->perm; break;
    case 1: v = &h->brick_row_begin; break;
    case 2: v = &h->brick_step_ptr; break;
    case 3: v = &h->step_begin; break;
    case 4: v = &h->brick_ext_begin; break;
    case 5: v = &h->ext_col; break;
    case 6: v = &h->pred_ptr; break;
    case 7: v = &h->pred_idx; break;
    case 8: v = &h->row_rank; break;
    case 9: v = &h->inv_local; break;
    case 10: v = &h->ext_row_off; break;
    default: return GKOMI_EINVAL;
    }
    *data = v->data();
    *count = static_cast<int64_t>(v->size());
    return GKOMI_SUCCESS;
}

extern "C" void gkomi_trs_bricks_destroy(gkomi_trs_bricks* h) { delete h; }

extern "C" size_t gkomi_trs_bricks_plan_bytes(const gkomi_trs_bricks* h)
{
    return h == nullptr ? 0 : make_layout(*h).total;
}

extern "C" int gkomi_trs_bricks_info(const gkomi_trs_bricks* h, int64_t* out)
{
    if (h == nullptr || out == nullptr) return GKOMI_EINVAL;
    out[0] = h->nbricks;
    out[1] = h->coarse_levels;
    out[2] = h->nsteps;
    out[3] = h->critical_steps;
    out[4] = h->lds_bytes_max;
    out[5] = h->width;
    out[6] = h->threads;
    out[7] = h->mode;
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_trs_bricks_numeric_f64_i32(gkomi_stream_t s, gkomi_trs_bricks* h, const int32_t* row_ptrs,
                                                const int32_t* col_idxs, const double* vals, void* plan,
                                                size_t plan_bytes)
{
    if (h == nullptr) return GKOMI_EINVAL;
    const brick_layout l = make_layout(*h);
    if (plan == nullptr || plan_bytes < l.total) return GKOMI_EWORKSPACE;
    hipStream_t stream = to_stream(s);
    char* p = static_cast<char*>(plan);
    brick_header hd{};
    int err = 0;
    bool index_arrays_there = false;
    if (h->uploaded_to == plan && h->token != 0) {
        err = static_cast<int>(hipMemcpyAsync(&hd, plan, sizeof(hd), hipMemcpyDeviceToHost, stream));
        if (!err) err = static_cast<int>(hipStreamSynchronize(stream));
        if (err) return err;
        index_arrays_there = hd.token == h->token && hd.n == h->n && hd.nbricks == h->nbricks;
    }
    if (!index_arrays_there) {
        static std::atomic<uint64_t> counter{1};
        h->token = ((reinterpret_cast<uint64_t>(h) << 16) ^ (counter.fetch_add(1) * 0x9e3779b97f4a7c15ull)) | 1ull;
    }
    hd = brick_header{};
    hd.n = h->n; hd.nbricks = h->nbricks; hd.ticket = 0; hd.finished = 0; hd.overrun = 0; hd.lower = h->lower;
    hd.token = h->token;
    err = static_cast<int>(hipMemcpyAsync(plan, &hd, sizeof(hd), hipMemcpyHostToDevice, stream));
    if (err) return err;
    if (!index_arrays_there && h->dev_index != nullptr) {
        // analysed on the device: the index arrays move device to device
        auto copy = [&](size_t to, size_t from, int64_t count) {
            return count > 0 ? static_cast<int>(hipMemcpyAsync(p + to, h->dev_index + from, sizeof(int32_t) * count,
                                                               hipMemcpyDeviceToDevice, stream))
                             : 0;
        };
        err = copy(l.perm, h->dev_perm, h->n);
        if (!err) err = copy(l.brick_row_begin, h->dev_brick_row_begin, h->nbricks + 1);
        if (!err) err = copy(l.brick_step_ptr, h->dev_brick_step_ptr, h->nbricks + 1);
        if (!err) err = copy(l.step_begin, h->dev_step_begin, h->n_step_begin);
        if (!err) err = copy(l.brick_ext_begin, h->dev_brick_ext_begin, h->nbricks + 1);
        if (!err) err = copy(l.ext_col, h->dev_ext_col, h->n_ext_col);
        if (!err) err = copy(l.row_rank, h->dev_row_rank, h->n);
        if (!err) err = copy(l.inv_local, h->dev_inv_local, h->n);
        if (!err) err = copy(l.ext_row_off, h->dev_ext_row_off, h->n);
        if (!err) err = upload(stream, p, l.pred_ptr, h->pred_ptr);
        if (!err) err = upload(stream, p, l.pred_idx, h->pred_idx);
        if (!err) err = upload(stream, p, l.image_off, h->image_off);
        if (err) return err;
        h->uploaded_to = plan;
    } else if (!index_arrays_there) {
        err = upload(stream, p, l.perm, h->perm);
        if (!err) err = upload(stream, p, l.brick_row_begin, h->brick_row_begin);
        if (!err) err = upload(stream, p, l.brick_step_ptr, h->brick_step_ptr);
        if (!err) err = upload(stream, p, l.step_begin, h->step_begin);
        if (!err) err = upload(stream, p, l.brick_ext_begin, h->brick_ext_begin);
        if (!err) err = upload(stream, p, l.ext_col, h->ext_col);
        if (!err) err = upload(stream, p, l.pred_ptr, h->pred_ptr);
        if (!err) err = upload(stream, p, l.pred_idx, h->pred_idx);
        if (!err) err = upload(stream, p, l.row_rank, h->row_rank);
        if (!err) err = upload(stream, p, l.inv_local, h->inv_local);
        if (!err) err = upload(stream, p, l.ext_row_off, h->ext_row_off);
        if (!err) err = upload(stream, p, l.image_off, h->image_off);
        if (err) return err;
        h->uploaded_to = plan;
    }
    err = static_cast<int>(hipMemsetAsync(p + l.done, 0, sizeof(int32_t) * h->nbricks, stream));
    if (err) return err;
    h->epoch = 0;
    hipLaunchKernelGGL(trs_brick_fill_kernel, dim3(static_cast<unsigned>(ceildiv(h->n, 256))), dim3(256), 0, stream,
                       static_cast<int32_t>(h->n), h->width, h->lower != 0, row_ptrs, col_idxs, vals,
                       reinterpret_cast<const int32_t*>(p + l.perm), reinterpret_cast<const int32_t*>(p + l.row_rank),
                       reinterpret_cast<const int32_t*>(p + l.inv_local),
                       reinterpret_cast<const int32_t*>(p + l.ext_row_off),
                       reinterpret_cast<const int32_t*>(p + l.brick_row_begin),
                       reinterpret_cast<const int32_t*>(p + l.brick_ext_begin), reinterpret_cast<double*>(p + l.diag),
                       reinterpret_cast<double*>(p + l.rdiag), reinterpret_cast<int32_t*>(p + l.cols), reinterpret_cast<double*>(p + l.vals));
    err = check_launch();
    if (err) return err;
    if (h->mode == 2) {
#define GKOMI_BRICK_IMAGE(K)                                                                                          \
    hipLaunchKernelGGL((trs_brick_image_kernel<K>), dim3(static_cast<unsigned>(h->nbricks)), dim3(128), 0, stream,     \
                       static_cast<int32_t>(h->n), reinterpret_cast<const int32_t*>(p + l.perm),                      \
                       reinterpret_cast<const double*>(p + l.diag), reinterpret_cast<const double*>(p + l.rdiag),    \
                       reinterpret_cast<const int32_t*>(p + l.cols), reinterpret_cast<const double*>(p + l.vals),    \
                       reinterpret_cast<const int32_t*>(p + l.brick_row_begin),                                      \
                       reinterpret_cast<const int32_t*>(p + l.brick_step_ptr),                                       \
                       reinterpret_cast<const int32_t*>(p + l.step_begin),                                           \
                       reinterpret_cast<const int32_t*>(p + l.brick_ext_begin),                                      \
                       reinterpret_cast<const int32_t*>(p + l.ext_col),                                              \
                       reinterpret_cast<const int32_t*>(p + l.ext_row_off), p + l.image,                             \
                       reinterpret_cast<const int64_t*>(p + l.image_off))
        if (h->width <= 2) {
            GKOMI_BRICK_IMAGE(2);
        } else if (h->width <= 3) {
            GKOMI_BRICK_IMAGE(3);
        } else if (h->width <= 4) {
            GKOMI_BRICK_IMAGE(4);
        } else {
            GKOMI_BRICK_IMAGE(8);
        }
#undef GKOMI_BRICK_IMAGE
        err = check_launch();
        if (err) return err;
    }
    return static_cast<int>(hipStreamSynchronize(stream));  // the header and the host vectors were sources
}

namespace {

template <int T, int K, bool Unit>
int launch_solve(hipStream_t stream, gkomi_trs_bricks* h, char* p, const brick_layout& l, const double* b, int64_t b_stride, double* x, int64_t x_stride, long long max_polls)
{
    const size_t lds_bytes = static_cast<size_t>(h->lds_bytes_max > 0 ? h->lds_bytes_max : 16);  // bytes
    if (lds_bytes > 64 * 1024) {
        const int err = static_cast<int>(hipFuncSetAttribute(reinterpret_cast<const void*>(&trs_brick_solve_kernel<T, K, Unit>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             static_cast<int>(lds_bytes)));
        if (err) return err;
    }
    ++h->epoch;
    if (h->epoch == 0) h->epoch = 1;  // 0 = "never finished" (after 2^32 solves a stale flag could match: re-run numeric)
    hipLaunchKernelGGL((trs_brick_solve_kernel<T, K, Unit>), dim3(static_cast<unsigned>(h->nbricks)), dim3(T), lds_bytes,
                       stream, reinterpret_cast<brick_header*>(p), reinterpret_cast<const int32_t*>(p + l.perm),
                       reinterpret_cast<const double*>(p + l.diag), reinterpret_cast<const double*>(p + l.rdiag),
                       reinterpret_cast<const int32_t*>(p + l.cols),
                       reinterpret_cast<const double*>(p + l.vals),
                       reinterpret_cast<const int32_t*>(p + l.brick_row_begin),
                       reinterpret_cast<const int32_t*>(p + l.brick_step_ptr),
                       reinterpret_cast<const int32_t*>(p + l.step_begin),
                       reinterpret_cast<const int32_t*>(p + l.brick_ext_begin),
                       reinterpret_cast<const int32_t*>(p + l.ext_col), reinterpret_cast<const int32_t*>(p + l.pred_ptr),
                       reinterpret_cast<const int32_t*>(p + l.pred_idx), reinterpret_cast<unsigned int*>(p + l.done),
                       static_cast<int32_t>(h->n), b, b_stride, x, x_stride, h->epoch, max_polls);
    return check_launch();
}

template <int K, bool Unit>
int launch_pipelined(hipStream_t stream, gkomi_trs_bricks* h, char* p, const brick_layout& l, const double* b,
                     int64_t b_stride, double* x, int64_t x_stride, long long max_polls, bool x_is_armed = false,
                     double* arm = nullptr, double* rearm_b = nullptr)
{
    const size_t lds_bytes = static_cast<size_t>(h->lds_bytes_max > 0 ? h->lds_bytes_max : 16);
    if (lds_bytes > 64 * 1024) {
        const int err = static_cast<int>(hipFuncSetAttribute(reinterpret_cast<const void*>(&trs_brick_pipelined_kernel<K, Unit>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             static_cast<int>(lds_bytes)));
        if (err) return err;
    }
    const char* env_nap = getenv("GKOMI_TRS_BRICK_NAP");  // tuning knob (tools/trs_bricks_probe.py)
    const int nap_max = env_nap != nullptr && env_nap[0] != 0 ? std::max(1, atoi(env_nap)) : 8;
    if (!x_is_armed) {
        hipLaunchKernelGGL(trs_brick_prepare_kernel, dim3(grid_for(h->n, 256)), dim3(256), 0, stream, h->n, x, x_stride);
    }
    const char* env_stamps = getenv("GKOMI_TRS_BRICK_STAMPS");  // tools/trs_bricks_probe.py stamps
    if (!Unit && env_stamps != nullptr && env_stamps[0] != 0) {
        if (lds_bytes > 64 * 1024) {
            const int err = static_cast<int>(hipFuncSetAttribute(
                reinterpret_cast<const void*>(&trs_brick_pipelined_kernel<K, false, true>),
                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
            if (err) return err;
        }
        hipLaunchKernelGGL((trs_brick_pipelined_kernel<K, false, true>), dim3(static_cast<unsigned>(h->nbricks)), dim3(128),
                           lds_bytes, stream, reinterpret_cast<brick_header*>(p), reinterpret_cast<const int32_t*>(p + l.perm),
                           reinterpret_cast<const double*>(p + l.diag), reinterpret_cast<const double*>(p + l.rdiag),
                           reinterpret_cast<const int32_t*>(p + l.cols), reinterpret_cast<const double*>(p + l.vals),
                           reinterpret_cast<const int32_t*>(p + l.brick_row_begin),
                           reinterpret_cast<const int32_t*>(p + l.brick_step_ptr),
                           reinterpret_cast<const int32_t*>(p + l.step_begin),
                           reinterpret_cast<const int32_t*>(p + l.brick_ext_begin),
                           reinterpret_cast<const int32_t*>(p + l.ext_col),
                           reinterpret_cast<const int32_t*>(p + l.ext_row_off), static_cast<int32_t>(h->n), b, b_stride, x,
                           x_stride, max_polls, nap_max, p + l.image, reinterpret_cast<const int64_t*>(p + l.image_off),
                           reinterpret_cast<long long*>(p + l.stamps), atoi(env_stamps));
        return check_launch();
    }
    hipLaunchKernelGGL((trs_brick_pipelined_kernel<K, Unit>), dim3(static_cast<unsigned>(h->nbricks)), dim3(128), lds_bytes,
                       stream, reinterpret_cast<brick_header*>(p), reinterpret_cast<const int32_t*>(p + l.perm),
                       reinterpret_cast<const double*>(p + l.diag), reinterpret_cast<const double*>(p + l.rdiag),
                       reinterpret_cast<const int32_t*>(p + l.cols), reinterpret_cast<const double*>(p + l.vals),
                       reinterpret_cast<const int32_t*>(p + l.brick_row_begin),
                       reinterpret_cast<const int32_t*>(p + l.brick_step_ptr),
                       reinterpret_cast<const int32_t*>(p + l.step_begin),
                       reinterpret_cast<const int32_t*>(p + l.brick_ext_begin),
                       reinterpret_cast<const int32_t*>(p + l.ext_col),
                       reinterpret_cast<const int32_t*>(p + l.ext_row_off), static_cast<int32_t>(h->n), b, b_stride, x,
                       x_stride, max_polls, nap_max, p + l.image, reinterpret_cast<const int64_t*>(p + l.image_off),
                       static_cast<long long*>(nullptr), 0, arm, rearm_b);
    return check_launch();
}

template <bool Unit>
int launch_pipelined_width(hipStream_t stream, gkomi_trs_bricks* h, char* p, const brick_layout& l, const double* b,
                           int64_t b_stride, double* x, int64_t x_stride, long long max_polls, bool x_is_armed = false,
                           double* arm = nullptr, double* rearm_b = nullptr)
{
    if (h->width <= 2) return launch_pipelined<2, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls, x_is_armed, arm, rearm_b);
    if (h->width <= 3) return launch_pipelined<3, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls, x_is_armed, arm, rearm_b);
    if (h->width <= 4) return launch_pipelined<4, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls, x_is_armed, arm, rearm_b);
    return launch_pipelined<8, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls, x_is_armed, arm, rearm_b);
}

template <int T, bool Unit>
int launch_solve_width(hipStream_t stream, gkomi_trs_bricks* h, char* p, const brick_layout& l, const double* b,
                       int64_t b_stride, double* x, int64_t x_stride, long long max_polls)
{
    if (h->width <= 2) return launch_solve<T, 2, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
    if (h->width <= 3) return launch_solve<T, 3, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
    if (h->width <= 4) return launch_solve<T, 4, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
    return launch_solve<T, 8, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
}

template <bool Unit>
int launch_solve_threads(hipStream_t stream, gkomi_trs_bricks* h, char* p, const brick_layout& l, const double* b,
                         int64_t b_stride, double* x, int64_t x_stride, long long max_polls)
{
    switch (h->threads) {
    case 64: return launch_solve_width<64, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
    case 128: return launch_solve_width<128, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
    default: return launch_solve_width<256, Unit>(stream, h, p, l, b, b_stride, x, x_stride, max_polls);
    }
}

}  // namespace

extern "C" int gkomi_trs_bricks_solve_f64(gkomi_stream_t s, gkomi_trs_bricks* h, void* plan, int64_t nrhs,
                                          int unit_diag, const double* b, int64_t b_stride, double* x,
                                          int64_t x_stride)
{
    if (h == nullptr || plan == nullptr || nrhs < 0 || b_stride < nrhs || x_stride < nrhs) return GKOMI_EINVAL;
    if (h->uploaded_to != plan) return GKOMI_EINVAL;  // numeric phase first
    // the pipelined solve keeps its ready flags in x and 31-bit byte offsets into x in its records: a solve
    // in place, or on a very wide / very long x, takes the other kernel (every plan has what it needs)
    const bool pipelined = h->mode == 2 && x != b && h->n * x_stride * 8 <= INT32_MAX;
    if (nrhs == 0) return GKOMI_SUCCESS;
    const brick_layout l = make_layout(*h);
    char* p = static_cast<char*>(plan);
    hipStream_t stream = to_stream(s);
    const char* env_polls = getenv("GKOMI_TRS_MAX_POLLS");
    const long long max_polls = env_polls != nullptr && env_polls[0] != 0 ? atoll(env_polls) : default_max_polls;
    for (int64_t j = 0; j < nrhs; ++j) {
        int err;
        if (pipelined) {
            err = unit_diag != 0
                      ? launch_pipelined_width<true>(stream, h, p, l, b + j, b_stride, x + j, x_stride, max_polls)
                      : launch_pipelined_width<false>(stream, h, p, l, b + j, b_stride, x + j, x_stride, max_polls);
        } else {
            err = unit_diag != 0
                      ? launch_solve_threads<true>(stream, h, p, l, b + j, b_stride, x + j, x_stride, max_polls)
                      : launch_solve_threads<false>(stream, h, p, l, b + j, b_stride, x + j, x_stride, max_polls);
        }
        if (err) return err;
    }
    return GKOMI_SUCCESS;
}

// One solve of a CHAIN of brick solves on contiguous vectors (internal.hpp; precond.hip chains L^-1 and U^-1):
// x_is_armed = x already holds the sentinel in every row (no pre-fill launch); arm = a vector whose rows every
// brick arms for a later solve; rearm_b = b itself, re-armed row by row once it has been read.
// GKOMI_ENOTSUPPORTED when the plan is not the pipelined one (the caller takes the ordinary entry).
int gkomi::trs_bricks_solve_chained(gkomi_stream_t s, gkomi_trs_bricks* h, void* plan, int unit_diag, const double* b,
                                    double* x, bool x_is_armed, double* arm, double* rearm_b)
{
    if (h == nullptr || plan == nullptr) return GKOMI_EINVAL;
    if (h->uploaded_to != plan) return GKOMI_EINVAL;
    if (h->mode != 2 || x == b || h->n * 8 > INT32_MAX) return GKOMI_ENOTSUPPORTED;
    const brick_layout l = make_layout(*h);
    static const long long max_polls_default = default_max_polls;
    const char* env_polls = getenv("GKOMI_TRS_MAX_POLLS");
    const long long max_polls = env_polls != nullptr && env_polls[0] != 0 ? atoll(env_polls) : max_polls_default;
    char* p = static_cast<char*>(plan);
    return unit_diag != 0 ? launch_pipelined_width<true>(to_stream(s), h, p, l, b, 1, x, 1, max_polls, x_is_armed, arm, rearm_b)
                          : launch_pipelined_width<false>(to_stream(s), h, p, l, b, 1, x, 1, max_polls, x_is_armed, arm, rearm_b);
}

// tools only (not in gkomi.h): the stamps a solve under GKOMI_TRS_BRICK_STAMPS=<brick> left, 1024 entries:
// [0] = steps of that brick, [1 + k] = shader clock at the top of step 2 k
extern "C" int gkomi_trs_bricks_debug_stamps(gkomi_stream_t s, const gkomi_trs_bricks* h, const void* plan,
                                             long long* host_out)
{
    if (h == nullptr || plan == nullptr || host_out == nullptr) return GKOMI_EINVAL;
    const brick_layout l = make_layout(*h);
    hipStream_t stream = to_stream(s);
    // 8 KiB in the one-brick mode; 64 bytes per brick when every brick stamped (GKOMI_TRS_BRICK_STAMPS=-1)
    const char* env_stamps = getenv("GKOMI_TRS_BRICK_STAMPS");
    const size_t bytes = env_stamps != nullptr && atoi(env_stamps) < 0 ? 64 * static_cast<size_t>(h->nbricks) : 8 * 1024;
    int err = static_cast<int>(hipMemcpyAsync(host_out, static_cast<const char*>(plan) + l.stamps, bytes,
                                              hipMemcpyDeviceToHost, stream));
    if (err) return err;
    return static_cast<int>(hipStreamSynchronize(stream));
}

extern "C" int gkomi_trs_bricks_check_overrun(gkomi_stream_t s, const void* plan, int* host_flag)
{
    if (plan == nullptr || host_flag == nullptr) return GKOMI_EINVAL;
    brick_header hd{};
    hipStream_t stream = to_stream(s);
    int err = static_cast<int>(hipMemcpyAsync(&hd, plan, sizeof(hd), hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    *host_flag = static_cast<int>(hd.overrun);
    return err;
}
