// Persistent CG (one GPU, CSR, Identity preconditioner, one right-hand side):
// the whole solve of core/solver/cg.cpp:107-193 in ONE launch.  One workgroup of
// 1024 threads per CU owns a contiguous chunk of rows for the life of the
// solve and keeps its part of x, r, p (and q) in registers; per iteration the
// chip reads the matrix (once, nontemporal), writes the new search direction p
// (the only vector other workgroups need: their gathers) and exchanges two
// scalars -- 8 n + matrix bytes instead of the 11 n + matrix of the
// three-launch iteration of cg_solver.hip, and no launch gaps.
//
// The workgroups meet three times per iteration, without atomics and without
// a kernel boundary: every workgroup publishes its partial sum in its own slot
// (one 16-B agent-scope store of the value and a check word keyed by the number
// of the meeting, written through this XCD's L2), and thread t of every
// workgroup polls slot t until value and check word fit the meeting.  All workgroups add the
// same 256 partials in the same order: rho, p.q, the stopping decision and the
// iteration count are bit-identical everywhere, and run-to-run.
//   1. rho = r.r             -> criterion (Combined(Iteration, ResidualNorm),
//                               as K1 of cg_fused.hpp), p = r + (rho/prev) p
//   2. p written             -> q = A p (row-block SpMV through LDS tiles, the
//                               reference's order inside a row)
//   3. p.q                   -> x += (rho/pq) p, r -= (rho/pq) q
// Slots are double-buffered by the parity of the meeting: a workgroup can be
// at most one meeting ahead of the slowest one.
//
// Preconditions (the driver falls back to the three-launch iteration
// otherwise): n <= 1024 * 8 rows per workgroup, every workgroup resident at
// once (grid = number of CUs), CSR arrays 16-/8-B aligned.  Spins are
// bounded: a workgroup that waits longer than `max_polls` raises scal->status'
// sibling `overrun` and everybody leaves; the driver then recomputes r and
// finishes with the three-launch iteration.
#pragma once
#include "cg_fused.hpp"

namespace gkomi {
namespace {

constexpr int pcg_block = 1024;
constexpr int pcg_items = 6;                       // nonzeros per thread and tile
constexpr int pcg_tile = pcg_block * pcg_items;    // 6144 products = 48 KB of LDS
constexpr int pcg_max_rows_per_thread = 8;
constexpr int pcg_max_stride = 256;     // slots up to 4 KB apart (16-B units)
constexpr int pcg_default_stride = 16;  // 256 B

// One slot = 16 bytes: the value and the value's bits XOR a key derived from
// the number of the meeting.  The pair is written with ONE 16-B agent-scope
// store and read with ONE 16-B agent-scope load; a reader accepts it only when
// the two words fit the meeting it waits for, so a torn or stale pair (old value
// with new check word, or the reverse) is simply polled again -- no separate
// tag, no store -> acknowledge -> store chain on the writer's side, no second
// load on the reader's side.
struct pcg_slot {
    unsigned long long v_bits;
    unsigned long long check;
};

struct pcg_control {
    unsigned int overrun;  // a workgroup gave up waiting
    unsigned int pad_[15];
    // GKOMI_PCG_PROFILE builds: 10-ns ticks workgroup 0 spent in each phase of the iteration
    unsigned long long ticks[8];
    unsigned long long pad2_[16];
};

#ifdef GKOMI_PCG_PROFILE
#define PCG_STAMP(i)                                      \
    do {                                                  \
        const unsigned long long now_ = wall_clock64();   \
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->ticks[i] += now_ - stamp_; \
        stamp_ = now_;                                    \
    } while (0)
#else
#define PCG_STAMP(i)
#endif

typedef unsigned int pcg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long pcg_key(long long meeting)
{
    return static_cast<unsigned long long>(meeting) * 0x9e3779b97f4a7c15ull;  // never 0 for meeting >= 1
}

__device__ __forceinline__ void pcg_publish(pcg_slot* slot, double v, long long meeting)
{
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v));
    const unsigned long long chk = bits ^ pcg_key(meeting);
    pcg_u32x4 w;
    w.x = static_cast<unsigned int>(bits);
    w.y = static_cast<unsigned int>(bits >> 32);
    w.z = static_cast<unsigned int>(chk);
    w.w = static_cast<unsigned int>(chk >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(slot), "v"(w) : "memory");
}

// true when the slot carries the value of `meeting`
__device__ __forceinline__ bool pcg_fetch(const pcg_slot* slot, long long meeting, double* v)
{
    pcg_u32x4 w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(slot) : "memory");
    const unsigned long long bits = (static_cast<unsigned long long>(w.y) << 32) | w.x;
    const unsigned long long chk = (static_cast<unsigned long long>(w.w) << 32) | w.z;
    *v = __longlong_as_double(static_cast<long long>(bits));
    return (bits ^ chk) == pcg_key(meeting);
}

constexpr int pcg_copies = 32;  // copies of the total the workgroups read it from (8 readers each)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// every outstanding global load of the wave (vmcnt(0)) -- exactly what must NOT
// happen while a prefetch for the next step is in flight behind a meeting.
__device__ __forceinline__ void pcg_sync_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// smem: Block / 64 + 1 doubles.  The sum of v over the workgroup, identical in every thread
// (one value per wave through LDS, added in wave order).
template <int Block>
__device__ __forceinline__ double pcg_block_sum(double v, double* smem)
{
    v = wave_reduce_sum(v);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    pcg_sync_lds();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < Block / wave_size; ++w) total += smem[w];
    pcg_sync_lds();  // smem is free again
    return total;
}

// Every workgroup's `mine` (thread 0's argument), added in slot order by workgroup 0, which
// hands the total back through pcg_copies slots: two memory hand-offs per meeting, 256 + 255
// polls in flight instead of 256 x 256 on 256 cache lines (3.9 us per meeting when every
// workgroup gathered all partials itself).  The total is computed once: every workgroup
// continues with the same bits.  smem: Block / 64 + 1 doubles.
template <int Block>
__device__ __forceinline__ bool pcg_meet(pcg_slot* slots, int stride, int nap, int nwg, long long meeting,
                                         double mine, double* smem, pcg_control* ctl, long long max_polls,
                                         double* total)
{
    constexpr int nwaves = Block / wave_size;
    // per parity: nwg slots of partials, then pcg_copies slots of the total
    pcg_slot* bank = slots + (meeting & 1) * static_cast<int64_t>(nwg + pcg_copies) * stride;
    pcg_slot* back = bank + static_cast<int64_t>(nwg) * stride;
#ifdef GKOMI_PCG_PROFILE
    const unsigned long long m0_ = wall_clock64();
#endif
    if (threadIdx.x == 0) pcg_publish(bank + blockIdx.x * stride, mine, meeting);
    bool ok = true;
    auto wait_for = [&](const pcg_slot* slot, double* v) {
        long long polls = 0;
        while (!pcg_fetch(slot, meeting, v)) {
            for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
            if (++polls > max_polls ||
                ((polls & 255) == 0 &&
                 __hip_atomic_load(&ctl->overrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                *v = 0.0;
                return false;
            }
        }
        return true;
    };
    if (threadIdx.x == 0) smem[nwaves] = 1.0;  // "nobody gave up"
    pcg_sync_lds();
    double sum = 0.0;
    if (blockIdx.x == 0) {
        // nwg <= 1024: thread t holds workgroup t's partial; one sum per wave goes through LDS and
        // every thread adds them in wave order
        double part = 0.0;
        if (static_cast<int>(threadIdx.x) < nwg) ok = wait_for(bank + threadIdx.x * stride, &part);
#ifdef GKOMI_PCG_PROFILE
        if (threadIdx.x == 0) ctl->pad2_[0] += wall_clock64() - m0_;                 // own slot seen (store -> load round trip)
        if (threadIdx.x == nwg - 1) ctl->pad2_[1] += wall_clock64() - m0_;           // last workgroup's slot seen
#endif
        if (!ok) smem[nwaves] = 0.0;
        part = wave_reduce_sum(part);
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = part;
        pcg_sync_lds();
#pragma unroll
        for (int w = 0; w < nwaves; ++w) sum += smem[w];
        ok = smem[nwaves] != 0.0;
        if (ok && threadIdx.x < pcg_copies) pcg_publish(back + threadIdx.x * stride, sum, meeting);
#ifdef GKOMI_PCG_PROFILE
        if (threadIdx.x == 0) ctl->pad2_[2] += wall_clock64() - m0_;                 // total published
#endif
    } else {
        if (threadIdx.x == 0) {
            ok = wait_for(back + (blockIdx.x % pcg_copies) * stride, &sum);
#ifdef GKOMI_PCG_PROFILE
            if (blockIdx.x == 101) ctl->pad2_[3] += wall_clock64() - m0_;            // workgroup 101: total received
            if (blockIdx.x == 101) ctl->pad2_[4] += 1;
#endif
            smem[0] = sum;
            if (!ok) smem[nwaves] = 0.0;
        }
        pcg_sync_lds();
        sum = smem[0];
        ok = smem[nwaves] != 0.0;
    }
    pcg_sync_lds();  // smem is free again
    if (!ok) {
        if (threadIdx.x == 0) __hip_atomic_store(&ctl->overrun, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    *total = sum;
    return true;
}

// A meeting over NV values at once (the blocked Gram-Schmidt sweep of gmres.hip): workgroup w's values in the NV
// slots behind its own, the totals in the NV slots behind each copy.  lsum[0 .. NV): in = this workgroup's sums, out =
// the totals (the same bits in every workgroup: workgroup 0 adds in a fixed order, once).  lred: (Block / 64) * NV + 1
// doubles; lval: nwg * NV doubles of scratch (workgroup 0 only).  The polls of workgroup 0 are plain agent-scope loads of the two words of a slot -- all 2 NV in flight
// together; a pair that does not fit the meeting (stale, torn) is asked for again, as in pcg_fetch.
// Published: the caller has stored this workgroup's NV values into its slots of `meeting` itself, has set
// lred[(Block / 64) * NV] to 1.0 and has passed a workgroup barrier since (gmres.hip: the wave that adds up a value
// publishes it -- one barrier less per meeting); lsum is then output only.
template <int Block, int NV, bool Published = false, typename F>
__device__ __forceinline__ bool pcg_meet_values(pcg_slot* slots, int stride, int nap, int nwg, long long meeting,
                                                double* lsum, double* lred, double* lval, pcg_control* ctl,
                                                long long max_polls, F after_publish)
{
    static_assert(NV <= pcg_default_stride, "a workgroup's values share its slot line");
    static_assert(pcg_copies * NV <= Block, "one thread per copy and value");
    constexpr int nwaves = Block / wave_size;
    const int tid = threadIdx.x;
    pcg_slot* bank = slots + (meeting & 1) * static_cast<int64_t>(nwg + pcg_copies) * stride;
    pcg_slot* back = bank + static_cast<int64_t>(nwg) * stride;
    if (!Published && tid < NV) pcg_publish(bank + blockIdx.x * stride + tid, lsum[tid], meeting);
    after_publish();  // (the caller's loads for its next step: behind the store, so they do not hold it back)
    if (!Published) {
        if (tid == 0) lred[nwaves * NV] = 1.0;  // "nobody gave up"
        pcg_sync_lds();
    }
    const unsigned long long key = pcg_key(meeting);
    bool ok = true;
    if (blockIdx.x == 0) {
        if (Published) pcg_sync_lds();  // lval may be the buffer the caller's waves have just added their sums from
        // pair p = (workgroup p / NV, value p % NV); U pairs per thread in flight together, the values into lval
        constexpr int U = 5;
        const int npairs = nwg * NV;
        for (int base = 0; base < npairs && ok; base += U * Block) {
            unsigned int pending = 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * Block + tid < npairs) pending |= 1u << u;
            }
            long long polls = 0;
            while (pending != 0u) {
                unsigned long long bits[U], chk[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if ((pending >> u) & 1u) {
                        const int pr = base + u * Block + tid;
                        const pcg_slot* slot = bank + (pr / NV) * stride + pr % NV;
                        bits[u] = __hip_atomic_load(&slot->v_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        chk[u] = __hip_atomic_load(&slot->check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (((pending >> u) & 1u) && (bits[u] ^ chk[u]) == key) {
                        lval[base + u * Block + tid] = __longlong_as_double(static_cast<long long>(bits[u]));
                        pending &= ~(1u << u);
                    }
                }
                if (pending != 0u) {
                    for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
                    if (++polls > max_polls ||
                        ((polls & 255) == 0 &&
                         __hip_atomic_load(&ctl->overrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        ok = false;
                        break;
                    }
                }
            }
        }
        if (!ok) lred[nwaves * NV] = 0.0;
        pcg_sync_lds();
        ok = lred[nwaves * NV] != 0.0;
        // value v: workgroups lane, lane + 64, ... in order, then the wave's fixed tree
        if (ok) {
            for (int v = tid >> 6; v < NV; v += nwaves) {
                double t = 0.0;
                for (int g = tid & 63; g < nwg; g += wave_size) t += lval[g * NV + v];
                t = wave_reduce_sum(t);
                if ((tid & 63) == 0) lsum[v] = t;
            }
        }
        pcg_sync_lds();
        ok = lred[nwaves * NV] != 0.0;
        if (ok && tid < pcg_copies * NV) pcg_publish(back + (tid / NV) * stride + tid % NV, lsum[tid % NV], meeting);
    } else {
        if (tid < NV) {
            double total = 0.0;
            long long polls = 0;
            while (!pcg_fetch(back + (blockIdx.x % pcg_copies) * stride + tid, meeting, &total)) {
                for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
                if (++polls > max_polls ||
                    ((polls & 255) == 0 &&
                     __hip_atomic_load(&ctl->overrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    lred[nwaves * NV] = 0.0;
                    total = 0.0;
                    break;
                }
            }
            lsum[tid] = total;
        }
        pcg_sync_lds();
        ok = lred[nwaves * NV] != 0.0;
    }
    pcg_sync_lds();  // lred is free again, lsum holds the totals
    if (!ok) {
        if (tid == 0) __hip_atomic_store(&ctl->overrun, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}

// a meeting without a value: "everybody's p is in memory"
template <int Block>
__device__ __forceinline__ bool pcg_barrier(pcg_slot* slots, int stride, int nap, int nwg, long long meeting,
                                            double* smem, pcg_control* ctl, long long max_polls)
{
    // every wave's write-through stores are acknowledged before the workgroup's slot goes out
    __builtin_amdgcn_s_waitcnt(0);
    pcg_sync_lds();
    double unused;
    return pcg_meet<Block>(slots, stride, nap, nwg, meeting, 0.0, smem, ctl, max_polls, &unused);
}

// R = rows per thread (row of slot k: chunk start + k * Block + thread); Block = 1024 threads
// (128 registers each) or 512 (256 each: room for the matrix values of 8 rows).
// K > 0: no row has more than K nonzeros and the thread keeps its rows' values and
// column indices in registers for the whole solve (1M rows x 5 nonzeros = 60 MB in
// the 128 MB of register files): an iteration then reads no matrix at all, the
// SpMV is R*K gathers of p per thread and the row sums in storage order.  A row
// longer than K makes every workgroup leave before anything is changed (the
// driver falls back).  The rows come from CSR arrays or, ell_stored > 0, from ELL
// arrays.  K = 0: the matrix streams from memory through LDS tiles.
// XL: x lives in LDS instead of registers (7 nonzeros x 8 rows per thread leave no room for it)
template <int R, int K, int Block, bool XL = false>
__global__ __launch_bounds__(Block) void cg_persistent_kernel(
    int n, int chunk, const int32_t* __restrict__ row_ptrs, const int32_t* __restrict__ col_idxs,
    const double* __restrict__ vals, double* __restrict__ x, double* __restrict__ r, double* pbuf0,
    double* pbuf1, pcg_slot* slots, int stride, int nap, pcg_control* ctl, cg_scalars* scal,
    long long max_iters, double goal, long long max_polls, int ell_stored, int64_t ell_stride)
{
    typedef double nt_double2 __attribute__((ext_vector_type(2)));
    typedef int nt_int2 __attribute__((ext_vector_type(2)));
    constexpr int pairs = pcg_items / 2;
    __shared__ __attribute__((aligned(16))) double prod[K > 0 ? 2 : (Block * pcg_items)];
    // K > 0: byte offsets of the rows' columns (5 x 4 x 1024 x 4 B = 80 KB), thread-minor: no bank conflicts
    __shared__ unsigned int lcol[K > 0 ? K * R * Block : 1];
    __shared__ double lx[XL ? R * Block : 1];
    __shared__ double smem[Block / wave_size + 1];
    const int nwg = gridDim.x;
    const int tid = threadIdx.x;
    const int b0 = min(static_cast<int>(blockIdx.x) * chunk, n);
    const int b1 = min(b0 + chunk, n);
    // ell_stored > 0 (K > 0 only): col_idxs / vals are ELL arrays (entry e of row r at r + e * ell_stride,
    // column -1 = padding), row_ptrs is not read
    const bool ell = K > 0 && ell_stored > 0;
    const int nz0 = ell ? 0 : row_ptrs[b0];
    const int nz1 = ell ? 0 : row_ptrs[b1];
    const int nnz_total = ell ? 0 : row_ptrs[n];
    int ra[R], rb[R];
    double xr[XL ? 1 : R], rr[R], pr[R], qr[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = b0 + k * Block + tid;
        ra[k] = rb[k] = nz1;
        rr[k] = 0.0;
        if (XL) {
            lx[k * Block + tid] = 0.0;
        } else {
            xr[k] = 0.0;
        }
        pr[k] = 0.0;  // cg::initialize: p = 0
        if (row < b1) {
            if (!ell) {
                ra[k] = row_ptrs[row];
                rb[k] = row_ptrs[row + 1];
            }
            if (XL) {
                lx[k * Block + tid] = x[row];
            } else {
                xr[k] = x[row];
            }
            rr[k] = r[row];
        }
    }
    // K > 0: the rows of this thread, in registers
    constexpr int KK = K > 0 ? K : 1;
    double mv[R][KK];
    unsigned int present[R];  // bit e: entry e of the row exists (CSR: the first len; ELL: not padding)
    if (K > 0) {
        bool fits = n <= (1 << 28) && (!ell || ell_stored <= K);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int row = b0 + k * Block + tid;
            const int len = rb[k] - ra[k];
            fits &= ell || len <= K;
            present[k] = 0;
#pragma unroll
            for (int e = 0; e < KK; ++e) {
                bool have = false;
                double val = 0.0;
                int col = 0;
                if (ell) {
                    if (row < b1 && e < ell_stored) {
                        col = col_idxs[row + e * ell_stride];
                        val = vals[row + e * ell_stride];
                        have = col != -1;
                    }
                } else if (e < len) {
                    col = col_idxs[ra[k] + e];
                    val = vals[ra[k] + e];
                    have = true;
                }
                mv[k][e] = have ? val : 0.0;
                present[k] |= have ? (1u << e) : 0u;
                // byte offset of p(col): the gather is base (scalar) + one 32-bit register
                lcol[(e * R + k) * Block + tid] = have ? static_cast<unsigned int>(col) * 8u : 0u;
            }
        }
        // everybody learns whether every row fits before anything is changed: a meeting of its own
        double misfits = 0.0;
        const double mine = pcg_block_sum<Block>(fits ? 0.0 : 1.0, smem);
        if (!pcg_meet<Block>(slots, stride, nap, gridDim.x, 1, mine, smem, ctl, max_polls, &misfits)) return;
        if (misfits != 0.0) {
            if (blockIdx.x == 0 && tid == 0) {
                __hip_atomic_store(&ctl->overrun, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;  // x and r untouched: the driver runs the three-launch iteration
        }
    }
    const double orig = scal->orig_tau;
    double prev = scal->rho[1];  // 1.0 (reference cg::initialize)
    long long meeting = K > 0 ? 1 : 0;
    long long it = 0;
    uint8_t st = 0;
    double rho = 0.0, tau = 0.0;
    bool broken = false;
#ifdef GKOMI_PCG_PROFILE
    unsigned long long stamp_ = wall_clock64();
#endif
    while (true) {
        // Forget what this CU's L1 and this XCD's L2 know of the p buffer of this iteration: its
        // lines date from the gathers two iterations ago.  Nobody gathers between here and meeting
        // 2 (every workgroup has left the SpMV of the iteration before: it published p.q), so the
        // invalidate may run now, off the critical path, instead of between meeting 2 and the gathers.
        // (The last wave issues it: the waves that poll in meeting 1 would wait for it.)
        if (tid >= Block - wave_size) asm volatile("buffer_inv sc1" ::: "memory");
        // 1. rho = r.r (z = r), the criterion, the new search direction
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) acc += rr[k] * rr[k];
        const double mine = pcg_block_sum<Block>(acc, smem);
        if (!pcg_meet<Block>(slots, stride, nap, nwg, ++meeting, mine, smem, ctl, max_polls, &rho)) {
            broken = true;
            break;
        }
        PCG_STAMP(0);
        tau = sqrt(rho);
        // Combined: Iteration is asked first, then ResidualNorm
        if (it >= max_iters) {
            st = id_iteration | GKOMI_STATUS_FINALIZED;
        } else if (tau < goal * orig) {
            st = GKOMI_STATUS_CONVERGED | id_residual | GKOMI_STATUS_FINALIZED;
        }
        if (st) break;
        const bool restart = prev == 0.0;
        const double factor = restart ? 0.0 : rho / prev;
        double* pbuf = (it & 1) ? pbuf1 : pbuf0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int row = b0 + k * Block + tid;
            pr[k] = restart ? rr[k] : rr[k] + factor * pr[k];
            // written through to memory: the other XCDs gather it after the meeting
            if (row < b1) __hip_atomic_store(pbuf + row, pr[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // 2. everybody's p is in memory
        if (!pcg_barrier<Block>(slots, stride, nap, nwg, ++meeting, smem, ctl, max_polls)) {
            broken = true;
            break;
        }
        PCG_STAMP(1);
        PCG_STAMP(2);
#pragma unroll
        for (int k = 0; k < R; ++k) qr[k] = 0.0;
        if (K > 0) {
            // two rows at a time: 2 K gathers in flight per thread (the register budget has no
            // room for all R K of them next to the values of the matrix)
            constexpr int G = R >= 2 ? 2 : 1;
#pragma unroll
            for (int k0 = 0; k0 < R; k0 += G) {
                double pv[G][KK];
#pragma unroll
                for (int g = 0; g < G; ++g) {
#pragma unroll
                    for (int e = 0; e < KK; ++e) {
                        const unsigned int off = lcol[(e * R + k0 + g) * Block + tid];
                        pv[g][e] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(pbuf) + off);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    double acc_row = 0.0;
#pragma unroll
                    for (int e = 0; e < KK; ++e) {
                        // storage order; entries past the row's end are skipped, not added as zeros
                        acc_row = (present[k0 + g] >> e) & 1u ? acc_row + mv[k0 + g][e] * pv[g][e] : acc_row;
                    }
                    qr[k0 + g] = acc_row;
                }
            }
        } else {
            for (int t0 = nz0 & ~1; t0 < nz1; t0 += (Block * pcg_items)) {
                double2 v[pairs];
                int2 ci[pairs];
    #pragma unroll
                for (int u = 0; u < pairs; ++u) {
                    const int k = t0 + 2 * (tid + u * Block);
                    v[u] = make_double2(0.0, 0.0);
                    ci[u] = make_int2(0, 0);
                    if (k < nz1) {
                        if (k + 1 < nnz_total) {
                            const nt_double2 tv =
                                __builtin_nontemporal_load(reinterpret_cast<const nt_double2*>(vals + k));
                            const nt_int2 tc =
                                __builtin_nontemporal_load(reinterpret_cast<const nt_int2*>(col_idxs + k));
                            v[u] = make_double2(tv.x, tv.y);
                            ci[u] = make_int2(tc.x, tc.y);
                        } else {
                            v[u].x = vals[k];
                            ci[u].x = col_idxs[k];
                        }
                    }
                }
                double2 pv[pairs];
    #pragma unroll
                for (int u = 0; u < pairs; ++u) {
                    pv[u].x = pbuf[ci[u].x];
                    pv[u].y = pbuf[ci[u].y];
                }
    #pragma unroll
                for (int u = 0; u < pairs; ++u) {
                    *reinterpret_cast<double2*>(prod + 2 * (tid + u * Block)) =
                        make_double2(v[u].x * pv[u].x, v[u].y * pv[u].y);
                }
                __syncthreads();
                const int t1 = t0 + (Block * pcg_items);
    #pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int lo = max(ra[k], t0), hi = min(rb[k], t1);
                    for (int j = lo; j < hi; ++j) qr[k] += prod[j - t0];
                }
                __syncthreads();
            }
        }
        PCG_STAMP(3);
        // 3. p.q, the step
        acc = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) acc += pr[k] * qr[k];
        const double mine_pq = pcg_block_sum<Block>(acc, smem);
        double pq = 0.0;
        if (!pcg_meet<Block>(slots, stride, nap, nwg, ++meeting, mine_pq, smem, ctl, max_polls, &pq)) {
            broken = true;
            break;
        }
        PCG_STAMP(4);
        if (pq != 0.0) {
            const double alpha = rho / pq;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (XL) {
                    lx[k * Block + tid] += alpha * pr[k];
                } else {
                    xr[k] += alpha * pr[k];
                }
                rr[k] -= alpha * qr[k];
            }
        }
        prev = rho;
        ++it;
        PCG_STAMP(5);
    }
    // the state goes back to memory whatever happened (a broken meeting leaves a consistent
    // enough x: the driver recomputes r from it)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = b0 + k * Block + tid;
        if (row < b1) {
            x[row] = XL ? lx[k * Block + tid] : xr[k];
            r[row] = rr[k];
        }
    }
    if (blockIdx.x == 0 && tid == 0 && !broken) {
        scal->rho[it & 1] = rho;
        scal->rho[(it + 1) & 1] = prev;
        scal->tau = tau;
        scal->stop_iter = it;
        scal->status = st;
    }
}

__global__ void pcg_clear_kernel(pcg_slot* slots, int stride, int count, pcg_control* ctl)
{
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        slots[static_cast<int64_t>(i) * stride].v_bits = 0;
        slots[static_cast<int64_t>(i) * stride].check = 0;  // fits no meeting: pcg_key(m) != 0 for m >= 1
    }
    if (threadIdx.x == 0) ctl->overrun = 0;
    if (threadIdx.x < 8) ctl->ticks[threadIdx.x] = 0;
    if (threadIdx.x < 16) ctl->pad2_[threadIdx.x] = 0;
}

}  // namespace
}  // namespace gkomi
