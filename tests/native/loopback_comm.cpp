// TEST INFRASTRUCTURE (not shipped): a `gkomi_comm` whose ranks are the THREADS
// of one process sharing one GPU.  It lets the native distributed drivers
// (csrc/dist_cg.hip) run with world sizes > 1 on a one-GPU box: the halo plan,
// pack / unpack offsets, the reductions and the iteration control are exercised
// exactly as over RCCL; only the transport differs (device-to-device copies and
// a host barrier instead of xGMI).  Built by tests/native/Makefile.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdint.h>

#include <vector>

#include "../../include/gkomi.h"

namespace {

struct world {
    int size;
    pthread_barrier_t bar;
    std::vector<const char*> send;
    std::vector<const int64_t*> send_counts, send_offsets;
    std::vector<std::vector<double>> red;
};

struct rank_ctx {
    world* w;
    int rank;
};

int allreduce(void* self, gkomi_stream_t s, double* buf, int64_t count)
{
    rank_ctx* c = static_cast<rank_ctx*>(self);
    world* w = c->w;
    hipStream_t stream = static_cast<hipStream_t>(s);
    std::vector<double>& mine = w->red[c->rank];
    mine.resize(count);
    int err = static_cast<int>(hipMemcpyAsync(mine.data(), buf, sizeof(double) * count, hipMemcpyDeviceToHost, stream));
    if (err) return err;
    err = static_cast<int>(hipStreamSynchronize(stream));
    pthread_barrier_wait(&w->bar);
    std::vector<double> total(count, 0.0);
    for (int r = 0; r < w->size; ++r) {  // rank order on every rank: identical bits everywhere
        for (int64_t i = 0; i < count; ++i) total[i] += w->red[r][i];
    }
    pthread_barrier_wait(&w->bar);  // nobody overwrites its contribution before all have read it
    if (err) return err;
    err = static_cast<int>(hipMemcpyAsync(buf, total.data(), sizeof(double) * count, hipMemcpyHostToDevice, stream));
    if (err) return err;
    return static_cast<int>(hipStreamSynchronize(stream));  // `total` lives on this stack frame
}

int alltoallv(void* self, gkomi_stream_t s, const void* send, const int64_t* send_counts,
              const int64_t* send_offsets, void* recv, const int64_t* recv_counts,
              const int64_t* recv_offsets, int elem_bytes)
{
    rank_ctx* c = static_cast<rank_ctx*>(self);
    world* w = c->w;
    hipStream_t stream = static_cast<hipStream_t>(s);
    int err = static_cast<int>(hipStreamSynchronize(stream));  // my send buffer is packed
    w->send[c->rank] = static_cast<const char*>(send);
    w->send_counts[c->rank] = send_counts;
    w->send_offsets[c->rank] = send_offsets;
    pthread_barrier_wait(&w->bar);
    int bad = 0;
    for (int p = 0; p < w->size && !err; ++p) {
        if (recv_counts[p] != w->send_counts[p][c->rank]) bad = 1;  // the two sides of the plan disagree
        if (recv_counts[p] <= 0 || bad) continue;
        err = static_cast<int>(hipMemcpyAsync(static_cast<char*>(recv) + recv_offsets[p] * elem_bytes,
                                              w->send[p] + w->send_offsets[p][c->rank] * elem_bytes,
                                              static_cast<size_t>(recv_counts[p]) * elem_bytes,
                                              hipMemcpyDeviceToDevice, stream));
    }
    if (!err) err = static_cast<int>(hipStreamSynchronize(stream));
    pthread_barrier_wait(&w->bar);  // send buffers may be reused from here on
    if (bad) return GKOMI_ECOMM;
    return err;
}

}  // namespace

extern "C" void* loopback_world_create(int size)
{
    world* w = new world;
    w->size = size;
    pthread_barrier_init(&w->bar, nullptr, size);
    w->send.assign(size, nullptr);
    w->send_counts.assign(size, nullptr);
    w->send_offsets.assign(size, nullptr);
    w->red.assign(size, std::vector<double>());
    return w;
}

extern "C" void loopback_world_destroy(void* wp)
{
    world* w = static_cast<world*>(wp);
    pthread_barrier_destroy(&w->bar);
    delete w;
}

extern "C" int loopback_comm_init(void* wp, int rank, gkomi_comm* out)
{
    world* w = static_cast<world*>(wp);
    if (w == nullptr || out == nullptr || rank < 0 || rank >= w->size) return GKOMI_EINVAL;
    rank_ctx* c = new rank_ctx{w, rank};
    out->self = c;
    out->rank = rank;
    out->size = w->size;
    out->allreduce_sum_f64 = allreduce;
    out->alltoallv = alltoallv;
    return 0;
}

extern "C" void loopback_comm_free(gkomi_comm* comm)
{
    delete static_cast<rank_ctx*>(comm->self);
    comm->self = nullptr;
}
