// Communicator of the distributed layer: the collectives that
// gko::experimental::mpi::communicator provides to
// experimental::distributed::{Matrix, Vector} (core/distributed/matrix.cpp:198-224,
// 263-303 all_to_all_v; core/distributed/vector.cpp:317-409 all_reduce), here
// over RCCL on the GPUs of one node (xGMI), one rank per GPU.
//
// The drivers of this library only ever see a `gkomi_comm` record: a context
// pointer plus two function pointers (all-reduce of doubles, all-to-all-v of
// bytes), so that any transport can stand behind it.  This file provides the
// RCCL one.  RCCL is opened at run time (dlopen): the kernel library itself has
// no link dependency on it, and a process that already carries an RCCL (PyTorch
// ships one) shares that copy.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace gkomi {
namespace {

struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};

rccl_api& api()
{
    static rccl_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                               "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (a.handle != nullptr) break;
        }
        if (a.handle == nullptr) return;
#define GKOMI_SYM(field, name) \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name))
        GKOMI_SYM(GetUniqueId, "ncclGetUniqueId");
        GKOMI_SYM(CommInitRank, "ncclCommInitRank");
        GKOMI_SYM(CommDestroy, "ncclCommDestroy");
        GKOMI_SYM(AllReduce, "ncclAllReduce");
        GKOMI_SYM(Send, "ncclSend");
        GKOMI_SYM(Recv, "ncclRecv");
        GKOMI_SYM(GroupStart, "ncclGroupStart");
        GKOMI_SYM(GroupEnd, "ncclGroupEnd");
        GKOMI_SYM(GetErrorString, "ncclGetErrorString");
        GKOMI_SYM(CommCount, "ncclCommCount");
        GKOMI_SYM(CommUserRank, "ncclCommUserRank");
#undef GKOMI_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.Send && a.Recv &&
               a.GroupStart && a.GroupEnd;
    });
    return a;
}

struct rccl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
};

inline int nccl_code(ncclResult_t r) { return r == ncclSuccess ? GKOMI_SUCCESS : GKOMI_ECOMM; }

int rccl_allreduce_sum_f64(void* self, gkomi_stream_t s, double* buf, int64_t count)
{
    rccl_comm* c = static_cast<rccl_comm*>(self);
    if (c == nullptr || count < 0) return GKOMI_EINVAL;
    if (count == 0) return GKOMI_SUCCESS;
    return nccl_code(api().AllReduce(buf, buf, static_cast<size_t>(count), ncclFloat64, ncclSum, c->comm,
                                     to_stream(s)));
}

// every rank sends send_counts[p] elements starting at send_offsets[p] to rank p
// and receives recv_counts[p] elements at recv_offsets[p] from it (elements of
// elem_bytes bytes; counts and offsets are host arrays of comm-size entries).
// One grouped send/recv per peer with something to move: what all_to_all_v does
// for a row partition in which every rank talks to a few neighbours.
int rccl_alltoallv(void* self, gkomi_stream_t s, const void* send, const int64_t* send_counts,
                   const int64_t* send_offsets, void* recv, const int64_t* recv_counts,
                   const int64_t* recv_offsets, int elem_bytes)
{
    rccl_comm* c = static_cast<rccl_comm*>(self);
    if (c == nullptr || elem_bytes <= 0) return GKOMI_EINVAL;
    const rccl_api& a = api();
    hipStream_t stream = to_stream(s);
    const char* sb = static_cast<const char*>(send);
    char* rb = static_cast<char*>(recv);
    // a rank's own share never leaves the device
    if (send_counts[c->rank] != recv_counts[c->rank]) return GKOMI_EINVAL;
    if (send_counts[c->rank] > 0) {
        const int err = static_cast<int>(hipMemcpyAsync(
            rb + recv_offsets[c->rank] * elem_bytes, sb + send_offsets[c->rank] * elem_bytes,
            static_cast<size_t>(send_counts[c->rank]) * elem_bytes, hipMemcpyDeviceToDevice, stream));
        if (err) return err;
    }
    bool any = false;
    for (int p = 0; p < c->size; ++p) {
        if (p != c->rank && (send_counts[p] > 0 || recv_counts[p] > 0)) any = true;
    }
    if (!any) return GKOMI_SUCCESS;
    ncclResult_t r = a.GroupStart();
    if (r != ncclSuccess) return GKOMI_ECOMM;
    for (int p = 0; p < c->size && r == ncclSuccess; ++p) {
        if (p == c->rank) continue;
        if (send_counts[p] > 0) {
            r = a.Send(sb + send_offsets[p] * elem_bytes, static_cast<size_t>(send_counts[p]) * elem_bytes,
                       ncclInt8, p, c->comm, stream);
        }
        if (r == ncclSuccess && recv_counts[p] > 0) {
            r = a.Recv(rb + recv_offsets[p] * elem_bytes, static_cast<size_t>(recv_counts[p]) * elem_bytes,
                       ncclInt8, p, c->comm, stream);
        }
    }
    const ncclResult_t e = a.GroupEnd();
    return nccl_code(r != ncclSuccess ? r : e);
}

}  // namespace
}  // namespace gkomi

using namespace gkomi;

extern "C" int64_t gkomi_comm_unique_id_bytes(void) { return static_cast<int64_t>(sizeof(ncclUniqueId)); }

extern "C" int64_t gkomi_comm_rccl_available(void) { return api().ok ? 1 : 0; }

extern "C" int gkomi_comm_rccl_unique_id(void* id_out)
{
    if (id_out == nullptr) return GKOMI_EINVAL;
    if (!api().ok) return GKOMI_ENOTSUPPORTED;
    ncclUniqueId id;
    const ncclResult_t r = api().GetUniqueId(&id);
    if (r != ncclSuccess) return GKOMI_ECOMM;
    std::memcpy(id_out, &id, sizeof(id));
    return GKOMI_SUCCESS;
}

extern "C" int gkomi_comm_rccl_create(const void* id_in, int rank, int size, gkomi_comm* out)
{
    if (id_in == nullptr || out == nullptr || size < 1 || rank < 0 || rank >= size) return GKOMI_EINVAL;
    if (!api().ok) return GKOMI_ENOTSUPPORTED;
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    rccl_comm* c = new rccl_comm;
    c->rank = rank;
    c->size = size;
    const ncclResult_t r = api().CommInitRank(&c->comm, size, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return GKOMI_ECOMM;
    }
    out->self = c;
    out->rank = rank;
    out->size = size;
    out->allreduce_sum_f64 = rccl_allreduce_sum_f64;
    out->alltoallv = rccl_alltoallv;
    return GKOMI_SUCCESS;
}

// what RCCL itself says about the communicator (not what the launcher said): ranks and this rank
extern "C" int gkomi_comm_rccl_query(const gkomi_comm* comm, int* count, int* user_rank)
{
    if (comm == nullptr || comm->self == nullptr || count == nullptr || user_rank == nullptr) return GKOMI_EINVAL;
    if (comm->allreduce_sum_f64 != rccl_allreduce_sum_f64) return GKOMI_EINVAL;  // not an RCCL communicator
    const rccl_api& a = api();
    if (a.CommCount == nullptr || a.CommUserRank == nullptr) return GKOMI_ENOTSUPPORTED;
    const rccl_comm* c = static_cast<const rccl_comm*>(comm->self);
    if (a.CommCount(c->comm, count) != ncclSuccess) return GKOMI_ECOMM;
    return nccl_code(a.CommUserRank(c->comm, user_rank));
}

extern "C" int gkomi_comm_rccl_destroy(gkomi_comm* comm)
{
    if (comm == nullptr || comm->self == nullptr) return GKOMI_SUCCESS;
    rccl_comm* c = static_cast<rccl_comm*>(comm->self);
    const ncclResult_t r = api().CommDestroy(c->comm);
    delete c;
    comm->self = nullptr;
    return nccl_code(r);
}

extern "C" int gkomi_comm_allreduce_sum_f64(const gkomi_comm* comm, gkomi_stream_t s, double* buf,
                                            int64_t count)
{
    if (comm == nullptr || comm->allreduce_sum_f64 == nullptr) return GKOMI_EINVAL;
    return comm->allreduce_sum_f64(comm->self, s, buf, count);
}

extern "C" int gkomi_comm_alltoallv(const gkomi_comm* comm, gkomi_stream_t s, const void* send,
                                    const int64_t* send_counts, const int64_t* send_offsets, void* recv,
                                    const int64_t* recv_counts, const int64_t* recv_offsets,
                                    int elem_bytes)
{
    if (comm == nullptr || comm->alltoallv == nullptr) return GKOMI_EINVAL;
    return comm->alltoallv(comm->self, s, send, send_counts, send_offsets, recv, recv_counts, recv_offsets,
                           elem_bytes);
}
