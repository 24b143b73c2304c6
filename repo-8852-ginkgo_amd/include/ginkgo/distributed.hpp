// gko::experimental::{mpi, distributed} of the host mirror: the interface of
// include/ginkgo/core/base/mpi.hpp and include/ginkgo/core/distributed/
// {partition,vector,matrix}.hpp that the row-partitioned hot path uses, over the
// C ABI's communicator (gkomi_comm on RCCL, csrc/comm.hip) and distributed
// drivers (csrc/dist_cg.hip).  Included by ginkgo.hpp; the reference's own
// examples/distributed-solver/distributed-solver.cpp compiles against it.
//
// There is no MPI in this build.  One process per GPU is started by any launcher
// that sets RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR and MASTER_PORT (torchrun
// --no-python, srun, a shell loop); `mpi::environment` reads them, rank 0 hands
// the RCCL unique id to the others over one TCP connection each (port
// MASTER_PORT + 1), and from then on every collective is RCCL over xGMI.  A
// single process needs no variables at all.  When a real <mpi.h> was included
// first, its MPI_Comm / MPI_COMM_WORLD are left alone.
#pragma once

#include <arpa/inet.h>
#include <netdb.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <thread>

#ifndef MPI_VERSION
using MPI_Comm = int;
#ifndef MPI_COMM_WORLD
#define MPI_COMM_WORLD 0
#endif
#endif

namespace gko {
namespace experimental {
namespace mpi {

enum class thread_type { serialized, funneled, single, multiple };

namespace detail {

// process-wide state of the "MPI" world of this build
struct world_state {
    int rank{0}, size{1}, local_rank{0};
    std::string master_addr{"127.0.0.1"};
    int master_port{29500};
    bool comm_ready{false};
    gkomi_comm comm{};
    double* scratch{nullptr};  // one device double for barriers
    ~world_state()
    {
        if (comm_ready) {
            if (scratch) gkomi_raw_free(scratch);
            gkomi_comm_rccl_destroy(&comm);
        }
    }
};
inline world_state& world()
{
    static world_state w;
    return w;
}

inline void throw_errno(const char* what) { throw Error(__FILE__, __LINE__, std::string("mpi bootstrap: ") + what); }

inline void send_all(int fd, const char* p, size_t n)
{
    while (n > 0) {
        const ssize_t k = ::send(fd, p, n, 0);
        if (k <= 0) throw_errno("send failed");
        p += k; n -= static_cast<size_t>(k);
    }
}
inline void recv_all(int fd, char* p, size_t n)
{
    while (n > 0) {
        const ssize_t k = ::recv(fd, p, n, 0);
        if (k <= 0) throw_errno("recv failed");
        p += k; n -= static_cast<size_t>(k);
    }
}

// rank 0 -> everybody: `bytes` bytes, one TCP connection per receiver
inline void bootstrap_broadcast(world_state& w, char* data, size_t bytes)
{
    if (w.size == 1) return;
    const int port = w.master_port + 1;
    if (w.rank == 0) {
        const int srv = ::socket(AF_INET, SOCK_STREAM, 0);
        if (srv < 0) throw_errno("socket");
        int yes = 1;
        ::setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &yes, sizeof(yes));
        sockaddr_in addr{};
        addr.sin_family = AF_INET;
        addr.sin_addr.s_addr = htonl(INADDR_ANY);
        addr.sin_port = htons(static_cast<uint16_t>(port));
        if (::bind(srv, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0) throw_errno("bind (MASTER_PORT + 1 busy?)");
        if (::listen(srv, w.size) != 0) throw_errno("listen");
        for (int i = 1; i < w.size; ++i) {
            const int fd = ::accept(srv, nullptr, nullptr);
            if (fd < 0) throw_errno("accept");
            send_all(fd, data, bytes);
            ::close(fd);
        }
        ::close(srv);
    } else {
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        if (::getaddrinfo(w.master_addr.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || res == nullptr) {
            throw_errno("cannot resolve MASTER_ADDR");
        }
        int fd = -1;
        for (int attempt = 0; attempt < 600; ++attempt) {  // rank 0 may not listen yet: retry for a minute
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
            if (fd >= 0) ::close(fd);
            fd = -1;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        ::freeaddrinfo(res);
        if (fd < 0) throw_errno("cannot connect to rank 0");
        recv_all(fd, data, bytes);
        ::close(fd);
    }
}

// the RCCL communicator is created at first use: by then the rank's executor has
// selected its device (HipExecutor::create calls hipSetDevice)
inline const gkomi_comm* native_comm()
{
    world_state& w = world();
    if (!w.comm_ready) {
        std::vector<char> id(static_cast<size_t>(gkomi_comm_unique_id_bytes()));
        if (w.rank == 0) GKOMI_CALL(gkomi_comm_rccl_unique_id(id.data()));
        bootstrap_broadcast(w, id.data(), id.size());
        GKOMI_CALL(gkomi_comm_rccl_create(id.data(), w.rank, w.size, &w.comm));
        void* p = nullptr;
        GKOMI_CALL(gkomi_raw_alloc(sizeof(double), &p));
        w.scratch = static_cast<double*>(p);
        w.comm_ready = true;
    }
    return &w.comm;
}

inline int env_int(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return v != nullptr && v[0] != 0 ? std::atoi(v) : fallback;
}

}  // namespace detail

// include/ginkgo/core/base/mpi.hpp:  RAII init / finalize
class environment {
public:
    environment(int&, char**&, thread_type = thread_type::serialized)
    {
        auto& w = detail::world();
        w.rank = detail::env_int("RANK", 0);
        w.size = detail::env_int("WORLD_SIZE", 1);
        w.local_rank = detail::env_int("LOCAL_RANK", w.rank);
        w.master_port = detail::env_int("MASTER_PORT", 29500);
        if (const char* a = std::getenv("MASTER_ADDR")) w.master_addr = a;
        if (w.size < 1 || w.rank < 0 || w.rank >= w.size) throw BadDimension(__FILE__, __LINE__, "RANK / WORLD_SIZE inconsistent");
    }
    environment(const environment&) = delete;
    environment& operator=(const environment&) = delete;
};

inline double get_walltime()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ranks of one node take the node's devices round robin (mpi.hpp map_rank_to_device_id)
inline int map_rank_to_device_id(MPI_Comm, int num_devices)
{
    return num_devices > 0 ? detail::world().local_rank % num_devices : 0;
}

class communicator {
public:
    communicator(MPI_Comm = MPI_COMM_WORLD) {}
    int rank() const { return detail::world().rank; }
    int size() const { return detail::world().size; }
    // MPI_Barrier: a one-element all-reduce, then wait for it
    void synchronize() const
    {
        if (size() == 1 && !detail::world().comm_ready) return;
        const gkomi_comm* c = detail::native_comm();
        GKOMI_CALL(gkomi_comm_allreduce_sum_f64(c, nullptr, detail::world().scratch, 1));
        GKOMI_CALL(gkomi_synchronize(nullptr));
    }
    const gkomi_comm* native() const { return detail::native_comm(); }
    bool operator==(const communicator&) const { return true; }
};

}  // namespace mpi


namespace distributed {

using comm_index_type = int32;

// include/ginkgo/core/distributed/partition.hpp: host-side metadata, O(#ranges)
template <typename LocalIndexType = int32, typename GlobalIndexType = int64>
class Partition {
public:
    static std::unique_ptr<Partition> build_from_global_size_uniform(std::shared_ptr<const Executor> exec, comm_index_type num_parts,
                                                                      GlobalIndexType global_size)
    {
        std::vector<int64_t> ranges(num_parts + 1);
        GKOMI_CALL(gkomi_partition_build_ranges_from_global_size(num_parts, global_size, ranges.data()));
        return build_from_contiguous(std::move(exec), ranges);
    }
    static std::unique_ptr<Partition> build_from_contiguous(std::shared_ptr<const Executor> exec, const std::vector<int64_t>& ranges)
    {
        auto p = std::unique_ptr<Partition>(new Partition(std::move(exec)));
        const int64_t num_parts = static_cast<int64_t>(ranges.size()) - 1;
        p->bounds_.resize(num_parts + 1);
        p->part_ids_.resize(num_parts);
        GKOMI_CALL(gkomi_partition_build_from_contiguous(num_parts, ranges.data(), p->bounds_.data(), p->part_ids_.data()));
        p->finish(num_parts);
        return p;
    }
    static std::unique_ptr<Partition> build_from_mapping(std::shared_ptr<const Executor> exec, const array<comm_index_type>& mapping,
                                                         comm_index_type num_parts)
    {
        auto p = std::unique_ptr<Partition>(new Partition(std::move(exec)));
        const auto host = mapping.to_host();
        p->bounds_.resize(host.size() + 1);
        p->part_ids_.resize(std::max<size_t>(host.size(), 1));
        int64_t nr = 0;
        GKOMI_CALL(gkomi_partition_build_from_mapping(static_cast<int64_t>(host.size()), host.data(), p->bounds_.data(), p->part_ids_.data(), &nr));
        p->bounds_.resize(nr + 1);
        p->part_ids_.resize(nr);
        p->finish(num_parts);
        return p;
    }
    size_type get_size() const { return static_cast<size_type>(bounds_.back()); }
    size_type get_num_ranges() const { return part_ids_.size(); }
    comm_index_type get_num_parts() const { return static_cast<comm_index_type>(part_sizes_.size()); }
    // As in the reference the arrays live on the partition's executor (partition.hpp:300-340): device pointers for a
    // partition built on a HipExecutor -- what the kernels (build_local_nonlocal, build_local) take --, host
    // pointers for one built on exec->get_master().  host_*(): the host copies this mirror's own set-up code reads.
    const GlobalIndexType* get_range_bounds() const { return exec_->is_device() ? d_bounds_.get_const_data() : bounds_.data(); }
    const comm_index_type* get_part_ids() const { return exec_->is_device() ? d_part_ids_.get_const_data() : part_ids_.data(); }
    const LocalIndexType* get_range_starting_indices() const { return exec_->is_device() ? d_starts_.get_const_data() : starts_.data(); }
    const LocalIndexType* get_part_sizes() const { return exec_->is_device() ? d_part_sizes_.get_const_data() : part_sizes_.data(); }
    const GlobalIndexType* host_range_bounds() const { return bounds_.data(); }
    const comm_index_type* host_part_ids() const { return part_ids_.data(); }
    const LocalIndexType* host_range_starting_indices() const { return starts_.data(); }
    LocalIndexType get_part_size(comm_index_type part) const { return part_sizes_[part]; }
    comm_index_type get_num_empty_parts() const { return static_cast<comm_index_type>(num_empty_parts_); }
    // core/distributed/partition.cpp:120-138
    bool has_connected_parts() const { return static_cast<size_type>(get_num_parts() - get_num_empty_parts()) == get_num_ranges(); }
    bool has_ordered_parts() const
    {
        if (!has_connected_parts()) return false;
        int64_t ordered = 0;
        GKOMI_CALL(gkomi_partition_has_ordered_parts(part_ids_.data(), static_cast<int64_t>(part_ids_.size()), &ordered));
        return ordered != 0;
    }
    std::shared_ptr<const Executor> get_executor() const { return exec_; }
private:
    explicit Partition(std::shared_ptr<const Executor> exec) : exec_(std::move(exec)) {}
    void finish(int64_t num_parts)
    {
        starts_.resize(part_ids_.size());
        part_sizes_.resize(num_parts);
        GKOMI_CALL(gkomi_partition_build_starting_indices(bounds_.data(), part_ids_.data(), static_cast<int64_t>(part_ids_.size()), num_parts,
                                                          starts_.data(), part_sizes_.data(), &num_empty_parts_));
        if (exec_->is_device()) {
            auto host = exec_->get_master();
            auto up = [&](auto& dst, const auto& src) {
                using T = typename std::decay<decltype(src)>::type::value_type;
                dst = array<T>(exec_, std::max<size_type>(src.size(), 1));
                if (!src.empty()) exec_->copy_from(host.get(), src.size(), src.data(), dst.get_data());
            };
            up(d_bounds_, bounds_);
            up(d_part_ids_, part_ids_);
            up(d_starts_, starts_);
            up(d_part_sizes_, part_sizes_);
        }
    }
    std::shared_ptr<const Executor> exec_;
    std::vector<int64_t> bounds_;
    std::vector<int32_t> part_ids_, starts_, part_sizes_;
    array<int64_t> d_bounds_;
    array<int32_t> d_part_ids_, d_starts_, d_part_sizes_;
    int64_t num_empty_parts_{0};
    static_assert(std::is_same<GlobalIndexType, int64>::value && std::is_same<LocalIndexType, int32>::value,
                  "this backend instantiates Partition<int32, int64>");
};


// include/ginkgo/core/distributed/vector.hpp: the local rows of a global vector

template <typename ValueType = double>
class Vector : public LinOp {
public:
    using value_type = ValueType;
    using local_vector_type = matrix::Dense<ValueType>;
    static std::unique_ptr<Vector> create(std::shared_ptr<const Executor> exec, mpi::communicator comm, dim<2> global_size = dim<2>{},
                                          dim<2> local_size = dim<2>{})
    {
        return std::unique_ptr<Vector>(new Vector(std::move(exec), comm, global_size, local_size));
    }
    // Vector::read_distributed (core/distributed/vector.cpp:120-170): keeps the rows this rank owns -- any partition
    // (several ranges per part, parts in any order); the entries go to the device as they are and
    // distributed_vector::build_local scatters the local ones (gkomi_dist_vector_build_local_f64)
    template <typename GlobalIndexType>
    void read_distributed(const matrix_data<ValueType, GlobalIndexType>& data, const Partition<int32, GlobalIndexType>* partition)
    {
        const int rank = comm_.rank();
        if (comm_.size() != partition->get_num_parts()) {  // core/distributed/vector.cpp:172
            throw BadDimension(__FILE__, __LINE__, "Vector::read_distributed: one part per rank");
        }
        const size_type nloc = static_cast<size_type>(partition->get_part_size(rank)), ncols = data.size[1];
        const size_type nnz = data.nonzeros.size(), nr = partition->get_num_ranges();
        local_ = local_vector_type::create(exec_, dim<2>(nloc, ncols));
        local_->fill(ValueType{});
        set_size(dim<2>(data.size[0], ncols));
        if (nnz == 0 || nloc == 0) return;
        if (!exec_->is_device()) {
            // staging on a host executor (the example's flow: read on the host, clone to the device): the same
            // map, entry by entry
            const auto* bounds = partition->host_range_bounds();
            for (const auto& e : data.nonzeros) {
                const size_type r = static_cast<size_type>(std::upper_bound(bounds + 1, bounds + nr + 1, e.row) - (bounds + 1));
                if (r < nr && partition->host_part_ids()[r] == rank) {
                    local_->at(static_cast<size_type>(e.row - bounds[r]) + partition->host_range_starting_indices()[r], static_cast<size_type>(e.column)) = e.value;
                }
            }
            return;
        }
        std::vector<int64_t> rows(nnz), cols(nnz);
        std::vector<ValueType> vals(nnz);
        for (size_type i = 0; i < nnz; ++i) {
            rows[i] = data.nonzeros[i].row;
            cols[i] = data.nonzeros[i].column;
            vals[i] = data.nonzeros[i].value;
        }
        auto host = exec_->get_master();
        array<int64_t> d_rows(exec_, nnz), d_cols(exec_, nnz), d_bounds(exec_, nr + 1);
        array<ValueType> d_vals(exec_, nnz);
        array<int32_t> d_ids(exec_, nr), d_starts(exec_, nr);
        exec_->copy_from(host.get(), nnz, rows.data(), d_rows.get_data());
        exec_->copy_from(host.get(), nnz, cols.data(), d_cols.get_data());
        exec_->copy_from(host.get(), nnz, vals.data(), d_vals.get_data());
        exec_->copy_from(host.get(), nr + 1, reinterpret_cast<const int64_t*>(partition->host_range_bounds()), d_bounds.get_data());
        exec_->copy_from(host.get(), nr, partition->host_part_ids(), d_ids.get_data());
        exec_->copy_from(host.get(), nr, partition->host_range_starting_indices(), d_starts.get_data());
        GKOMI_CALL(gkomi_dist_vector_build_local_f64(nullptr, static_cast<int64_t>(nnz), d_rows.get_const_data(), d_cols.get_const_data(),
                                                     d_vals.get_const_data(), d_bounds.get_const_data(), d_ids.get_const_data(),
                                                     d_starts.get_const_data(), static_cast<int64_t>(nr), rank, local_->get_values(),
                                                     static_cast<int64_t>(local_->get_stride())));
        GKOMI_CALL(gkomi_synchronize(nullptr));  // the staging arrays go out of scope
    }
    void copy_from(const Vector* other)
    {
        local_ = other->local_->clone(exec_);
        set_size(other->get_size());
    }
    std::unique_ptr<Vector> clone(std::shared_ptr<const Executor> exec = nullptr) const
    {
        auto v = Vector::create(exec ? exec : exec_, comm_);
        v->copy_from(this);
        return v;
    }
    local_vector_type* get_local_vector() { return local_.get(); }
    const local_vector_type* get_local_vector() const { return local_.get(); }
    mpi::communicator get_communicator() const { return comm_; }
    void fill(ValueType v) { local_->fill(v); }
    void scale(const matrix::Dense<ValueType>* alpha) { local_->scale(alpha); }
    void add_scaled(const matrix::Dense<ValueType>* alpha, const Vector* b) { local_->add_scaled(alpha, b->local_.get()); }
    void sub_scaled(const matrix::Dense<ValueType>* alpha, const Vector* b) { local_->sub_scaled(alpha, b->local_.get()); }
    // Vector::compute_dot / compute_norm2 (core/distributed/vector.cpp:317-409): local result, all-reduce
    void compute_dot(const Vector* b, matrix::Dense<ValueType>* result) const
    {
        auto dev = device_result(result);
        local_->compute_dot(b->local_.get(), dev.get());
        GKOMI_CALL(gkomi_comm_allreduce_sum_f64(comm_.native(), nullptr, dev->get_values(), static_cast<int64_t>(get_size()[1])));
        result->copy_from(dev.get());
    }
    void compute_conj_dot(const Vector* b, matrix::Dense<ValueType>* result) const { compute_dot(b, result); }
    void compute_norm2(matrix::Dense<ValueType>* result) const
    {
        auto dev = device_result(result);
        const int64_t k = static_cast<int64_t>(get_size()[1]);
        array<char> tmp(exec_, gkomi_dense_reduction_workspace_bytes(local_->rows(), k) + 8);
        GKOMI_CALL(gkomi_dense_compute_squared_norm2_f64(nullptr, local_->rows(), k, local_->get_const_values(), local_->get_stride(), dev->get_values(),
                                                         tmp.get_data(), tmp.get_num_elems()));
        GKOMI_CALL(gkomi_comm_allreduce_sum_f64(comm_.native(), nullptr, dev->get_values(), k));
        GKOMI_CALL(gkomi_dense_compute_sqrt_f64(nullptr, 1, k, dev->get_values(), k));
        result->copy_from(dev.get());
    }
protected:
    Vector(std::shared_ptr<const Executor> exec, mpi::communicator comm, dim<2> global_size, dim<2> local_size)
        : LinOp(exec, global_size), comm_(comm), local_(local_vector_type::create(exec, local_size)) {}
    // the reductions run on the device; a result on the host is filled through a device staging value
    std::unique_ptr<matrix::Dense<ValueType>> device_result(const matrix::Dense<ValueType>* result) const
    {
        ::gko::detail::require_device(exec_, "distributed::Vector reduction");
        if (result->get_size() != dim<2>(1, get_size()[1])) throw DimensionMismatch(__FILE__, __LINE__, "result must be 1 x #columns");
        return matrix::Dense<ValueType>::create(exec_, result->get_size());
    }
    void apply_impl(const LinOp*, LinOp*) const override { GKO_NOT_IMPLEMENTED; }
    void apply_impl(const LinOp*, const LinOp*, const LinOp*, LinOp*) const override { GKO_NOT_IMPLEMENTED; }
    mpi::communicator comm_;
    std::unique_ptr<local_vector_type> local_;
};


// include/ginkgo/core/distributed/matrix.hpp: local block + non-local block + halo plan
template <typename ValueType = double, typename LocalIndexType = int32, typename GlobalIndexType = int64>
class Matrix : public LinOp, public ::gko::detail::distributed_system {
public:
    using value_type = ValueType;
    using global_vector_type = Vector<ValueType>;
    static std::unique_ptr<Matrix> create(std::shared_ptr<const Executor> exec, mpi::communicator comm)
    {
        return std::unique_ptr<Matrix>(new Matrix(std::move(exec), comm));
    }
    ~Matrix() override
    {
        if (ctx_) gkomi_dist_ctx_destroy(ctx_);
    }
    // Matrix::read_distributed (core/distributed/matrix.cpp:142-260).  On a host executor the
    // entries of this rank are kept for a later copy to a device (the example's flow); on the
    // device the blocks are built and the two setup exchanges run (collective).
    void read_distributed(const matrix_data<ValueType, GlobalIndexType>& data, const Partition<LocalIndexType, GlobalIndexType>* partition)
    {
        const int rank = comm_.rank();
        // core/distributed/matrix.cpp:148-151
        if (static_cast<size_type>(data.size[0]) != partition->get_size() || comm_.size() != partition->get_num_parts()) {
            throw BadDimension(__FILE__, __LINE__, "Matrix::read_distributed: the partition must cover the rows and have one part per rank");
        }
        // ranges map to parts through part_ids (reference/distributed/partition_kernels.cpp:42-95): this rank's rows
        // are those of EVERY range whose part id is the rank (a part may own several ranges, in any order); entries of
        // other parts are dropped here, the device kernel (build_local_nonlocal) maps the rest to local indices
        const auto* rb = partition->host_range_bounds();
        const size_type nr = partition->get_num_ranges();
        staged_.rows.clear(); staged_.cols.clear(); staged_.vals.clear();
        auto sorted = data;
        sorted.ensure_row_major_order();
        size_type hint = 0;
        for (const auto& e : sorted.nonzeros) {
            if (!(nr > 0 && rb[hint] <= e.row && e.row < rb[hint + 1])) {
                hint = static_cast<size_type>(std::upper_bound(rb + 1, rb + nr + 1, e.row) - (rb + 1));
            }
            if (hint < nr && partition->host_part_ids()[hint] == rank) {
                staged_.rows.push_back(e.row); staged_.cols.push_back(e.column); staged_.vals.push_back(e.value);
            }
            if (hint >= nr) hint = 0;
        }
        staged_.bounds.assign(partition->host_range_bounds(), partition->host_range_bounds() + partition->get_num_ranges() + 1);
        staged_.part_ids.assign(partition->host_part_ids(), partition->host_part_ids() + partition->get_num_ranges());
        staged_.starts.assign(partition->host_range_starting_indices(), partition->host_range_starting_indices() + partition->get_num_ranges());
        staged_.num_parts = partition->get_num_parts();
        staged_.n_local = static_cast<int64_t>(partition->get_part_size(rank));
        staged_.valid = true;
        set_size(data.size);
        if (exec_->is_device()) build_on_device();
    }
    void copy_from(const Matrix* other)
    {
        staged_ = other->staged_;
        set_size(other->get_size());
        if (exec_->is_device() && staged_.valid) build_on_device();
    }
    size_type get_num_local_rows() const { return static_cast<size_type>(staged_.n_local); }
    size_type get_num_halo_entries() const { return static_cast<size_type>(plan_.n_halo); }
    size_type get_num_send_entries() const { return static_cast<size_type>(plan_.send_total); }
    mpi::communicator get_communicator() const { return comm_; }

    // Cg on distributed vectors: the fused native driver (csrc/dist_cg.hip)
    void cg_solve(const LinOp* b, LinOp* x, const stop::criterion_settings& st, const LinOp* precond, int64_t* iters, bool* converged) const override
    {
        ::gko::detail::require_device(exec_, "distributed cg");
        if (precond != nullptr) GKO_NOT_SUPPORTED("distributed Cg: preconditioners are rank-local gkomi_apply_fn callbacks of the C ABI");
        auto db = as<const Vector<ValueType>>(b);
        auto dx = as<Vector<ValueType>>(x);
        if (db->get_size()[1] != 1) GKO_NOT_SUPPORTED("distributed Cg: one right-hand side");
        array<char> ws(exec_, gkomi_dist_cg_workspace_bytes(plan_.n_local, plan_.nl_rows));
        double info[4] = {};
        GKOMI_CALL(gkomi_dist_cg_solve_f64(nullptr, comm_.native(), ctx_, &plan_, nullptr, nullptr, db->get_local_vector()->get_const_values(),
                                           dx->get_local_vector()->get_values(), st.max_iters, st.reduction_factor,
                                           st.baseline == stop::mode::rhs_norm ? 0 : (st.baseline == stop::mode::initial_resnorm ? 1 : 2), 16,
                                           ws.get_data(), ws.get_num_elems(), info));
        *iters = static_cast<int64_t>(info[0]);
        *converged = info[1] != 0.0;
    }
protected:
    Matrix(std::shared_ptr<const Executor> exec, mpi::communicator comm) : LinOp(exec, dim<2>{}), comm_(comm) {}
    // x = A b (matrix.cpp:307-335)
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "distributed::Matrix::apply");
        auto db = as<const Vector<ValueType>>(b);
        auto dx = as<Vector<ValueType>>(x);
        if (db->get_size()[1] != 1) GKO_NOT_SUPPORTED("distributed apply: one right-hand side");
        GKOMI_CALL(gkomi_dist_matrix_apply_f64(nullptr, comm_.native(), ctx_, &plan_, db->get_local_vector()->get_const_values(),
                                               dx->get_local_vector()->get_values()));
    }
    // x = alpha A b + beta x (matrix.cpp:338-369)
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "distributed::Matrix::apply");
        auto dx = as<Vector<ValueType>>(x);
        auto tmp = dx->clone();
        this->apply_impl(b, tmp.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), tmp.get());
    }

    void build_on_device()
    {
        const int rank = comm_.rank(), world = comm_.size();
        const int64_t nnz = static_cast<int64_t>(staged_.vals.size()), n_loc = staged_.n_local;
        const int64_t nr = static_cast<int64_t>(staged_.part_ids.size());
        auto host = exec_->get_master();
        auto up = [&](const auto& v) { using T = typename std::decay<decltype(v)>::type::value_type; return array<T>(exec_, v.begin(), v.end()); };
        array<int64_t> rows = up(staged_.rows), cols = up(staged_.cols), bounds = up(staged_.bounds);
        array<double> vals = up(staged_.vals);
        array<int32_t> part_ids = up(staged_.part_ids), starts = up(staged_.starts);
        array<char> ws(exec_, gkomi_dist_build_workspace_bytes(nnz));
        int64_t sizes[3] = {};
        GKOMI_CALL(gkomi_dist_build_local_nonlocal_sizes(nullptr, nnz, rows.get_const_data(), cols.get_const_data(), bounds.get_const_data(), part_ids.get_const_data(),
                                                         starts.get_const_data(), nr, bounds.get_const_data(), part_ids.get_const_data(), starts.get_const_data(), nr, rank,
                                                         ws.get_data(), ws.get_num_elems(), sizes));
        const int64_t nl = sizes[0], nn = sizes[1], nu = sizes[2];
        auto at_least_one = [](int64_t v) { return static_cast<size_type>(v > 0 ? v : 1); };
        array<int32_t> l_rows(exec_, at_least_one(nl)), nl_rows(exec_, at_least_one(nn)), gather_recv(exec_, at_least_one(nu)), recv_sizes(exec_, world);
        l_cols_ = array<int32_t>(exec_, at_least_one(nl)); l_vals_ = array<double>(exec_, at_least_one(nl));
        nl_cols_ = array<int32_t>(exec_, at_least_one(nn)); nl_vals_ = array<double>(exec_, at_least_one(nn));
        array<int64_t> n2g(exec_, at_least_one(nu));
        GKOMI_CALL(gkomi_dist_build_local_nonlocal_fill(nullptr, nnz, rows.get_const_data(), cols.get_const_data(), vals.get_const_data(), bounds.get_const_data(),
                                                        part_ids.get_const_data(), starts.get_const_data(), nr, bounds.get_const_data(), part_ids.get_const_data(),
                                                        starts.get_const_data(), nr, staged_.num_parts, ws.get_const_data(), nu, l_rows.get_data(), l_cols_.get_data(),
                                                        l_vals_.get_data(), nl_rows.get_data(), nl_cols_.get_data(), nl_vals_.get_data(), gather_recv.get_data(),
                                                        recv_sizes.get_data(), n2g.get_data()));
        // Csr::read(device_matrix_data): row indices -> row pointers, for both blocks
        array<char> pws(exec_, std::max<size_t>(gkomi_prefix_sum_workspace_bytes(n_loc + 1), 8));
        l_rp_ = array<int32_t>(exec_, n_loc + 1);
        array<int32_t> nl_rp_full(exec_, n_loc + 1);
        GKOMI_CALL(gkomi_convert_idxs_to_ptrs_i32(nullptr, l_rows.get_const_data(), nl, n_loc, l_rp_.get_data(), pws.get_data(), pws.get_num_elems()));
        GKOMI_CALL(gkomi_convert_idxs_to_ptrs_i32(nullptr, nl_rows.get_const_data(), nn, n_loc, nl_rp_full.get_data(), pws.get_data(), pws.get_num_elems()));
        // rows of the non-local block that have entries
        array<char> rws(exec_, gkomi_dist_nonlocal_rows_workspace_bytes(n_loc));
        nl_row_idxs_ = array<int32_t>(exec_, n_loc + 1);
        nl_row_ptrs_ = array<int32_t>(exec_, n_loc + 1);
        int64_t nz_rows = 0;
        GKOMI_CALL(gkomi_dist_nonlocal_rows_i32(nullptr, n_loc, nl_rp_full.get_const_data(), nl_row_idxs_.get_data(), nl_row_ptrs_.get_data(), rws.get_data(),
                                                rws.get_num_elems(), &nz_rows));
        // setup exchange 1: who sends how much (matrix.cpp:198-209), on device buffers over the communicator
        const auto recv_host32 = recv_sizes.to_host();
        recv_counts_.assign(recv_host32.begin(), recv_host32.end());
        send_counts_.assign(world, 0);
        {
            std::vector<double> mine(recv_counts_.begin(), recv_counts_.end()), theirs(world, 0.0);
            array<double> sbuf(exec_, mine.begin(), mine.end()), rbuf(exec_, world);
            std::vector<int64_t> ones(world, 1), offs(world);
            for (int p = 0; p < world; ++p) offs[p] = p;
            GKOMI_CALL(gkomi_comm_alltoallv(comm_.native(), nullptr, sbuf.get_const_data(), ones.data(), offs.data(), rbuf.get_data(), ones.data(), offs.data(), 8));
            GKOMI_CALL(gkomi_synchronize(nullptr));
            theirs = rbuf.to_host();
            for (int p = 0; p < world; ++p) send_counts_[p] = static_cast<int64_t>(theirs[p]);
        }
        send_offsets_.assign(world, 0); recv_offsets_.assign(world, 0);
        for (int p = 1; p < world; ++p) {
            send_offsets_[p] = send_offsets_[p - 1] + send_counts_[p - 1];
            recv_offsets_[p] = recv_offsets_[p - 1] + recv_counts_[p - 1];
        }
        const int64_t send_total = send_offsets_[world - 1] + send_counts_[world - 1];
        // setup exchange 2: receivers tell senders which of their rows they need (:211-224)
        gather_idxs_ = array<int32_t>(exec_, at_least_one(send_total));
        GKOMI_CALL(gkomi_comm_alltoallv(comm_.native(), nullptr, gather_recv.get_const_data(), recv_counts_.data(), recv_offsets_.data(), gather_idxs_.get_data(),
                                        send_counts_.data(), send_offsets_.data(), 4));
        // Csr::make_srow + row statistics of the local block
        const int64_t tile = gkomi_csr_srow_tile();
        srow_ = array<int32_t>(exec_, at_least_one(gkomi_csr_srow_entries(nl, tile)));
        array<int32_t> mx(exec_, 1);
        int64_t max_row = -1;
        if (n_loc > 0) {
            GKOMI_CALL(gkomi_csr_max_row_nnz_i32(nullptr, n_loc, l_rp_.get_const_data(), mx.get_data()));
            max_row = exec_->copy_val_to_host(mx.get_const_data());
        }
        if (nl >= 2) GKOMI_CALL(gkomi_csr_make_srow_i32(nullptr, n_loc, nl, l_rp_.get_const_data(), tile, srow_.get_data(), static_cast<int64_t>(srow_.get_num_elems())));
        send_buf_ = array<double>(exec_, at_least_one(send_total));
        recv_buf_ = array<double>(exec_, at_least_one(nu));
        GKOMI_CALL(gkomi_synchronize(nullptr));
        plan_ = gkomi_dist_matrix{};
        plan_.n_local = n_loc; plan_.n_halo = nu; plan_.l_nnz = nl;
        plan_.l_row_ptrs = l_rp_.get_const_data(); plan_.l_col_idxs = l_cols_.get_const_data(); plan_.l_vals = l_vals_.get_const_data();
        plan_.l_max_row_nnz = max_row; plan_.l_srow = nl >= 2 ? srow_.get_const_data() : nullptr; plan_.l_srow_tile = tile;
        plan_.nl_rows = nz_rows; plan_.nl_nnz = nn; plan_.nl_row_idxs = nl_row_idxs_.get_const_data(); plan_.nl_row_ptrs = nl_row_ptrs_.get_const_data();
        plan_.nl_col_idxs = nl_cols_.get_const_data(); plan_.nl_vals = nl_vals_.get_const_data();
        plan_.send_total = send_total; plan_.gather_idxs = gather_idxs_.get_const_data();
        plan_.send_counts = send_counts_.data(); plan_.send_offsets = send_offsets_.data();
        plan_.recv_counts = recv_counts_.data(); plan_.recv_offsets = recv_offsets_.data();
        plan_.send_buf = send_buf_.get_data(); plan_.recv_buf = recv_buf_.get_data();
        if (!ctx_) GKOMI_CALL(gkomi_dist_ctx_create(&ctx_));
    }

    struct staged_data {
        std::vector<int64_t> rows, cols, bounds;
        std::vector<double> vals;
        std::vector<int32_t> part_ids, starts;
        int64_t num_parts{0}, n_local{0};
        bool valid{false};
    };
    mpi::communicator comm_;
    staged_data staged_;
    array<int32_t> l_rp_, l_cols_, nl_cols_, nl_row_idxs_, nl_row_ptrs_, gather_idxs_, srow_;
    array<double> l_vals_, nl_vals_, send_buf_, recv_buf_;
    std::vector<int64_t> send_counts_, send_offsets_, recv_counts_, recv_offsets_;
    gkomi_dist_matrix plan_{};
    gkomi_dist_ctx* ctx_{nullptr};
    static_assert(std::is_same<ValueType, double>::value && std::is_same<LocalIndexType, int32>::value && std::is_same<GlobalIndexType, int64>::value,
                  "this backend instantiates distributed::Matrix<double, int32, int64>");
};

}  // namespace distributed
}  // namespace experimental
}  // namespace gko
