// Host-side mirror of the Ginkgo public interface for the SpMV + Krylov hot
// path, on top of the C ABI in include/gkomi.h (header-only, C++14).
//
// It reproduces the names, argument meaning and error behaviour of the
// reference's operator interface so that code written against
// <ginkgo/ginkgo.hpp> for this path -- e.g. the reference's
// examples/simple-solver/simple-solver.cpp -- compiles and runs unchanged:
//   gko::Executor / OmpExecutor / ReferenceExecutor / HipExecutor
//     (include/ginkgo/core/base/executor.hpp:602-1017, 1591-1781),
//   gko::array, gko::dim, gko::LinOp, gko::LinOpFactory
//     (include/ginkgo/core/base/{array,dim,lin_op}.hpp),
//   gko::matrix::{Dense, Csr, Coo, Ell, Sellp, Hybrid},
//   gko::stop::{Iteration, ResidualNorm, Combined},
//   gko::solver::{Cg, Gmres, LowerTrs, UpperTrs},
//   gko::preconditioner::{Jacobi, Ilu}, gko::factorization::ParIlu,
//   gko::read / gko::write (MatrixMarket), gko::initialize, gko::share, lend.
//
// Kernels exist only for HipExecutor (the MI355X backend).  The host executors
// are memory spaces: running a kernel on them throws gko::NotCompiled, exactly
// what a Ginkgo build without the reference/omp modules does
// (core/device_hooks/common_kernels.inc.cpp:94).  Only fp64 values / int32
// indices are instantiated (north_star scope).
#ifndef GKOMI_GINKGO_HPP_
#define GKOMI_GINKGO_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <cstdlib>
#include <functional>
#include <initializer_list>
#include <iomanip>
#include <iostream>
#include <istream>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/gkomi.h"

namespace gko {

using size_type = std::size_t;
using int32 = std::int32_t;
using int64 = std::int64_t;
using uint8 = std::uint8_t;
using uint64 = std::uint64_t;
using uint32 = std::uint32_t;
template <typename T>
using remove_complex = T;
template <typename T>
constexpr T zero() { return T{}; }
template <typename T>
constexpr T one() { return T(1); }

// ---- exceptions (include/ginkgo/core/base/exception.hpp:86-632) -------------
class Error : public std::exception {
public:
    Error(const std::string& file, int line, const std::string& what)
        : what_(file + ":" + std::to_string(line) + ": " + what) {}
    const char* what() const noexcept override { return what_.c_str(); }
private:
    std::string what_;
};
#define GKOMI_DEFINE_ERROR(Name)                                              \
    class Name : public Error {                                               \
    public:                                                                   \
        Name(const std::string& f, int l, const std::string& w) : Error(f, l, #Name ": " + w) {} \
    }
GKOMI_DEFINE_ERROR(NotImplemented);
GKOMI_DEFINE_ERROR(NotCompiled);
GKOMI_DEFINE_ERROR(NotSupported);
GKOMI_DEFINE_ERROR(HipError);
GKOMI_DEFINE_ERROR(DimensionMismatch);
GKOMI_DEFINE_ERROR(BadDimension);
GKOMI_DEFINE_ERROR(AllocationError);
GKOMI_DEFINE_ERROR(StreamError);
GKOMI_DEFINE_ERROR(ValueMismatch);
#undef GKOMI_DEFINE_ERROR
#define GKO_NOT_COMPILED(module) throw ::gko::NotCompiled(__FILE__, __LINE__, #module " kernels are not part of this backend")
#define GKO_NOT_SUPPORTED(what) throw ::gko::NotSupported(__FILE__, __LINE__, what)
#define GKO_NOT_IMPLEMENTED throw ::gko::NotImplemented(__FILE__, __LINE__, __func__)

namespace detail {
// translate a C-ABI return code into the gko::Error family
inline void check(int code, const char* file, int line, const char* call)
{
    if (code == 0) return;
    const std::string msg = std::string(call) + ": " + gkomi_error_string(code);
    if (code > 0) throw HipError(file, line, msg);
    if (code == GKOMI_ENOTSUPPORTED) throw NotSupported(file, line, msg);
    if (code == GKOMI_ENOTIMPL) throw NotImplemented(file, line, msg);
    if (code == GKOMI_EINVAL) throw BadDimension(file, line, msg);
    throw Error(file, line, msg);
}
}  // namespace detail

class Executor;
// ---- logging hook (include/ginkgo/core/log/logger.hpp: operation_launched / operation_completed,
// fired by Executor::run around every kernel, executor.hpp:1153-1158).  Here every C-ABI call of
// the mirror is such an operation; its name is the entry point.  Loggers are attached with
// Executor::add_logger like in the reference; GKOMI_ROCTX=1 additionally brackets every operation
// with a roctx range for rocprofv3 --marker-trace.
namespace log {
class Logger {
public:
    virtual ~Logger() = default;
    virtual void on_operation_launched(const Executor*, const char* /*operation*/) const {}
    virtual void on_operation_completed(const Executor*, const char* /*operation*/) const {}
};
}  // namespace log

namespace detail {
struct logger_registry {
    std::vector<std::pair<const Executor*, std::shared_ptr<const log::Logger>>> loggers;
    bool roctx{false};
    logger_registry()
    {
        const char* v = std::getenv("GKOMI_ROCTX");
        roctx = v != nullptr && v[0] == '1' && gkomi_roctx_available();
    }
    bool active() const { return roctx || !loggers.empty(); }
};
inline logger_registry& registry()
{
    static logger_registry r;
    return r;
}
// "gkomi_csr_spmv_f64_i32(nullptr, ...)" -> "gkomi_csr_spmv_f64_i32"
inline std::string operation_name(const char* call)
{
    std::string s(call);
    const auto p = s.find('(');
    return p == std::string::npos ? s : s.substr(0, p);
}
struct operation_scope {
    explicit operation_scope(const char* call)
    {
        if (!registry().active()) return;
        name_ = operation_name(call);
        if (registry().roctx) gkomi_roctx_push(name_.c_str());
        for (const auto& l : registry().loggers) l.second->on_operation_launched(l.first, name_.c_str());
        armed_ = true;
    }
    ~operation_scope()
    {
        if (!armed_) return;
        for (const auto& l : registry().loggers) l.second->on_operation_completed(l.first, name_.c_str());
        if (registry().roctx) gkomi_roctx_pop();
    }
    std::string name_;
    bool armed_{false};
};
}  // namespace detail
#define GKOMI_CALL(expr)                                                   \
    do {                                                                   \
        ::gko::detail::operation_scope gkomi_scope_(#expr);                \
        ::gko::detail::check((expr), __FILE__, __LINE__, #expr);           \
    } while (0)

// ---- dim / version ------------------------------------------------------------
template <size_type N = 2>
struct dim {
    std::array<size_type, N> d{};
    dim() = default;
    dim(size_type s) { d.fill(s); }
    dim(size_type r, size_type c) { d[0] = r; d[1] = c; }
    size_type& operator[](size_type i) { return d[i]; }
    const size_type& operator[](size_type i) const { return d[i]; }
    explicit operator bool() const { return d[0] != 0 && d[1] != 0; }
    friend bool operator==(const dim& a, const dim& b) { return a.d == b.d; }
    friend bool operator!=(const dim& a, const dim& b) { return !(a == b); }
};
inline dim<2> transpose(const dim<2>& x) { return {x[1], x[0]}; }

class version_info {
public:
    static const version_info& get() { static version_info v; return v; }
    friend std::ostream& operator<<(std::ostream& os, const version_info&)
    {
        return os << "This is the MI355X-native hot-path backend with the Ginkgo 1.5.0 interface\n"
                  << "    running with core module 1.5.0 (hot path only)\n"
                  << "    the hip  module is   " << gkomi_version() << "\n"
                  << "    the reference/omp/cuda/dpcpp modules are not compiled";
    }
};

// ---- executors ------------------------------------------------------------------
class Executor : public std::enable_shared_from_this<Executor> {
public:
    virtual ~Executor() = default;
    virtual std::shared_ptr<Executor> get_master() noexcept = 0;
    virtual std::shared_ptr<const Executor> get_master() const noexcept = 0;
    virtual void synchronize() const = 0;
    virtual bool is_device() const noexcept = 0;
    template <typename T>
    T* alloc(size_type n) const { return static_cast<T*>(this->raw_alloc(n * sizeof(T))); }
    void free(void* p) const noexcept { this->raw_free(p); }
    template <typename T>
    void copy_from(const Executor* src_exec, size_type n, const T* src, T* dst) const
    {
        if (n == 0) return;
        const size_type bytes = n * sizeof(T);
        if (!src_exec->is_device() && !this->is_device()) {
            std::memcpy(dst, src, bytes);
        } else {
            const int kind = src_exec->is_device() ? (this->is_device() ? 2 : 1) : 0;
            GKOMI_CALL(gkomi_raw_copy(dst, src, bytes, kind));
        }
    }
    template <typename T>
    void copy(size_type n, const T* src, T* dst) const { this->copy_from(this, n, src, dst); }
    template <typename T>
    T copy_val_to_host(const T* ptr) const
    {
        T out{};
        this->get_master()->copy_from(this, 1, ptr, &out);
        return out;
    }
    // log::EnableLogging (include/ginkgo/core/log/logger.hpp:560-620)
    void add_logger(std::shared_ptr<const log::Logger> logger) const { detail::registry().loggers.emplace_back(this, std::move(logger)); }
    void remove_logger(const log::Logger* logger) const
    {
        auto& v = detail::registry().loggers;
        v.erase(std::remove_if(v.begin(), v.end(), [&](const std::pair<const Executor*, std::shared_ptr<const log::Logger>>& e) {
                    return e.first == this && e.second.get() == logger; }), v.end());
    }
protected:
    virtual void* raw_alloc(size_type bytes) const = 0;
    virtual void raw_free(void* p) const noexcept = 0;
};

class OmpExecutor : public Executor {
public:
    static std::shared_ptr<OmpExecutor> create() { return std::shared_ptr<OmpExecutor>(new OmpExecutor()); }
    std::shared_ptr<Executor> get_master() noexcept override { return this->shared_from_this(); }
    std::shared_ptr<const Executor> get_master() const noexcept override { return this->shared_from_this(); }
    void synchronize() const override {}
    bool is_device() const noexcept override { return false; }
protected:
    OmpExecutor() = default;
    void* raw_alloc(size_type bytes) const override
    {
        void* p = bytes ? std::malloc(bytes) : nullptr;
        if (bytes && !p) throw AllocationError(__FILE__, __LINE__, "host allocation failed");
        return p;
    }
    void raw_free(void* p) const noexcept override { std::free(p); }
};

class ReferenceExecutor : public OmpExecutor {
public:
    static std::shared_ptr<ReferenceExecutor> create() { return std::shared_ptr<ReferenceExecutor>(new ReferenceExecutor()); }
protected:
    ReferenceExecutor() = default;
};

enum class allocation_mode { device, unified_global, unified_host };

class HipExecutor : public Executor {
public:
    static std::shared_ptr<HipExecutor> create(int device_id, std::shared_ptr<Executor> master,
                                               bool device_reset = false,
                                               allocation_mode = allocation_mode::device)
    {
        (void)device_reset;
        if (device_id < 0 || device_id >= get_num_devices()) {
            throw HipError(__FILE__, __LINE__, "invalid device id " + std::to_string(device_id));
        }
        GKOMI_CALL(gkomi_set_device(device_id));
        auto e = std::shared_ptr<HipExecutor>(new HipExecutor(device_id, std::move(master)));
        int64_t prop[4] = {};
        GKOMI_CALL(gkomi_device_properties(device_id, prop));
        e->num_multiprocessor_ = static_cast<int>(prop[0]);
        e->warp_size_ = static_cast<int>(prop[1]);
        return e;
    }
    static int get_num_devices()
    {
        int n = 0;
        GKOMI_CALL(gkomi_get_num_devices(&n));
        return n;
    }
    std::shared_ptr<Executor> get_master() noexcept override { return master_; }
    std::shared_ptr<const Executor> get_master() const noexcept override { return master_; }
    void synchronize() const override { GKOMI_CALL(gkomi_synchronize(nullptr)); }
    bool is_device() const noexcept override { return true; }
    int get_device_id() const noexcept { return device_id_; }
    int get_num_multiprocessor() const noexcept { return num_multiprocessor_; }
    int get_warp_size() const noexcept { return warp_size_; }
    int get_num_warps_per_sm() const noexcept { return 4; }  // num_pu_per_cu on AMD
    int get_num_warps() const noexcept { return num_multiprocessor_ * 4; }
protected:
    HipExecutor(int id, std::shared_ptr<Executor> master) : device_id_(id), master_(std::move(master)) {}
    void* raw_alloc(size_type bytes) const override
    {
        void* p = nullptr;
        const int rc = gkomi_raw_alloc(bytes, &p);
        if (rc) throw AllocationError(__FILE__, __LINE__, std::string("hip: ") + gkomi_error_string(rc));
        return p;
    }
    void raw_free(void* p) const noexcept override
    {
        // like the reference: a failing free is fatal, never an exception
        if (gkomi_raw_free(p) != 0) { std::cerr << "Unrecoverable HIP error on hipFree\n"; std::exit(1); }
    }
private:
    int device_id_;
    std::shared_ptr<Executor> master_;
    int num_multiprocessor_{0};
    int warp_size_{64};
};

// backends outside this build: same factory signatures, NotCompiled on use
class CudaExecutor {
public:
    static std::shared_ptr<Executor> create(int, std::shared_ptr<Executor>, bool = false,
                                            allocation_mode = allocation_mode::device) { GKO_NOT_COMPILED(cuda); }
    static int get_num_devices() { return 0; }
};
class DpcppExecutor {
public:
    static std::shared_ptr<Executor> create(int, std::shared_ptr<Executor>, std::string = "all") { GKO_NOT_COMPILED(dpcpp); }
    static int get_num_devices(std::string = "all") { return 0; }
};

namespace detail {
inline void require_device(const std::shared_ptr<const Executor>& exec, const char* what)
{
    if (!exec->is_device()) {
        throw NotCompiled(__FILE__, __LINE__, std::string(what) +
                          ": only the hip (MI355X) kernels are compiled; reference/omp kernels are not part of this backend");
    }
}
}  // namespace detail

// ---- array ----------------------------------------------------------------------
template <typename T>
class array {
public:
    array() = default;
    explicit array(std::shared_ptr<const Executor> exec, size_type n = 0) : exec_(std::move(exec)) { resize_and_reset(n); }
    array(std::shared_ptr<const Executor> exec, std::initializer_list<T> init) : exec_(std::move(exec))
    {
        resize_and_reset(init.size());
        std::vector<T> tmp(init);
        auto host = exec_->get_master();
        exec_->copy_from(host.get(), tmp.size(), tmp.data(), data_);
    }
    template <typename It>
    array(std::shared_ptr<const Executor> exec, It b, It e) : exec_(std::move(exec))
    {
        std::vector<T> tmp(b, e);
        resize_and_reset(tmp.size());
        exec_->copy_from(exec_->get_master().get(), tmp.size(), tmp.data(), data_);
    }
    array(std::shared_ptr<const Executor> exec, const array& other) : exec_(std::move(exec)) { *this = other; }
    array(const array& other) : exec_(other.exec_) { *this = other; }
    array(array&& other) noexcept { *this = std::move(other); }
    ~array() { clear(); }
    array& operator=(const array& other)
    {
        if (this == &other) return *this;
        if (!exec_) exec_ = other.exec_;
        resize_and_reset(other.n_);
        if (n_) exec_->copy_from(other.exec_.get(), n_, other.data_, data_);
        return *this;
    }
    array& operator=(array&& other) noexcept
    {
        if (this == &other) return *this;
        if (exec_ && other.exec_ && exec_ != other.exec_) { *this = static_cast<const array&>(other); return *this; }
        clear();
        exec_ = other.exec_; data_ = other.data_; n_ = other.n_; owns_ = other.owns_;
        other.data_ = nullptr; other.n_ = 0;
        return *this;
    }
    static array view(std::shared_ptr<const Executor> exec, size_type n, T* data)
    {
        array a; a.exec_ = std::move(exec); a.data_ = data; a.n_ = n; a.owns_ = false; return a;
    }
    void resize_and_reset(size_type n)
    {
        if (n == n_) return;
        clear();
        if (n) data_ = exec_->template alloc<T>(n);
        n_ = n; owns_ = true;
    }
    void clear() noexcept
    {
        if (owns_ && data_ && exec_) exec_->free(data_);
        data_ = nullptr; n_ = 0;
    }
    void fill(T v)
    {
        std::vector<T> tmp(n_, v);
        exec_->copy_from(exec_->get_master().get(), n_, tmp.data(), data_);
    }
    void set_executor(std::shared_ptr<const Executor> exec)
    {
        if (exec == exec_) return;
        array tmp(exec, *this);  // copy of the data in the new memory space
        clear();
        exec_ = std::move(exec);
        data_ = tmp.data_; n_ = tmp.n_; owns_ = true;
        tmp.data_ = nullptr; tmp.n_ = 0;
    }
    T* get_data() noexcept { return data_; }
    const T* get_const_data() const noexcept { return data_; }
    size_type get_num_elems() const noexcept { return n_; }
    std::shared_ptr<const Executor> get_executor() const noexcept { return exec_; }
    std::vector<T> to_host() const
    {
        std::vector<T> out(n_);
        if (n_) exec_->get_master()->copy_from(exec_.get(), n_, data_, out.data());
        return out;
    }
private:
    std::shared_ptr<const Executor> exec_;
    T* data_{nullptr};
    size_type n_{0};
    bool owns_{true};
};
template <typename T>
array<T> make_array_view(std::shared_ptr<const Executor> exec, size_type n, T* data) { return array<T>::view(std::move(exec), n, data); }

// ---- pointer helpers --------------------------------------------------------------
template <typename T>
T* lend(const std::unique_ptr<T>& p) { return p.get(); }
template <typename T>
T* lend(const std::shared_ptr<T>& p) { return p.get(); }
template <typename T>
T* lend(T* p) { return p; }
template <typename T>
std::shared_ptr<T> share(std::unique_ptr<T>&& p) { return std::shared_ptr<T>(std::move(p)); }
template <typename T>
std::shared_ptr<T> share(std::shared_ptr<T> p) { return p; }
template <typename T>
std::unique_ptr<T> give(std::unique_ptr<T>&& p) { return std::move(p); }
template <typename T, typename U>
T* as(U* p)
{
    auto r = dynamic_cast<T*>(p);
    if (!r) throw NotSupported(__FILE__, __LINE__, "object cannot be cast to the requested type");
    return r;
}

// ---- matrix_data + MatrixMarket I/O (include/ginkgo/core/base/mtx_io.hpp) -----------
template <typename V = double, typename I = int32>
struct matrix_data {
    struct nonzero_type {
        nonzero_type() = default;
        nonzero_type(I r, I c, V v) : row(r), column(c), value(v) {}
        I row{};
        I column{};
        V value{};
    };
    matrix_data() = default;
    explicit matrix_data(dim<2> size_) : size(size_) {}  // include/ginkgo/core/base/matrix_data.hpp:159 (no fill value: empty)
    dim<2> size;
    std::vector<nonzero_type> nonzeros;
    void ensure_row_major_order()
    {
        std::stable_sort(nonzeros.begin(), nonzeros.end(), [](const nonzero_type& a, const nonzero_type& b) {
            return std::tie(a.row, a.column) < std::tie(b.row, b.column);
        });
    }
    // include/ginkgo/core/base/matrix_data.hpp: remove_zeros / sum_duplicates
    void remove_zeros()
    {
        nonzeros.erase(std::remove_if(nonzeros.begin(), nonzeros.end(), [](const nonzero_type& e) { return e.value == V{}; }), nonzeros.end());
    }
    void sum_duplicates()
    {
        ensure_row_major_order();
        std::vector<nonzero_type> out;
        for (const auto& e : nonzeros) {
            if (out.empty() || out.back().row != e.row || out.back().column != e.column) out.push_back({e.row, e.column, V{}});
            out.back().value += e.value;
        }
        nonzeros = std::move(out);
    }
};

// ---- device_matrix_data (include/ginkgo/core/base/device_matrix_data.hpp:63-270) -------
// SoA triplets in the executor's memory; sort / de-duplicate / drop zeros run on
// the device (core/base/device_matrix_data.cpp:115-141 -> csrc/assembly.hip).
template <typename V = double, typename I = int32>
class device_matrix_data {
public:
    using value_type = V;
    using index_type = I;
    using host_type = matrix_data<V, I>;
    struct arrays { array<I> row_idxs; array<I> col_idxs; array<V> values; };
    explicit device_matrix_data(std::shared_ptr<const Executor> exec, dim<2> size = {}, size_type num_entries = 0)
        : size_(size), row_idxs_(exec, num_entries), col_idxs_(exec, num_entries), values_(exec, num_entries) {}
    device_matrix_data(std::shared_ptr<const Executor> exec, const device_matrix_data& data)
        : size_(data.size_), row_idxs_(exec, data.row_idxs_), col_idxs_(exec, data.col_idxs_), values_(exec, data.values_) {}
    device_matrix_data(dim<2> size, array<I> row_idxs, array<I> col_idxs, array<V> values)
        : size_(size), row_idxs_(std::move(row_idxs)), col_idxs_(std::move(col_idxs)), values_(std::move(values))
    {
        if (values_.get_num_elems() != row_idxs_.get_num_elems() || values_.get_num_elems() != col_idxs_.get_num_elems())
            throw BadDimension(__FILE__, __LINE__, "device_matrix_data: array sizes differ");
    }
    host_type copy_to_host() const
    {
        host_type result;
        result.size = size_;
        auto r = row_idxs_.to_host(); auto c = col_idxs_.to_host(); auto v = values_.to_host();
        result.nonzeros.resize(v.size());
        for (size_type i = 0; i < v.size(); ++i) result.nonzeros[i] = {r[i], c[i], v[i]};
        return result;
    }
    static device_matrix_data create_from_host(std::shared_ptr<const Executor> exec, const host_type& data)
    {
        std::vector<I> r(data.nonzeros.size()), c(data.nonzeros.size());
        std::vector<V> v(data.nonzeros.size());
        for (size_type i = 0; i < v.size(); ++i) { r[i] = data.nonzeros[i].row; c[i] = data.nonzeros[i].column; v[i] = data.nonzeros[i].value; }
        return device_matrix_data(data.size, array<I>(exec, r.begin(), r.end()), array<I>(exec, c.begin(), c.end()), array<V>(exec, v.begin(), v.end()));
    }
    void sort_row_major()
    {
        auto exec = device("components::sort_row_major");
        array<char> ws(exec, gkomi_matrix_data_workspace_bytes(get_num_elems()));
        GKOMI_CALL(gkomi_matrix_data_sort_row_major_f64_i32(nullptr, get_num_elems(), row_idxs_.get_data(), col_idxs_.get_data(), values_.get_data(),
                                                            ws.get_data(), ws.get_num_elems()));
    }
    void remove_zeros() { compact("components::remove_zeros", gkomi_matrix_data_remove_zeros_f64_i32); }
    void sum_duplicates()
    {
        sort_row_major();
        compact("components::sum_duplicates", gkomi_matrix_data_sum_duplicates_f64_i32);
    }
    std::shared_ptr<const Executor> get_executor() const { return values_.get_executor(); }
    dim<2> get_size() const { return size_; }
    size_type get_num_elems() const { return values_.get_num_elems(); }
    I* get_row_idxs() { return row_idxs_.get_data(); }
    const I* get_const_row_idxs() const { return row_idxs_.get_const_data(); }
    I* get_col_idxs() { return col_idxs_.get_data(); }
    const I* get_const_col_idxs() const { return col_idxs_.get_const_data(); }
    V* get_values() { return values_.get_data(); }
    const V* get_const_values() const { return values_.get_const_data(); }
    void resize_and_reset(size_type n) { row_idxs_.resize_and_reset(n); col_idxs_.resize_and_reset(n); values_.resize_and_reset(n); }
    void resize_and_reset(dim<2> new_size, size_type n) { size_ = new_size; resize_and_reset(n); }
    arrays empty_out()
    {
        arrays result{std::move(row_idxs_), std::move(col_idxs_), std::move(values_)};
        size_ = {};
        return result;
    }
private:
    std::shared_ptr<const Executor> device(const char* what) const
    {
        auto exec = get_executor();
        detail::require_device(exec, what);
        return exec;
    }
    template <typename Fn>
    void compact(const char* what, Fn kernel)
    {
        auto exec = device(what);
        const size_type n = get_num_elems();
        array<I> r(exec, n), c(exec, n);
        array<V> v(exec, n);
        array<char> ws(exec, gkomi_matrix_data_workspace_bytes(n));
        int64_t kept = 0;
        GKOMI_CALL(kernel(nullptr, n, row_idxs_.get_const_data(), col_idxs_.get_const_data(), values_.get_const_data(), r.get_data(), c.get_data(),
                          v.get_data(), ws.get_data(), ws.get_num_elems(), &kept));
        if (static_cast<size_type>(kept) == n) return;  // nothing removed: keep the arrays (no reallocation, as the reference)
        array<I> r2(exec, kept), c2(exec, kept);
        array<V> v2(exec, kept);
        if (kept) {
            exec->copy_from(exec.get(), kept, r.get_const_data(), r2.get_data());
            exec->copy_from(exec.get(), kept, c.get_const_data(), c2.get_data());
            exec->copy_from(exec.get(), kept, v.get_const_data(), v2.get_data());
        }
        row_idxs_ = std::move(r2); col_idxs_ = std::move(c2); values_ = std::move(v2);
    }
    dim<2> size_;
    array<I> row_idxs_;
    array<I> col_idxs_;
    array<V> values_;
};

template <typename V = double, typename I = int32>
matrix_data<V, I> read_raw(std::istream& is)
{
    std::string line;
    if (!std::getline(is, line)) throw StreamError(__FILE__, __LINE__, "empty MatrixMarket stream");
    std::istringstream hdr(line);
    std::string banner, object, layout, field, symmetry;
    hdr >> banner >> object >> layout >> field >> symmetry;
    std::transform(layout.begin(), layout.end(), layout.begin(), ::tolower);
    std::transform(field.begin(), field.end(), field.begin(), ::tolower);
    std::transform(symmetry.begin(), symmetry.end(), symmetry.begin(), ::tolower);
    if (banner != "%%MatrixMarket") throw StreamError(__FILE__, __LINE__, "not a MatrixMarket header");
    while (std::getline(is, line) && (line.empty() || line[0] == '%')) {}
    std::istringstream dims(line);
    matrix_data<V, I> data;
    if (layout == "array") {
        size_type r, c;
        dims >> r >> c;
        data.size = dim<2>(r, c);
        for (size_type j = 0; j < c; ++j) {
            for (size_type i = 0; i < r; ++i) {
                double v;
                if (!(is >> v)) throw StreamError(__FILE__, __LINE__, "truncated array data");
                if (v != 0.0) data.nonzeros.push_back({static_cast<I>(i), static_cast<I>(j), static_cast<V>(v)});
            }
        }
    } else {
        size_type r, c, nnz;
        dims >> r >> c >> nnz;
        data.size = dim<2>(r, c);
        for (size_type k = 0; k < nnz; ++k) {
            long long i, j;
            double v = 1.0;
            is >> i >> j;
            if (field != "pattern") is >> v;
            if (!is) throw StreamError(__FILE__, __LINE__, "truncated coordinate data");
            data.nonzeros.push_back({static_cast<I>(i - 1), static_cast<I>(j - 1), static_cast<V>(v)});
            if (symmetry == "symmetric" && i != j) {
                data.nonzeros.push_back({static_cast<I>(j - 1), static_cast<I>(i - 1), static_cast<V>(v)});
            }
        }
    }
    data.ensure_row_major_order();
    return data;
}

// ---- LinOp ------------------------------------------------------------------------
namespace matrix {
template <typename V>
class Dense;
}

class LinOp {
public:
    virtual ~LinOp() = default;
    // lin_op.hpp:158-224: validate, then apply_impl
    LinOp* apply(const LinOp* b, LinOp* x)
    {
        this->validate(b, x);
        this->apply_impl(b, x);
        return this;
    }
    const LinOp* apply(const LinOp* b, LinOp* x) const
    {
        this->validate(b, x);
        this->apply_impl(b, x);
        return this;
    }
    const LinOp* apply(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        this->validate(b, x);
        if (alpha->get_size() != dim<2>(1, 1) || beta->get_size() != dim<2>(1, 1)) {
            throw DimensionMismatch(__FILE__, __LINE__, "alpha and beta must be 1x1");
        }
        this->apply_impl(alpha, b, beta, x);
        return this;
    }
    const dim<2>& get_size() const noexcept { return size_; }
    std::shared_ptr<const Executor> get_executor() const noexcept { return exec_; }
protected:
    LinOp(std::shared_ptr<const Executor> exec, const dim<2>& size = dim<2>{}) : exec_(std::move(exec)), size_(size) {}
    void set_size(const dim<2>& s) { size_ = s; }
    virtual void apply_impl(const LinOp* b, LinOp* x) const = 0;
    virtual void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const = 0;
    void validate(const LinOp* b, const LinOp* x) const
    {
        // GKO_ASSERT_CONFORMANT / EQUAL_ROWS / EQUAL_COLS (lin_op.hpp:323-345)
        if (size_[1] != b->get_size()[0] || size_[0] != x->get_size()[0] || b->get_size()[1] != x->get_size()[1]) {
            std::ostringstream os;
            os << "apply: operator " << size_[0] << "x" << size_[1] << ", b " << b->get_size()[0] << "x" << b->get_size()[1]
               << ", x " << x->get_size()[0] << "x" << x->get_size()[1];
            throw DimensionMismatch(__FILE__, __LINE__, os.str());
        }
    }
    std::shared_ptr<const Executor> exec_;
    dim<2> size_;
};

class LinOpFactory {
public:
    virtual ~LinOpFactory() = default;
    virtual std::unique_ptr<LinOp> generate_impl(std::shared_ptr<const LinOp> input) const = 0;
    std::shared_ptr<const Executor> get_executor() const noexcept { return exec_; }
protected:
    explicit LinOpFactory(std::shared_ptr<const Executor> exec) : exec_(std::move(exec)) {}
    std::shared_ptr<const Executor> exec_;
};

// ---- matrix formats -----------------------------------------------------------------
namespace matrix {

template <typename V = double>
class Dense : public LinOp {
public:
    using value_type = V;
    static std::unique_ptr<Dense> create(std::shared_ptr<const Executor> exec, const dim<2>& size = dim<2>{}, size_type stride = 0)
    {
        return std::unique_ptr<Dense>(new Dense(std::move(exec), size, stride ? stride : size[1]));
    }
    // view over existing memory (Dense::create(exec, size, array view, stride))
    static std::unique_ptr<Dense> create(std::shared_ptr<const Executor> exec, const dim<2>& size, array<V> values, size_type stride)
    {
        auto d = std::unique_ptr<Dense>(new Dense(std::move(exec), dim<2>{}, 0));
        d->values_ = std::move(values);
        d->stride_ = stride;
        d->set_size(size);
        return d;
    }
    V* get_values() noexcept { return values_.get_data(); }
    const V* get_const_values() const noexcept { return values_.get_const_data(); }
    size_type get_stride() const noexcept { return stride_; }
    size_type get_num_stored_elements() const noexcept { return values_.get_num_elems(); }
    // host access only (like the reference: undefined on device memory)
    V& at(size_type r, size_type c = 0) { return values_.get_data()[r * stride_ + c]; }
    V at(size_type r, size_type c = 0) const { return values_.get_const_data()[r * stride_ + c]; }

    void fill(V v)
    {
        if (on_device()) GKOMI_CALL(gkomi_dense_fill_f64(nullptr, rows(), cols(), get_values(), stride_, v));
        else for (size_type i = 0; i < size_[0]; ++i) for (size_type j = 0; j < size_[1]; ++j) at(i, j) = v;
    }
    void copy_from(const Dense* other)
    {
        if (other->get_size() != size_ || other->get_stride() != stride_) {
            values_.resize_and_reset(other->get_size()[0] * other->get_stride());
            stride_ = other->get_stride();
            set_size(other->get_size());
        }
        exec_->copy_from(other->get_executor().get(), values_.get_num_elems(), other->get_const_values(), get_values());
    }
    std::unique_ptr<Dense> clone(std::shared_ptr<const Executor> exec = nullptr) const
    {
        auto d = Dense::create(exec ? exec : exec_, size_, stride_);
        d->copy_from(this);
        return d;
    }
    void scale(const Dense* alpha) { kernel("dense::scale"); GKOMI_CALL(gkomi_dense_scale_f64(nullptr, rows(), cols(), alpha->get_const_values(), alpha->cols(), get_values(), stride_)); }
    void inv_scale(const Dense* alpha) { kernel("dense::inv_scale"); GKOMI_CALL(gkomi_dense_inv_scale_f64(nullptr, rows(), cols(), alpha->get_const_values(), alpha->cols(), get_values(), stride_)); }
    void add_scaled(const Dense* alpha, const Dense* b)
    {
        kernel("dense::add_scaled"); same_size(b);
        GKOMI_CALL(gkomi_dense_add_scaled_f64(nullptr, rows(), cols(), alpha->get_const_values(), alpha->cols(), b->get_const_values(), b->get_stride(), get_values(), stride_));
    }
    void sub_scaled(const Dense* alpha, const Dense* b)
    {
        kernel("dense::sub_scaled"); same_size(b);
        GKOMI_CALL(gkomi_dense_sub_scaled_f64(nullptr, rows(), cols(), alpha->get_const_values(), alpha->cols(), b->get_const_values(), b->get_stride(), get_values(), stride_));
    }
    void compute_dot(const Dense* b, Dense* result) const
    {
        kernel("dense::compute_dot"); same_size(b); result_size(result);
        array<char> tmp(exec_, gkomi_dense_reduction_workspace_bytes(rows(), cols()) + 8);
        GKOMI_CALL(gkomi_dense_compute_dot_f64(nullptr, rows(), cols(), get_const_values(), stride_, b->get_const_values(), b->get_stride(), result->get_values(), tmp.get_data(), tmp.get_num_elems()));
    }
    void compute_conj_dot(const Dense* b, Dense* result) const { compute_dot(b, result); }
    void compute_norm2(Dense* result) const
    {
        kernel("dense::compute_norm2"); result_size(result);
        array<char> tmp(exec_, gkomi_dense_reduction_workspace_bytes(rows(), cols()) + 8);
        GKOMI_CALL(gkomi_dense_compute_norm2_f64(nullptr, rows(), cols(), get_const_values(), stride_, result->get_values(), tmp.get_data(), tmp.get_num_elems()));
    }
    void read(const matrix_data<V, int32>& data)
    {
        values_.set_executor(exec_->get_master());
        values_.resize_and_reset(data.size[0] * data.size[1]);
        stride_ = data.size[1];
        set_size(data.size);
        std::fill_n(values_.get_data(), values_.get_num_elems(), V{});
        for (const auto& e : data.nonzeros) values_.get_data()[e.row * stride_ + e.column] = e.value;
        values_.set_executor(exec_);
    }
    void write(matrix_data<V, int32>& data) const
    {
        auto host = values_.to_host();
        data.size = size_;
        data.nonzeros.clear();
        for (size_type i = 0; i < size_[0]; ++i) for (size_type j = 0; j < size_[1]; ++j) {
            data.nonzeros.push_back({static_cast<int32>(i), static_cast<int32>(j), host[i * stride_ + j]});
        }
    }
    int64_t rows() const { return static_cast<int64_t>(size_[0]); }
    int64_t cols() const { return static_cast<int64_t>(size_[1]); }
protected:
    Dense(std::shared_ptr<const Executor> exec, const dim<2>& size, size_type stride)
        : LinOp(exec, size), values_(exec, size[0] * stride), stride_(stride) {}
    bool on_device() const { return exec_->is_device(); }
    void kernel(const char* name) const { detail::require_device(exec_, name); }
    void same_size(const Dense* b) const { if (b->get_size() != size_) throw DimensionMismatch(__FILE__, __LINE__, "operands differ in size"); }
    void result_size(const Dense* r) const { if (r->get_size() != dim<2>(1, size_[1])) throw DimensionMismatch(__FILE__, __LINE__, "result must be 1 x #columns"); }
    void apply_impl(const LinOp*, LinOp*) const override { GKO_NOT_IMPLEMENTED; }  // dense GEMM: off the hot path
    void apply_impl(const LinOp*, const LinOp*, const LinOp*, LinOp*) const override { GKO_NOT_IMPLEMENTED; }
    array<V> values_;
    size_type stride_;
};

namespace detail_fmt {
inline const Dense<double>* dense(const LinOp* op) { return as<const Dense<double>>(op); }
inline Dense<double>* dense(LinOp* op) { return as<Dense<double>>(op); }
}  // namespace detail_fmt

template <typename V, typename I>
class Coo;
template <typename V, typename I>
class Ell;
template <typename V, typename I>
class Sellp;
template <typename V, typename I>
class Hybrid;
template <typename V, typename I>
class CsrBuilder;

namespace detail_abi {
// the C-ABI entry points of the two index types this backend instantiates for the SpMV path
// (GKO_INSTANTIATE_FOR_EACH_VALUE_AND_INDEX_TYPE, include/ginkgo/core/base/types.hpp:544-560: <double, int32> and
// <double, int64>; the other format / factorisation kernels exist for int32 only and do not compile for int64)
template <typename I>
struct csr_abi;
template <>
struct csr_abi<int32> {
    static constexpr auto spmv_srow = &gkomi_csr_spmv_srow_f64_i32;
    static constexpr auto make_srow = &gkomi_csr_make_srow_i32;
    static constexpr auto max_row_nnz = &gkomi_csr_max_row_nnz_i32;
    static constexpr auto idxs_to_ptrs = &gkomi_convert_idxs_to_ptrs_i32;
    // the column-pattern statistic of the strategy objects (GKOMI_CSR_COLBLOCK or 0): blocking, once per matrix
    static int gather_flags(int64_t ncols, int64_t nnz, const int32* col_idxs, double* scratch, int64_t* footprint)
    {
        int flags = 0;
        GKOMI_CALL(gkomi_csr_analyse_gather_i32(nullptr, ncols, nnz, col_idxs, scratch, &flags, footprint));
        return flags;
    }
    // the column-partitioned copy of the "gkomi_partitioned" strategy (gkomi_csr_colpart_*): nullptr = does not pay
    static gkomi_csr_colpart* colpart_create(std::shared_ptr<const Executor> exec, int64_t nrows, int64_t ncols, int64_t nnz, const int32* rp,
                                             const int32* ci, const double* v, array<char>& plan)
    {
        if (gkomi_csr_colpart_blocks_for(nrows, ncols, nnz) == 0) return nullptr;
        plan = array<char>(exec, gkomi_csr_colpart_plan_bytes(nrows, nnz, 0));
        gkomi_csr_colpart* h = nullptr;  // (0 blocks: the analysis times two block counts and keeps the faster)
        const int rc = gkomi_csr_colpart_create_f64_i32(nullptr, nrows, ncols, nnz, rp, ci, v, 0, plan.get_data(), plan.get_num_elems(), &h);
        if (rc == GKOMI_ENOTSUPPORTED) {  // timed against the matrix's own kernel: the copy does not pay
            plan = array<char>(exec, 0);
            return nullptr;
        }
        GKOMI_CALL(rc);
        return h;
    }
};
template <>
struct csr_abi<int64> {
    static constexpr auto spmv_srow = &gkomi_csr_spmv_srow_f64_i64;
    static constexpr auto make_srow = &gkomi_csr_make_srow_i64;
    static constexpr auto max_row_nnz = &gkomi_csr_max_row_nnz_i64;
    static constexpr auto idxs_to_ptrs = &gkomi_convert_idxs_to_ptrs_i64;
    static int gather_flags(int64_t, int64_t, const int64*, double*, int64_t* footprint)  // (column windows: int32 kernels only)
    {
        *footprint = 0;
        return 0;
    }
    static gkomi_csr_colpart* colpart_create(std::shared_ptr<const Executor>, int64_t, int64_t, int64_t, const int64*, const int64*, const double*,
                                             array<char>&)
    {
        return nullptr;
    }
};
}  // namespace detail_abi

template <typename V = double, typename I = int32>
class Csr : public LinOp {
    friend class CsrBuilder<V, I>;
public:
    using value_type = V;
    using index_type = I;
    using mat_data = matrix_data<V, I>;
    // kernel selection objects (include/ginkgo/core/matrix/csr.hpp:170-705)
    class strategy_type {
    public:
        explicit strategy_type(std::string name, int code) : name_(std::move(name)), code_(code) {}
        virtual ~strategy_type() = default;
        const std::string& get_name() const { return name_; }
        int get_code() const { return code_; }
    private:
        std::string name_;
        int code_;
    };
    struct classical : strategy_type { classical() : strategy_type("classical", GKOMI_CSR_VECTOR) {} };
    struct load_balance : strategy_type {
        load_balance(int64_t = 0) : strategy_type("load_balance", GKOMI_CSR_BALANCED) {}
        // include/ginkgo/core/matrix/csr.hpp:355-372: built from an executor (its warp count sized the reference's srow)
        template <typename Exec>
        load_balance(std::shared_ptr<Exec>) : load_balance(int64_t{0}) {}
    };
    struct merge_path : strategy_type { merge_path() : strategy_type("merge_path", GKOMI_CSR_STREAM) {} };
    struct automatical : strategy_type {
        automatical(int64_t = 0) : strategy_type("automatical", GKOMI_CSR_AUTO) {}
        template <typename Exec>
        automatical(std::shared_ptr<Exec>) : automatical(int64_t{0}) {}  // csr.hpp:571-588
    };
    // the vendor-library strategy (csr.hpp:299-330): no hipSPARSE behind this backend, the automatic choice serves it
    struct sparselib : strategy_type { sparselib() : strategy_type("sparselib", GKOMI_CSR_AUTO) {} };
    // An analysis-based strategy of this backend (the role hipSPARSE's analysis plays behind `sparselib` in the reference):
    // the matrix keeps a column-partitioned COPY (gkomi_csr_colpart_*, csrc/csr_colpart.hip) when its column pattern is
    // scattered and its shape fits, and applies one-column products through it (1.3-1.6 x on uniformly random / power-law
    // patterns of ~1 M columns; tolerance parity like load_balance); otherwise the automatic kernels.  The copy holds
    // values: it is re-gathered after get_values() (non-const) was asked for.
    struct gkomi_partitioned : strategy_type { gkomi_partitioned() : strategy_type("gkomi_partitioned", GKOMI_CSR_AUTO) {} };
    struct cusparse : strategy_type { cusparse() : strategy_type("cusparse", GKOMI_CSR_AUTO) {} };

    static std::unique_ptr<Csr> create(std::shared_ptr<const Executor> exec, const dim<2>& size = dim<2>{}, size_type nnz = 0,
                                       std::shared_ptr<strategy_type> strategy = std::make_shared<automatical>())
    {
        return std::unique_ptr<Csr>(new Csr(std::move(exec), size, nnz, std::move(strategy)));
    }
    // include/ginkgo/core/matrix/csr.hpp:1036-1039: an empty matrix with a strategy
    static std::unique_ptr<Csr> create(std::shared_ptr<const Executor> exec, std::shared_ptr<strategy_type> strategy)
    {
        return std::unique_ptr<Csr>(new Csr(std::move(exec), dim<2>{}, 0, std::move(strategy)));
    }
    V* get_values() noexcept { colpart_dirty_ = true; return values_.get_data(); }
    const V* get_const_values() const noexcept { return values_.get_const_data(); }
    I* get_col_idxs() noexcept { return col_idxs_.get_data(); }
    const I* get_const_col_idxs() const noexcept { return col_idxs_.get_const_data(); }
    I* get_row_ptrs() noexcept { srow_valid_ = false; return row_ptrs_.get_data(); }  // the caller may rewrite them
    const I* get_const_row_ptrs() const noexcept { return row_ptrs_.get_const_data(); }
    size_type get_num_stored_elements() const noexcept { return values_.get_num_elems(); }
    std::shared_ptr<strategy_type> get_strategy() const noexcept { return strategy_; }
    void set_strategy(std::shared_ptr<strategy_type> s) { strategy_ = std::move(s); }
    int64_t get_max_row_nnz() const noexcept { return max_row_nnz_; }

    // Csr::read(matrix_data) (core/matrix/csr.cpp:438-470)
    void read(const mat_data& data)
    {
        const size_type nnz = data.nonzeros.size();
        std::vector<I> rp(data.size[0] + 1, 0), ci(nnz);
        std::vector<V> v(nnz);
        for (size_type k = 0; k < nnz; ++k) {
            rp[data.nonzeros[k].row + 1]++;
            ci[k] = data.nonzeros[k].column;
            v[k] = data.nonzeros[k].value;
        }
        int64_t mx = 0;
        for (size_type r = 0; r < data.size[0]; ++r) { mx = std::max<int64_t>(mx, rp[r + 1]); rp[r + 1] += rp[r]; }
        max_row_nnz_ = mx;
        auto host = exec_->get_master();
        row_ptrs_ = array<I>(exec_, rp.begin(), rp.end());
        col_idxs_ = array<I>(exec_, ci.begin(), ci.end());
        values_ = array<V>(exec_, v.begin(), v.end());
        set_size(data.size);
        invalidate_srow();
    }
    // Csr::read(device_matrix_data) (core/matrix/csr.cpp:445-470): takes the
    // arrays over as they are (row-major order is the caller's job) and builds
    // row_ptrs with convert_idxs_to_ptrs on the device
    using device_mat_data = device_matrix_data<V, I>;
    void read(const device_mat_data& data) { this->read(device_mat_data{exec_, data}); }
    void read(device_mat_data&& data)
    {
        detail::require_device(exec_, "csr::convert_idxs_to_ptrs");
        const auto size = data.get_size();
        auto arrays = data.empty_out();
        arrays.row_idxs.set_executor(exec_); arrays.col_idxs.set_executor(exec_); arrays.values.set_executor(exec_);
        row_ptrs_.resize_and_reset(size[0] + 1);
        set_size(size);
        values_ = std::move(arrays.values);
        col_idxs_ = std::move(arrays.col_idxs);
        array<char> ws(exec_, gkomi_prefix_sum_workspace_bytes(size[0] + 1));
        GKOMI_CALL(detail_abi::csr_abi<I>::idxs_to_ptrs(nullptr, arrays.row_idxs.get_const_data(), arrays.row_idxs.get_num_elems(), size[0], row_ptrs_.get_data(),
                                                    ws.get_data(), ws.get_num_elems()));
        max_row_nnz_ = -1;
        invalidate_srow();
    }
    void write(mat_data& data) const
    {
        auto rp = row_ptrs_.to_host(); auto ci = col_idxs_.to_host(); auto v = values_.to_host();
        data.size = size_;
        data.nonzeros.clear();
        for (size_type r = 0; r < size_[0]; ++r) for (I k = rp[r]; k < rp[r + 1]; ++k) data.nonzeros.push_back({static_cast<I>(r), ci[k], v[k]});
    }
    void convert_to(Coo<V, I>* result) const;
    void convert_to(Ell<V, I>* result) const;
    void convert_to(Sellp<V, I>* result) const;
    void convert_to(Hybrid<V, I>* result) const;
    std::unique_ptr<Csr> transpose() const
    {
        detail::require_device(exec_, "csr::transpose");
        auto t = Csr::create(exec_, gko::transpose(size_), get_num_stored_elements(), strategy_);
        array<char> ws(exec_, gkomi_csr_transpose_workspace_bytes(size_[1]));
        GKOMI_CALL(gkomi_csr_transpose_f64_i32(nullptr, size_[0], size_[1], get_num_stored_elements(), get_const_row_ptrs(), get_const_col_idxs(),
                                               get_const_values(), t->get_row_ptrs(), t->get_col_idxs(), t->get_values(), ws.get_data(), ws.get_num_elems()));
        t->max_row_nnz_ = -1;
        return t;
    }
    // csr::sort_by_column_index / is_sorted_by_column_index (core/matrix/csr.cpp)
    void sort_by_column_index()
    {
        detail::require_device(exec_, "csr::sort_by_column_index");
        GKOMI_CALL(gkomi_csr_sort_by_column_index_f64_i32(nullptr, size_[0], get_const_row_ptrs(), get_col_idxs(), get_values()));
    }
    bool is_sorted_by_column_index() const
    {
        detail::require_device(exec_, "csr::is_sorted_by_column_index");
        array<char> ws(exec_, 8);
        int sorted = 1;
        GKOMI_CALL(gkomi_csr_is_sorted_by_column_index_i32(nullptr, size_[0], get_const_row_ptrs(), get_const_col_idxs(), ws.get_data(), ws.get_num_elems(), &sorted));
        return sorted != 0;
    }
    // arrays handed over by kernels that size their own outputs
    void adopt(const dim<2>& size, array<I> rp, array<I> ci, array<V> v)
    {
        row_ptrs_ = std::move(rp); col_idxs_ = std::move(ci); values_ = std::move(v); set_size(size); max_row_nnz_ = -1;
        invalidate_srow();
    }
protected:
    Csr(std::shared_ptr<const Executor> exec, const dim<2>& size, size_type nnz, std::shared_ptr<strategy_type> strategy)
        : LinOp(exec, size), values_(exec, nnz), col_idxs_(exec, nnz), row_ptrs_(exec, size[0] + 1), strategy_(std::move(strategy)) {}
    void apply_impl(const LinOp* b, LinOp* x) const override { spmv(nullptr, b, nullptr, x); }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { spmv(alpha, b, beta, x); }
    void spmv(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const { spmv_as(V{}, alpha, b, beta, x); }
    // <float, int32>: csr::spmv / advanced_spmv of the single-precision instantiation (gkomi_csr_spmv_f32_i32); no srow
    void spmv_as(float, const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        detail::require_device(exec_, "csr::spmv");
        static_assert(!std::is_same<V, float>::value || std::is_same<I, int32>::value, "float: int32 indices");
        auto db = dynamic_cast<const Dense<float>*>(b);
        auto dx = dynamic_cast<Dense<float>*>(x);
        auto da = alpha ? dynamic_cast<const Dense<float>*>(alpha) : nullptr;
        auto dbeta = beta ? dynamic_cast<const Dense<float>*>(beta) : nullptr;
        if (!db || !dx || (alpha && !da) || (beta && !dbeta)) GKO_NOT_SUPPORTED("Csr<float>::apply takes Dense<float> operands");
        GKOMI_CALL(gkomi_csr_spmv_f32_i32(nullptr, size_[0], size_[1], db->get_size()[1], get_num_stored_elements(),
                                          reinterpret_cast<const int32_t*>(get_const_row_ptrs()), reinterpret_cast<const int32_t*>(get_const_col_idxs()),
                                          reinterpret_cast<const float*>(get_const_values()), db->get_const_values(), db->get_stride(), dx->get_values(),
                                          dx->get_stride(), da ? da->get_const_values() : nullptr, dbeta ? dbeta->get_const_values() : nullptr));
    }
    void spmv_as(double, const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        detail::require_device(exec_, "csr::spmv");
        auto db = detail_fmt::dense(b); auto dx = detail_fmt::dense(x);
        make_srow();
        if (colpart_ && db->cols() == 1) {
            if (colpart_dirty_) {
                GKOMI_CALL(gkomi_csr_colpart_refresh_f64(nullptr, colpart_.get(), get_const_values()));
                colpart_dirty_ = false;
            }
            GKOMI_CALL(gkomi_csr_colpart_spmv_f64(nullptr, colpart_.get(), db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride(),
                                                  alpha ? detail_fmt::dense(alpha)->get_const_values() : nullptr,
                                                  beta ? detail_fmt::dense(beta)->get_const_values() : nullptr));
            return;
        }
        GKOMI_CALL(detail_abi::csr_abi<I>::spmv_srow(nullptr, size_[0], size_[1], db->cols(), get_num_stored_elements(), get_const_row_ptrs(), get_const_col_idxs(),
                                                 get_const_values(), db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride(),
                                                 alpha ? detail_fmt::dense(alpha)->get_const_values() : nullptr,
                                                 beta ? detail_fmt::dense(beta)->get_const_values() : nullptr, strategy_->get_code() | gather_flags_, max_row_nnz_,
                                                 get_num_srow_elements() ? get_const_srow() : nullptr, srow_tile_));
    }
public:
    // Csr::srow_ / make_srow (csr.hpp:1139-1157, 1265-1266): the tile start rows of the nonzero-split
    // kernel, rebuilt whenever row_ptrs may have changed (read / adopt / conversions reset it)
    const I* get_const_srow() const noexcept { return srow_.get_const_data(); }
    size_type get_num_srow_elements() const noexcept { return srow_.get_num_elems(); }
    int64_t get_srow_tile() const noexcept { return srow_tile_; }
    void make_srow() const
    {
        if (srow_valid_ || !exec_->is_device()) return;
        const int64_t nnz = static_cast<int64_t>(get_num_stored_elements());
        if (nnz >= 2 && max_row_nnz_ < 0) {  // row statistics of the strategy objects (csr.hpp:526-705)
            array<I> mx(exec_, 1);
            GKOMI_CALL(detail_abi::csr_abi<I>::max_row_nnz(nullptr, size_[0], get_const_row_ptrs(), mx.get_data()));
            max_row_nnz_ = exec_->copy_val_to_host(mx.get_const_data());
        }
        srow_tile_ = gkomi_csr_srow_tile_for(static_cast<int64_t>(get_num_stored_elements()));
        if (nnz >= 2) {
            srow_ = array<I>(exec_, static_cast<size_type>(gkomi_csr_srow_entries(nnz, srow_tile_)));
            GKOMI_CALL(detail_abi::csr_abi<I>::make_srow(nullptr, size_[0], nnz, get_const_row_ptrs(), srow_tile_, srow_.get_data(),
                                                     static_cast<int64_t>(srow_.get_num_elems())));
        } else {
            srow_ = array<I>(exec_, 0);
        }
        // ... and where the gathers of b go: matrices whose long rows select the load-balanced kernel get its column
        // windows when a tile's gathers overflow an XCD's L2 (gkomi_csr_analyse_gather_i32; acts on that kernel only)
        gather_flags_ = 0;
        gather_footprint_ = 0;
        const int code = strategy_->get_code() & 0xff;
        if (nnz >= 2 && (code == GKOMI_CSR_AUTO || code == GKOMI_CSR_BALANCED)) {
            array<double> scratch(exec_, 2);
            gather_flags_ = detail_abi::csr_abi<I>::gather_flags(static_cast<int64_t>(size_[1]), nnz, get_const_col_idxs(), scratch.get_data(), &gather_footprint_);
        }
        // the analysis of the gkomi_partitioned strategy: a scattered column pattern (the statistic above) on a shape that fits
        colpart_.reset();
        if (strategy_->get_name() == "gkomi_partitioned" && gather_footprint_ > (int64_t{3} << 20)) {  // gathers beyond an L2
            colpart_.reset(detail_abi::csr_abi<I>::colpart_create(exec_, static_cast<int64_t>(size_[0]), static_cast<int64_t>(size_[1]), nnz, get_const_row_ptrs(),
                                                                  get_const_col_idxs(), get_const_values(), colpart_plan_));
            colpart_dirty_ = false;
        }
        srow_valid_ = true;
    }
    bool has_partitioned_copy() const noexcept { return static_cast<bool>(colpart_); }
    void invalidate_srow() const { srow_valid_ = false; }
protected:
    array<V> values_;
    array<I> col_idxs_;
    array<I> row_ptrs_;
    std::shared_ptr<strategy_type> strategy_;
    mutable int64_t max_row_nnz_{-1};
    mutable array<I> srow_;
    mutable int64_t srow_tile_{0};
    mutable int gather_flags_{0};
    mutable int64_t gather_footprint_{0};
    mutable bool srow_valid_{false};
    struct colpart_deleter {
        void operator()(gkomi_csr_colpart* h) const { gkomi_csr_colpart_destroy(h); }
    };
    mutable std::unique_ptr<gkomi_csr_colpart, colpart_deleter> colpart_;
    mutable array<char> colpart_plan_;
    mutable bool colpart_dirty_{false};
};

// core/matrix/csr_builder.hpp:47-83: intrusive access to a Csr's arrays for kernels that rebuild them
// (factorization::add_diagonal_elements); the srow is rebuilt when the builder goes away
template <typename V = double, typename I = int32>
class CsrBuilder {
public:
    array<I>& get_col_idx_array() { return matrix_->col_idxs_; }
    array<V>& get_value_array() { return matrix_->values_; }
    explicit CsrBuilder(Csr<V, I>* matrix) : matrix_{matrix} {}
    ~CsrBuilder()
    {
        matrix_->max_row_nnz_ = -1;
        matrix_->invalidate_srow();
    }
    CsrBuilder(const CsrBuilder&) = delete;
    CsrBuilder& operator=(const CsrBuilder&) = delete;
private:
    Csr<V, I>* matrix_;
};

template <typename V = double, typename I = int32>
class Coo : public LinOp {
public:
    static std::unique_ptr<Coo> create(std::shared_ptr<const Executor> exec, const dim<2>& size = dim<2>{}, size_type nnz = 0) { return std::unique_ptr<Coo>(new Coo(std::move(exec), size, nnz)); }
    V* get_values() noexcept { return values_.get_data(); }
    const V* get_const_values() const noexcept { return values_.get_const_data(); }
    I* get_col_idxs() noexcept { return col_idxs_.get_data(); }
    const I* get_const_col_idxs() const noexcept { return col_idxs_.get_const_data(); }
    // handing out writable row indices forgets what is known about their order
    I* get_row_idxs() noexcept { sorted_state_ = -1; return row_idxs_.get_data(); }
    const I* get_const_row_idxs() const noexcept { return row_idxs_.get_const_data(); }
    size_type get_num_stored_elements() const noexcept { return values_.get_num_elems(); }
    void resize(const dim<2>& size, size_type nnz) { values_.resize_and_reset(nnz); col_idxs_.resize_and_reset(nnz); row_idxs_.resize_and_reset(nnz); set_size(size); sorted_state_ = -1; }
    // row indices non-decreasing (checked on the device once per set of indices):
    // apply / apply2 then run the atomic-free kernels of csrc/coo_spmv.hip
    bool is_sorted_by_row() const { return sorted_workspace(1) != nullptr; }
    // x += A b  (Coo::apply2, include/ginkgo/core/matrix/coo.hpp)
    void apply2(const LinOp* b, LinOp* x) const { validate(b, x); run2(nullptr, b, x); }
    void apply2(const LinOp* alpha, const LinOp* b, LinOp* x) const { validate(b, x); run2(alpha, b, x); }
protected:
    Coo(std::shared_ptr<const Executor> exec, const dim<2>& size, size_type nnz) : LinOp(exec, size), values_(exec, nnz), col_idxs_(exec, nnz), row_idxs_(exec, nnz), sorted_ws_(exec) {}
    // workspace of the sorted kernels, or nullptr when the rows are not sorted (or misaligned)
    void* sorted_workspace(size_type nrhs) const
    {
        if (!exec_->is_device() || sorted_state_ == 0) return nullptr;
        const size_type nnz = get_num_stored_elements();
        const size_type need = std::max<size_type>(gkomi_coo_sorted_workspace_bytes(nnz, nrhs), 16);
        if (sorted_ws_.get_num_elems() < need) sorted_ws_.resize_and_reset(need);
        if (sorted_state_ < 0) {
            int flag = 0;
            int64_t longest = 0;
            GKOMI_CALL(gkomi_coo_analyse_rows_i32(nullptr, nnz, get_const_row_idxs(), sorted_ws_.get_data(), sorted_ws_.get_num_elems(), &flag, &longest));
            max_row_nnz_ = longest;  // capped at 65: rows of up to 64 nonzeros take the one-launch kernel
            const bool aligned = reinterpret_cast<std::uintptr_t>(get_const_values()) % 16 == 0 && reinterpret_cast<std::uintptr_t>(get_const_row_idxs()) % 8 == 0 &&
                                 reinterpret_cast<std::uintptr_t>(get_const_col_idxs()) % 8 == 0;
            sorted_state_ = flag && aligned ? 1 : 0;
        }
        return sorted_state_ == 1 ? sorted_ws_.get_data() : nullptr;
    }
    void run2(const LinOp* alpha, const LinOp* b, LinOp* x) const
    {
        detail::require_device(exec_, "coo::spmv2");
        auto db = detail_fmt::dense(b); auto dx = detail_fmt::dense(x);
        auto al = alpha ? detail_fmt::dense(alpha)->get_const_values() : nullptr;
        if (void* ws = sorted_workspace(db->cols())) {
            GKOMI_CALL(gkomi_coo_spmv2_sorted_f64_i32(nullptr, size_[0], size_[1], db->cols(), get_num_stored_elements(), get_const_row_idxs(), get_const_col_idxs(),
                                                      get_const_values(), db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride(), al, max_row_nnz_, ws,
                                                      sorted_ws_.get_num_elems()));
            return;
        }
        GKOMI_CALL(gkomi_coo_spmv2_f64_i32(nullptr, size_[0], size_[1], db->cols(), get_num_stored_elements(), get_const_row_idxs(), get_const_col_idxs(),
                                           get_const_values(), db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride(), al));
    }
    void run(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        detail::require_device(exec_, "coo::spmv");
        auto db = detail_fmt::dense(b); auto dx = detail_fmt::dense(x);
        auto al = alpha ? detail_fmt::dense(alpha)->get_const_values() : nullptr;
        auto be = beta ? detail_fmt::dense(beta)->get_const_values() : nullptr;
        if (void* ws = sorted_workspace(db->cols())) {
            GKOMI_CALL(gkomi_coo_spmv_sorted_f64_i32(nullptr, size_[0], size_[1], db->cols(), get_num_stored_elements(), get_const_row_idxs(), get_const_col_idxs(),
                                                     get_const_values(), db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride(), al, be, max_row_nnz_, ws,
                                                     sorted_ws_.get_num_elems()));
            return;
        }
        GKOMI_CALL(gkomi_coo_spmv_f64_i32(nullptr, size_[0], size_[1], db->cols(), get_num_stored_elements(), get_const_row_idxs(), get_const_col_idxs(),
                                          get_const_values(), db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride(), al, be));
    }
    void apply_impl(const LinOp* b, LinOp* x) const override { run(nullptr, b, nullptr, x); }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { run(alpha, b, beta, x); }
    array<V> values_;
    array<I> col_idxs_;
    array<I> row_idxs_;
    mutable int sorted_state_{-1};   // -1 unknown, 0 not sorted, 1 sorted
    mutable int64_t max_row_nnz_{-1};
    mutable array<char> sorted_ws_;
};

template <typename V = double, typename I = int32>
class Ell : public LinOp {
public:
    static std::unique_ptr<Ell> create(std::shared_ptr<const Executor> exec, const dim<2>& size = dim<2>{}, size_type num_stored_per_row = 0, size_type stride = 0)
    {
        return std::unique_ptr<Ell>(new Ell(std::move(exec), size, num_stored_per_row, stride ? stride : size[0]));
    }
    V* get_values() noexcept { return values_.get_data(); }
    const V* get_const_values() const noexcept { return values_.get_const_data(); }
    I* get_col_idxs() noexcept { return col_idxs_.get_data(); }
    const I* get_const_col_idxs() const noexcept { return col_idxs_.get_const_data(); }
    size_type get_num_stored_elements_per_row() const noexcept { return k_; }
    size_type get_stride() const noexcept { return stride_; }
    size_type get_num_stored_elements() const noexcept { return values_.get_num_elems(); }
    void resize(const dim<2>& size, size_type k, size_type stride) { k_ = k; stride_ = stride; values_.resize_and_reset(k * stride); col_idxs_.resize_and_reset(k * stride); set_size(size); }
protected:
    Ell(std::shared_ptr<const Executor> exec, const dim<2>& size, size_type k, size_type stride) : LinOp(exec, size), values_(exec, k * stride), col_idxs_(exec, k * stride), k_(k), stride_(stride) {}
    void run(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        detail::require_device(exec_, "ell::spmv");
        auto db = detail_fmt::dense(b); auto dx = detail_fmt::dense(x);
        GKOMI_CALL(gkomi_ell_spmv_f64_i32(nullptr, size_[0], size_[1], db->cols(), k_, stride_, get_const_col_idxs(), get_const_values(), db->get_const_values(), db->get_stride(),
                                          dx->get_values(), dx->get_stride(), alpha ? detail_fmt::dense(alpha)->get_const_values() : nullptr, beta ? detail_fmt::dense(beta)->get_const_values() : nullptr));
    }
    void apply_impl(const LinOp* b, LinOp* x) const override { run(nullptr, b, nullptr, x); }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { run(alpha, b, beta, x); }
    array<V> values_;
    array<I> col_idxs_;
    size_type k_, stride_;
};

constexpr size_type default_slice_size = 64;
constexpr size_type default_stride_factor = 1;

template <typename V = double, typename I = int32>
class Sellp : public LinOp {
public:
    static std::unique_ptr<Sellp> create(std::shared_ptr<const Executor> exec, const dim<2>& size = dim<2>{}, size_type slice_size = default_slice_size,
                                         size_type stride_factor = default_stride_factor, size_type total_cols = 0)
    {
        return std::unique_ptr<Sellp>(new Sellp(std::move(exec), size, slice_size, stride_factor, total_cols));
    }
    const V* get_const_values() const noexcept { return values_.get_const_data(); }
    const I* get_const_col_idxs() const noexcept { return col_idxs_.get_const_data(); }
    const size_type* get_const_slice_sets() const noexcept { return slice_sets_.get_const_data(); }
    const size_type* get_const_slice_lengths() const noexcept { return slice_lengths_.get_const_data(); }
    size_type get_slice_size() const noexcept { return slice_size_; }
    size_type get_stride_factor() const noexcept { return stride_factor_; }
    size_type get_total_cols() const noexcept { return values_.get_num_elems() / slice_size_; }
    size_type get_num_stored_elements() const noexcept { return values_.get_num_elems(); }
    friend class Csr<V, I>;
protected:
    Sellp(std::shared_ptr<const Executor> exec, const dim<2>& size, size_type slice_size, size_type stride_factor, size_type total_cols)
        : LinOp(exec, size), values_(exec, slice_size * total_cols), col_idxs_(exec, slice_size * total_cols),
          slice_lengths_(exec, (size[0] + slice_size - 1) / slice_size), slice_sets_(exec, (size[0] + slice_size - 1) / slice_size + 1),
          slice_size_(slice_size), stride_factor_(stride_factor) {}
    void run(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        detail::require_device(exec_, "sellp::spmv");
        auto db = detail_fmt::dense(b); auto dx = detail_fmt::dense(x);
        GKOMI_CALL(gkomi_sellp_spmv_f64_i32(nullptr, size_[0], size_[1], db->cols(), slice_size_, reinterpret_cast<const uint64_t*>(slice_sets_.get_const_data()),
                                            reinterpret_cast<const uint64_t*>(slice_lengths_.get_const_data()), get_const_col_idxs(), get_const_values(), db->get_const_values(), db->get_stride(),
                                            dx->get_values(), dx->get_stride(), alpha ? detail_fmt::dense(alpha)->get_const_values() : nullptr, beta ? detail_fmt::dense(beta)->get_const_values() : nullptr));
    }
    void apply_impl(const LinOp* b, LinOp* x) const override { run(nullptr, b, nullptr, x); }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { run(alpha, b, beta, x); }
    array<V> values_;
    array<I> col_idxs_;
    array<size_type> slice_lengths_;
    array<size_type> slice_sets_;
    size_type slice_size_, stride_factor_;
};

template <typename V = double, typename I = int32>
class Hybrid : public LinOp {
public:
    // ELL-width strategies (include/ginkgo/core/matrix/hybrid.hpp:206-370)
    struct strategy_type { int kind; double percent; double ratio; int64_t num_columns; virtual ~strategy_type() = default;
        strategy_type(int k, double p, double r, int64_t c) : kind(k), percent(p), ratio(r), num_columns(c) {} };
    struct column_limit : strategy_type { explicit column_limit(size_type n = 0) : strategy_type(0, 0, 0, static_cast<int64_t>(n)) {} };
    struct imbalance_limit : strategy_type { explicit imbalance_limit(double p = 0.8) : strategy_type(1, p, 0, 0) {} };
    struct imbalance_bounded_limit : strategy_type { imbalance_bounded_limit(double p = 0.8, double r = 0.0001) : strategy_type(2, p, r, 0) {} };
    struct minimal_storage_limit : strategy_type { minimal_storage_limit() : strategy_type(3, 0, 0, 0) {} };
    struct automatic : strategy_type { automatic() : strategy_type(4, 0, 0, 0) {} };
    static std::unique_ptr<Hybrid> create(std::shared_ptr<const Executor> exec, std::shared_ptr<strategy_type> strategy = std::make_shared<automatic>())
    {
        return std::unique_ptr<Hybrid>(new Hybrid(std::move(exec), std::move(strategy)));
    }
    const Ell<V, I>* get_ell() const noexcept { return ell_.get(); }
    const Coo<V, I>* get_coo() const noexcept { return coo_.get(); }
    size_type get_ell_num_stored_elements_per_row() const noexcept { return ell_->get_num_stored_elements_per_row(); }
    size_type get_coo_num_stored_elements() const noexcept { return coo_->get_num_stored_elements(); }
    std::shared_ptr<strategy_type> get_strategy() const noexcept { return strategy_; }
    friend class Csr<V, I>;
protected:
    Hybrid(std::shared_ptr<const Executor> exec, std::shared_ptr<strategy_type> strategy)
        : LinOp(exec), ell_(Ell<V, I>::create(exec)), coo_(Coo<V, I>::create(exec)), strategy_(std::move(strategy)) {}
    // core/matrix/hybrid.cpp:133-159
    void apply_impl(const LinOp* b, LinOp* x) const override { ell_->apply(b, x); coo_->apply2(b, x); }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { ell_->apply(alpha, b, beta, x); coo_->apply2(alpha, b, x); }
    std::unique_ptr<Ell<V, I>> ell_;
    std::unique_ptr<Coo<V, I>> coo_;
    std::shared_ptr<strategy_type> strategy_;
};

// conversions (core/matrix/csr.cpp:257-405)
template <typename V, typename I>
void Csr<V, I>::convert_to(Coo<V, I>* result) const
{
    detail::require_device(exec_, "csr::convert_to_coo");
    result->resize(size_, get_num_stored_elements());
    exec_->copy(get_num_stored_elements(), get_const_values(), result->get_values());
    exec_->copy(get_num_stored_elements(), get_const_col_idxs(), result->get_col_idxs());
    GKOMI_CALL(gkomi_convert_ptrs_to_idxs_i32(nullptr, get_const_row_ptrs(), size_[0], result->get_row_idxs()));
}
template <typename V, typename I>
void Csr<V, I>::convert_to(Ell<V, I>* result) const
{
    detail::require_device(exec_, "csr::convert_to_ell");
    array<I> mx(exec_, 1);
    GKOMI_CALL(gkomi_csr_max_row_nnz_i32(nullptr, size_[0], get_const_row_ptrs(), mx.get_data()));
    const size_type k = static_cast<size_type>(exec_->copy_val_to_host(mx.get_const_data()));
    result->resize(size_, k, size_[0]);
    GKOMI_CALL(gkomi_csr_convert_to_ell_f64_i32(nullptr, size_[0], get_const_row_ptrs(), get_const_col_idxs(), get_const_values(), k, size_[0], result->get_col_idxs(), result->get_values()));
}
template <typename V, typename I>
void Csr<V, I>::convert_to(Sellp<V, I>* result) const
{
    detail::require_device(exec_, "csr::convert_to_sellp");
    const size_type ss = result->slice_size_, sf = result->stride_factor_;
    const size_type nslices = (size_[0] + ss - 1) / ss;
    result->slice_sets_.resize_and_reset(nslices + 1);
    result->slice_lengths_.resize_and_reset(nslices);
    array<char> ws(exec_, gkomi_prefix_sum_workspace_bytes(nslices + 1) + 8);
    GKOMI_CALL(gkomi_sellp_compute_slice_sets_i32(nullptr, get_const_row_ptrs(), size_[0], ss, sf, reinterpret_cast<uint64_t*>(result->slice_sets_.get_data()),
                                                  reinterpret_cast<uint64_t*>(result->slice_lengths_.get_data()), ws.get_data(), ws.get_num_elems()));
    const size_type total = exec_->copy_val_to_host(result->slice_sets_.get_const_data() + nslices);
    result->values_.resize_and_reset(total * ss);
    result->col_idxs_.resize_and_reset(total * ss);
    result->set_size(size_);
    GKOMI_CALL(gkomi_csr_convert_to_sellp_f64_i32(nullptr, size_[0], get_const_row_ptrs(), get_const_col_idxs(), get_const_values(), ss,
                                                  reinterpret_cast<const uint64_t*>(result->slice_sets_.get_const_data()), reinterpret_cast<const uint64_t*>(result->slice_lengths_.get_const_data()),
                                                  result->col_idxs_.get_data(), result->values_.get_data()));
}
template <typename V, typename I>
void Csr<V, I>::convert_to(Hybrid<V, I>* result) const
{
    detail::require_device(exec_, "csr::convert_to_hybrid");
    const auto& st = *result->strategy_;
    int64_t ell_lim = 0;
    GKOMI_CALL(gkomi_hybrid_ell_width_i32(nullptr, get_const_row_ptrs(), size_[0], st.kind, st.percent, st.ratio, st.num_columns, &ell_lim));
    if (ell_lim > static_cast<int64_t>(size_[1])) ell_lim = size_[1];
    array<int64> crp(exec_, size_[0] + 1);
    array<char> ws(exec_, gkomi_prefix_sum_workspace_bytes(size_[0] + 1) + 8);
    GKOMI_CALL(gkomi_hybrid_compute_coo_row_ptrs_i32(nullptr, get_const_row_ptrs(), size_[0], ell_lim, crp.get_data(), ws.get_data(), ws.get_num_elems()));
    const size_type coo_nnz = static_cast<size_type>(exec_->copy_val_to_host(crp.get_const_data() + size_[0]));
    result->ell_->resize(size_, ell_lim, size_[0]);
    result->coo_->resize(size_, coo_nnz);
    result->set_size(size_);
    GKOMI_CALL(gkomi_csr_convert_to_hybrid_f64_i32(nullptr, size_[0], get_const_row_ptrs(), get_const_col_idxs(), get_const_values(), crp.get_const_data(), ell_lim, size_[0],
                                                   result->ell_->get_col_idxs(), result->ell_->get_values(), result->coo_->get_row_idxs(), result->coo_->get_col_idxs(), result->coo_->get_values()));
}

}  // namespace matrix

// ---- binary matrix I/O (core/base/mtx_io.cpp:768-960): 32-byte header = magic
// "GINKGO" + value type (D/S) + index type (I/L), rows, cols, entries as uint64,
// then (row, column, value) records; complex files are refused ----------------------
namespace detail {
inline uint64 binary_magic(char value_bit, char index_bit)
{
    const char m[8] = {'G', 'I', 'N', 'K', 'G', 'O', value_bit, index_bit};
    uint64 v;
    std::memcpy(&v, m, 8);
    return v;
}
template <typename FV, typename FI, typename V, typename I>
matrix_data<V, I> read_binary_entries(std::istream& is, uint64 rows, uint64 cols, uint64 entries)
{
    if (rows > static_cast<uint64>(std::numeric_limits<I>::max()) || cols > static_cast<uint64>(std::numeric_limits<I>::max()))
        throw StreamError(__FILE__, __LINE__, "cannot read into this format, its index type would overflow");
    matrix_data<V, I> result;
    result.size = dim<2>(rows, cols);
    result.nonzeros.resize(entries);
    for (uint64 i = 0; i < entries; ++i) {
        char block[sizeof(FV) + 2 * sizeof(FI)];
        if (!is.read(block, sizeof(block))) throw StreamError(__FILE__, __LINE__, "failed reading entry " + std::to_string(i));
        FI row, column; FV value;
        std::memcpy(&row, block, sizeof(FI));
        std::memcpy(&column, block + sizeof(FI), sizeof(FI));
        std::memcpy(&value, block + 2 * sizeof(FI), sizeof(FV));
        result.nonzeros[i] = {static_cast<I>(row), static_cast<I>(column), static_cast<V>(value)};
    }
    result.ensure_row_major_order();
    return result;
}
}  // namespace detail

template <typename V = double, typename I = int32>
matrix_data<V, I> read_binary_raw(std::istream& is)
{
    char header[32];
    if (!is.read(header, 32)) throw StreamError(__FILE__, __LINE__, "failed reading header");
    uint64 magic, rows, cols, entries;
    std::memcpy(&magic, header, 8); std::memcpy(&rows, header + 8, 8); std::memcpy(&cols, header + 16, 8); std::memcpy(&entries, header + 24, 8);
    if (magic == detail::binary_magic('D', 'I')) return detail::read_binary_entries<double, int32, V, I>(is, rows, cols, entries);
    if (magic == detail::binary_magic('S', 'I')) return detail::read_binary_entries<float, int32, V, I>(is, rows, cols, entries);
    if (magic == detail::binary_magic('D', 'L')) return detail::read_binary_entries<double, int64, V, I>(is, rows, cols, entries);
    if (magic == detail::binary_magic('S', 'L')) return detail::read_binary_entries<float, int64, V, I>(is, rows, cols, entries);
    if (magic == detail::binary_magic('Z', 'I') || magic == detail::binary_magic('C', 'I') || magic == detail::binary_magic('Z', 'L') || magic == detail::binary_magic('C', 'L'))
        throw StreamError(__FILE__, __LINE__, "cannot read into this format, would assign complex to real");
    throw StreamError(__FILE__, __LINE__, "invalid header magic number '" + std::string(header, 8) + "'");
}
template <typename V = double, typename I = int32>
matrix_data<V, I> read_generic_raw(std::istream& is)
{
    const auto first = is.peek();
    if (!is) throw StreamError(__FILE__, __LINE__, "failed reading from stream");
    return first == '%' ? read_raw<V, I>(is) : read_binary_raw<V, I>(is);
}
template <typename V, typename I>
void write_binary_raw(std::ostream& os, const matrix_data<V, I>& data)
{
    static_assert(std::is_same<V, double>::value && std::is_same<I, int32>::value, "this mirror stores double / int32");
    const uint64 hdr[4] = {detail::binary_magic('D', 'I'), data.size[0], data.size[1], data.nonzeros.size()};
    os.write(reinterpret_cast<const char*>(hdr), 32);
    for (const auto& e : data.nonzeros) {
        char block[16];
        std::memcpy(block, &e.row, 4); std::memcpy(block + 4, &e.column, 4); std::memcpy(block + 8, &e.value, 8);
        os.write(block, 16);
    }
    if (!os) throw StreamError(__FILE__, __LINE__, "failed writing the matrix");
}

// ---- read / write / initialize ------------------------------------------------------
template <typename M, typename Stream>
std::unique_ptr<M> read_binary(Stream&& is, std::shared_ptr<const Executor> exec)
{
    auto data = read_binary_raw<double, int32>(is);
    auto m = M::create(std::move(exec));
    m->read(data);
    return m;
}
template <typename M, typename Stream>
std::unique_ptr<M> read_generic(Stream&& is, std::shared_ptr<const Executor> exec)
{
    auto data = read_generic_raw<double, int32>(is);
    auto m = M::create(std::move(exec));
    m->read(data);
    return m;
}
template <typename M>
void write_binary(std::ostream& os, const M* m)
{
    matrix_data<double, int32> data;
    m->write(data);
    write_binary_raw(os, data);
}
template <typename M, typename Stream>
std::unique_ptr<M> read(Stream&& is, std::shared_ptr<const Executor> exec)
{
    auto data = read_raw<double, int32>(is);
    auto m = M::create(std::move(exec));
    m->read(data);
    return m;
}
template <typename M>
void write(std::ostream& os, const M* m)
{
    matrix_data<double, int32> data;
    m->write(data);
    // Dense is written in array layout, sparse formats in coordinate layout
    if (std::is_same<M, matrix::Dense<double>>::value) {
        os << "%%MatrixMarket matrix array real general\n" << data.size[0] << " " << data.size[1] << "\n";
        std::vector<double> colmajor(data.size[0] * data.size[1]);
        for (const auto& e : data.nonzeros) colmajor[e.column * data.size[0] + e.row] = e.value;
        for (double v : colmajor) os << v << "\n";
    } else {
        os << "%%MatrixMarket matrix coordinate real general\n" << data.size[0] << " " << data.size[1] << " " << data.nonzeros.size() << "\n";
        for (const auto& e : data.nonzeros) os << e.row + 1 << " " << e.column + 1 << " " << e.value << "\n";
    }
}
template <typename M>
std::unique_ptr<M> initialize(size_type stride, std::initializer_list<typename M::value_type> vals, std::shared_ptr<const Executor> exec)
{
    auto host = exec->get_master();
    auto tmp = matrix::Dense<double>::create(host, dim<2>(vals.size(), 1), stride);
    size_type i = 0;
    for (auto v : vals) tmp->at(i++, 0) = v;
    auto m = M::create(exec, dim<2>(vals.size(), 1), stride);
    m->copy_from(tmp.get());
    return m;
}
template <typename M>
std::unique_ptr<M> initialize(std::initializer_list<typename M::value_type> vals, std::shared_ptr<const Executor> exec) { return initialize<M>(1, vals, std::move(exec)); }
template <typename M>
std::unique_ptr<M> initialize(std::initializer_list<std::initializer_list<typename M::value_type>> vals, std::shared_ptr<const Executor> exec)
{
    const size_type r = vals.size(), c = r ? vals.begin()->size() : 0;
    auto tmp = matrix::Dense<double>::create(exec->get_master(), dim<2>(r, c));
    size_type i = 0;
    for (const auto& row : vals) { size_type j = 0; for (auto v : row) tmp->at(i, j++) = v; ++i; }
    auto m = M::create(exec, dim<2>(r, c));
    m->copy_from(tmp.get());
    return m;
}

// ---- stopping criteria (include/ginkgo/core/stop/*.hpp) ---------------------------------
namespace stop {
enum class mode { absolute, initial_resnorm, rhs_norm };
struct criterion_settings {
    int64_t max_iters{std::numeric_limits<int64_t>::max() / 4};
    double reduction_factor{-1.0};  // < 0: no residual criterion
    mode baseline{mode::rhs_norm};
    bool implicit{false};           // ImplicitResidualNorm: sqrt of the solver's implicit r.z instead of ||r||
};
class CriterionFactory {
public:
    virtual ~CriterionFactory() = default;
    virtual void contribute(criterion_settings& s) const = 0;
};
template <typename Concrete>
struct builder_base {
    std::shared_ptr<const CriterionFactory> on(std::shared_ptr<const Executor>) const { return std::make_shared<Concrete>(static_cast<const Concrete&>(*this)); }
};
class Iteration : public CriterionFactory, public builder_base<Iteration> {
public:
    static Iteration build() { return {}; }
    Iteration& with_max_iters(size_type n) { max_iters_ = n; return *this; }
    void contribute(criterion_settings& s) const override { s.max_iters = std::min<int64_t>(s.max_iters, static_cast<int64_t>(max_iters_)); }
private:
    size_type max_iters_{0};
};
template <typename V = double>
class ResidualNorm : public CriterionFactory, public builder_base<ResidualNorm<V>> {
public:
    static ResidualNorm build() { return {}; }
    ResidualNorm& with_reduction_factor(V f) { factor_ = f; return *this; }
    ResidualNorm& with_baseline(mode m) { baseline_ = m; return *this; }
    void contribute(criterion_settings& s) const override { s.reduction_factor = factor_; s.baseline = baseline_; }
private:
    V factor_{static_cast<V>(1e-15)};  // residual_norm.hpp:65
    mode baseline_{mode::rhs_norm};
};
// stop::ImplicitResidualNorm (include/ginkgo/core/stop/residual_norm.hpp:193-244; kernel
// stop::implicit_residual_norm): the criterion on sqrt(|r.z|), the quantity Cg / Fcg carry anyway.
// With the Identity preconditioner that IS ||r||, and the native drivers evaluate it as such; with
// another preconditioner the solvers refuse it (their drivers evaluate ResidualNorm).
template <typename V = double>
class ImplicitResidualNorm : public CriterionFactory, public builder_base<ImplicitResidualNorm<V>> {
public:
    static ImplicitResidualNorm build() { return {}; }
    ImplicitResidualNorm& with_reduction_factor(V f) { factor_ = f; return *this; }
    ImplicitResidualNorm& with_baseline(mode m) { baseline_ = m; return *this; }
    void contribute(criterion_settings& s) const override { s.reduction_factor = factor_; s.baseline = baseline_; s.implicit = true; }
private:
    V factor_{static_cast<V>(1e-15)};
    mode baseline_{mode::rhs_norm};
};
// stop::Combined (include/ginkgo/core/stop/combined.hpp): stops when ANY of its criteria does,
// which is how the drivers treat a list of criteria anyway (Iteration is asked first)
class Combined : public CriterionFactory, public builder_base<Combined> {
public:
    static Combined build() { return {}; }
    template <typename... Criteria>
    Combined& with_criteria(Criteria&&... c)
    {
        criteria_ = {std::shared_ptr<const CriterionFactory>(std::forward<Criteria>(c))...};
        return *this;
    }
    Combined& with_criteria(std::vector<std::shared_ptr<const CriterionFactory>> c) { criteria_ = std::move(c); return *this; }
    void contribute(criterion_settings& s) const override { for (const auto& c : criteria_) if (c) c->contribute(s); }
private:
    std::vector<std::shared_ptr<const CriterionFactory>> criteria_;
};
// stop::combine (combined.hpp:130-160)
template <typename FactoryContainer>
std::shared_ptr<const CriterionFactory> combine(FactoryContainer&& factories)
{
    if (factories.size() == 1) return factories[0];
    auto exec = std::shared_ptr<const Executor>();
    return Combined::build().with_criteria(std::vector<std::shared_ptr<const CriterionFactory>>(factories.begin(), factories.end())).on(exec);
}
}  // namespace stop

namespace detail {
// A system whose solve is a collective: implemented by experimental::distributed::Matrix
// (distributed.hpp); the solvers hand such a system to its own driver.
struct distributed_system {
    virtual ~distributed_system() = default;
    virtual void cg_solve(const LinOp* b, LinOp* x, const stop::criterion_settings& settings, const LinOp* precond, int64_t* iters,
                          bool* converged) const = 0;
};
}  // namespace detail

// ---- preconditioners -----------------------------------------------------------------
namespace detail {
// a preconditioner the library has a callback of its own for (gkomi_jacobi_apply_cb, gkomi_ilu_apply_cb): it hands
// the native drivers that callback and its record instead of an opaque LinOp::apply
struct native_preconditioner {
    virtual ~native_preconditioner() = default;
    // false: not in this configuration (the caller wraps LinOp::apply)
    virtual bool native_callback(gkomi_apply_fn& fn, void*& ctx, size_type nrhs) const = 0;
};
// a LinOp as a gkomi_apply_fn for the native solver drivers
struct linop_callback {
    const LinOp* op;
    std::shared_ptr<const Executor> exec;
    size_type n, nrhs;
    static int call(void* ctx, gkomi_stream_t, const double* in, double* out)
    {
        auto* c = static_cast<linop_callback*>(ctx);
        try {
            auto vin = matrix::Dense<double>::create(c->exec, dim<2>(c->n, c->nrhs), array<double>::view(c->exec, c->n * c->nrhs, const_cast<double*>(in)), c->nrhs);
            auto vout = matrix::Dense<double>::create(c->exec, dim<2>(c->n, c->nrhs), array<double>::view(c->exec, c->n * c->nrhs, out), c->nrhs);
            c->op->apply(vin.get(), vout.get());
        } catch (const std::exception&) {
            return GKOMI_EINVAL;
        }
        return 0;
    }
};
// any LinOp as a gkomi_matrix_apply_fn: the system matrix of the *_solve_op_f64
// drivers (Ell, Sellp, Coo, Hybrid, ... -- config 4 of BASELINE.json)
struct matrix_callback {
    const LinOp* op;
    std::shared_ptr<const Executor> exec;
    size_type n;
    static int call(void* ctx, gkomi_stream_t, int64_t nrhs, const double* alpha, const double* b, int64_t b_stride, const double* beta, double* c, int64_t c_stride)
    {
        auto* m = static_cast<matrix_callback*>(ctx);
        try {
            const size_type k = static_cast<size_type>(nrhs);
            auto vb = matrix::Dense<double>::create(m->exec, dim<2>(m->n, k), array<double>::view(m->exec, m->n * b_stride, const_cast<double*>(b)), b_stride);
            auto vc = matrix::Dense<double>::create(m->exec, dim<2>(m->n, k), array<double>::view(m->exec, m->n * c_stride, c), c_stride);
            if (alpha == nullptr) {
                m->op->apply(vb.get(), vc.get());
            } else {
                auto va = matrix::Dense<double>::create(m->exec, dim<2>(1, 1), array<double>::view(m->exec, 1, const_cast<double*>(alpha)), 1);
                auto vbeta = matrix::Dense<double>::create(m->exec, dim<2>(1, 1), array<double>::view(m->exec, 1, const_cast<double*>(beta)), 1);
                m->op->apply(va.get(), vb.get(), vbeta.get(), vc.get());
            }
        } catch (const std::exception&) {
            return GKOMI_EINVAL;
        }
        return 0;
    }
};
// The system matrix of a solver as (gkomi_matrix_apply_fn, context): Ell and
// Sellp go by the library's own callbacks + records, which the fused drivers
// recognise (SpMV with the dot-product epilogue, csrc/formats.hip); every other
// LinOp goes through matrix_callback.
struct system_callback {
    matrix_callback generic;
    gkomi_csr_ctx csr{};
    gkomi_ell_ctx ell{};
    gkomi_sellp_ctx sellp{};
    gkomi_matrix_apply_fn fn;
    void* ctx;
    system_callback(const LinOp* A, std::shared_ptr<const Executor> exec, size_type n)
        : generic{A, std::move(exec), n}, fn(&matrix_callback::call), ctx(&generic)
    {
        if (auto c = dynamic_cast<const matrix::Csr<double, int32>*>(A)) {
            // the matrix travels with its srow and its row statistic: the drivers then run the kernel
            // Csr::apply runs, also inside their fused SpMV + dot launches
            c->make_srow();
            csr.nrows = static_cast<int64_t>(c->get_size()[0]);
            csr.ncols = static_cast<int64_t>(c->get_size()[1]);
            csr.nnz = static_cast<int64_t>(c->get_num_stored_elements());
            csr.row_ptrs = c->get_const_row_ptrs();
            csr.col_idxs = c->get_const_col_idxs();
            csr.vals = c->get_const_values();
            csr.strategy = c->get_strategy()->get_code();
            csr.max_row_nnz_hint = c->get_max_row_nnz();
            csr.srow = c->get_num_srow_elements() ? c->get_const_srow() : nullptr;
            csr.srow_tile = c->get_num_srow_elements() ? c->get_srow_tile() : 0;
            fn = &gkomi_csr_matrix_apply_cb;
            ctx = &csr;
        } else if (auto e = dynamic_cast<const matrix::Ell<double, int32>*>(A)) {
            ell.nrows = static_cast<int64_t>(e->get_size()[0]);
            ell.ncols = static_cast<int64_t>(e->get_size()[1]);
            ell.num_stored_per_row = static_cast<int64_t>(e->get_num_stored_elements_per_row());
            ell.stride = static_cast<int64_t>(e->get_stride());
            ell.col_idxs = e->get_const_col_idxs();
            ell.vals = e->get_const_values();
            fn = &gkomi_ell_matrix_apply_cb;
            ctx = &ell;
        } else if (auto sp = dynamic_cast<const matrix::Sellp<double, int32>*>(A)) {
            sellp.nrows = static_cast<int64_t>(sp->get_size()[0]);
            sellp.ncols = static_cast<int64_t>(sp->get_size()[1]);
            sellp.slice_size = static_cast<int64_t>(sp->get_slice_size());
            sellp.slice_sets = reinterpret_cast<const uint64_t*>(sp->get_const_slice_sets());
            sellp.slice_lengths = reinterpret_cast<const uint64_t*>(sp->get_const_slice_lengths());
            sellp.col_idxs = sp->get_const_col_idxs();
            sellp.vals = sp->get_const_values();
            fn = &gkomi_sellp_matrix_apply_cb;
            ctx = &sellp;
        }
    }
    system_callback(const system_callback&) = delete;
    system_callback& operator=(const system_callback&) = delete;
};
}  // namespace detail

// include/ginkgo/core/base/lin_op.hpp:433-455 (real values: conj_transpose == transpose)
class Transposable {
public:
    virtual ~Transposable() = default;
    virtual std::unique_ptr<LinOp> transpose() const = 0;
    virtual std::unique_ptr<LinOp> conj_transpose() const { return this->transpose(); }
};

// include/ginkgo/core/base/types.hpp:257-400
class precision_reduction {
public:
    using storage_type = uint8;
    constexpr precision_reduction() noexcept : data_{0} {}
    constexpr precision_reduction(storage_type preserving, storage_type nonpreserving) noexcept
        : data_(static_cast<storage_type>((preserving << 4) | nonpreserving)) {}
    constexpr operator storage_type() const noexcept { return data_; }
    constexpr storage_type get_preserving() const noexcept { return data_ >> 4; }
    constexpr storage_type get_nonpreserving() const noexcept { return data_ & 0xf; }
    constexpr static precision_reduction autodetect() noexcept { return from_bits(0xff); }
    constexpr static precision_reduction from_bits(storage_type b) noexcept { precision_reduction p; p.data_ = b; return p; }
private:
    storage_type data_;
};

namespace preconditioner {
// include/ginkgo/core/preconditioner/jacobi.hpp:62-167
template <typename IndexType>
struct block_interleaved_storage_scheme {
    block_interleaved_storage_scheme() = default;
    block_interleaved_storage_scheme(IndexType block_offset_, IndexType group_offset_, uint32 group_power_)
        : block_offset{block_offset_}, group_offset{group_offset_}, group_power{group_power_} {}
    IndexType block_offset{};
    IndexType group_offset{};
    uint32 group_power{};
    IndexType get_group_size() const noexcept { return IndexType{1} << group_power; }
    size_type compute_storage_space(size_type num_blocks) const noexcept
    {
        return (num_blocks + 1 == size_type{0}) ? size_type{0} : ((num_blocks + get_group_size() - 1) / get_group_size()) * group_offset;
    }
    IndexType get_group_offset(IndexType block_id) const noexcept { return group_offset * (block_id >> group_power); }
    IndexType get_block_offset(IndexType block_id) const noexcept { return block_offset * (block_id & (get_group_size() - 1)); }
    IndexType get_global_block_offset(IndexType block_id) const noexcept { return get_group_offset(block_id) + get_block_offset(block_id); }
    IndexType get_stride() const noexcept { return block_offset << group_power; }
};

template <typename V = double, typename I = int32>
class Jacobi : public LinOp, public Transposable, public ::gko::detail::native_preconditioner {
public:
    class Factory : public LinOpFactory {
    public:
        Factory& with_max_block_size(uint32 n) { max_block_size_ = n; return *this; }
        // storage_optimization (jacobi.hpp:262-330): one precision for all blocks
        // (autodetect() = adaptive) or a list repeated over the blocks
        Factory& with_storage_optimization(precision_reduction p) { storage_.assign(1, p); adaptive_ = true; return *this; }
        Factory& with_storage_optimization(const std::vector<precision_reduction>& p) { storage_ = p; adaptive_ = !p.empty(); return *this; }
        Factory& with_accuracy(double a) { accuracy_ = a; return *this; }
        // jacobi.hpp:338-349: 0 = the executor's default = the wavefront size (64 on HIP), the only
        // stride the kernels of this backend lay the blocks out for
        Factory& with_max_block_stride(uint32 s)
        {
            if (s != 0 && s != 64) GKO_NOT_SUPPORTED("max_block_stride: 0 or 64 (the HIP wavefront size)");
            return *this;
        }
        std::shared_ptr<Factory> on(std::shared_ptr<const Executor> exec) const { auto f = std::make_shared<Factory>(*this); f->exec_ = std::move(exec); return f; }
        std::unique_ptr<Jacobi> generate(std::shared_ptr<const LinOp> A) const
        {
            return std::unique_ptr<Jacobi>(new Jacobi(this->exec_, max_block_size_, adaptive_ ? &storage_ : nullptr, accuracy_, std::move(A)));
        }
        std::unique_ptr<LinOp> generate_impl(std::shared_ptr<const LinOp> A) const override { return generate(std::move(A)); }
        Factory() : LinOpFactory(nullptr) {}
    private:
        uint32 max_block_size_{32};  // jacobi.hpp:338-349
        std::vector<precision_reduction> storage_;
        bool adaptive_{false};
        double accuracy_{1e-1};      // jacobi.hpp:332-336
    };
    static Factory build() { return Factory{}; }
    size_type get_num_blocks() const noexcept { return num_blocks_; }
    uint32 get_max_block_size() const noexcept { return max_block_size_; }
    // per-block storage precisions actually used / condition numbers (adaptive storage only)
    std::vector<precision_reduction> get_block_precisions() const
    {
        std::vector<precision_reduction> out;
        for (auto b : precisions_.to_host()) out.push_back(precision_reduction::from_bits(b));
        return out;
    }
    std::vector<double> get_conditioning() const { return conditioning_.to_host(); }
    // jacobi.hpp:557-609: the scheme of max_block_stride = 64
    block_interleaved_storage_scheme<I> get_storage_scheme() const
    {
        int64_t o[4] = {};
        GKOMI_CALL(gkomi_jacobi_storage_scheme(static_cast<int>(max_block_size_), o));
        return {static_cast<I>(o[0]), static_cast<I>(o[1]), static_cast<uint32>(o[2])};
    }
    const V* get_blocks() const noexcept { return blocks_.get_const_data(); }
    size_type get_num_stored_elements() const noexcept { return blocks_.get_num_elems(); }
    const I* get_const_block_pointers() const noexcept { return block_ptrs_.get_const_data(); }
    // Jacobi::transpose (core/preconditioner/jacobi.cpp): same blocks structure,
    // every stored block transposed in its storage precision (jacobi::transpose_jacobi)
    std::unique_ptr<LinOp> transpose() const override
    {
        std::unique_ptr<Jacobi> t(new Jacobi(*this));
        if (max_block_size_ > 1 && num_blocks_ > 0) {
            t->blocks_ = array<V>(exec_, blocks_.get_num_elems());
            GKOMI_CALL(gkomi_jacobi_transpose_f64_i32(nullptr, num_blocks_, max_block_size_, block_ptrs_.get_const_data(),
                                                      precisions_.get_num_elems() > 0 ? precisions_.get_const_data() : nullptr, blocks_.get_const_data(),
                                                      t->blocks_.get_data()));
        }  // scalar Jacobi: a diagonal is its own transpose
        return std::unique_ptr<LinOp>(t.release());
    }
protected:
    Jacobi(const Jacobi&) = default;
    Jacobi(std::shared_ptr<const Executor> exec, uint32 max_bs, const std::vector<precision_reduction>* storage, double accuracy, std::shared_ptr<const LinOp> A)
        : LinOp(exec, A->get_size()), max_block_size_(max_bs), block_ptrs_(exec), blocks_(exec), precisions_(exec), conditioning_(exec)
    {
        ::gko::detail::require_device(exec_, "jacobi::generate");
        if (max_bs < 1 || max_bs > 32) GKO_NOT_SUPPORTED("max_block_size must be in [1, 32]");
        auto csr = as<const matrix::Csr<V, I>>(A.get());
        const size_type n = size_[0];
        if (max_bs == 1) {
            array<V> diag(exec_, n);
            blocks_.resize_and_reset(n);
            GKOMI_CALL(gkomi_csr_extract_diagonal_f64_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(), diag.get_data()));
            GKOMI_CALL(gkomi_jacobi_invert_diagonal_f64(nullptr, n, diag.get_const_data(), blocks_.get_data()));
            num_blocks_ = n;
            return;
        }
        block_ptrs_.resize_and_reset(n + 1);
        array<int64> nb(exec_, 1);
        array<char> ws(exec_, gkomi_jacobi_find_blocks_workspace_bytes(n));
        int64_t host_nb = 0;
        GKOMI_CALL(gkomi_jacobi_find_blocks_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), max_bs, block_ptrs_.get_data(), nb.get_data(), ws.get_data(), ws.get_num_elems(), &host_nb));
        num_blocks_ = static_cast<size_type>(host_nb);
        blocks_.resize_and_reset(gkomi_jacobi_storage_elements(max_bs, host_nb));
        if (storage == nullptr) {
            GKOMI_CALL(gkomi_jacobi_generate_f64_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(), host_nb, max_bs, block_ptrs_.get_const_data(), nullptr, blocks_.get_data()));
            return;
        }
        // jacobi::initialize_precisions (jacobi_kernels.cpp:485-493): the list repeats over the blocks
        std::vector<uint8> req(num_blocks_);
        for (size_type i = 0; i < num_blocks_; ++i) req[i] = (*storage)[i % storage->size()];
        precisions_ = array<uint8>(exec_, req.begin(), req.end());
        conditioning_.resize_and_reset(num_blocks_);
        GKOMI_CALL(gkomi_jacobi_generate_adaptive_f64_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(), host_nb, max_bs,
                                                          block_ptrs_.get_const_data(), accuracy, conditioning_.get_data(), precisions_.get_data(), blocks_.get_data()));
    }
    void run(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        const double* a = alpha ? matrix::detail_fmt::dense(alpha)->get_const_values() : nullptr;
        const double* be = beta ? matrix::detail_fmt::dense(beta)->get_const_values() : nullptr;
        if (max_block_size_ == 1) {
            GKOMI_CALL(gkomi_jacobi_scalar_apply_f64(nullptr, size_[0], db->cols(), blocks_.get_const_data(), a, db->get_const_values(), db->get_stride(), be, dx->get_values(), dx->get_stride()));
        } else if (precisions_.get_num_elems() > 0) {
            GKOMI_CALL(gkomi_jacobi_apply_adaptive_f64_i32(nullptr, num_blocks_, max_block_size_, block_ptrs_.get_const_data(), precisions_.get_const_data(), blocks_.get_const_data(),
                                                           db->cols(), a, db->get_const_values(), db->get_stride(), be, dx->get_values(), dx->get_stride()));
        } else {
            GKOMI_CALL(gkomi_jacobi_apply_f64_i32(nullptr, num_blocks_, max_block_size_, block_ptrs_.get_const_data(), blocks_.get_const_data(), db->cols(), a, db->get_const_values(), db->get_stride(), be, dx->get_values(), dx->get_stride()));
        }
    }
    void apply_impl(const LinOp* b, LinOp* x) const override { run(nullptr, b, nullptr, x); }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { run(alpha, b, beta, x); }
public:
    // this preconditioner as the record of the library's own callback (gkomi_jacobi_apply_cb): the native drivers
    // then see a block-Jacobi, not an opaque operator (the fused CG lets its apply carry r.z and r.r).  The record
    // lives in the object: one solve at a time per preconditioner object and column count.
    bool native_callback(gkomi_apply_fn& fn, void*& ctx, size_type nrhs) const override
    {
        if (!std::is_same<V, double>::value || !std::is_same<I, int32>::value) return false;
        gkomi_jacobi_ctx& c = record_;
        c.n = static_cast<int64_t>(size_[0]);
        c.nrhs = static_cast<int64_t>(nrhs);
        c.num_blocks = static_cast<int64_t>(num_blocks_);
        c.max_block_size = static_cast<int32_t>(max_block_size_);
        c.pad_ = 0;
        c.block_ptrs = max_block_size_ == 1 ? nullptr : block_ptrs_.get_const_data();
        c.blocks = blocks_.get_const_data();
        c.block_precisions = precisions_.get_num_elems() > 0 ? precisions_.get_const_data() : nullptr;
        fn = &gkomi_jacobi_apply_cb;
        ctx = &record_;
        return true;
    }
private:
    mutable gkomi_jacobi_ctx record_{};
    uint32 max_block_size_;
    size_type num_blocks_{0};
    array<I> block_ptrs_;
    array<V> blocks_;
    array<uint8> precisions_;
    array<double> conditioning_;
};
}  // namespace preconditioner

namespace detail {
// The preconditioner of a native solver driver as (gkomi_apply_fn, context): a preconditioner::Jacobi or Ilu
// <double, int32> goes by the library's own callback + record (native_preconditioner), any other LinOp by linop_callback.
struct precond_callback {
    linop_callback generic;
    gkomi_apply_fn fn = nullptr;
    void* ctx = nullptr;
    precond_callback(const LinOp* op, std::shared_ptr<const Executor> exec, size_type n, size_type nrhs) : generic{op, std::move(exec), n, nrhs}
    {
        if (op == nullptr) return;
        if (auto native = dynamic_cast<const native_preconditioner*>(op)) {
            if (native->native_callback(fn, ctx, nrhs)) return;
        }
        fn = &linop_callback::call;
        ctx = &generic;
    }
    precond_callback(const precond_callback&) = delete;
    precond_callback& operator=(const precond_callback&) = delete;
};
}  // namespace detail

// ---- solvers ------------------------------------------------------------------------------
namespace solver {
namespace detail {
template <typename Solver>
class factory_base : public LinOpFactory {
public:
    factory_base() : LinOpFactory(nullptr) {}
    template <typename... Criteria>
    typename Solver::Factory& with_criteria(Criteria&&... c)
    {
        criteria_ = {std::shared_ptr<const stop::CriterionFactory>(std::forward<Criteria>(c))...};
        return static_cast<typename Solver::Factory&>(*this);
    }
    typename Solver::Factory& with_preconditioner(std::shared_ptr<const LinOpFactory> f) { precond_factory_ = std::move(f); return static_cast<typename Solver::Factory&>(*this); }
    typename Solver::Factory& with_generated_preconditioner(std::shared_ptr<const LinOp> p) { precond_ = std::move(p); return static_cast<typename Solver::Factory&>(*this); }
    std::shared_ptr<typename Solver::Factory> on(std::shared_ptr<const Executor> exec) const
    {
        auto f = std::make_shared<typename Solver::Factory>(static_cast<const typename Solver::Factory&>(*this));
        f->exec_ = std::move(exec);
        return f;
    }
    std::unique_ptr<Solver> generate(std::shared_ptr<const LinOp> A) const { return std::unique_ptr<Solver>(new Solver(static_cast<const typename Solver::Factory*>(this), std::move(A))); }
    std::unique_ptr<LinOp> generate_impl(std::shared_ptr<const LinOp> A) const override { return generate(std::move(A)); }
    stop::criterion_settings settings() const
    {
        stop::criterion_settings s;
        for (const auto& c : criteria_) c->contribute(s);
        if (s.reduction_factor < 0) s.reduction_factor = 0.0;  // Iteration only
        if (s.implicit && (precond_ || precond_factory_)) {
            GKO_NOT_SUPPORTED("ImplicitResidualNorm with a preconditioner: the native drivers evaluate ResidualNorm (identical without one)");
        }
        return s;
    }
    std::vector<std::shared_ptr<const stop::CriterionFactory>> criteria_;
    std::shared_ptr<const LinOpFactory> precond_factory_;
    std::shared_ptr<const LinOp> precond_;
};

inline int baseline_code(stop::mode m) { return m == stop::mode::rhs_norm ? 0 : (m == stop::mode::initial_resnorm ? 1 : 2); }
}  // namespace detail

template <typename V = double>
class Cg : public LinOp {
public:
    class Factory : public detail::factory_base<Cg> {};
    static Factory build() { return Factory{}; }
    std::shared_ptr<const LinOp> get_system_matrix() const { return A_; }
    std::shared_ptr<const LinOp> get_preconditioner() const { return precond_; }
    // filled by the last apply
    int64_t get_last_iteration_count() const noexcept { return last_iters_; }
    bool has_converged() const noexcept { return last_converged_; }
protected:
    friend class detail::factory_base<Cg>;
    Cg(const Factory* f, std::shared_ptr<const LinOp> A) : LinOp(f->get_executor(), gko::transpose(A->get_size())), A_(std::move(A)), settings_(f->settings())
    {
        if (size_[0] != size_[1]) throw DimensionMismatch(__FILE__, __LINE__, "Cg needs a square system matrix");
        precond_ = f->precond_ ? f->precond_ : (f->precond_factory_ ? std::shared_ptr<const LinOp>(f->precond_factory_->generate_impl(A_)) : nullptr);
    }
    void apply_impl(const LinOp* b, LinOp* x) const override { solve_as(V{}, b, x); }
    // Cg<float>: the single-precision instantiation -- the reference's kernel sequence on the _f32 kernels (gkomi_cg_solve_f32):
    // a Csr<float, int32> system matrix, Dense<float> vectors with one column, no preconditioner, Iteration + ResidualNorm
    void solve_as(float, const LinOp* b, LinOp* x) const
    {
        ::gko::detail::require_device(exec_, "cg::apply");
        auto A = dynamic_cast<const matrix::Csr<float, int32>*>(A_.get());
        auto db = dynamic_cast<const matrix::Dense<float>*>(b);
        auto dx = dynamic_cast<matrix::Dense<float>*>(x);
        if (!A || !db || !dx || precond_ || settings_.implicit || db->get_size()[1] != 1 || db->get_stride() != 1 || dx->get_stride() != 1) {
            GKO_NOT_SUPPORTED("Cg<float>: Csr<float, int32> system, Dense<float> vectors of one contiguous column, Iteration + ResidualNorm, no preconditioner");
        }
        const int64_t n = size_[0];
        array<char> ws(exec_, gkomi_cg_workspace_bytes_f32(n));
        double info[4] = {};
        GKOMI_CALL(gkomi_cg_solve_f32(nullptr, n, static_cast<int64_t>(A->get_num_stored_elements()), A->get_const_row_ptrs(), A->get_const_col_idxs(),
                                      A->get_const_values(), db->get_const_values(), dx->get_values(), settings_.max_iters,
                                      static_cast<float>(settings_.reduction_factor), detail::baseline_code(settings_.baseline), ws.get_data(),
                                      ws.get_num_elems(), info));
        last_iters_ = static_cast<int64_t>(info[0]);
        last_converged_ = info[1] != 0.0;
    }
    void solve_as(double, const LinOp* b, LinOp* x) const
    {
        ::gko::detail::require_device(exec_, "cg::apply");
        if (auto ds = dynamic_cast<const ::gko::detail::distributed_system*>(A_.get())) {
            // experimental::distributed::Matrix + Vectors: the row-partitioned driver
            ds->cg_solve(b, x, settings_, precond_.get(), &last_iters_, &last_converged_);
            return;
        }
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        const int64_t n = size_[0], nrhs = db->cols();
        if (db->get_stride() != static_cast<size_type>(nrhs) || dx->get_stride() != static_cast<size_type>(nrhs)) GKO_NOT_SUPPORTED("solver vectors must be contiguous (stride == #columns)");
        array<char> ws(exec_, gkomi_cg_workspace_bytes(n, nrhs));
        std::vector<double> info(2 + 2 * nrhs, 0.0);
        ::gko::detail::precond_callback cb(precond_.get(), exec_, static_cast<size_type>(n), static_cast<size_type>(nrhs));
        auto pfn = cb.fn;
        void* pctx = cb.ctx;
        // a Csr system matrix travels as its record (srow, row statistic): detail::system_callback
        ::gko::detail::system_callback mcb(A_.get(), exec_, static_cast<size_type>(n));
        if (nrhs == 1) {  // the fused loop; Ell / Sellp with the dot in the SpMV's epilogue
            GKOMI_CALL(gkomi_cg_solve_fused_op_f64(nullptr, n, mcb.fn, mcb.ctx, pfn, pctx, db->get_const_values(), dx->get_values(), settings_.max_iters,
                                                   settings_.reduction_factor, detail::baseline_code(settings_.baseline), 8, ws.get_data(), ws.get_num_elems(), info.data()));
        } else {
            GKOMI_CALL(gkomi_cg_solve_op_f64(nullptr, n, nrhs, mcb.fn, mcb.ctx, pfn, pctx, db->get_const_values(), dx->get_values(), settings_.max_iters,
                                             settings_.reduction_factor, detail::baseline_code(settings_.baseline), ws.get_data(), ws.get_num_elems(), info.data()));
        }
        last_iters_ = static_cast<int64_t>(info[0]);
        last_converged_ = info[1] != 0.0;
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override { advanced_as(V{}, alpha, b, beta, x); }
    void advanced_as(float, const LinOp*, const LinOp*, const LinOp*, LinOp*) const { GKO_NOT_SUPPORTED("Cg<float>: plain apply only"); }
    void advanced_as(double, const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const
    {
        // x = alpha * solve(b) + beta * x  (core/solver/cg.cpp:196-210)
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const LinOp> A_;
    std::shared_ptr<const LinOp> precond_;
    stop::criterion_settings settings_;
    mutable int64_t last_iters_{-1};
    mutable bool last_converged_{false};
};

// Bicgstab / Fcg / Cgs (include/ginkgo/core/solver/{bicgstab,fcg,cgs}.hpp): same
// factory parameters as Cg, native drivers of csrc/krylov.hip
namespace detail {
using krylov_driver = int (*)(gkomi_stream_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, const double*, int, int64_t, gkomi_apply_fn, void*,
                              const double*, double*, int64_t, double, int, int64_t, void*, size_t, double*);
using krylov_op_driver = int (*)(gkomi_stream_t, int64_t, int64_t, gkomi_matrix_apply_fn, void*, gkomi_apply_fn, void*, const double*, double*, int64_t, double, int, int64_t,
                                 void*, size_t, double*);
template <typename Derived, krylov_driver Driver, krylov_op_driver OpDriver>
class krylov_solver : public LinOp {
public:
    class Factory : public factory_base<Derived> {};
    static Factory build() { return Factory{}; }
    std::shared_ptr<const LinOp> get_system_matrix() const { return A_; }
    std::shared_ptr<const LinOp> get_preconditioner() const { return precond_; }
    int64_t get_last_iteration_count() const noexcept { return last_iters_; }
    bool has_converged() const noexcept { return last_converged_; }
protected:
    krylov_solver(const Factory* f, std::shared_ptr<const LinOp> A) : LinOp(f->get_executor(), gko::transpose(A->get_size())), A_(std::move(A)), settings_(f->settings())
    {
        if (size_[0] != size_[1]) throw DimensionMismatch(__FILE__, __LINE__, "the solver needs a square system matrix");
        precond_ = f->precond_ ? f->precond_ : (f->precond_factory_ ? std::shared_ptr<const LinOp>(f->precond_factory_->generate_impl(A_)) : nullptr);
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "solver::apply");
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        const int64_t n = size_[0], nrhs = db->cols();
        if (db->get_stride() != static_cast<size_type>(nrhs) || dx->get_stride() != static_cast<size_type>(nrhs)) GKO_NOT_SUPPORTED("solver vectors must be contiguous (stride == #columns)");
        array<char> ws(exec_, gkomi_krylov_workspace_bytes(n, nrhs));
        std::vector<double> info(2 + 2 * nrhs, 0.0);
        ::gko::detail::precond_callback cb(precond_.get(), exec_, static_cast<size_type>(n), static_cast<size_type>(nrhs));
        auto pfn = cb.fn;
        void* pctx = cb.ctx;
        // a Csr system matrix travels as its record (srow, row statistic): detail::system_callback
        ::gko::detail::system_callback mcb(A_.get(), exec_, static_cast<size_type>(n));
        GKOMI_CALL(OpDriver(nullptr, n, nrhs, mcb.fn, mcb.ctx, pfn, pctx, db->get_const_values(), dx->get_values(), settings_.max_iters,
                            settings_.reduction_factor, baseline_code(settings_.baseline), 8, ws.get_data(), ws.get_num_elems(), info.data()));
        last_iters_ = static_cast<int64_t>(info[0]);
        last_converged_ = info[1] != 0.0;
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const LinOp> A_;
    std::shared_ptr<const LinOp> precond_;
    stop::criterion_settings settings_;
    mutable int64_t last_iters_{-1};
    mutable bool last_converged_{false};
};
}  // namespace detail

#define GKOMI_KRYLOV_SOLVER(Name, driver, op_driver)                                               \
    template <typename V = double>                                                                 \
    class Name : public detail::krylov_solver<Name<V>, driver, op_driver> {                        \
        using base = detail::krylov_solver<Name<V>, driver, op_driver>;                            \
        friend class detail::factory_base<Name>;                                                   \
        Name(const typename base::Factory* f, std::shared_ptr<const LinOp> A) : base(f, std::move(A)) {} \
    }
namespace detail {
// one right-hand side: the fused 6-launch BiCGSTAB driver (csrc/krylov.hip)
inline int bicgstab_driver(gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* rp, const int32_t* ci, const double* v, int strategy, int64_t hint,
                           gkomi_apply_fn pfn, void* pctx, const double* b, double* x, int64_t max_iters, double reduction, int baseline, int64_t check_every, void* ws,
                           size_t ws_bytes, double* info)
{
    if (nrhs == 1) return gkomi_bicgstab_solve_fused_f64_i32(s, n, nnz, rp, ci, v, strategy, hint, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
    return gkomi_bicgstab_solve_f64_i32(s, n, nrhs, nnz, rp, ci, v, strategy, hint, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
}
inline int bicgstab_op_driver(gkomi_stream_t s, int64_t n, int64_t nrhs, gkomi_matrix_apply_fn mfn, void* mctx, gkomi_apply_fn pfn, void* pctx, const double* b, double* x,
                              int64_t max_iters, double reduction, int baseline, int64_t check_every, void* ws, size_t ws_bytes, double* info)
{
    if (nrhs == 1) return gkomi_bicgstab_solve_fused_op_f64(s, n, mfn, mctx, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
    return gkomi_bicgstab_solve_op_f64(s, n, nrhs, mfn, mctx, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
}
inline int fcg_driver(gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* rp, const int32_t* ci, const double* v, int strategy, int64_t hint,
                      gkomi_apply_fn pfn, void* pctx, const double* b, double* x, int64_t max_iters, double reduction, int baseline, int64_t check_every, void* ws,
                      size_t ws_bytes, double* info)
{
    if (nrhs == 1) return gkomi_fcg_solve_fused_f64_i32(s, n, nnz, rp, ci, v, strategy, hint, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
    return gkomi_fcg_solve_f64_i32(s, n, nrhs, nnz, rp, ci, v, strategy, hint, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
}
inline int fcg_op_driver(gkomi_stream_t s, int64_t n, int64_t nrhs, gkomi_matrix_apply_fn mfn, void* mctx, gkomi_apply_fn pfn, void* pctx, const double* b, double* x,
                         int64_t max_iters, double reduction, int baseline, int64_t check_every, void* ws, size_t ws_bytes, double* info)
{
    if (nrhs == 1) return gkomi_fcg_solve_fused_op_f64(s, n, mfn, mctx, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
    return gkomi_fcg_solve_op_f64(s, n, nrhs, mfn, mctx, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
}
inline int cgs_driver(gkomi_stream_t s, int64_t n, int64_t nrhs, int64_t nnz, const int32_t* rp, const int32_t* ci, const double* v, int strategy, int64_t hint,
                      gkomi_apply_fn pfn, void* pctx, const double* b, double* x, int64_t max_iters, double reduction, int baseline, int64_t check_every, void* ws,
                      size_t ws_bytes, double* info)
{
    if (nrhs == 1) return gkomi_cgs_solve_fused_f64_i32(s, n, nnz, rp, ci, v, strategy, hint, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
    return gkomi_cgs_solve_f64_i32(s, n, nrhs, nnz, rp, ci, v, strategy, hint, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
}
inline int cgs_op_driver(gkomi_stream_t s, int64_t n, int64_t nrhs, gkomi_matrix_apply_fn mfn, void* mctx, gkomi_apply_fn pfn, void* pctx, const double* b, double* x,
                         int64_t max_iters, double reduction, int baseline, int64_t check_every, void* ws, size_t ws_bytes, double* info)
{
    if (nrhs == 1) return gkomi_cgs_solve_fused_op_f64(s, n, mfn, mctx, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
    return gkomi_cgs_solve_op_f64(s, n, nrhs, mfn, mctx, pfn, pctx, b, x, max_iters, reduction, baseline, check_every, ws, ws_bytes, info);
}
}  // namespace detail
GKOMI_KRYLOV_SOLVER(Bicgstab, detail::bicgstab_driver, detail::bicgstab_op_driver);
GKOMI_KRYLOV_SOLVER(Fcg, detail::fcg_driver, detail::fcg_op_driver);
GKOMI_KRYLOV_SOLVER(Cgs, detail::cgs_driver, detail::cgs_op_driver);
#undef GKOMI_KRYLOV_SOLVER

// Bicg (include/ginkgo/core/solver/bicg.hpp): the transposed system matrix is
// built once at generate (the reference rebuilds it in every apply), and so is
// the transposed preconditioner (bicg.cpp:169-171 asks the preconditioner for its
// Transposable interface: Jacobi here).
template <typename V = double>
class Bicg : public LinOp {
public:
    class Factory : public detail::factory_base<Bicg> {};
    static Factory build() { return Factory{}; }
    std::shared_ptr<const LinOp> get_system_matrix() const { return A_; }
    int64_t get_last_iteration_count() const noexcept { return last_iters_; }
    bool has_converged() const noexcept { return last_converged_; }
protected:
    friend class detail::factory_base<Bicg>;
    Bicg(const Factory* f, std::shared_ptr<const LinOp> A) : LinOp(f->get_executor(), gko::transpose(A->get_size())), A_(std::move(A)), settings_(f->settings())
    {
        if (size_[0] != size_[1]) throw DimensionMismatch(__FILE__, __LINE__, "Bicg needs a square system matrix");
        precond_ = f->precond_ ? f->precond_ : (f->precond_factory_ ? std::shared_ptr<const LinOp>(f->precond_factory_->generate_impl(A_)) : nullptr);
        if (precond_) {
            auto tr = dynamic_cast<const Transposable*>(precond_.get());
            if (tr == nullptr) GKO_NOT_SUPPORTED("Bicg: the preconditioner must be Transposable");
            precond_t_ = std::shared_ptr<const LinOp>(tr->conj_transpose());
        }
        At_ = as<const matrix::Csr<V, int32>>(A_.get())->transpose();
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "bicg::apply");
        auto csr = as<const matrix::Csr<V, int32>>(A_.get());
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        const int64_t n = size_[0], nrhs = db->cols();
        if (db->get_stride() != static_cast<size_type>(nrhs) || dx->get_stride() != static_cast<size_type>(nrhs)) GKO_NOT_SUPPORTED("solver vectors must be contiguous (stride == #columns)");
        array<char> ws(exec_, gkomi_krylov_workspace_bytes(n, nrhs));
        std::vector<double> info(2 + 2 * nrhs, 0.0);
        ::gko::detail::linop_callback cb{precond_.get(), exec_, static_cast<size_type>(n), static_cast<size_type>(nrhs)};
        ::gko::detail::linop_callback cbt{precond_t_.get(), exec_, static_cast<size_type>(n), static_cast<size_type>(nrhs)};
        auto pfn = precond_ ? &::gko::detail::linop_callback::call : nullptr;
        GKOMI_CALL(gkomi_bicg_solve_f64_i32(nullptr, n, nrhs, csr->get_num_stored_elements(), csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(),
                                            At_->get_const_row_ptrs(), At_->get_const_col_idxs(), At_->get_const_values(), csr->get_strategy()->get_code(), csr->get_max_row_nnz(),
                                            pfn, precond_ ? &cb : nullptr, pfn, precond_ ? &cbt : nullptr, db->get_const_values(), dx->get_values(), settings_.max_iters, settings_.reduction_factor,
                                            detail::baseline_code(settings_.baseline), 8, ws.get_data(), ws.get_num_elems(), info.data()));
        last_iters_ = static_cast<int64_t>(info[0]);
        last_converged_ = info[1] != 0.0;
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const LinOp> A_;
    std::shared_ptr<const LinOp> precond_, precond_t_;
    std::unique_ptr<matrix::Csr<V, int32>> At_;
    stop::criterion_settings settings_;
    mutable int64_t last_iters_{-1};
    mutable bool last_converged_{false};
};

// Ir (include/ginkgo/core/solver/ir.hpp): x += relaxation_factor * solver(b - A x);
// without an inner solver it is the Richardson iteration.  with_preconditioner /
// with_generated_preconditioner of the common factory play with_solver /
// with_generated_solver.
template <typename V = double>
class Ir : public LinOp {
public:
    class Factory : public detail::factory_base<Ir> {
    public:
        Factory& with_relaxation_factor(V r) { relaxation_factor_ = r; return *this; }
        Factory& with_solver(std::shared_ptr<const LinOpFactory> f) { this->precond_factory_ = std::move(f); return *this; }
        Factory& with_generated_solver(std::shared_ptr<const LinOp> p) { this->precond_ = std::move(p); return *this; }
        V relaxation_factor_{1.0};
    };
    static Factory build() { return Factory{}; }
    std::shared_ptr<const LinOp> get_system_matrix() const { return A_; }
    std::shared_ptr<const LinOp> get_solver() const { return inner_; }
    int64_t get_last_iteration_count() const noexcept { return last_iters_; }
    bool has_converged() const noexcept { return last_converged_; }
protected:
    friend class detail::factory_base<Ir>;
    Ir(const Factory* f, std::shared_ptr<const LinOp> A) : LinOp(f->get_executor(), gko::transpose(A->get_size())), A_(std::move(A)), settings_(f->settings()), relaxation_factor_(f->relaxation_factor_)
    {
        if (size_[0] != size_[1]) throw DimensionMismatch(__FILE__, __LINE__, "Ir needs a square system matrix");
        inner_ = f->precond_ ? f->precond_ : (f->precond_factory_ ? std::shared_ptr<const LinOp>(f->precond_factory_->generate_impl(A_)) : nullptr);
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "ir::apply");
        auto csr = as<const matrix::Csr<V, int32>>(A_.get());
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        const int64_t n = size_[0], nrhs = db->cols();
        if (db->get_stride() != static_cast<size_type>(nrhs) || dx->get_stride() != static_cast<size_type>(nrhs)) GKO_NOT_SUPPORTED("solver vectors must be contiguous (stride == #columns)");
        array<char> ws(exec_, gkomi_krylov_workspace_bytes(n, nrhs));
        std::vector<double> info(2 + 2 * nrhs, 0.0);
        ::gko::detail::linop_callback cb{inner_.get(), exec_, static_cast<size_type>(n), static_cast<size_type>(nrhs)};
        GKOMI_CALL(gkomi_ir_solve_f64_i32(nullptr, n, nrhs, csr->get_num_stored_elements(), csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(),
                                          csr->get_strategy()->get_code(), csr->get_max_row_nnz(), inner_ ? &::gko::detail::linop_callback::call : nullptr, inner_ ? &cb : nullptr,
                                          relaxation_factor_, db->get_const_values(), dx->get_values(), settings_.max_iters, settings_.reduction_factor,
                                          detail::baseline_code(settings_.baseline), ws.get_data(), ws.get_num_elems(), info.data()));
        last_iters_ = static_cast<int64_t>(info[0]);
        last_converged_ = info[1] != 0.0;
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const LinOp> A_;
    std::shared_ptr<const LinOp> inner_;
    stop::criterion_settings settings_;
    V relaxation_factor_;
    mutable int64_t last_iters_{-1};
    mutable bool last_converged_{false};
};

template <typename V = double>
class Gmres : public LinOp {
public:
    class Factory : public detail::factory_base<Gmres> {
    public:
        Factory& with_krylov_dim(size_type d) { krylov_dim_ = d; return *this; }
        size_type krylov_dim_{100};  // gmres.hpp:57,161
    };
    static Factory build() { return Factory{}; }
    size_type get_krylov_dim() const noexcept { return krylov_dim_; }
    int64_t get_last_iteration_count() const noexcept { return last_iters_; }
    bool has_converged() const noexcept { return last_converged_; }
protected:
    friend class detail::factory_base<Gmres>;
    Gmres(const Factory* f, std::shared_ptr<const LinOp> A) : LinOp(f->get_executor(), gko::transpose(A->get_size())), A_(std::move(A)), settings_(f->settings()), krylov_dim_(f->krylov_dim_)
    {
        precond_ = f->precond_ ? f->precond_ : (f->precond_factory_ ? std::shared_ptr<const LinOp>(f->precond_factory_->generate_impl(A_)) : nullptr);
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "gmres::apply");
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        const int64_t n = size_[0], nrhs = db->cols();
        if (db->get_stride() != static_cast<size_type>(nrhs) || dx->get_stride() != static_cast<size_type>(nrhs)) GKO_NOT_SUPPORTED("solver vectors must be contiguous (stride == #columns)");
        array<char> ws(exec_, gkomi_gmres_workspace_bytes(n, nrhs, krylov_dim_));
        std::vector<double> info(2 + 2 * nrhs, 0.0);
        ::gko::detail::precond_callback cb(precond_.get(), exec_, static_cast<size_type>(n), static_cast<size_type>(nrhs));
        auto pfn = cb.fn;
        void* pctx = cb.ctx;
        // a Csr system matrix travels as its record (srow, row statistic): detail::system_callback
        ::gko::detail::system_callback mcb(A_.get(), exec_, static_cast<size_type>(n));
        GKOMI_CALL(gkomi_gmres_solve_op_f64(nullptr, n, nrhs, mcb.fn, mcb.ctx, pfn, pctx, db->get_const_values(), dx->get_values(), krylov_dim_,
                                            settings_.max_iters, settings_.reduction_factor, detail::baseline_code(settings_.baseline), ws.get_data(), ws.get_num_elems(), info.data()));
        last_iters_ = static_cast<int64_t>(info[0]);
        last_converged_ = info[1] != 0.0;
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const LinOp> A_;
    std::shared_ptr<const LinOp> precond_;
    stop::criterion_settings settings_;
    size_type krylov_dim_;
    mutable int64_t last_iters_{-1};
    mutable bool last_converged_{false};
};

// LowerTrs / UpperTrs (include/ginkgo/core/solver/triangular.hpp)
template <bool Lower, typename V = double, typename I = int32>
class Trs : public LinOp {
public:
    class Factory : public LinOpFactory {
    public:
        Factory() : LinOpFactory(nullptr) {}
        Factory& with_unit_diagonal(bool u) { unit_ = u; return *this; }
        Factory& with_num_rhs(size_type) { return *this; }
        std::shared_ptr<Factory> on(std::shared_ptr<const Executor> exec) const { auto f = std::make_shared<Factory>(*this); f->exec_ = std::move(exec); return f; }
        std::unique_ptr<Trs> generate(std::shared_ptr<const LinOp> A) const { return std::unique_ptr<Trs>(new Trs(this->exec_, unit_, std::move(A))); }
        std::unique_ptr<LinOp> generate_impl(std::shared_ptr<const LinOp> A) const override { return generate(std::move(A)); }
        bool unit_{false};
    };
    static Factory build() { return Factory{}; }
protected:
public:
    // what generate() found: dependency levels of the factor and whether the level-scheduled solve is used
    int64_t get_num_levels() const noexcept { return nlevels_; }
    bool uses_level_schedule() const noexcept { return planned_; }
    bool uses_brick_plan() const noexcept { return bricks_ != nullptr; }
    bool has_unit_diagonal() const noexcept { return unit_; }
    // what gkomi_ilu_ctx carries of an analysed factor: its brick plan, else its level plan, else nothing
    void fill_callback_record(void*& level_plan, int64_t& nslices, int64_t& entries, int64_t& max_deps, gkomi_trs_bricks*& bricks, void*& bricks_plan) const
    {
        level_plan = nullptr; bricks = nullptr; bricks_plan = nullptr; nslices = 0; entries = 0; max_deps = -1;
        if (bricks_ != nullptr) {
            bricks = bricks_;
            bricks_plan = const_cast<char*>(plan_.get_const_data());
        } else if (planned_) {
            level_plan = const_cast<char*>(plan_.get_const_data());
            nslices = nslices_; entries = entries_; max_deps = max_deps_;
        }
    }
    ~Trs() override { gkomi_trs_bricks_destroy(bricks_); }
    // a solve that gave up (spin bound) left NaNs in x; sticky until the next generate
    bool has_overrun() const
    {
        int flag = 0;
        if (bricks_ != nullptr) GKOMI_CALL(gkomi_trs_bricks_check_overrun(nullptr, plan_.get_const_data(), &flag));
        else if (planned_) GKOMI_CALL(gkomi_trs_plan_check_overrun(nullptr, plan_.get_const_data(), &flag));
        else GKOMI_CALL(gkomi_trs_check_overrun(nullptr, ws_.get_const_data(), &flag));
        return flag != 0;
    }
protected:
    // LowerTrs / UpperTrs::generate (core/solver/lower_trs.cpp generate -> lower_trs::generate): the
    // dependency-level analysis of the factor, kept for every apply (the reference's SolveStruct)
    Trs(std::shared_ptr<const Executor> exec, bool unit, std::shared_ptr<const LinOp> A)
        : LinOp(exec, A->get_size()), unit_(unit), A_(std::move(A)), ws_(exec, gkomi_trs_workspace_bytes()), plan_(exec)
    {
        if (!exec_->is_device()) return;
        ws_.fill(0);
        auto csr = as<const matrix::Csr<V, I>>(A_.get());
        const int64_t n = static_cast<int64_t>(size_[0]);
        // factors of grid problems (many levels, stencil-shaped): the brick plan -- bricks solved out of LDS,
        // about one LDS step per level instead of one memory hand-off (csrc/trs_bricks.hip).  Its analysis runs
        // on the device and says itself how many levels the factor has if its box geometry holds, so a factor
        // that takes the brick plan never pays for the level analysis (generate on the 108^3 ILU: 55 -> 7 ms).
        if (n >= 2) {
            const int err = gkomi_trs_bricks_create_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), Lower ? 1 : 0, 0, 0, 0, &bricks_);
            if (err != GKOMI_SUCCESS && err != GKOMI_ENOTSUPPORTED) GKOMI_CALL(err);
            if (bricks_ != nullptr) {
                int64_t info[8] = {};
                GKOMI_CALL(gkomi_trs_bricks_info(bricks_, info));
                const int64_t levels = gkomi_trs_bricks_levels_estimate(bricks_);
                if (gkomi_trs_prefer_bricks(n, levels, info[1]) != 0) {  // the cost model of profiles/r02_trs_bricks.md (+ the single-workgroup solve of small factors)
                    nlevels_ = levels;
                    plan_.resize_and_reset(gkomi_trs_bricks_plan_bytes(bricks_));
                    GKOMI_CALL(gkomi_trs_bricks_numeric_f64_i32(nullptr, bricks_, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(), plan_.get_data(),
                                                                plan_.get_num_elems()));
                    return;
                }
                gkomi_trs_bricks_destroy(bricks_);
                bricks_ = nullptr;
            }
        }
        array<char> symbolic(exec_, gkomi_trs_symbolic_workspace_bytes(n));
        int64_t out[4] = {};
        GKOMI_CALL(gkomi_trs_analyse_symbolic_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), Lower ? 1 : 0, symbolic.get_data(),
                                                  symbolic.get_num_elems(), out));
        nslices_ = out[0]; entries_ = out[1]; nlevels_ = out[2]; max_deps_ = out[3];
        // wide levels: level-scheduled; chains and narrow bands: the analysis-free kernel's in-workgroup hand-offs
        planned_ = gkomi_trs_use_plan(n, nlevels_, max_deps_) != 0;
        if (planned_) {
            plan_.resize_and_reset(gkomi_trs_plan_bytes(nslices_, entries_));
            GKOMI_CALL(gkomi_trs_analyse_numeric_f64_i32(nullptr, n, csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(), Lower ? 1 : 0,
                                                         symbolic.get_const_data(), nslices_, entries_, nlevels_, plan_.get_data(), plan_.get_num_elems()));
            GKOMI_CALL(gkomi_synchronize(nullptr));  // `symbolic` goes out of scope
        }
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        ::gko::detail::require_device(exec_, "trs::solve");
        auto csr = as<const matrix::Csr<V, I>>(A_.get());
        auto db = matrix::detail_fmt::dense(b); auto dx = matrix::detail_fmt::dense(x);
        if (bricks_ != nullptr) {
            GKOMI_CALL(gkomi_trs_bricks_solve_f64(nullptr, bricks_, const_cast<char*>(plan_.get_const_data()), db->cols(), unit_ ? 1 : 0, db->get_const_values(), db->get_stride(),
                                                  dx->get_values(), dx->get_stride()));
            return;
        }
        if (planned_) {
            GKOMI_CALL(gkomi_trs_solve_plan_f64(nullptr, size_[0], db->cols(), const_cast<char*>(plan_.get_const_data()), nslices_, entries_, max_deps_, unit_ ? 1 : 0,
                                                db->get_const_values(), db->get_stride(), dx->get_values(), dx->get_stride()));
            return;
        }
        auto fn = Lower ? gkomi_lower_trs_solve_f64_i32 : gkomi_upper_trs_solve_f64_i32;
        GKOMI_CALL(fn(nullptr, size_[0], db->cols(), csr->get_const_row_ptrs(), csr->get_const_col_idxs(), csr->get_const_values(), unit_ ? 1 : 0, db->get_const_values(), db->get_stride(),
                      dx->get_values(), dx->get_stride(), const_cast<char*>(ws_.get_const_data()), ws_.get_num_elems()));
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    bool unit_;
    std::shared_ptr<const LinOp> A_;
    array<char> ws_;
    array<char> plan_;
    int64_t nslices_{0}, entries_{0}, nlevels_{0}, max_deps_{-1};
    bool planned_{false};
    gkomi_trs_bricks* bricks_{nullptr};  // host-side analysis; its device plan lives in plan_
};
template <typename V = double, typename I = int32>
using LowerTrs = Trs<true, V, I>;
template <typename V = double, typename I = int32>
using UpperTrs = Trs<false, V, I>;
}  // namespace solver

// ---- ParILU + Ilu (core/factorization/par_ilu.cpp:74-163, preconditioner/ilu.hpp:265-305) ------
namespace factorization {
template <typename V = double, typename I = int32>
class ParIlu {
public:
    using matrix_type = matrix::Csr<V, I>;
    class Factory {
    public:
        Factory& with_iterations(size_type n) { iterations_ = n; return *this; }
        Factory& with_skip_sorting(bool) { return *this; }
        std::shared_ptr<Factory> on(std::shared_ptr<const Executor> exec) const { auto f = std::make_shared<Factory>(*this); f->exec_ = std::move(exec); return f; }
        std::unique_ptr<ParIlu> generate(std::shared_ptr<const LinOp> A) const { return std::unique_ptr<ParIlu>(new ParIlu(exec_, iterations_, std::move(A))); }
        std::shared_ptr<const Executor> exec_;
        size_type iterations_{0};
    };
    static Factory build() { return Factory{}; }
    std::shared_ptr<const matrix_type> get_l_factor() const { return l_; }
    std::shared_ptr<const matrix_type> get_u_factor() const { return u_; }
protected:
    ParIlu(std::shared_ptr<const Executor> exec, size_type iterations, std::shared_ptr<const LinOp> A)
    {
        detail::require_device(exec, "par_ilu_factorization");
        auto src = as<const matrix_type>(A.get());
        const size_type n = src->get_size()[0];
        if (n != src->get_size()[1]) throw DimensionMismatch(__FILE__, __LINE__, "ParIlu needs a square matrix");
        // working copy with explicit diagonal
        array<I> rp(exec, n + 1);
        exec->copy(n + 1, src->get_const_row_ptrs(), rp.get_data());
        size_type nnz = src->get_num_stored_elements();
        array<char> fws(exec, gkomi_factorization_workspace_bytes(n));
        int64_t missing = 0;
        GKOMI_CALL(gkomi_factorization_count_missing_diagonal_i32(nullptr, n, n, rp.get_const_data(), src->get_const_col_idxs(), fws.get_data(), fws.get_num_elems(), &missing));
        array<I> ci(exec, nnz + missing);
        array<V> v(exec, nnz + missing);
        if (missing) {
            GKOMI_CALL(gkomi_factorization_add_diagonal_elements_f64_i32(nullptr, n, n, rp.get_data(), src->get_const_col_idxs(), src->get_const_values(), ci.get_data(), v.get_data(), fws.get_const_data()));
        } else {
            exec->copy(nnz, src->get_const_col_idxs(), ci.get_data());
            exec->copy(nnz, src->get_const_values(), v.get_data());
        }
        nnz += missing;
        array<I> lrp(exec, n + 1), urp(exec, n + 1);
        array<char> sws(exec, gkomi_prefix_sum_workspace_bytes(n + 1) + 8);
        GKOMI_CALL(gkomi_factorization_initialize_row_ptrs_l_u_i32(nullptr, n, rp.get_const_data(), ci.get_const_data(), lrp.get_data(), urp.get_data(), sws.get_data(), sws.get_num_elems()));
        const size_type lnnz = exec->copy_val_to_host(lrp.get_const_data() + n), unnz = exec->copy_val_to_host(urp.get_const_data() + n);
        array<I> lc(exec, lnnz), uc(exec, unnz);
        array<V> lv(exec, lnnz), uv(exec, unnz);
        GKOMI_CALL(gkomi_factorization_initialize_l_u_f64_i32(nullptr, n, rp.get_const_data(), ci.get_const_data(), v.get_const_data(), lrp.get_const_data(), lc.get_data(), lv.get_data(), urp.get_const_data(), uc.get_data(), uv.get_data()));
        array<I> utrp(exec, n + 1), utc(exec, unnz);
        array<V> utv(exec, unnz);
        array<char> tws(exec, gkomi_csr_transpose_workspace_bytes(n));
        GKOMI_CALL(gkomi_csr_transpose_f64_i32(nullptr, n, n, unnz, urp.get_const_data(), uc.get_const_data(), uv.get_const_data(), utrp.get_data(), utc.get_data(), utv.get_data(), tws.get_data(), tws.get_num_elems()));
        array<I> rows(exec, nnz);
        GKOMI_CALL(gkomi_convert_ptrs_to_idxs_i32(nullptr, rp.get_const_data(), n, rows.get_data()));
        GKOMI_CALL(gkomi_par_ilu_compute_l_u_factors_f64_i32(nullptr, iterations, nnz, rows.get_const_data(), ci.get_const_data(), v.get_const_data(), lrp.get_const_data(), lc.get_const_data(), lv.get_data(),
                                                             utrp.get_const_data(), utc.get_const_data(), utv.get_data()));
        GKOMI_CALL(gkomi_csr_transpose_f64_i32(nullptr, n, n, unnz, utrp.get_const_data(), utc.get_const_data(), utv.get_const_data(), urp.get_data(), uc.get_data(), uv.get_data(), tws.get_data(), tws.get_num_elems()));
        auto l = matrix_type::create(exec); auto u = matrix_type::create(exec);
        l->adopt(dim<2>(n, n), std::move(lrp), std::move(lc), std::move(lv));
        u->adopt(dim<2>(n, n), std::move(urp), std::move(uc), std::move(uv));
        l_ = std::move(l); u_ = std::move(u);
    }
    std::shared_ptr<matrix_type> l_, u_;
};
}  // namespace factorization

namespace preconditioner {
template <typename V = double, typename I = int32>
class Ilu : public LinOp, public Transposable, public ::gko::detail::native_preconditioner {
public:
    // L^-1 then U^-1 as gkomi_ilu_apply_cb + record: the two brick solves of an apply then run as a chain (no launch
    // that pre-fills an output with the ready flags), and no Dense objects are built per apply.  The record, the
    // intermediate vector and the analysis-free kernels' workspace live in the object (the reference's Ilu caches its
    // intermediate too): one solve at a time per preconditioner object.
    bool native_callback(gkomi_apply_fn& fn, void*& ctx, size_type nrhs) const override
    {
        if (!std::is_same<V, double>::value || !std::is_same<I, int32>::value) return false;
        auto l = dynamic_cast<const solver::LowerTrs<V, I>*>(l_solver_.get());
        auto u = dynamic_cast<const solver::UpperTrs<V, I>*>(u_solver_.get());
        if (l == nullptr || u == nullptr || u->has_unit_diagonal()) return false;
        const size_type n = size_[0];
        if (mid_.get_num_elems() != n * nrhs || mid_.get_executor() == nullptr) {
            mid_ = array<V>(exec_, n * nrhs);
            mid_.fill(V{0});
            record_.pad_ = 0;
        }
        if (tws_.get_num_elems() == 0) {
            tws_ = array<char>(exec_, gkomi_trs_workspace_bytes());
            tws_.fill(0);
        }
        const int32_t chain_state = record_.pad_;
        record_ = gkomi_ilu_ctx{};
        record_.pad_ = chain_state;
        record_.n = static_cast<int64_t>(n);
        record_.nrhs = static_cast<int64_t>(nrhs);
        record_.l_row_ptrs = l_factor_->get_const_row_ptrs();
        record_.l_col_idxs = l_factor_->get_const_col_idxs();
        record_.l_vals = l_factor_->get_const_values();
        record_.u_row_ptrs = u_factor_->get_const_row_ptrs();
        record_.u_col_idxs = u_factor_->get_const_col_idxs();
        record_.u_vals = u_factor_->get_const_values();
        record_.intermediate = mid_.get_data();
        record_.trs_workspace = tws_.get_data();
        record_.trs_workspace_bytes = tws_.get_num_elems();
        record_.l_unit_diag = l->has_unit_diagonal() ? 1 : 0;
        l->fill_callback_record(record_.l_plan, record_.l_nslices, record_.l_entries, record_.l_max_deps, record_.l_bricks, record_.l_bricks_plan);
        u->fill_callback_record(record_.u_plan, record_.u_nslices, record_.u_entries, record_.u_max_deps, record_.u_bricks, record_.u_bricks_plan);
        fn = &gkomi_ilu_apply_cb;
        ctx = &record_;
        return true;
    }
    class Factory : public LinOpFactory {
    public:
        Factory() : LinOpFactory(nullptr) {}
        Factory& with_factorization_iterations(size_type n) { iterations_ = n; return *this; }
        std::shared_ptr<Factory> on(std::shared_ptr<const Executor> exec) const { auto f = std::make_shared<Factory>(*this); f->exec_ = std::move(exec); return f; }
        std::unique_ptr<Ilu> generate(std::shared_ptr<const LinOp> A) const { return std::unique_ptr<Ilu>(new Ilu(this->exec_, iterations_, std::move(A))); }
        std::unique_ptr<LinOp> generate_impl(std::shared_ptr<const LinOp> A) const override { return generate(std::move(A)); }
        size_type iterations_{0};
    };
    static Factory build() { return Factory{}; }
    std::shared_ptr<const LinOp> get_l_solver() const { return l_solver_; }
    std::shared_ptr<const LinOp> get_u_solver() const { return u_solver_; }
    // Ilu::transpose (include/ginkgo/core/preconditioner/ilu.hpp:200-230): (U^-1 L^-1)^T =
    // L^-T U^-T, i.e. the lower solver of U^T followed by the upper solver of L^T
    std::unique_ptr<LinOp> transpose() const override
    {
        std::unique_ptr<Ilu> t(new Ilu(exec_, size_));
        t->l_factor_ = std::shared_ptr<const matrix::Csr<V, I>>(u_factor_->transpose());
        t->u_factor_ = std::shared_ptr<const matrix::Csr<V, I>>(l_factor_->transpose());
        t->l_solver_ = solver::LowerTrs<V, I>::build().on(exec_)->generate(t->l_factor_);
        t->u_solver_ = solver::UpperTrs<V, I>::build().on(exec_)->generate(t->u_factor_);
        return std::unique_ptr<LinOp>(t.release());
    }
protected:
    Ilu(std::shared_ptr<const Executor> exec, dim<2> size) : LinOp(std::move(exec), size) {}
    Ilu(std::shared_ptr<const Executor> exec, size_type iterations, std::shared_ptr<const LinOp> A) : LinOp(exec, A->get_size())
    {
        auto fact = factorization::ParIlu<V, I>::build().with_iterations(iterations).on(exec)->generate(std::move(A));
        l_factor_ = fact->get_l_factor();
        u_factor_ = fact->get_u_factor();
        l_solver_ = solver::LowerTrs<V, I>::build().on(exec)->generate(l_factor_);
        u_solver_ = solver::UpperTrs<V, I>::build().on(exec)->generate(u_factor_);
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        auto db = matrix::detail_fmt::dense(b);
        auto mid = matrix::Dense<V>::create(exec_, db->get_size());
        l_solver_->apply(b, mid.get());
        u_solver_->apply(mid.get(), x);
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const matrix::Csr<V, I>> l_factor_, u_factor_;
    std::shared_ptr<const LinOp> l_solver_, u_solver_;
    mutable gkomi_ilu_ctx record_{};
    mutable array<V> mid_;
    mutable array<char> tws_;
};
}  // namespace preconditioner

// ---- ParIc + Ic (core/factorization/par_ic.cpp:70-145, include/ginkgo/core/preconditioner/ic.hpp) ------
namespace factorization {
template <typename V = double, typename I = int32>
class ParIc {
public:
    using matrix_type = matrix::Csr<V, I>;
    class Factory {
    public:
        Factory& with_iterations(size_type n) { iterations_ = n; return *this; }
        Factory& with_skip_sorting(bool) { return *this; }
        Factory& with_both_factors(bool) { return *this; }
        std::shared_ptr<Factory> on(std::shared_ptr<const Executor> exec) const { auto f = std::make_shared<Factory>(*this); f->exec_ = std::move(exec); return f; }
        std::unique_ptr<ParIc> generate(std::shared_ptr<const LinOp> A) const { return std::unique_ptr<ParIc>(new ParIc(exec_, iterations_, std::move(A))); }
        std::shared_ptr<const Executor> exec_;
        size_type iterations_{0};
    };
    static Factory build() { return Factory{}; }
    std::shared_ptr<const matrix_type> get_l_factor() const { return l_; }
    std::shared_ptr<const matrix_type> get_lt_factor() const { return lt_; }
protected:
    ParIc(std::shared_ptr<const Executor> exec, size_type iterations, std::shared_ptr<const LinOp> A)
    {
        detail::require_device(exec, "par_ic_factorization");
        auto src = as<const matrix_type>(A.get());
        const size_type n = src->get_size()[0];
        if (n != src->get_size()[1]) throw DimensionMismatch(__FILE__, __LINE__, "ParIc needs a square matrix");
        array<I> rp(exec, n + 1);
        exec->copy(n + 1, src->get_const_row_ptrs(), rp.get_data());
        size_type nnz = src->get_num_stored_elements();
        array<char> fws(exec, gkomi_factorization_workspace_bytes(n));
        int64_t missing = 0;
        GKOMI_CALL(gkomi_factorization_count_missing_diagonal_i32(nullptr, n, n, rp.get_const_data(), src->get_const_col_idxs(), fws.get_data(), fws.get_num_elems(), &missing));
        array<I> ci(exec, nnz + missing);
        array<V> v(exec, nnz + missing);
        if (missing) {
            GKOMI_CALL(gkomi_factorization_add_diagonal_elements_f64_i32(nullptr, n, n, rp.get_data(), src->get_const_col_idxs(), src->get_const_values(), ci.get_data(), v.get_data(), fws.get_const_data()));
        } else {
            exec->copy(nnz, src->get_const_col_idxs(), ci.get_data());
            exec->copy(nnz, src->get_const_values(), v.get_data());
        }
        array<I> lrp(exec, n + 1);
        array<char> sws(exec, gkomi_prefix_sum_workspace_bytes(n + 1) + 8);
        GKOMI_CALL(gkomi_factorization_initialize_row_ptrs_l_i32(nullptr, n, rp.get_const_data(), ci.get_const_data(), lrp.get_data(), sws.get_data(), sws.get_num_elems()));
        const size_type lnnz = exec->copy_val_to_host(lrp.get_const_data() + n);
        array<I> lc(exec, lnnz), rows(exec, lnnz);
        array<V> lv(exec, lnnz);
        GKOMI_CALL(gkomi_factorization_initialize_l_f64_i32(nullptr, n, rp.get_const_data(), ci.get_const_data(), v.get_const_data(), lrp.get_const_data(), lc.get_data(), lv.get_data(), 0));
        array<V> a_vals(exec, lv);  // the COO copy of the lower triangle (par_ic.cpp:121-131)
        GKOMI_CALL(gkomi_convert_ptrs_to_idxs_i32(nullptr, lrp.get_const_data(), n, rows.get_data()));
        GKOMI_CALL(gkomi_par_ic_init_factor_f64_i32(nullptr, n, lrp.get_const_data(), lc.get_const_data(), lv.get_data()));
        GKOMI_CALL(gkomi_par_ic_compute_factor_f64_i32(nullptr, iterations, lnnz, rows.get_const_data(), a_vals.get_const_data(), lrp.get_const_data(), lc.get_const_data(), lv.get_data()));
        auto l = matrix_type::create(exec);
        l->adopt(dim<2>(n, n), std::move(lrp), std::move(lc), std::move(lv));
        lt_ = l->transpose();
        l_ = std::move(l);
    }
    std::shared_ptr<matrix_type> l_, lt_;
};
}  // namespace factorization

namespace preconditioner {
// Ic::apply = L^-1 then L^-H (ic.hpp: two triangular solves, like Ilu)
template <typename V = double, typename I = int32>
class Ic : public LinOp, public Transposable {
public:
    // (L L^T)^-1 is symmetric: the transpose applies the same two solves
    std::unique_ptr<LinOp> transpose() const override { return std::unique_ptr<LinOp>(new Ic(*this)); }
    class Factory : public LinOpFactory {
    public:
        Factory() : LinOpFactory(nullptr) {}
        Factory& with_factorization_iterations(size_type n) { iterations_ = n; return *this; }
        std::shared_ptr<Factory> on(std::shared_ptr<const Executor> exec) const { auto f = std::make_shared<Factory>(*this); f->exec_ = std::move(exec); return f; }
        std::unique_ptr<Ic> generate(std::shared_ptr<const LinOp> A) const { return std::unique_ptr<Ic>(new Ic(this->exec_, iterations_, std::move(A))); }
        std::unique_ptr<LinOp> generate_impl(std::shared_ptr<const LinOp> A) const override { return generate(std::move(A)); }
        size_type iterations_{0};
    };
    static Factory build() { return Factory{}; }
    std::shared_ptr<const LinOp> get_l_solver() const { return l_solver_; }
    std::shared_ptr<const LinOp> get_lh_solver() const { return lh_solver_; }
protected:
    Ic(const Ic&) = default;
    Ic(std::shared_ptr<const Executor> exec, size_type iterations, std::shared_ptr<const LinOp> A) : LinOp(exec, A->get_size())
    {
        auto fact = factorization::ParIc<V, I>::build().with_iterations(iterations).on(exec)->generate(std::move(A));
        l_solver_ = solver::LowerTrs<V, I>::build().on(exec)->generate(fact->get_l_factor());
        lh_solver_ = solver::UpperTrs<V, I>::build().on(exec)->generate(fact->get_lt_factor());
    }
    void apply_impl(const LinOp* b, LinOp* x) const override
    {
        auto db = matrix::detail_fmt::dense(b);
        auto mid = matrix::Dense<V>::create(exec_, db->get_size());
        l_solver_->apply(b, mid.get());
        lh_solver_->apply(mid.get(), x);
    }
    void apply_impl(const LinOp* alpha, const LinOp* b, const LinOp* beta, LinOp* x) const override
    {
        auto dx = matrix::detail_fmt::dense(x);
        auto x_clone = dx->clone();
        this->apply_impl(b, x_clone.get());
        dx->scale(matrix::detail_fmt::dense(beta));
        dx->add_scaled(matrix::detail_fmt::dense(alpha), x_clone.get());
    }
    std::shared_ptr<const LinOp> l_solver_, lh_solver_;
};
}  // namespace preconditioner

}  // namespace gko

#include "distributed.hpp"

#endif  // GKOMI_GINKGO_HPP_
