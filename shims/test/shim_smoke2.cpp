// Runs the REST of the shims on the device (round 4; shim_smoke.cpp has csr::spmv, compute_norm2, lower_trs, ell, sellp,
// coo::spmv2, prefix_sum<int32>, fcg::step_1): every kernel INTEGRATION.md's table binds, each against the mirror's own
// apply or a closed-form answer.  Prints one "ran <namespace>::<kernel> ok|WRONG" line per kernel
// (tests/test_cpp_mirror.py compares the list with INTEGRATION.md) and returns the number of wrong ones.
// (The distributed bindings read the Partition's arrays as DEVICE pointers -- they live on the partition's executor in
// a reference tree, partition.hpp:300-340, and in the mirror since round 4.)
#include "prelude_mirror.hpp"
#include <cmath>
#include <cstdio>
#include <vector>

namespace gko { namespace kernels { namespace hip {
using Vec = matrix::Dense<double>;
using Mtx = matrix::Csr<double, int32>;
using Status = array<stopping_status>;
namespace csr {
void spmv(std::shared_ptr<const HipExecutor>, const Mtx*, const Vec*, Vec*);
void advanced_spmv(std::shared_ptr<const HipExecutor>, const Vec*, const Mtx*, const Vec*, const Vec*, Vec*);
void spmv(std::shared_ptr<const HipExecutor>, const matrix::Csr<double, int64>*, const Vec*, Vec*);
void advanced_spmv(std::shared_ptr<const HipExecutor>, const Vec*, const matrix::Csr<double, int64>*, const Vec*, const Vec*, Vec*);
}
namespace dense {
void fill(std::shared_ptr<const HipExecutor>, Vec*, double);
void copy(std::shared_ptr<const HipExecutor>, const Vec*, Vec*);
void scale(std::shared_ptr<const HipExecutor>, const Vec*, Vec*);
void inv_scale(std::shared_ptr<const HipExecutor>, const Vec*, Vec*);
void add_scaled(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*);
void sub_scaled(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*);
void compute_dot(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, array<char>&);
void compute_conj_dot(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, array<char>&);
void compute_dot_dispatch(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, array<char>&);
void compute_norm2(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, array<char>&);
void compute_norm2_dispatch(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, array<char>&);
void compute_squared_norm2(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, array<char>&);
void compute_sqrt(std::shared_ptr<const HipExecutor>, Vec*);
void row_gather(std::shared_ptr<const HipExecutor>, const array<int32>*, const Vec*, Vec*);
}
namespace ell { void advanced_spmv(std::shared_ptr<const HipExecutor>, const Vec*, const matrix::Ell<double, int32>*, const Vec*, const Vec*, Vec*); }
namespace sellp {
void spmv(std::shared_ptr<const HipExecutor>, const matrix::Sellp<double, int32>*, const Vec*, Vec*);
void compute_slice_sets(std::shared_ptr<const HipExecutor>, const array<int32>&, size_type, size_type, size_type*, size_type*);
}
namespace coo {
void spmv(std::shared_ptr<const HipExecutor>, const matrix::Coo<double, int32>*, const Vec*, Vec*);
void advanced_spmv(std::shared_ptr<const HipExecutor>, const Vec*, const matrix::Coo<double, int32>*, const Vec*, const Vec*, Vec*);
void advanced_spmv2(std::shared_ptr<const HipExecutor>, const Vec*, const matrix::Coo<double, int32>*, const Vec*, Vec*);
}
namespace hybrid { void compute_coo_row_ptrs(std::shared_ptr<const HipExecutor>, const array<size_type>&, size_type, int64*); }
namespace components {
void prefix_sum(std::shared_ptr<const HipExecutor>, int64*, size_type);
void fill_array(std::shared_ptr<const HipExecutor>, double*, size_type, double);
void convert_ptrs_to_idxs(std::shared_ptr<const HipExecutor>, const int32*, size_type, int32*);
void convert_idxs_to_ptrs(std::shared_ptr<const HipExecutor>, const int32*, size_type, size_type, int32*);
void convert_ptrs_to_sizes(std::shared_ptr<const HipExecutor>, const int32*, size_type, size_type*);
void sort_row_major(std::shared_ptr<const HipExecutor>, device_matrix_data<double, int32>&);
void remove_zeros(std::shared_ptr<const HipExecutor>, array<double>&, array<int32>&, array<int32>&);
void sum_duplicates(std::shared_ptr<const HipExecutor>, size_type, array<double>&, array<int32>&, array<int32>&);
}
namespace cg {
void initialize(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Status*);
void step_1(std::shared_ptr<const HipExecutor>, Vec*, const Vec*, const Vec*, const Vec*, const Status*);
void step_2(std::shared_ptr<const HipExecutor>, Vec*, Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Status*);
}
namespace fcg {
void initialize(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Status*);
void step_2(std::shared_ptr<const HipExecutor>, Vec*, Vec*, Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Status*);
}
namespace bicgstab {
void initialize(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Status*);
void step_1(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Status*);
void step_2(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, const Vec*, const Vec*, Vec*, const Vec*, const Status*);
void step_3(std::shared_ptr<const HipExecutor>, Vec*, Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Vec*, Vec*, const Status*);
void finalize(std::shared_ptr<const HipExecutor>, Vec*, const Vec*, const Vec*, Status*);
}
namespace cgs {
void initialize(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Status*);
void step_1(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, const Vec*, Vec*, const Vec*, const Vec*, const Status*);
void step_2(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, Vec*, Vec*, const Vec*, const Vec*, const Status*);
void step_3(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, Vec*, const Vec*, const Status*);
}
namespace bicg {
void initialize(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Vec*, Status*);
void step_1(std::shared_ptr<const HipExecutor>, Vec*, const Vec*, Vec*, const Vec*, const Vec*, const Vec*, const Status*);
void step_2(std::shared_ptr<const HipExecutor>, Vec*, Vec*, Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Vec*, const Status*);
}
namespace ir { void initialize(std::shared_ptr<const HipExecutor>, Status*); }
namespace set_all_statuses { void set_all_statuses(std::shared_ptr<const HipExecutor>, uint8, bool, Status*); }
namespace residual_norm { void residual_norm(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, double, uint8, bool, Status*, array<bool>*, bool*, bool*); }
namespace implicit_residual_norm { void implicit_residual_norm(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, double, uint8, bool, Status*, array<bool>*, bool*, bool*); }
namespace common_gmres {
void initialize(std::shared_ptr<const HipExecutor>, const Vec*, Vec*, Vec*, Vec*, stopping_status*);
void hessenberg_qr(std::shared_ptr<const HipExecutor>, Vec*, Vec*, Vec*, Vec*, Vec*, size_type, size_type*, const stopping_status*);
void solve_krylov(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, const size_type*, const stopping_status*);
}
namespace gmres {
void restart(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, Vec*, size_type*);
void multi_axpy(std::shared_ptr<const HipExecutor>, const Vec*, const Vec*, Vec*, const size_type*, stopping_status*);
}
namespace jacobi {
using scheme = preconditioner::block_interleaved_storage_scheme<int32>;
void find_blocks(std::shared_ptr<const HipExecutor>, const Mtx*, uint32, size_type&, array<int32>&);
void generate(std::shared_ptr<const HipExecutor>, const Mtx*, size_type, uint32, double, const scheme&, array<double>&, array<precision_reduction>&,
              const array<int32>&, array<double>&);
void simple_apply(std::shared_ptr<const HipExecutor>, size_type, uint32, const scheme&, const array<precision_reduction>&, const array<int32>&,
                  const array<double>&, const Vec*, Vec*);
void apply(std::shared_ptr<const HipExecutor>, size_type, uint32, const scheme&, const array<precision_reduction>&, const array<int32>&,
           const array<double>&, const Vec*, const Vec*, const Vec*, Vec*);
void invert_diagonal(std::shared_ptr<const HipExecutor>, const array<double>&, array<double>&);
void simple_scalar_apply(std::shared_ptr<const HipExecutor>, const array<double>&, const Vec*, Vec*);
void scalar_apply(std::shared_ptr<const HipExecutor>, const array<double>&, const Vec*, const Vec*, const Vec*, Vec*);
void transpose_jacobi(std::shared_ptr<const HipExecutor>, size_type, uint32, const array<precision_reduction>&, const array<int32>&, const array<double>&,
                      const scheme&, array<double>&);
void conj_transpose_jacobi(std::shared_ptr<const HipExecutor>, size_type, uint32, const array<precision_reduction>&, const array<int32>&,
                           const array<double>&, const scheme&, array<double>&);
}
namespace factorization {
void add_diagonal_elements(std::shared_ptr<const HipExecutor>, Mtx*, bool);
void initialize_row_ptrs_l_u(std::shared_ptr<const HipExecutor>, const Mtx*, int32*, int32*);
void initialize_l_u(std::shared_ptr<const HipExecutor>, const Mtx*, Mtx*, Mtx*);
void initialize_row_ptrs_l(std::shared_ptr<const HipExecutor>, const Mtx*, int32*);
void initialize_l(std::shared_ptr<const HipExecutor>, const Mtx*, Mtx*, bool);
}
namespace par_ilu_factorization { void compute_l_u_factors(std::shared_ptr<const HipExecutor>, size_type, const matrix::Coo<double, int32>*, Mtx*, Mtx*); }
namespace upper_trs {
void should_perform_transpose(std::shared_ptr<const HipExecutor>, bool&);
void generate(std::shared_ptr<const HipExecutor>, const Mtx*, std::shared_ptr<solver::SolveStruct>&, bool, const solver::trisolve_algorithm, const size_type);
void solve(std::shared_ptr<const HipExecutor>, const Mtx*, const solver::SolveStruct*, bool, const solver::trisolve_algorithm, Vec*, Vec*, const Vec*, Vec*);
}
namespace lower_trs {
void should_perform_transpose(std::shared_ptr<const HipExecutor>, bool&);
void generate(std::shared_ptr<const HipExecutor>, const Mtx*, std::shared_ptr<solver::SolveStruct>&, bool, const solver::trisolve_algorithm, const size_type);
void solve(std::shared_ptr<const HipExecutor>, const Mtx*, const solver::SolveStruct*, bool, const solver::trisolve_algorithm, Vec*, Vec*, const Vec*, Vec*);
}
namespace partition {
void build_starting_indices(std::shared_ptr<const HipExecutor>, const int64*, const int*, size_type, experimental::distributed::comm_index_type,
                            experimental::distributed::comm_index_type&, int32*, int32*);
void has_ordered_parts(std::shared_ptr<const HipExecutor>, const experimental::distributed::Partition<int32, int64>*, bool*);
}
namespace distributed_vector {
void build_local(std::shared_ptr<const HipExecutor>, const device_matrix_data<double, int64>&, const experimental::distributed::Partition<int32, int64>*,
                 experimental::distributed::comm_index_type, Vec*);
}
namespace csr {
void spmv(std::shared_ptr<const HipExecutor>, const matrix::Csr<float, int32>*, const matrix::Dense<float>*, matrix::Dense<float>*);
void advanced_spmv(std::shared_ptr<const HipExecutor>, const matrix::Dense<float>*, const matrix::Csr<float, int32>*, const matrix::Dense<float>*,
                   const matrix::Dense<float>*, matrix::Dense<float>*);
}
namespace dense {
void fill(std::shared_ptr<const HipExecutor>, matrix::Dense<float>*, float);
void add_scaled(std::shared_ptr<const HipExecutor>, const matrix::Dense<float>*, const matrix::Dense<float>*, matrix::Dense<float>*);
void compute_dot(std::shared_ptr<const HipExecutor>, const matrix::Dense<float>*, const matrix::Dense<float>*, matrix::Dense<float>*, array<char>&);
void compute_norm2(std::shared_ptr<const HipExecutor>, const matrix::Dense<float>*, matrix::Dense<float>*, array<char>&);
}
namespace cg {
void initialize(std::shared_ptr<const HipExecutor>, const matrix::Dense<float>*, matrix::Dense<float>*, matrix::Dense<float>*, matrix::Dense<float>*,
                matrix::Dense<float>*, matrix::Dense<float>*, matrix::Dense<float>*, array<stopping_status>*);
void step_1(std::shared_ptr<const HipExecutor>, matrix::Dense<float>*, const matrix::Dense<float>*, const matrix::Dense<float>*,
            const matrix::Dense<float>*, const array<stopping_status>*);
void step_2(std::shared_ptr<const HipExecutor>, matrix::Dense<float>*, matrix::Dense<float>*, const matrix::Dense<float>*, const matrix::Dense<float>*,
            const matrix::Dense<float>*, const matrix::Dense<float>*, const array<stopping_status>*);
}
namespace residual_norm {
void residual_norm(std::shared_ptr<const HipExecutor>, const matrix::Dense<float>*, const matrix::Dense<float>*, float, uint8, bool,
                   array<stopping_status>*, array<bool>*, bool*, bool*);
}
namespace distributed_matrix {
void build_local_nonlocal(std::shared_ptr<const HipExecutor>, const device_matrix_data<double, int64>&, const experimental::distributed::Partition<int32, int64>*,
                          const experimental::distributed::Partition<int32, int64>*, experimental::distributed::comm_index_type, array<int32>&, array<int32>&,
                          array<double>&, array<int32>&, array<int32>&, array<double>&, array<int32>&, array<experimental::distributed::comm_index_type>&,
                          array<int64>&);
}
}}}

using namespace gko;
namespace k = gko::kernels::hip;
using Vec = matrix::Dense<double>;
using Mtx = matrix::Csr<double, int32>;

static int wrong = 0;
static void ran(const char* name, bool ok)
{
    std::printf("ran %s %s\n", name, ok ? "ok" : "WRONG");
    if (!ok) ++wrong;
}

static std::vector<double> host_of(const Vec* v)
{
    auto h = v->clone(v->get_executor()->get_master());
    std::vector<double> out(v->get_size()[0] * v->get_size()[1]);
    for (size_type i = 0; i < v->get_size()[0]; ++i)
        for (size_type j = 0; j < v->get_size()[1]; ++j) out[i * v->get_size()[1] + j] = h->at(i, j);
    return out;
}
static bool all_equal(const Vec* v, double value, double tol = 0.0)
{
    for (double x : host_of(v)) if (!(std::abs(x - value) <= tol)) return false;
    return true;
}
static std::unique_ptr<Vec> filled(std::shared_ptr<const Executor> exec, size_type n, double value, size_type cols = 1)
{
    auto v = Vec::create(exec, dim<2>(n, cols));
    v->fill(value);
    return v;
}
static std::unique_ptr<Vec> scalar(std::shared_ptr<const Executor> exec, double value) { return initialize<Vec>({value}, exec); }
static double value_of(std::shared_ptr<const Executor> exec, const Vec* s) { return exec->copy_val_to_host(s->get_const_values()); }

static matrix_data<double, int32> banded(size_type n)
{
    matrix_data<double, int32> d;
    d.size = {n, n};
    for (size_type i = 0; i < n; ++i) {
        if (i >= 37) d.nonzeros.emplace_back(i, i - 37, -0.5);
        if (i > 0) d.nonzeros.emplace_back(i, i - 1, -1.0);
        d.nonzeros.emplace_back(i, i, 4.0 + 0.001 * (i % 13));
        if (i + 1 < n) d.nonzeros.emplace_back(i, i + 1, -1.0);
        if (i + 37 < n) d.nonzeros.emplace_back(i, i + 37, -0.5);
    }
    return d;
}

int main()
{
    auto hip = HipExecutor::create(0, ReferenceExecutor::create());
    std::shared_ptr<const Executor> exec = hip;
    const size_type n = 3000;
    array<char> tmp(hip);
    auto one = scalar(hip, 1.0), zero = scalar(hip, 0.0), two = scalar(hip, 2.0), half = scalar(hip, 0.5);
    auto nrm = Vec::create(hip, dim<2>(1, 1));
    auto diff_norm = [&](const Vec* a, const Vec* b) {
        auto d = a->clone();
        k::dense::sub_scaled(hip, one.get(), b, d.get());
        k::dense::compute_norm2(hip, d.get(), nrm.get(), tmp);
        return value_of(hip, nrm.get());
    };

    // ---- dense ----------------------------------------------------------------------------------
    {
        auto x = Vec::create(hip, dim<2>(n, 1)), y = Vec::create(hip, dim<2>(n, 1));
        k::dense::fill(hip, x.get(), 3.0);
        ran("dense::fill", all_equal(x.get(), 3.0));
        k::dense::copy(hip, x.get(), y.get());
        ran("dense::copy", all_equal(y.get(), 3.0));
        k::dense::scale(hip, two.get(), y.get());
        ran("dense::scale", all_equal(y.get(), 6.0));
        k::dense::inv_scale(hip, two.get(), y.get());
        ran("dense::inv_scale", all_equal(y.get(), 3.0));
        k::dense::add_scaled(hip, half.get(), x.get(), y.get());
        ran("dense::add_scaled", all_equal(y.get(), 4.5));
        k::dense::sub_scaled(hip, two.get(), x.get(), y.get());
        ran("dense::sub_scaled", all_equal(y.get(), -1.5));
        auto r = Vec::create(hip, dim<2>(1, 1));
        k::dense::compute_dot(hip, x.get(), y.get(), r.get(), tmp);
        ran("dense::compute_dot", value_of(hip, r.get()) == -4.5 * n);
        k::dense::compute_conj_dot(hip, x.get(), y.get(), r.get(), tmp);
        ran("dense::compute_conj_dot", value_of(hip, r.get()) == -4.5 * n);
        k::dense::compute_dot_dispatch(hip, x.get(), x.get(), r.get(), tmp);
        ran("dense::compute_dot_dispatch", value_of(hip, r.get()) == 9.0 * n);
        k::dense::compute_norm2_dispatch(hip, x.get(), r.get(), tmp);
        ran("dense::compute_norm2_dispatch", std::abs(value_of(hip, r.get()) - 3.0 * std::sqrt(double(n))) < 1e-10);
        k::dense::compute_squared_norm2(hip, y.get(), r.get(), tmp);
        ran("dense::compute_squared_norm2", value_of(hip, r.get()) == 2.25 * n);
        k::dense::compute_sqrt(hip, r.get());
        ran("dense::compute_sqrt", std::abs(value_of(hip, r.get()) - 1.5 * std::sqrt(double(n))) < 1e-10);
        auto host_src = Vec::create(hip->get_master(), dim<2>(n, 2));
        for (size_type i = 0; i < n; ++i) { host_src->at(i, 0) = double(i); host_src->at(i, 1) = -double(i); }
        auto src = host_src->clone(hip);
        std::vector<int32> idx = {5, 0, 2999, 17, 17};
        array<int32> gi(hip, idx.begin(), idx.end());
        auto got = Vec::create(hip, dim<2>(idx.size(), 2));
        k::dense::row_gather(hip, &gi, src.get(), got.get());
        auto g = host_of(got.get());
        bool ok = true;
        for (size_t i = 0; i < idx.size(); ++i) ok = ok && g[2 * i] == idx[i] && g[2 * i + 1] == -idx[i];
        ran("dense::row_gather", ok);
    }

    // ---- matrices: every remaining format kernel against the mirror's CSR apply ---------------------
    auto A = share(Mtx::create(hip));
    const auto data = banded(n);
    A->read(data);
    auto host_x = Vec::create(hip->get_master(), dim<2>(n, 1));
    for (size_type i = 0; i < n; ++i) host_x->at(i) = std::sin(0.01 * i);
    auto x = host_x->clone(hip);
    auto y_ref = Vec::create(hip, dim<2>(n, 1));
    A->apply(x.get(), y_ref.get());
    auto y3_ref = y_ref->clone();            // 2 A x + 1 y
    A->apply(two.get(), x.get(), one.get(), y3_ref.get());
    {
        auto y = y_ref->clone();
        A->make_srow();
        k::csr::advanced_spmv(hip, two.get(), A.get(), x.get(), one.get(), y.get());
        ran("csr::advanced_spmv", diff_norm(y.get(), y3_ref.get()) == 0.0);
        matrix_data<double, int64> d64;
        d64.size = data.size;
        for (const auto& e : data.nonzeros) d64.nonzeros.push_back({e.row, e.column, e.value});
        auto A64 = matrix::Csr<double, int64>::create(hip);
        A64->read(d64);
        A64->make_srow();
        auto y64 = Vec::create(hip, dim<2>(n, 1));
        k::csr::spmv(hip, A64.get(), x.get(), y64.get());
        ran("csr::spmv<int64>", diff_norm(y64.get(), y_ref.get()) == 0.0);
        y64 = y_ref->clone();
        k::csr::advanced_spmv(hip, two.get(), A64.get(), x.get(), one.get(), y64.get());
        ran("csr::advanced_spmv<int64>", diff_norm(y64.get(), y3_ref.get()) == 0.0);
        auto E = matrix::Ell<double, int32>::create(hip);
        A->convert_to(E.get());
        y = y_ref->clone();
        k::ell::advanced_spmv(hip, two.get(), E.get(), x.get(), one.get(), y.get());
        ran("ell::advanced_spmv", diff_norm(y.get(), y3_ref.get()) < 1e-12);
        auto S = matrix::Sellp<double, int32>::create(hip);
        A->convert_to(S.get());
        k::sellp::spmv(hip, S.get(), x.get(), y.get());
        ran("sellp::spmv", diff_norm(y.get(), y_ref.get()) == 0.0);
        auto C = matrix::Coo<double, int32>::create(hip);
        A->convert_to(C.get());
        k::coo::spmv(hip, C.get(), x.get(), y.get());
        ran("coo::spmv", diff_norm(y.get(), y_ref.get()) < 1e-12);
        y = y_ref->clone();
        k::coo::advanced_spmv(hip, two.get(), C.get(), x.get(), one.get(), y.get());
        ran("coo::advanced_spmv", diff_norm(y.get(), y3_ref.get()) < 1e-12);
        y = y_ref->clone();
        k::coo::advanced_spmv2(hip, two.get(), C.get(), x.get(), y.get());
        ran("coo::advanced_spmv2", diff_norm(y.get(), y3_ref.get()) < 1e-12);
        // sellp::compute_slice_sets / hybrid::compute_coo_row_ptrs against the row lengths
        auto rp = array<int32>(hip, n + 1);
        hip->copy(n + 1, A->get_const_row_ptrs(), rp.get_data());
        const size_type nslices = (n + 63) / 64;
        array<size_type> sets(hip, nslices + 1), lens(hip, nslices);
        k::sellp::compute_slice_sets(hip, rp, 64, 1, sets.get_data(), lens.get_data());
        auto hrp = rp.to_host();
        auto hsets = sets.to_host();
        auto hlens = lens.to_host();
        bool ok = hsets[0] == 0;
        for (size_type s = 0; s < nslices; ++s) {
            size_type longest = 0;
            for (size_type r = s * 64; r < std::min<size_type>(n, (s + 1) * 64); ++r) longest = std::max<size_type>(longest, hrp[r + 1] - hrp[r]);
            ok = ok && hlens[s] == longest && hsets[s + 1] == hsets[s] + longest;
        }
        ran("sellp::compute_slice_sets", ok);
        std::vector<size_type> row_nnz(n);
        for (size_type r = 0; r < n; ++r) row_nnz[r] = hrp[r + 1] - hrp[r];
        array<size_type> dnnz(hip, row_nnz.begin(), row_nnz.end());
        array<int64> coo_ptrs(hip, n + 1);
        k::hybrid::compute_coo_row_ptrs(hip, dnnz, 3, coo_ptrs.get_data());
        auto hc = coo_ptrs.to_host();
        ok = hc[0] == 0;
        for (size_type r = 0; r < n; ++r) ok = ok && hc[r + 1] == hc[r] + (row_nnz[r] > 3 ? int64(row_nnz[r] - 3) : 0);
        ran("hybrid::compute_coo_row_ptrs", ok);
    }

    // ---- components ---------------------------------------------------------------------------------
    {
        std::vector<int64> counts(5000);
        for (size_t i = 0; i < counts.size(); ++i) counts[i] = int64(i % 11) << 28;   // sums beyond 2^31
        array<int64> dc(hip, counts.begin(), counts.end());
        k::components::prefix_sum(hip, dc.get_data(), counts.size());
        auto s = dc.to_host();
        bool ok = true;
        int64 run = 0;
        for (size_t i = 0; i < counts.size(); ++i) { ok = ok && s[i] == run; run += counts[i]; }
        ran("components::prefix_sum<int64>", ok && run > (int64(1) << 32));
        array<double> f(hip, 777);
        k::components::fill_array(hip, f.get_data(), 777, -2.5);
        ok = true;
        for (double v : f.to_host()) ok = ok && v == -2.5;
        ran("components::fill_array", ok);
        auto hrp = std::vector<int32>{0, 2, 2, 5, 6};
        array<int32> rp(hip, hrp.begin(), hrp.end()), idxs(hip, 6), back(hip, 5);
        k::components::convert_ptrs_to_idxs(hip, rp.get_const_data(), 4, idxs.get_data());
        ran("components::convert_ptrs_to_idxs", idxs.to_host() == std::vector<int32>({0, 0, 2, 2, 2, 3}));
        k::components::convert_idxs_to_ptrs(hip, idxs.get_const_data(), 6, 4, back.get_data());
        ran("components::convert_idxs_to_ptrs", back.to_host() == hrp);
        array<size_type> sizes(hip, 4);
        k::components::convert_ptrs_to_sizes(hip, rp.get_const_data(), 4, sizes.get_data());
        ran("components::convert_ptrs_to_sizes", sizes.to_host() == std::vector<size_type>({2, 0, 3, 1}));
        // device_matrix_data: shuffled entries with duplicates and explicit zeros -> the CSR matrix again
        matrix_data<double, int32> messy;
        messy.size = data.size;
        for (std::size_t i = 0; i < data.nonzeros.size(); ++i) {
            const auto& e = data.nonzeros[(i * 7919) % data.nonzeros.size()];
            messy.nonzeros.push_back({e.row, e.column, 0.25 * e.value});
            messy.nonzeros.push_back({e.row, e.column, 0.75 * e.value});
            if (i % 3 == 0) messy.nonzeros.push_back({e.row, int32((e.column + 5) % n), 0.0});
        }
        auto dd = device_matrix_data<double, int32>::create_from_host(hip, messy);
        auto parts = dd.empty_out();
        k::components::remove_zeros(hip, parts.values, parts.row_idxs, parts.col_idxs);
        ran("components::remove_zeros", parts.values.get_num_elems() == 2 * data.nonzeros.size());
        device_matrix_data<double, int32> d2(data.size, std::move(parts.row_idxs), std::move(parts.col_idxs), std::move(parts.values));
        k::components::sort_row_major(hip, d2);
        auto sorted = d2.copy_to_host();
        ok = true;
        for (size_t i = 1; i < sorted.nonzeros.size(); ++i) {
            const auto &a = sorted.nonzeros[i - 1], &b = sorted.nonzeros[i];
            ok = ok && (a.row < b.row || (a.row == b.row && a.column <= b.column));
        }
        ran("components::sort_row_major", ok);
        auto p2 = d2.empty_out();
        k::components::sum_duplicates(hip, n, p2.values, p2.row_idxs, p2.col_idxs);
        ok = p2.values.get_num_elems() == data.nonzeros.size();
        if (ok) {
            auto v = p2.values.to_host(); auto r = p2.row_idxs.to_host(); auto c = p2.col_idxs.to_host();
            auto sorted_ref = data;
            sorted_ref.ensure_row_major_order();
            for (size_t i = 0; i < v.size(); ++i) {
                ok = ok && r[i] == sorted_ref.nonzeros[i].row && c[i] == sorted_ref.nonzeros[i].column && std::abs(v[i] - sorted_ref.nonzeros[i].value) < 1e-15;
            }
        }
        ran("components::sum_duplicates", ok);
    }

    // ---- CG, kernel by kernel as core/solver/cg.cpp:107-193 drives them, against the mirror's Cg ------------
    auto b = filled(hip, n, 1.0);
    int cg_iters = 0;
    {
        auto xs = filled(hip, n, 0.0);
        auto r = Vec::create(hip, dim<2>(n, 1)), z = Vec::create(hip, dim<2>(n, 1)), p = Vec::create(hip, dim<2>(n, 1)), q = Vec::create(hip, dim<2>(n, 1));
        auto prev_rho = scalar(hip, 0.0), rho = scalar(hip, 0.0), beta = scalar(hip, 0.0), tau = scalar(hip, 0.0), b_norm = scalar(hip, 0.0);
        auto neg_one = scalar(hip, -1.0);
        std::vector<stopping_status> st0(1);
        array<stopping_status> status(hip, st0.begin(), st0.end());
        array<bool> device_storage(hip, 2);
        k::cg::initialize(hip, b.get(), r.get(), z.get(), p.get(), q.get(), prev_rho.get(), rho.get(), &status);
        bool init_ok = all_equal(p.get(), 0.0) && diff_norm(r.get(), b.get()) == 0.0 && value_of(hip, prev_rho.get()) == 1.0 && value_of(hip, rho.get()) == 0.0;
        ran("cg::initialize", init_ok);
        k::csr::advanced_spmv(hip, neg_one.get(), A.get(), xs.get(), one.get(), r.get());   // r = b - A x
        k::dense::compute_norm2(hip, b.get(), b_norm.get(), tmp);
        bool all_converged = false, one_changed = false, steps_ok = true, crit_ok = true;
        for (int it = 0; it < 500; ++it) {
            k::dense::copy(hip, r.get(), z.get());                          // identity preconditioner
            k::dense::compute_dot(hip, r.get(), z.get(), rho.get(), tmp);
            k::dense::compute_norm2(hip, r.get(), tau.get(), tmp);
            k::residual_norm::residual_norm(hip, tau.get(), b_norm.get(), 1e-10, 1, true, &status, &device_storage, &all_converged, &one_changed);
            crit_ok = crit_ok && all_converged == (value_of(hip, tau.get()) < 1e-10 * value_of(hip, b_norm.get()));
            if (all_converged) { cg_iters = it; break; }
            k::cg::step_1(hip, p.get(), z.get(), rho.get(), prev_rho.get(), &status);
            k::csr::spmv(hip, A.get(), p.get(), q.get());
            k::dense::compute_dot(hip, p.get(), q.get(), beta.get(), tmp);
            k::cg::step_2(hip, xs.get(), r.get(), p.get(), q.get(), beta.get(), rho.get(), &status);
            std::swap(prev_rho, rho);
        }
        auto cg = solver::Cg<double>::build()
                      .with_criteria(stop::Iteration::build().with_max_iters(500u).on(exec), stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                      .on(exec)->generate(A);
        auto xm = filled(hip, n, 0.0);
        cg->apply(b.get(), xm.get());
        k::dense::compute_norm2(hip, xm.get(), nrm.get(), tmp);
        const double xnorm = value_of(hip, nrm.get());
        steps_ok = all_converged && std::abs(cg_iters - int(cg->get_last_iteration_count())) <= 1 && diff_norm(xs.get(), xm.get()) <= 1e-9 * xnorm;
        ran("cg::step_1", steps_ok);
        ran("cg::step_2", steps_ok);
        ran("residual_norm::residual_norm", crit_ok && all_converged);
        // the implicit criterion: tau holds the SQUARED residual norm (reference/stop/residual_norm_kernels.cpp:100-126)
        std::vector<stopping_status> st1(1);
        array<stopping_status> status2(hip, st1.begin(), st1.end());
        auto tau2 = scalar(hip, 1e-22), orig = scalar(hip, 1.0);
        k::implicit_residual_norm::implicit_residual_norm(hip, tau2.get(), orig.get(), 1e-10, 1, true, &status2, &device_storage, &all_converged, &one_changed);
        bool ok = all_converged && one_changed;
        tau2 = scalar(hip, 1e-18);
        std::vector<stopping_status> st2(1);
        array<stopping_status> status3(hip, st2.begin(), st2.end());
        k::implicit_residual_norm::implicit_residual_norm(hip, tau2.get(), orig.get(), 1e-10, 1, true, &status3, &device_storage, &all_converged, &one_changed);
        ran("implicit_residual_norm::implicit_residual_norm", ok && !all_converged && !one_changed);
    }

    // ---- the other Krylov step kernels on constant vectors: closed forms of common/unified/solver/*_kernels.cpp -------
    {
        std::vector<stopping_status> st0(1);
        array<stopping_status> status(hip, st0.begin(), st0.end());
        auto v = [&](double a) { return filled(hip, n, a); };
        auto s = [&](double a) { return scalar(hip, a); };
        // fcg
        {
            auto bb = v(2.0), r = v(9), z = v(9), p = v(9), q = v(9), t = v(9);
            auto prev_rho = s(9), rho = s(9), rho_t = s(9);
            k::fcg::initialize(hip, bb.get(), r.get(), z.get(), p.get(), q.get(), t.get(), prev_rho.get(), rho.get(), rho_t.get(), &status);
            ran("fcg::initialize", all_equal(r.get(), 2.0) && all_equal(t.get(), 2.0) && all_equal(z.get(), 0.0) && all_equal(p.get(), 0.0) && all_equal(q.get(), 0.0) &&
                                       value_of(hip, rho.get()) == 0.0 && value_of(hip, prev_rho.get()) == 1.0 && value_of(hip, rho_t.get()) == 1.0);
            auto xx = v(1.0), rr = v(5.0), tt = v(0.0), pp = v(2.0), qq = v(4.0);
            auto beta = s(8.0), rh = s(4.0);   // tmp = 0.5: x = 2, r = 3, t = -2
            k::fcg::step_2(hip, xx.get(), rr.get(), tt.get(), pp.get(), qq.get(), beta.get(), rh.get(), &status);
            ran("fcg::step_2", all_equal(xx.get(), 2.0) && all_equal(rr.get(), 3.0) && all_equal(tt.get(), -2.0));
        }
        // bicgstab
        {
            auto bb = v(2.0), r = v(9), rr = v(9), y = v(9), sv = v(9), t = v(9), z = v(9), vv = v(9), p = v(9);
            auto prev_rho = s(9), rho = s(9), alpha = s(9), beta = s(9), gamma = s(9), omega = s(9);
            k::bicgstab::initialize(hip, bb.get(), r.get(), rr.get(), y.get(), sv.get(), t.get(), z.get(), vv.get(), p.get(), prev_rho.get(), rho.get(),
                                    alpha.get(), beta.get(), gamma.get(), omega.get(), &status);
            bool ok = all_equal(r.get(), 2.0) && all_equal(rr.get(), 0.0) && all_equal(p.get(), 0.0) && all_equal(vv.get(), 0.0) && all_equal(y.get(), 0.0);
            for (auto* q : {prev_rho.get(), rho.get(), alpha.get(), beta.get(), gamma.get(), omega.get()}) ok = ok && value_of(hip, q) == 1.0;
            ran("bicgstab::initialize", ok);
            // step_1: tmp = (rho / prev_rho) (alpha / omega) = (6/3)(2/4) = 1; p = r + tmp (p - omega v) = 2 + (5 - 4 * 0.5) = 5
            auto r1 = v(2.0), p1 = v(5.0), v1 = v(0.5);
            k::bicgstab::step_1(hip, r1.get(), p1.get(), v1.get(), s(6.0).get(), s(3.0).get(), s(2.0).get(), s(4.0).get(), &status);
            ran("bicgstab::step_1", all_equal(p1.get(), 5.0));
            // step_2: alpha = rho / beta = 1.5; s = r - alpha v = 2 - 0.75 = 1.25
            auto s2 = v(9), al = s(9);
            k::bicgstab::step_2(hip, r1.get(), s2.get(), v1.get(), s(6.0).get(), al.get(), s(4.0).get(), &status);
            ran("bicgstab::step_2", all_equal(s2.get(), 1.25) && value_of(hip, al.get()) == 1.5);
            // step_3: omega = gamma / beta = 0.5; x += alpha y + omega z = 1 + 2 * 3 + 0.5 * 4 = 9; r = s - omega t = 1.25 - 0.5 * 2 = 0.25
            auto x3 = v(1.0), r3 = v(9), t3 = v(2.0), y3 = v(3.0), z3 = v(4.0), om = s(9);
            k::bicgstab::step_3(hip, x3.get(), r3.get(), s2.get(), t3.get(), y3.get(), z3.get(), s(2.0).get(), s(8.0).get(), s(4.0).get(), om.get(), &status);
            ran("bicgstab::step_3", all_equal(x3.get(), 9.0) && all_equal(r3.get(), 0.25) && value_of(hip, om.get()) == 0.5);
            // finalize acts on a column that has stopped and is not finalized: x += alpha y
            std::vector<stopping_status> sst(1);
            array<stopping_status> stopped(hip, sst.begin(), sst.end());
            k::set_all_statuses::set_all_statuses(hip, 3, false, &stopped);
            const uint8 after = stopped.to_host()[0].data_;
            ran("set_all_statuses::set_all_statuses", (after & 0x3f) == 3 && (after & 0x40) == 0);
            k::bicgstab::finalize(hip, x3.get(), y3.get(), s(2.0).get(), &stopped);
            ran("bicgstab::finalize", all_equal(x3.get(), 15.0) && (stopped.to_host()[0].data_ & 0x40) != 0);
            k::ir::initialize(hip, &stopped);
            ran("ir::initialize", stopped.to_host()[0].data_ == 0);
        }
        // cgs
        {
            auto bb = v(2.0), r = v(9), rt = v(9), p = v(9), q = v(9), u = v(9), uh = v(9), vh = v(9), t = v(9);
            auto alpha = s(9), beta = s(9), gamma = s(9), prev_rho = s(9), rho = s(9);
            k::cgs::initialize(hip, bb.get(), r.get(), rt.get(), p.get(), q.get(), u.get(), uh.get(), vh.get(), t.get(), alpha.get(), beta.get(), gamma.get(),
                               prev_rho.get(), rho.get(), &status);
            ran("cgs::initialize", all_equal(r.get(), 2.0) && all_equal(rt.get(), 2.0) && all_equal(u.get(), 0.0) && all_equal(t.get(), 0.0) &&
                                       value_of(hip, rho.get()) == 0.0 && value_of(hip, prev_rho.get()) == 1.0 && value_of(hip, alpha.get()) == 1.0);
            // step_1: beta = rho / prev = 2; u = r + beta q = 2 + 2 = 4; p = u + beta (q + beta p) = 4 + 2 (1 + 2 * 0.5) = 8
            auto r1 = v(2.0), u1 = v(9), p1 = v(0.5), q1 = v(1.0), b1 = s(9);
            k::cgs::step_1(hip, r1.get(), u1.get(), p1.get(), q1.get(), b1.get(), s(6.0).get(), s(3.0).get(), &status);
            ran("cgs::step_1", all_equal(u1.get(), 4.0) && all_equal(p1.get(), 8.0) && value_of(hip, b1.get()) == 2.0);
            // step_2: alpha = rho / gamma = 0.5; q = u - alpha v_hat = 4 - 1 = 3; t = u + q = 7
            auto vh1 = v(2.0), q2 = v(9), t2 = v(9), a2 = s(9);
            k::cgs::step_2(hip, u1.get(), vh1.get(), q2.get(), t2.get(), a2.get(), s(6.0).get(), s(12.0).get(), &status);
            ran("cgs::step_2", all_equal(q2.get(), 3.0) && all_equal(t2.get(), 7.0) && value_of(hip, a2.get()) == 0.5);
            // step_3: x += alpha u_hat = 1 + 0.5 * 4 = 3; r -= alpha t = 2 - 3.5 = -1.5
            auto x3 = v(1.0), uh3 = v(4.0);
            k::cgs::step_3(hip, t2.get(), uh3.get(), r1.get(), x3.get(), a2.get(), &status);
            ran("cgs::step_3", all_equal(x3.get(), 3.0) && all_equal(r1.get(), -1.5));
        }
        // bicg
        {
            auto bb = v(2.0), r = v(9), z = v(9), p = v(9), q = v(9), r2 = v(9), z2 = v(9), p2 = v(9), q2 = v(9);
            auto prev_rho = s(9), rho = s(9);
            k::bicg::initialize(hip, bb.get(), r.get(), z.get(), p.get(), q.get(), prev_rho.get(), rho.get(), r2.get(), z2.get(), p2.get(), q2.get(), &status);
            ran("bicg::initialize", all_equal(r.get(), 2.0) && all_equal(r2.get(), 2.0) && all_equal(p2.get(), 0.0) && all_equal(q.get(), 0.0) &&
                                        value_of(hip, rho.get()) == 0.0 && value_of(hip, prev_rho.get()) == 1.0);
            // step_1: tmp = rho / prev = 2: p = z + 2 p = 1 + 6 = 7, p2 = z2 + 2 p2 = -1 + 1 = 0
            auto p1 = v(3.0), z1 = v(1.0), p21 = v(0.5), z21 = v(-1.0);
            k::bicg::step_1(hip, p1.get(), z1.get(), p21.get(), z21.get(), s(6.0).get(), s(3.0).get(), &status);
            ran("bicg::step_1", all_equal(p1.get(), 7.0) && all_equal(p21.get(), 0.0));
            // step_2: tmp = rho / beta = 0.5: x = 1 + 3.5, r = 2 - 1 = 1, r2 = 2 - 2 = 0
            auto x2 = v(1.0), rr = v(2.0), rr2 = v(2.0), qq = v(2.0), qq2 = v(4.0);
            k::bicg::step_2(hip, x2.get(), rr.get(), rr2.get(), p1.get(), qq.get(), qq2.get(), s(12.0).get(), s(6.0).get(), &status);
            ran("bicg::step_2", all_equal(x2.get(), 4.5) && all_equal(rr.get(), 1.0) && all_equal(rr2.get(), 0.0));
        }
    }

    // ---- GMRES(m), kernel by kernel as core/solver/gmres.cpp:139-372 drives them: one restart cycle and a half ----------
    {
        const size_type m = 20;
        auto xs = filled(hip, n, 0.0);
        auto residual = Vec::create(hip, dim<2>(n, 1));
        auto bases = Vec::create(hip, dim<2>((m + 1) * n, 1));
        auto hess = Vec::create(hip, dim<2>(m + 1, m));
        auto gsin = Vec::create(hip, dim<2>(m, 1)), gcos = Vec::create(hip, dim<2>(m, 1));
        auto rnc = Vec::create(hip, dim<2>(m + 1, 1)), yv = Vec::create(hip, dim<2>(m, 1));
        auto before = Vec::create(hip, dim<2>(n, 1));
        auto res_norm = scalar(hip, 0.0), b_norm = scalar(hip, 0.0), neg_one = scalar(hip, -1.0);
        std::vector<stopping_status> st0(1);
        array<stopping_status> status(hip, st0.begin(), st0.end());
        array<size_type> final_iter(hip, 1);
        array<bool> device_storage(hip, 2);
        auto basis = [&](size_type i) { return Vec::create(hip, dim<2>(n, 1), array<double>::view(hip, n, bases->get_values() + i * n), 1); };
        auto restart_cycle = [&] {
            k::dense::copy(hip, b.get(), residual.get());
            k::csr::advanced_spmv(hip, neg_one.get(), A.get(), xs.get(), one.get(), residual.get());
            k::dense::compute_norm2(hip, residual.get(), res_norm.get(), tmp);
            k::gmres::restart(hip, residual.get(), res_norm.get(), rnc.get(), bases.get(), final_iter.get_data());
        };
        k::common_gmres::initialize(hip, b.get(), residual.get(), gsin.get(), gcos.get(), status.get_data());
        const bool init_ok = diff_norm(residual.get(), b.get()) == 0.0 && all_equal(gsin.get(), 0.0) && all_equal(gcos.get(), 0.0);
        ran("common_gmres::initialize", init_ok);
        k::dense::compute_norm2(hip, b.get(), b_norm.get(), tmp);
        restart_cycle();
        k::dense::compute_norm2(hip, basis(0).get(), nrm.get(), tmp);
        ran("gmres::restart", std::abs(value_of(hip, nrm.get()) - 1.0) < 1e-14 && final_iter.to_host()[0] == 0 &&
                                  std::abs(value_of(hip, rnc.get()) - value_of(hip, res_norm.get())) == 0.0);
        bool all_converged = false, one_changed = false;
        size_type restart_iter = 0;
        int total = 0;
        for (; total < 200 && !all_converged; ++total) {
            if (restart_iter == m) {
                k::common_gmres::solve_krylov(hip, rnc.get(), hess.get(), yv.get(), final_iter.get_const_data(), status.get_const_data());
                k::gmres::multi_axpy(hip, bases.get(), yv.get(), before.get(), final_iter.get_const_data(), status.get_data());
                k::dense::add_scaled(hip, one.get(), before.get(), xs.get());
                restart_cycle();
                restart_iter = 0;
            }
            auto next = basis(restart_iter + 1);
            k::csr::spmv(hip, A.get(), basis(restart_iter).get(), next.get());
            // modified Gram-Schmidt into column restart_iter of the Hessenberg matrix (gmres.cpp:300-319)
            for (size_type i = 0; i <= restart_iter; ++i) {
                auto h = Vec::create(hip, dim<2>(1, 1), array<double>::view(hip, 1, hess->get_values() + i * m + restart_iter), 1);
                k::dense::compute_dot(hip, next.get(), basis(i).get(), h.get(), tmp);
                k::dense::sub_scaled(hip, h.get(), basis(i).get(), next.get());
            }
            auto hlast = Vec::create(hip, dim<2>(1, 1), array<double>::view(hip, 1, hess->get_values() + (restart_iter + 1) * m + restart_iter), 1);
            k::dense::compute_norm2(hip, next.get(), hlast.get(), tmp);
            k::dense::inv_scale(hip, hlast.get(), next.get());
            auto hess_iter = Vec::create(hip, dim<2>(restart_iter + 2, 1), array<double>::view(hip, (restart_iter + 2) * m, hess->get_values() + restart_iter), m);
            k::common_gmres::hessenberg_qr(hip, gsin.get(), gcos.get(), res_norm.get(), rnc.get(), hess_iter.get(), restart_iter, final_iter.get_data(),
                                           status.get_const_data());
            ++restart_iter;
            // (setFinalized = false, as core/solver/gmres.cpp does: the update of x below must still happen for a column
            // that has just converged; multi_axpy finalizes it)
            k::residual_norm::residual_norm(hip, res_norm.get(), b_norm.get(), 1e-10, 1, false, &status, &device_storage, &all_converged, &one_changed);
        }
        k::common_gmres::solve_krylov(hip, rnc.get(), hess.get(), yv.get(), final_iter.get_const_data(), status.get_const_data());
        k::gmres::multi_axpy(hip, bases.get(), yv.get(), before.get(), final_iter.get_const_data(), status.get_data());
        k::dense::add_scaled(hip, one.get(), before.get(), xs.get());
        auto r = b->clone();
        A->apply(neg_one.get(), xs.get(), one.get(), r.get());
        k::dense::compute_norm2(hip, r.get(), nrm.get(), tmp);
        const bool solved = all_converged && value_of(hip, nrm.get()) <= 1e-9 * value_of(hip, b_norm.get()) && total > int(m);
        ran("common_gmres::hessenberg_qr", solved);
        ran("common_gmres::solve_krylov", solved);
        ran("gmres::multi_axpy", solved);
    }

    // ---- block-Jacobi through its kernels against the mirror's preconditioner::Jacobi --------------------------------
    {
        auto jac = preconditioner::Jacobi<double, int32>::build().with_max_block_size(8u).on(exec)->generate(A);
        array<int32> ptrs(hip, n + 1);
        size_type nb = 0;
        k::jacobi::find_blocks(hip, A.get(), 8, nb, ptrs);
        bool ok = nb == jac->get_num_blocks();
        auto hp = ptrs.to_host();
        array<int32> mirror_ptrs(hip, nb + 1);
        hip->copy(nb + 1, jac->get_const_block_pointers(), mirror_ptrs.get_data());
        auto mp = mirror_ptrs.to_host();
        for (size_type i = 0; ok && i <= nb; ++i) ok = hp[i] == mp[i];
        ran("jacobi::find_blocks", ok);
        const auto scheme = jac->get_storage_scheme();
        array<double> blocks(hip, jac->get_num_stored_elements()), conditioning(hip, 0);
        array<precision_reduction> precisions(hip, 0);
        blocks.fill(0.0);
        k::jacobi::generate(hip, A.get(), nb, 8, 0.1, scheme, conditioning, precisions, ptrs, blocks);
        array<double> mirror_blocks(hip, jac->get_num_stored_elements());
        hip->copy(jac->get_num_stored_elements(), jac->get_blocks(), mirror_blocks.get_data());
        ran("jacobi::generate", blocks.to_host() == mirror_blocks.to_host());
        auto z = Vec::create(hip, dim<2>(n, 1)), zm = Vec::create(hip, dim<2>(n, 1));
        k::jacobi::simple_apply(hip, nb, 8, scheme, precisions, ptrs, blocks, x.get(), z.get());
        jac->apply(x.get(), zm.get());
        ran("jacobi::simple_apply", diff_norm(z.get(), zm.get()) == 0.0);
        auto z2 = y_ref->clone(), zm2 = y_ref->clone();
        k::jacobi::apply(hip, nb, 8, scheme, precisions, ptrs, blocks, two.get(), x.get(), half.get(), z2.get());
        jac->apply(two.get(), x.get(), half.get(), zm2.get());
        ran("jacobi::apply", diff_norm(z2.get(), zm2.get()) == 0.0);
        array<double> tblocks(hip, blocks.get_num_elems()), ttblocks(hip, blocks.get_num_elems());
        tblocks.fill(0.0);
        ttblocks.fill(0.0);
        k::jacobi::transpose_jacobi(hip, nb, 8, precisions, ptrs, blocks, scheme, tblocks);
        k::jacobi::transpose_jacobi(hip, nb, 8, precisions, ptrs, tblocks, scheme, ttblocks);
        ran("jacobi::transpose_jacobi", ttblocks.to_host() == blocks.to_host() && tblocks.to_host() != blocks.to_host());
        ttblocks.fill(0.0);
        k::jacobi::conj_transpose_jacobi(hip, nb, 8, precisions, ptrs, blocks, scheme, ttblocks);   // real values: the transpose
        ran("jacobi::conj_transpose_jacobi", ttblocks.to_host() == tblocks.to_host());
        std::vector<double> hd(n);
        for (size_type i = 0; i < n; ++i) hd[i] = 2.0 + (i % 4);
        array<double> diag(hip, hd.begin(), hd.end()), inv(hip, n);
        k::jacobi::invert_diagonal(hip, diag, inv);
        auto hi = inv.to_host();
        ok = true;
        for (size_type i = 0; i < n; ++i) ok = ok && hi[i] == 1.0 / hd[i];
        ran("jacobi::invert_diagonal", ok);
        auto ones = filled(hip, n, 1.0), sz = Vec::create(hip, dim<2>(n, 1));
        k::jacobi::simple_scalar_apply(hip, inv, ones.get(), sz.get());
        auto hs = host_of(sz.get());
        ok = true;
        for (size_type i = 0; i < n; ++i) ok = ok && hs[i] == hi[i];
        ran("jacobi::simple_scalar_apply", ok);
        k::jacobi::scalar_apply(hip, inv, two.get(), ones.get(), half.get(), sz.get());   // 2 D^-1 1 + 0.5 D^-1 1
        hs = host_of(sz.get());
        ok = true;
        for (size_type i = 0; i < n; ++i) ok = ok && std::abs(hs[i] - 2.5 * hi[i]) < 1e-15;
        ran("jacobi::scalar_apply", ok);
    }

    // ---- ParILU chain (core/factorization/par_ilu.cpp:74-163) and the triangular solves of its factors -------------------
    {
        // a matrix that lacks some diagonal entries: add_diagonal_elements restores them as explicit zeros
        matrix_data<double, int32> holes = data;
        holes.nonzeros.erase(std::remove_if(holes.nonzeros.begin(), holes.nonzeros.end(),
                                            [](const matrix_data<double, int32>::nonzero_type& e) { return e.row == e.column && e.row % 5 == 2; }),
                             holes.nonzeros.end());
        auto H = Mtx::create(hip);
        H->read(holes);
        k::factorization::add_diagonal_elements(hip, H.get(), true);
        matrix_data<double, int32> back;
        H->write(back);
        bool ok = back.nonzeros.size() == data.nonzeros.size();
        for (size_t i = 0; ok && i < back.nonzeros.size(); ++i) {
            const auto &e = back.nonzeros[i], &d = data.nonzeros[i];
            ok = e.row == d.row && e.column == d.column && e.value == ((d.row == d.column && d.row % 5 == 2) ? 0.0 : d.value);
        }
        ran("factorization::add_diagonal_elements", ok);
        // the factors of the mirror's ParIlu (20 sweeps: the fixed point = ILU(0)) against the chain driven through the shims
        auto fact = factorization::ParIlu<double, int32>::build().with_iterations(20u).on(exec)->generate(A);
        auto Lm = fact->get_l_factor();
        auto Um = fact->get_u_factor();
        array<int32> lp(hip, n + 1), up(hip, n + 1);
        k::factorization::initialize_row_ptrs_l_u(hip, A.get(), lp.get_data(), up.get_data());
        auto hl = lp.to_host(); auto hu = up.to_host();
        array<int32> mlp(hip, n + 1), mup(hip, n + 1);
        hip->copy(n + 1, Lm->get_const_row_ptrs(), mlp.get_data());
        hip->copy(n + 1, Um->get_const_row_ptrs(), mup.get_data());
        ran("factorization::initialize_row_ptrs_l_u", hl == mlp.to_host() && hu == mup.to_host());
        auto L = Mtx::create(hip, dim<2>(n, n), hl[n]), U = Mtx::create(hip, dim<2>(n, n), hu[n]);
        hip->copy(n + 1, lp.get_const_data(), L->get_row_ptrs());
        hip->copy(n + 1, up.get_const_data(), U->get_row_ptrs());
        k::factorization::initialize_l_u(hip, A.get(), L.get(), U.get());
        array<int32> lc(hip, hl[n]), mlc(hip, hl[n]);
        hip->copy(hl[n], L->get_const_col_idxs(), lc.get_data());
        hip->copy(hl[n], Lm->get_const_col_idxs(), mlc.get_data());
        ran("factorization::initialize_l_u", lc.to_host() == mlc.to_host());
        auto C = matrix::Coo<double, int32>::create(hip);
        A->convert_to(C.get());
        auto Ut = U->transpose();                                   // the sweeps work on U as CSC (par_ilu.cpp:128-150)
        k::par_ilu_factorization::compute_l_u_factors(hip, 20, C.get(), L.get(), Ut.get());
        auto U2 = Ut->transpose();
        // the fixed point of the sweeps is ILU(0): (L U)(i, j) = A(i, j) on the pattern of A -- checked on the host from the
        // factors the shims computed (the sweeps are asynchronous: digits beyond ~1e-10 vary from run to run), and the
        // factors agree with the mirror's ParIlu to the same level
        matrix_data<double, int32> ld, ud;
        L->write(ld);
        U2->write(ud);
        std::vector<std::vector<std::pair<int32, double>>> lrows(n), urows(n);
        for (const auto& e : ld.nonzeros) lrows[e.row].emplace_back(e.column, e.value);
        for (const auto& e : ud.nonzeros) urows[e.row].emplace_back(e.column, e.value);
        double worst = 0.0;
        std::vector<double> acc(n, 0.0);
        size_type at = 0;
        for (size_type i = 0; i < n; ++i) {
            std::vector<int32> touched;
            for (const auto& lk : lrows[i]) {
                for (const auto& uk : urows[lk.first]) {
                    if (acc[uk.first] == 0.0) touched.push_back(uk.first);
                    acc[uk.first] += lk.second * uk.second;
                }
            }
            for (; at < data.nonzeros.size() && static_cast<size_type>(data.nonzeros[at].row) == i; ++at) {
                worst = std::max(worst, std::abs(acc[data.nonzeros[at].column] - data.nonzeros[at].value));
            }
            for (int32 c : touched) acc[c] = 0.0;
        }
        auto vals_close = [&](const Mtx* a, const Mtx* bm, size_type nnz) {
            array<double> va(hip, nnz), vb(hip, nnz);
            hip->copy(nnz, a->get_const_values(), va.get_data());
            hip->copy(nnz, bm->get_const_values(), vb.get_data());
            auto ha = va.to_host(); auto hb = vb.to_host();
            double w = 0.0;
            for (size_type i = 0; i < nnz; ++i) w = std::max(w, std::abs(ha[i] - hb[i]));
            return w;
        };
        const double dl = vals_close(L.get(), Lm.get(), hl[n]), du = vals_close(U2.get(), Um.get(), hu[n]);
        std::printf("par_ilu: |LU - A| on the pattern %.3e, vs the mirror's factors L %.3e U %.3e\n", worst, dl, du);
        ran("par_ilu_factorization::compute_l_u_factors", worst < 1e-9 && dl < 1e-8 && du < 1e-8);
        // IC-style lower factor: structure of the lower triangle, sqrt of the diagonal
        array<int32> lp2(hip, n + 1);
        k::factorization::initialize_row_ptrs_l(hip, A.get(), lp2.get_data());
        ran("factorization::initialize_row_ptrs_l", lp2.to_host() == hl);
        auto L2 = Mtx::create(hip, dim<2>(n, n), hl[n]);
        hip->copy(n + 1, lp2.get_const_data(), L2->get_row_ptrs());
        k::factorization::initialize_l(hip, A.get(), L2.get(), true);
        matrix_data<double, int32> l2d;
        L2->write(l2d);
        ok = true;
        for (const auto& e : l2d.nonzeros) {
            if (e.row == e.column) ok = ok && std::abs(e.value - std::sqrt(4.0 + 0.001 * (e.row % 13))) < 1e-15;
            else ok = ok && e.column < e.row;
        }
        ran("factorization::initialize_l", ok);
        // upper_trs on U, lower_trs (unit diagonal) on L: residuals of the solves
        bool tr = true;
        k::upper_trs::should_perform_transpose(hip, tr);
        bool tr2 = true;
        k::lower_trs::should_perform_transpose(hip, tr2);
        ran("upper_trs::should_perform_transpose", !tr);
        ran("lower_trs::should_perform_transpose", !tr2);
        std::shared_ptr<solver::SolveStruct> su, sl;
        k::upper_trs::generate(hip, Um.get(), su, false, solver::trisolve_algorithm::syncfree, 1);
        ran("upper_trs::generate", su != nullptr);
        auto z = Vec::create(hip, dim<2>(n, 1)), backv = Vec::create(hip, dim<2>(n, 1));
        k::upper_trs::solve(hip, Um.get(), su.get(), false, solver::trisolve_algorithm::syncfree, nullptr, nullptr, x.get(), z.get());
        Um->apply(z.get(), backv.get());
        k::dense::compute_norm2(hip, x.get(), nrm.get(), tmp);
        const double xn = value_of(hip, nrm.get());
        ran("upper_trs::solve", diff_norm(backv.get(), x.get()) <= 1e-13 * xn);
        k::lower_trs::generate(hip, Lm.get(), sl, true, solver::trisolve_algorithm::syncfree, 1);
        k::lower_trs::solve(hip, Lm.get(), sl.get(), true, solver::trisolve_algorithm::syncfree, nullptr, nullptr, x.get(), z.get());
        Lm->apply(z.get(), backv.get());   // ParILU's L carries its unit diagonal explicitly
        ran("lower_trs::solve<unit_diag>", diff_norm(backv.get(), x.get()) <= 1e-13 * xn);
    }

    // ---- partition::build_starting_indices (reference/distributed/partition_kernels.cpp:114-160) -------------------------
    {
        // ranges [0,10) -> part 1, [10,30) -> part 0, [30,35) -> part 1, [35,60) -> part 2; part 3 is empty
        std::vector<int64> offsets = {0, 10, 30, 35, 60};
        std::vector<int> parts = {1, 0, 1, 2};
        array<int64> doff(hip, offsets.begin(), offsets.end());
        array<int> dparts(hip, parts.begin(), parts.end());
        array<int32> starts(hip, 4), sizes(hip, 4);
        experimental::distributed::comm_index_type empty = -1;
        k::partition::build_starting_indices(hip, doff.get_const_data(), dparts.get_const_data(), 4, 4, empty, starts.get_data(), sizes.get_data());
        ran("partition::build_starting_indices", starts.to_host() == std::vector<int32>({0, 0, 10, 0}) && sizes.to_host() == std::vector<int32>({20, 15, 25, 0}) && empty == 1);
    }
    // ---- the distributed set-up kernels on a Partition whose arrays live on the device --------------------------------------
    {
        using part_t = experimental::distributed::Partition<int32, int64>;
        // reference/test/distributed/matrix_kernels.cpp:301-330 (BuildsLocalNonLocalMixed), part 1
        array<int> mapping(hip, {1, 2, 0, 0, 2, 1});
        auto partition = part_t::build_from_mapping(hip, mapping, 3);
        std::vector<int64> rows = {0, 0, 0, 0, 1, 1, 1, 2, 3, 3, 4, 4, 5, 5}, cols = {0, 1, 3, 5, 1, 4, 5, 3, 1, 2, 3, 4, 0, 2};
        std::vector<double> vals = {11, 1, 2, 12, 13, 14, 5, 15, 6, 16, 7, 17, 18, 8};
        device_matrix_data<double, int64> input(dim<2>(6, 6), array<int64>(hip, rows.begin(), rows.end()), array<int64>(hip, cols.begin(), cols.end()),
                                                array<double>(hip, vals.begin(), vals.end()));
        array<int32> lr(hip), lc(hip), nr(hip), nc(hip), gather(hip);
        array<double> lv(hip), nv(hip);
        array<experimental::distributed::comm_index_type> recv(hip, 3);
        array<int64> n2g(hip);
        k::distributed_matrix::build_local_nonlocal(hip, input, partition.get(), partition.get(), 1, lr, lc, lv, nr, nc, nv, gather, recv, n2g);
        ran("distributed_matrix::build_local_nonlocal",
            lr.to_host() == std::vector<int32>({0, 0, 1}) && lc.to_host() == std::vector<int32>({0, 1, 0}) && lv.to_host() == std::vector<double>({11, 12, 18}) &&
                nr.to_host() == std::vector<int32>({0, 0, 1}) && nc.to_host() == std::vector<int32>({2, 1, 0}) && nv.to_host() == std::vector<double>({1, 2, 8}) &&
                gather.to_host() == std::vector<int32>({0, 1, 0}) && recv.to_host() == std::vector<int32>({2, 0, 1}) && n2g.to_host() == std::vector<int64>({2, 3, 1}));
        // reference/test/distributed/vector_kernels.cpp:137-152 (BuildsLocal), part 2
        std::vector<int64> vr = {0, 0, 1, 1, 2, 3, 4, 5}, vc = {0, 1, 2, 3, 4, 5, 6, 7};
        std::vector<double> vv = {1, 2, 3, 4, 5, 6, 7, 8};
        device_matrix_data<double, int64> vin(dim<2>(6, 8), array<int64>(hip, vr.begin(), vr.end()), array<int64>(hip, vc.begin(), vc.end()),
                                              array<double>(hip, vv.begin(), vv.end()));
        auto local = Vec::create(hip, dim<2>(2, 8));
        local->fill(0.0);
        k::distributed_vector::build_local(hip, vin, partition.get(), 2, local.get());
        auto hl = local->clone(hip->get_master());
        bool ok = true;
        const double expect[2][8] = {{0, 0, 3, 4, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 7, 0}};
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) ok = ok && hl->at(i, j) == expect[i][j];
        ran("distributed_vector::build_local", ok);
        // reference/test/distributed/partition_kernels.cpp:246-295: {1, 1, 0, 0, 2} is connected but not ordered, {0, 1, 1, 2, 2} ordered
        bool unordered = true, ordered = false;
        k::partition::has_ordered_parts(hip, part_t::build_from_mapping(hip, array<int>(hip, {1, 1, 0, 0, 2}), 3).get(), &unordered);
        k::partition::has_ordered_parts(hip, part_t::build_from_mapping(hip, array<int>(hip, {0, 1, 1, 2, 2}), 3).get(), &ordered);
        ran("partition::has_ordered_parts", !unordered && ordered);
    }
    // ---- the <float, int32> instantiations (shims/hip/float_kernels.hip.cpp): reference/test/matrix/csr_kernels.cpp:358-400 and
    //      reference/test/solver/cg_kernels.cpp:181-252 in single precision ---------------------------------------------------------
    {
        using FVec = matrix::Dense<float>;
        auto host = hip->get_master();
        auto fvec = [&](std::initializer_list<float> v, size_type rows, size_type cols) {
            auto d = FVec::create(hip, dim<2>(rows, cols));
            std::vector<float> h(v);
            hip->copy_from(host.get(), h.size(), h.data(), d->get_values());
            return d;
        };
        auto values_of = [&](const FVec* d) {
            std::vector<float> h(d->get_size()[0] * d->get_size()[1]);
            host->copy_from(hip.get(), h.size(), d->get_const_values(), h.data());
            return h;
        };
        // [1 3 2; 0 5 0] (2, 1, 4)^T = (13, 5)^T;  -1 A b + 2 (1, 2)^T = (-11, -1)^T
        auto A = matrix::Csr<float, int32>::create(hip, dim<2>(2, 3), 4);
        const int32 rp[3] = {0, 3, 4}, ci[4] = {0, 1, 2, 1};
        const float av[4] = {1.0f, 3.0f, 2.0f, 5.0f};
        hip->copy_from(host.get(), 3, rp, A->get_row_ptrs());
        hip->copy_from(host.get(), 4, ci, A->get_col_idxs());
        hip->copy_from(host.get(), 4, av, A->get_values());
        auto b = fvec({2.0f, 1.0f, 4.0f}, 3, 1);
        auto c = fvec({0.0f, 0.0f}, 2, 1);
        k::csr::spmv(hip, A.get(), b.get(), c.get());
        ran("csr::spmv<float>", values_of(c.get()) == std::vector<float>({13.0f, 5.0f}));
        auto y = fvec({1.0f, 2.0f}, 2, 1);
        auto al = fvec({-1.0f}, 1, 1), be = fvec({2.0f}, 1, 1);
        k::csr::advanced_spmv(hip, al.get(), A.get(), b.get(), be.get(), y.get());
        ran("csr::advanced_spmv<float>", values_of(y.get()) == std::vector<float>({-11.0f, -1.0f}));
        auto u = fvec({0.0f, 0.0f, 0.0f}, 3, 1);
        k::dense::fill(hip, u.get(), 1.5f);
        k::dense::add_scaled(hip, be.get(), b.get(), u.get());   // 1.5 + 2 b
        ran("dense::add_scaled<float>", values_of(u.get()) == std::vector<float>({5.5f, 3.5f, 9.5f}));
        array<char> tmp(hip, 0);
        auto res = fvec({0.0f}, 1, 1);
        k::dense::compute_dot(hip, b.get(), u.get(), res.get(), tmp);       // 11 + 3.5 + 38 = 52.5
        const bool dot_ok = values_of(res.get())[0] == 52.5f;
        k::dense::compute_norm2(hip, b.get(), res.get(), tmp);              // sqrt(21)
        ran("dense::compute_norm2<float>", dot_ok && std::abs(values_of(res.get())[0] - std::sqrt(21.0f)) < 1e-6f);
        // cg::step_1 / step_2 (cg_kernels.cpp:181-197, 214-234): p = z + rho / prev_rho p
        auto p = fvec({-2.0f, 3.0f, 5.0f, -4.0f}, 2, 2), z = fvec({4.0f, 3.0f, -2.0f, -1.0f}, 2, 2);
        auto rho = fvec({-4.0f, 3.0f}, 1, 2), prev = fvec({2.0f, -3.0f}, 1, 2);
        array<stopping_status> st(hip, 2);
        auto q = fvec({0.0f, 0.0f, 0.0f, 0.0f}, 2, 2), r = fvec({0.0f, 0.0f, 0.0f, 0.0f}, 2, 2), zz = fvec({0.0f, 0.0f, 0.0f, 0.0f}, 2, 2),
             pp = fvec({0.0f, 0.0f, 0.0f, 0.0f}, 2, 2), pr = fvec({0.0f, 0.0f}, 1, 2), rh = fvec({0.0f, 0.0f}, 1, 2);
        k::cg::initialize(hip, p.get(), r.get(), zz.get(), pp.get(), q.get(), pr.get(), rh.get(), &st);   // also clears the stopping statuses
        k::cg::step_1(hip, p.get(), z.get(), rho.get(), prev.get(), &st);
        ran("cg::step_1<float>", values_of(p.get()) == std::vector<float>({8.0f, 0.0f, -12.0f, 3.0f}) &&
                                     values_of(r.get()) == std::vector<float>({-2.0f, 3.0f, 5.0f, -4.0f}) && values_of(pr.get()) == std::vector<float>({1.0f, 1.0f}));
        auto x = fvec({-1.0f, 2.0f, 3.0f, -4.0f}, 2, 2), rr = fvec({4.0f, -2.0f, 1.0f, 3.0f}, 2, 2), beta = fvec({2.0f, -3.0f}, 1, 2);
        auto qq = fvec({1.0f, 2.0f, -3.0f, 4.0f}, 2, 2);
        k::cg::step_2(hip, x.get(), rr.get(), p.get(), qq.get(), beta.get(), rho.get(), &st);   // tmp = rho / beta = (-2, -1)
        ran("cg::step_2<float>", values_of(x.get()) == std::vector<float>({-17.0f, 2.0f, 27.0f, -7.0f}) &&
                                     values_of(rr.get()) == std::vector<float>({6.0f, 0.0f, -5.0f, 7.0f}));
        // residual_norm: tau (0.5, 3) against 0.1 * (10, 10): the first column converges
        auto tau = fvec({0.5f, 3.0f}, 1, 2), orig = fvec({10.0f, 10.0f}, 1, 2);
        array<bool> storage(hip, 2);
        bool all = true, one = false;
        k::residual_norm::residual_norm(hip, tau.get(), orig.get(), 0.1f, 1, true, &st, &storage, &all, &one);
        const auto sth = st.to_host();
        ran("residual_norm::residual_norm<float>", !all && one && (sth[0].data_ & GKOMI_STATUS_CONVERGED) != 0 && (sth[1].data_ & GKOMI_STATUS_ID_MASK) == 0);
    }
    std::printf("wrong: %d\n", wrong);
    return wrong;
}
