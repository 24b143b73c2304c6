// Runs two of the shims on the device against the mirror's own applies (tests/test_cpp_mirror.py).
#include "prelude_mirror.hpp"
namespace gko { namespace kernels { namespace hip {
namespace csr { void spmv(std::shared_ptr<const HipExecutor>, const matrix::Csr<double, int32>*, const matrix::Dense<double>*, matrix::Dense<double>*); }
namespace dense { void compute_norm2(std::shared_ptr<const HipExecutor>, const matrix::Dense<double>*, matrix::Dense<double>*, array<char>&); }
namespace lower_trs {
void generate(std::shared_ptr<const HipExecutor>, const matrix::Csr<double, int32>*, std::shared_ptr<solver::SolveStruct>&, bool, const solver::trisolve_algorithm, const size_type);
void solve(std::shared_ptr<const HipExecutor>, const matrix::Csr<double, int32>*, const solver::SolveStruct*, bool, const solver::trisolve_algorithm, matrix::Dense<double>*, matrix::Dense<double>*, const matrix::Dense<double>*, matrix::Dense<double>*);
}
}}}

int main()
{
    using namespace gko;
    auto hip = HipExecutor::create(0, ReferenceExecutor::create());
    const size_type n = 3000;
    matrix_data<double, int32> d;
    d.size = {n, n};
    for (size_type i = 0; i < n; ++i) {
        if (i > 0) d.nonzeros.emplace_back(i, i - 1, -1.0);
        d.nonzeros.emplace_back(i, i, 2.5);
        if (i + 40 < n) d.nonzeros.emplace_back(i, i + 40, -0.5);
    }
    auto A = matrix::Csr<double, int32>::create(hip);
    A->read(d);
    auto x = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    x->fill(1.0);
    auto y1 = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    auto y2 = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    A->apply(x.get(), y1.get());
    A->make_srow();
    kernels::hip::csr::spmv(hip, A.get(), x.get(), y2.get());
    auto one = initialize<matrix::Dense<double>>({1.0}, hip);
    y2->sub_scaled(one.get(), y1.get());
    auto nrm = matrix::Dense<double>::create(hip, dim<2>(1, 1));
    array<char> tmp(hip);
    kernels::hip::dense::compute_norm2(hip, y2.get(), nrm.get(), tmp);
    const double diff = hip->copy_val_to_host(nrm->get_const_values());
    // lower triangular solve through generate + solve, checked by multiplying back with the lower part
    matrix_data<double, int32> dl;
    dl.size = {n, n};
    for (const auto& e : d.nonzeros) if (e.column <= e.row) dl.nonzeros.push_back(e);
    auto L = matrix::Csr<double, int32>::create(hip);
    L->read(dl);
    std::shared_ptr<solver::SolveStruct> st;
    kernels::hip::lower_trs::generate(hip, L.get(), st, false, solver::trisolve_algorithm::syncfree, 1);
    auto z = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    kernels::hip::lower_trs::solve(hip, L.get(), st.get(), false, solver::trisolve_algorithm::syncfree, nullptr, nullptr, x.get(), z.get());
    auto back = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    L->apply(z.get(), back.get());
    back->sub_scaled(one.get(), x.get());
    kernels::hip::dense::compute_norm2(hip, back.get(), nrm.get(), tmp);
    const double res = hip->copy_val_to_host(nrm->get_const_values());
    std::cout << "spmv shim vs mirror apply: " << diff << "\ntrs shim residual: " << res << std::endl;
    return diff == 0.0 && res < 1e-10 ? 0 : 2;
}
