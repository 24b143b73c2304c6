// Runs shims on the device against the mirror's own applies (tests/test_cpp_mirror.py): csr::spmv, dense::compute_norm2,
// lower_trs::generate / solve, and (round 3) ell::spmv, sellp::advanced_spmv, coo::spmv2, components::prefix_sum,
// fcg::step_1.
#include "prelude_mirror.hpp"
#include <cmath>
#include <vector>
namespace gko { namespace kernels { namespace hip {
namespace csr { void spmv(std::shared_ptr<const HipExecutor>, const matrix::Csr<double, int32>*, const matrix::Dense<double>*, matrix::Dense<double>*); }
namespace dense { void compute_norm2(std::shared_ptr<const HipExecutor>, const matrix::Dense<double>*, matrix::Dense<double>*, array<char>&); }
namespace ell { void spmv(std::shared_ptr<const HipExecutor>, const matrix::Ell<double, int32>*, const matrix::Dense<double>*, matrix::Dense<double>*); }
namespace sellp { void advanced_spmv(std::shared_ptr<const HipExecutor>, const matrix::Dense<double>*, const matrix::Sellp<double, int32>*, const matrix::Dense<double>*, const matrix::Dense<double>*, matrix::Dense<double>*); }
namespace coo { void spmv2(std::shared_ptr<const HipExecutor>, const matrix::Coo<double, int32>*, const matrix::Dense<double>*, matrix::Dense<double>*); }
namespace components { void prefix_sum(std::shared_ptr<const HipExecutor>, int32*, size_type); }
namespace fcg { void step_1(std::shared_ptr<const HipExecutor>, matrix::Dense<double>*, const matrix::Dense<double>*, const matrix::Dense<double>*, const matrix::Dense<double>*, const array<stopping_status>*); }
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
    // ---- round 3: the other formats through their shims against the mirror's CSR apply (y1 = A x)
    auto norm_of_difference = [&](matrix::Dense<double>* y) {
        y->sub_scaled(one.get(), y1.get());
        kernels::hip::dense::compute_norm2(hip, y, nrm.get(), tmp);
        return hip->copy_val_to_host(nrm->get_const_values());
    };
    auto E = matrix::Ell<double, int32>::create(hip);
    A->convert_to(E.get());
    auto ye = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    kernels::hip::ell::spmv(hip, E.get(), x.get(), ye.get());
    const double d_ell = norm_of_difference(ye.get());
    auto S = matrix::Sellp<double, int32>::create(hip);
    A->convert_to(S.get());
    auto ys = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    ys->fill(0.0);
    auto zero = initialize<matrix::Dense<double>>({0.0}, hip);
    kernels::hip::sellp::advanced_spmv(hip, one.get(), S.get(), x.get(), zero.get(), ys.get());   // 1 * A x + 0 * y
    const double d_sellp = norm_of_difference(ys.get());
    auto Co = matrix::Coo<double, int32>::create(hip);
    A->convert_to(Co.get());
    auto yc = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    yc->fill(0.0);
    kernels::hip::coo::spmv2(hip, Co.get(), x.get(), yc.get());                                    // y += A x on y = 0
    const double d_coo = norm_of_difference(yc.get());
    // components::prefix_sum: exclusive scan in place
    std::vector<int32> counts(5000);
    for (size_t i = 0; i < counts.size(); ++i) counts[i] = static_cast<int32>(i % 7);
    array<int32> dc(hip, counts.begin(), counts.end());
    kernels::hip::components::prefix_sum(hip, dc.get_data(), counts.size());
    const auto scanned = dc.to_host();
    bool scan_ok = true;
    for (size_t i = 0, run = 0; i < counts.size(); run += counts[i], ++i) scan_ok = scan_ok && scanned[i] == static_cast<int32>(run);
    // fcg::step_1: p = z + (rho_t / prev_rho) p on a column that has not stopped
    auto p = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    auto zv = matrix::Dense<double>::create(hip, dim<2>(n, 1));
    p->fill(2.0);
    zv->fill(3.0);
    auto rho_t = initialize<matrix::Dense<double>>({6.0}, hip);
    auto prev_rho = initialize<matrix::Dense<double>>({4.0}, hip);
    std::vector<stopping_status> st0(1);
    array<stopping_status> status(hip, st0.begin(), st0.end());
    kernels::hip::fcg::step_1(hip, p.get(), zv.get(), rho_t.get(), prev_rho.get(), &status);
    kernels::hip::dense::compute_norm2(hip, p.get(), nrm.get(), tmp);
    const double pn = hip->copy_val_to_host(nrm->get_const_values());   // every entry 3 + 1.5 * 2 = 6
    const bool step_ok = std::abs(pn - 6.0 * std::sqrt(static_cast<double>(n))) < 1e-9;
    std::cout << "ell / sellp / coo shims vs mirror apply: " << d_ell << " / " << d_sellp << " / " << d_coo << "\nprefix_sum shim: "
              << (scan_ok ? "ok" : "WRONG") << "\nfcg::step_1 shim: " << (step_ok ? "ok" : "WRONG") << std::endl;
    // one line per kernel, like shim_smoke2.cpp (tests/test_cpp_mirror.py compares the list with INTEGRATION.md)
    auto ran = [](const char* name, bool ok) { std::cout << "ran " << name << (ok ? " ok" : " WRONG") << "\n"; };
    ran("csr::spmv", diff == 0.0);
    ran("dense::compute_norm2", diff == 0.0 && res < 1e-10);
    ran("lower_trs::generate", st != nullptr);
    ran("lower_trs::solve", res < 1e-10);
    ran("ell::spmv", d_ell == 0.0);
    ran("sellp::advanced_spmv", d_sellp == 0.0);
    ran("coo::spmv2", d_coo < 1e-10);
    ran("components::prefix_sum<int32>", scan_ok);
    ran("fcg::step_1", step_ok);
    return diff == 0.0 && res < 1e-10 && d_ell == 0.0 && d_sellp == 0.0 && d_coo < 1e-10 && scan_ok && step_ok ? 0 : 2;
}
