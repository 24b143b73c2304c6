// Exercises the rest of the gko:: mirror on a HipExecutor: format conversions
// and every format's apply, block-Jacobi-preconditioned CG, ParILU-preconditioned
// GMRES(30), on an n x n 5-pt Poisson / convection-diffusion matrix built in
// place.  Prints "name value" lines that tests/test_cpp_mirror.py checks.
#include <ginkgo/ginkgo.hpp>

#include <algorithm>
#include <cmath>
#include <iostream>

using vec = gko::matrix::Dense<double>;
using csr = gko::matrix::Csr<double, int>;

static gko::matrix_data<double, int> stencil(int g, double upwind)
{
    gko::matrix_data<double, int> d;
    d.size = gko::dim<2>(g * g, g * g);
    for (int i = 0; i < g; ++i) {
        for (int j = 0; j < g; ++j) {
            const int r = i * g + j;
            if (i > 0) d.nonzeros.push_back({r, r - g, -1.0});
            if (j > 0) d.nonzeros.push_back({r, r - 1, -1.0 - upwind});
            d.nonzeros.push_back({r, r, 4.0 + upwind});
            if (j < g - 1) d.nonzeros.push_back({r, r + 1, -1.0});
            if (i < g - 1) d.nonzeros.push_back({r, r + g, -1.0});
        }
    }
    return d;
}

static double diff_norm(std::shared_ptr<const gko::Executor> exec, const vec* a, const vec* b)
{
    auto d = a->clone();
    auto neg = gko::initialize<vec>({-1.0}, exec);
    d->add_scaled(neg.get(), b);
    auto n = vec::create(exec, gko::dim<2>(1, 1));
    d->compute_norm2(n.get());
    return exec->copy_val_to_host(n->get_const_values());
}

int main()
{
    try {
        auto exec = gko::HipExecutor::create(0, gko::OmpExecutor::create());
        const int g = 96, n = g * g;
        auto A = gko::share(csr::create(exec));
        A->read(stencil(g, 0.0));
        auto host_x = vec::create(exec->get_master(), gko::dim<2>(n, 1));
        for (int i = 0; i < n; ++i) host_x->at(i) = std::sin(0.01 * i);
        auto x = host_x->clone(exec);
        auto y = vec::create(exec, gko::dim<2>(n, 1));
        A->apply(x.get(), y.get());
        // every format agrees with CSR
        auto y2 = vec::create(exec, gko::dim<2>(n, 1));
        auto ell = gko::matrix::Ell<double, int>::create(exec);
        A->convert_to(ell.get());
        ell->apply(x.get(), y2.get());
        std::cout << "ell_diff " << diff_norm(exec, y.get(), y2.get()) << "\n";
        auto sellp = gko::matrix::Sellp<double, int>::create(exec);
        A->convert_to(sellp.get());
        sellp->apply(x.get(), y2.get());
        std::cout << "sellp_diff " << diff_norm(exec, y.get(), y2.get()) << "\n";
        auto coo = gko::matrix::Coo<double, int>::create(exec);
        A->convert_to(coo.get());
        coo->apply(x.get(), y2.get());
        std::cout << "coo_diff " << diff_norm(exec, y.get(), y2.get()) << "\n";
        // converted from CSR: sorted by row -> the atomic-free kernels; apply2 adds on top
        coo->apply2(x.get(), y2.get());
        y2->scale(gko::initialize<vec>({0.5}, exec).get());
        std::cout << "coo_sorted " << coo->is_sorted_by_row() << " coo_apply2_diff " << diff_norm(exec, y.get(), y2.get()) << "\n";
        auto hyb = gko::matrix::Hybrid<double, int>::create(exec, std::make_shared<gko::matrix::Hybrid<double, int>::column_limit>(3));
        A->convert_to(hyb.get());
        hyb->apply(x.get(), y2.get());
        std::cout << "hybrid_diff " << diff_norm(exec, y.get(), y2.get()) << " coo_nnz " << hyb->get_coo_num_stored_elements() << "\n";

        // Csr<double, int64>: the index type of matrices beyond 2^31 nonzeros, same bits as <double, int32>
        {
            gko::matrix_data<double, gko::int64> d64;
            auto d32 = stencil(g, 0.0);
            d64.size = d32.size;
            for (const auto& e : d32.nonzeros) d64.nonzeros.push_back({e.row, e.column, e.value});
            auto A64 = gko::matrix::Csr<double, gko::int64>::create(exec);
            A64->read(d64);
            A64->apply(x.get(), y2.get());
            std::cout << "csr_int64_diff " << diff_norm(exec, y.get(), y2.get()) << " srow_entries " << A64->get_num_srow_elements() << "\n";
            auto two = gko::initialize<vec>({2.0}, exec);
            auto one = gko::initialize<vec>({1.0}, exec);
            auto y3 = y->clone();
            A64->apply(two.get(), x.get(), one.get(), y2.get());   // y2 = 2 A x + y2 = 3 A x
            A->apply(two.get(), x.get(), one.get(), y3.get());
            std::cout << "csr_int64_advanced_diff " << diff_norm(exec, y3.get(), y2.get()) << "\n";
        }

        // long rows whose gathers span 8 MB of b: make_srow's column statistic switches the load-balanced kernel to
        // column windows (Csr::load_balance and the default strategy alike); results to rounding of the COO kernel's
        {
            const int nl = 1 << 20;
            gko::matrix_data<double, int> dl(gko::dim<2>(nl, nl));
            unsigned long long state = 12345;
            auto next = [&] { state = state * 6364136223846793005ull + 1442695040888963407ull; return static_cast<unsigned>(state >> 33); };
            for (int i = 0; i < nl; ++i) {
                if (i % 65536 == 7) {
                    std::vector<int> cs(50000);
                    for (auto& c : cs) c = static_cast<int>(next() % nl);
                    std::sort(cs.begin(), cs.end());
                    cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
                    for (int c : cs) dl.nonzeros.push_back({i, c, 1.0 + 1e-3 * (c % 97)});
                } else {
                    dl.nonzeros.push_back({i, i, 2.0});
                }
            }
            auto xl = vec::create(exec->get_master(), gko::dim<2>(nl, 1));
            for (int i = 0; i < nl; ++i) xl->at(i) = std::cos(0.001 * i);
            auto dxl = xl->clone(exec);
            auto cl = gko::matrix::Coo<double, int>::create(exec);
            {
                auto tmp = csr::create(exec);
                tmp->read(dl);
                tmp->convert_to(cl.get());
            }
            auto yc = vec::create(exec, gko::dim<2>(nl, 1));
            cl->apply(dxl.get(), yc.get());
            auto yl = vec::create(exec, gko::dim<2>(nl, 1));
            double worst = 0.0;
            for (int which = 0; which < 2; ++which) {
                auto Al = which ? csr::create(exec, std::make_shared<csr::load_balance>(exec)) : csr::create(exec);
                Al->read(dl);
                Al->apply(dxl.get(), yl.get());
                worst = std::max(worst, diff_norm(exec, yc.get(), yl.get()));
            }
            std::cout << "csr_long_rows_diff " << worst << " nnz " << dl.nonzeros.size() << "\n";
        }

        // Csr<float, int32>::apply: the single-precision instantiation of csr::spmv / advanced_spmv
        // (reference/test/matrix/csr_kernels.cpp:358-400 in float: [1 3 2; 0 5 0] (2, 1, 4)^T = (13, 5)^T, then -1 A b + 2 c)
        {
            using fvec = gko::matrix::Dense<float>;
            auto host = exec->get_master();
            auto Af = gko::matrix::Csr<float, int>::create(exec, gko::dim<2>(2, 3), 4);
            const int rp[3] = {0, 3, 4}, ci[4] = {0, 1, 2, 1};
            const float av[4] = {1.0f, 3.0f, 2.0f, 5.0f}, bv[3] = {2.0f, 1.0f, 4.0f}, two_v[1] = {2.0f}, neg_v[1] = {-1.0f};
            exec->copy_from(host.get(), 3, rp, Af->get_row_ptrs());
            exec->copy_from(host.get(), 4, ci, Af->get_col_idxs());
            exec->copy_from(host.get(), 4, av, Af->get_values());
            auto bf = fvec::create(exec, gko::dim<2>(3, 1)), cf = fvec::create(exec, gko::dim<2>(2, 1));
            auto two_f = fvec::create(exec, gko::dim<2>(1, 1)), neg_f = fvec::create(exec, gko::dim<2>(1, 1));
            exec->copy_from(host.get(), 3, bv, bf->get_values());
            exec->copy_from(host.get(), 1, two_v, two_f->get_values());
            exec->copy_from(host.get(), 1, neg_v, neg_f->get_values());
            Af->apply(bf.get(), cf.get());
            float out[2] = {0.0f, 0.0f}, out2[2] = {0.0f, 0.0f};
            host->copy_from(exec.get(), 2, cf->get_const_values(), out);
            Af->apply(neg_f.get(), bf.get(), two_f.get(), cf.get());     // -1 (13, 5) + 2 (13, 5)
            host->copy_from(exec.get(), 2, cf->get_const_values(), out2);
            std::cout << "csr_float " << out[0] << " " << out[1] << " advanced " << out2[0] << " " << out2[1] << "\n";
            // Cg<float> on the reference's stencil system (reference/test/solver/cg_kernels.cpp:255-266): x = (1, 3, 2)
            auto Sf = gko::share(gko::matrix::Csr<float, int>::create(exec, gko::dim<2>(3, 3), 7));
            const int srp[4] = {0, 2, 5, 7}, sci[7] = {0, 1, 0, 1, 2, 1, 2};
            const float sv[7] = {2.0f, -1.0f, -1.0f, 2.0f, -1.0f, -1.0f, 2.0f}, sb[3] = {-1.0f, 3.0f, 1.0f}, zero3[3] = {0.0f, 0.0f, 0.0f};
            exec->copy_from(host.get(), 4, srp, Sf->get_row_ptrs());
            exec->copy_from(host.get(), 7, sci, Sf->get_col_idxs());
            exec->copy_from(host.get(), 7, sv, Sf->get_values());
            auto rhs = fvec::create(exec, gko::dim<2>(3, 1)), sol3 = fvec::create(exec, gko::dim<2>(3, 1));
            exec->copy_from(host.get(), 3, sb, rhs->get_values());
            exec->copy_from(host.get(), 3, zero3, sol3->get_values());
            auto cgf = gko::solver::Cg<float>::build()
                           .with_criteria(gko::stop::Iteration::build().with_max_iters(400u).on(exec),
                                          gko::stop::ResidualNorm<float>::build().with_reduction_factor(1.2e-6f).on(exec))
                           .on(exec)
                           ->generate(Sf);
            cgf->apply(rhs.get(), sol3.get());
            float xs[3] = {};
            host->copy_from(exec.get(), 3, sol3->get_const_values(), xs);
            std::cout << "cg_float_iters " << cgf->get_last_iteration_count() << " converged " << cgf->has_converged() << " err "
                      << std::max(std::max(std::abs(xs[0] - 1.0f), std::abs(xs[1] - 3.0f)), std::abs(xs[2] - 2.0f)) << "\n";
        }

        // the analysis-based strategy of this backend: a column-partitioned copy for scattered column patterns
        // (uniformly random columns over 4.8 MB of b); same product to rounding, also after the values changed
        {
            const int np_ = 600000;
            gko::matrix_data<double, int> dp(gko::dim<2>(np_, np_));
            unsigned long long state = 987654321;
            auto next = [&] { state = state * 6364136223846793005ull + 1442695040888963407ull; return static_cast<unsigned>(state >> 33); };
            dp.nonzeros.reserve(static_cast<size_t>(np_) * 8);
            for (int i = 0; i < np_; ++i) {
                int cs[8];
                for (auto& c : cs) c = static_cast<int>(next() % np_);
                std::sort(cs, cs + 8);
                for (int k = 0; k < 8; ++k) {
                    if (k == 0 || cs[k] != cs[k - 1]) dp.nonzeros.push_back({i, cs[k], 0.5 + 1e-3 * (cs[k] % 89)});
                }
            }
            auto xp = vec::create(exec->get_master(), gko::dim<2>(np_, 1));
            for (int i = 0; i < np_; ++i) xp->at(i) = std::sin(0.003 * i);
            auto dxp = xp->clone(exec);
            auto plain = csr::create(exec);
            plain->read(dp);
            auto part = csr::create(exec, std::make_shared<csr::gkomi_partitioned>());
            part->read(dp);
            auto y0 = vec::create(exec, gko::dim<2>(np_, 1));
            auto y1 = vec::create(exec, gko::dim<2>(np_, 1));
            plain->apply(dxp.get(), y0.get());
            part->apply(dxp.get(), y1.get());
            const double d1 = diff_norm(exec, y0.get(), y1.get());
            // new values through the mutable accessor: the copy is re-gathered before the next apply
            std::vector<double> doubled(dp.nonzeros.size());
            for (size_t k = 0; k < doubled.size(); ++k) doubled[k] = 2.0 * dp.nonzeros[k].value;
            exec->copy_from(exec->get_master().get(), doubled.size(), doubled.data(), part->get_values());
            part->apply(dxp.get(), y1.get());
            y0->scale(gko::initialize<vec>({2.0}, exec).get());
            std::cout << "csr_partitioned_diff " << d1 << " has_copy " << part->has_partitioned_copy() << " after_new_values "
                      << diff_norm(exec, y0.get(), y1.get()) << " plain_has_copy " << plain->has_partitioned_copy() << "\n";
        }

        // CG + block-Jacobi
        auto b = vec::create(exec, gko::dim<2>(n, 1));
        b->fill(1.0);
        auto sol = vec::create(exec, gko::dim<2>(n, 1));
        sol->fill(0.0);
        auto cg = gko::solver::Cg<double>::build()
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(2000u).on(exec),
                                     gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                      .with_preconditioner(gko::preconditioner::Jacobi<double, int>::build().with_max_block_size(8u).on(exec))
                      .on(exec)
                      ->generate(A);
        cg->apply(b.get(), sol.get());
        auto r = b->clone();
        auto one = gko::initialize<vec>({1.0}, exec);
        auto neg = gko::initialize<vec>({-1.0}, exec);
        A->apply(neg.get(), sol.get(), one.get(), r.get());
        auto rn = vec::create(exec, gko::dim<2>(1, 1));
        r->compute_norm2(rn.get());
        std::cout << "cg_jacobi_iters " << cg->get_last_iteration_count() << " converged " << cg->has_converged()
                  << " true_residual " << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";

        // the same solve with adaptive-precision block storage (storage_optimization = autodetect)
        {
            sol->fill(0.0);
            auto bj = gko::share(gko::preconditioner::Jacobi<double, int>::build()
                                     .with_max_block_size(8u)
                                     .with_storage_optimization(gko::precision_reduction::autodetect())
                                     .with_accuracy(1e-1)
                                     .on(exec)
                                     ->generate(A));
            int reduced = 0;
            for (auto p : bj->get_block_precisions()) reduced += p != gko::precision_reduction(0, 0);
            auto cg2 = gko::solver::Cg<double>::build()
                           .with_criteria(gko::stop::Iteration::build().with_max_iters(2000u).on(exec),
                                          gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                           .with_generated_preconditioner(bj)
                           .on(exec)
                           ->generate(A);
            cg2->apply(b.get(), sol.get());
            std::cout << "cg_adaptive_jacobi_iters " << cg2->get_last_iteration_count() << " converged " << cg2->has_converged() << " reduced_blocks "
                      << reduced << " of " << bj->get_num_blocks() << "\n";
        }

        // GMRES(30) + ParILU on the nonsymmetric variant
        auto B = gko::share(csr::create(exec));
        B->read(stencil(g, 0.5));
        sol->fill(0.0);
        auto gm = gko::solver::Gmres<double>::build()
                      .with_krylov_dim(30u)
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(1000u).on(exec),
                                     gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                      .with_preconditioner(gko::preconditioner::Ilu<double, int>::build().with_factorization_iterations(20u).on(exec))
                      .on(exec)
                      ->generate(B);
        gm->apply(b.get(), sol.get());
        r->copy_from(b.get());
        B->apply(neg.get(), sol.get(), one.get(), r.get());
        r->compute_norm2(rn.get());
        std::cout << "gmres_ilu_iters " << gm->get_last_iteration_count() << " converged " << gm->has_converged()
                  << " true_residual " << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";

        // config 4 of BASELINE.json: GMRES(30) + ILU with the system matrix in SELL-P and ELL
        {
            auto Bs = gko::share(gko::matrix::Sellp<double, int>::create(exec));
            B->convert_to(Bs.get());
            auto Be = gko::share(gko::matrix::Ell<double, int>::create(exec));
            B->convert_to(Be.get());
            auto ilu = gko::share(gko::preconditioner::Ilu<double, int>::build().with_factorization_iterations(20u).on(exec)->generate(B));
            int k = 0;
            for (auto M : {std::shared_ptr<const gko::LinOp>(Bs), std::shared_ptr<const gko::LinOp>(Be)}) {
                sol->fill(0.0);
                auto g2 = gko::solver::Gmres<double>::build()
                              .with_krylov_dim(30u)
                              .with_criteria(gko::stop::Iteration::build().with_max_iters(1000u).on(exec),
                                             gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                              .with_generated_preconditioner(ilu)
                              .on(exec)->generate(M);
                g2->apply(b.get(), sol.get());
                r->copy_from(b.get());
                B->apply(neg.get(), sol.get(), one.get(), r.get());
                r->compute_norm2(rn.get());
                std::cout << (k++ == 0 ? "gmres_ilu_sellp_iters " : "gmres_ilu_ell_iters ") << g2->get_last_iteration_count() << " converged " << g2->has_converged()
                          << " true_residual " << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";
            }
        }

        // the other Krylov solvers on the same systems: FCG (SPD), BiCGSTAB and CGS (nonsymmetric)
        {
            auto crit = [&]() {
                return std::make_pair(gko::stop::Iteration::build().with_max_iters(2000u).on(exec),
                                      gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec));
            };
            auto report = [&](const char* name, const csr* M, int64_t iters, bool conv) {
                r->copy_from(b.get());
                M->apply(neg.get(), sol.get(), one.get(), r.get());
                r->compute_norm2(rn.get());
                std::cout << name << " " << iters << " converged " << conv << " true_residual "
                          << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";
            };
            sol->fill(0.0);
            auto c1 = crit();
            auto fcg = gko::solver::Fcg<double>::build().with_criteria(c1.first, c1.second)
                           .with_preconditioner(gko::preconditioner::Jacobi<double, int>::build().with_max_block_size(8u).on(exec)).on(exec)->generate(A);
            fcg->apply(b.get(), sol.get());
            report("fcg_jacobi_iters", A.get(), fcg->get_last_iteration_count(), fcg->has_converged());
            sol->fill(0.0);
            auto c4 = crit();
            auto iccg = gko::solver::Cg<double>::build().with_criteria(c4.first, c4.second)
                            .with_preconditioner(gko::preconditioner::Ic<double, int>::build().with_factorization_iterations(10u).on(exec)).on(exec)->generate(A);
            iccg->apply(b.get(), sol.get());
            report("cg_ic_iters", A.get(), iccg->get_last_iteration_count(), iccg->has_converged());
            sol->fill(0.0);
            auto c2 = crit();
            auto bicg = gko::solver::Bicgstab<double>::build().with_criteria(c2.first, c2.second)
                            .with_preconditioner(gko::preconditioner::Ilu<double, int>::build().with_factorization_iterations(20u).on(exec)).on(exec)->generate(B);
            bicg->apply(b.get(), sol.get());
            report("bicgstab_ilu_iters", B.get(), bicg->get_last_iteration_count(), bicg->has_converged());
            sol->fill(0.0);
            auto c3 = crit();
            // (unpreconditioned CGS loses its recurrence residual on this matrix -- in the reference too)
            auto cgs = gko::solver::Cgs<double>::build().with_criteria(c3.first, c3.second)
                           .with_preconditioner(gko::preconditioner::Ilu<double, int>::build().with_factorization_iterations(20u).on(exec)).on(exec)->generate(B);
            cgs->apply(b.get(), sol.get());
            report("cgs_ilu_iters", B.get(), cgs->get_last_iteration_count(), cgs->has_converged());
        }

        // BiCG on the nonsymmetric system; Richardson / IR with block-Jacobi as the inner "solver"
        {
            sol->fill(0.0);
            auto bicg = gko::solver::Bicg<double>::build()
                            .with_criteria(gko::stop::Iteration::build().with_max_iters(2000u).on(exec),
                                           gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                            .on(exec)->generate(B);
            bicg->apply(b.get(), sol.get());
            r->copy_from(b.get());
            B->apply(neg.get(), sol.get(), one.get(), r.get());
            r->compute_norm2(rn.get());
            std::cout << "bicg_iters " << bicg->get_last_iteration_count() << " converged " << bicg->has_converged() << " true_residual "
                      << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";
            // block-Jacobi is Transposable: BiCG applies M^-1 to r and M^-T to the shadow residual
            sol->fill(0.0);
            auto bicg_j = gko::solver::Bicg<double>::build()
                              .with_criteria(gko::stop::Iteration::build().with_max_iters(2000u).on(exec),
                                             gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                              .with_preconditioner(gko::preconditioner::Jacobi<double, int>::build().with_max_block_size(16u).on(exec))
                              .on(exec)->generate(B);
            bicg_j->apply(b.get(), sol.get());
            r->copy_from(b.get());
            B->apply(neg.get(), sol.get(), one.get(), r.get());
            r->compute_norm2(rn.get());
            std::cout << "bicg_jacobi_iters " << bicg_j->get_last_iteration_count() << " converged " << bicg_j->has_converged() << " true_residual "
                      << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";
            // ... and so is Ilu: (U^-1 L^-1)^T = the lower solve with U^T, then the upper solve with L^T
            sol->fill(0.0);
            auto bicg_i = gko::solver::Bicg<double>::build()
                              .with_criteria(gko::stop::Iteration::build().with_max_iters(2000u).on(exec),
                                             gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-10).on(exec))
                              .with_preconditioner(gko::preconditioner::Ilu<double, int>::build().with_factorization_iterations(20u).on(exec))
                              .on(exec)->generate(B);
            bicg_i->apply(b.get(), sol.get());
            r->copy_from(b.get());
            B->apply(neg.get(), sol.get(), one.get(), r.get());
            r->compute_norm2(rn.get());
            std::cout << "bicg_ilu_iters " << bicg_i->get_last_iteration_count() << " converged " << bicg_i->has_converged() << " true_residual "
                      << exec->copy_val_to_host(rn->get_const_values()) / std::sqrt(double(n)) << "\n";
            sol->fill(0.0);
            auto ir = gko::solver::Ir<double>::build()
                          .with_criteria(gko::stop::Iteration::build().with_max_iters(200u).on(exec),
                                         gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-2).on(exec))
                          .with_solver(gko::preconditioner::Jacobi<double, int>::build().with_max_block_size(8u).on(exec))
                          .with_relaxation_factor(0.9)
                          .on(exec)->generate(A);
            ir->apply(b.get(), sol.get());
            r->copy_from(b.get());
            A->apply(neg.get(), sol.get(), one.get(), r.get());
            r->compute_norm2(rn.get());
            auto bn = vec::create(exec, gko::dim<2>(1, 1));
            b->compute_norm2(bn.get());
            std::cout << "ir_jacobi_iters " << ir->get_last_iteration_count() << " converged " << ir->has_converged() << " rel_residual "
                      << exec->copy_val_to_host(rn->get_const_values()) / exec->copy_val_to_host(bn->get_const_values()) << "\n";
        }

        // assembly on the device: shuffled triplets with duplicates and zeros ->
        // device_matrix_data::sum_duplicates / remove_zeros -> Csr::read
        {
            auto data = stencil(g, 0.0);
            gko::matrix_data<double, int> messy;
            messy.size = data.size;
            for (std::size_t i = 0; i < data.nonzeros.size(); ++i) {
                const auto& e = data.nonzeros[(i * 7919) % data.nonzeros.size()];  // 7919 is coprime to nnz
                messy.nonzeros.push_back({e.row, e.column, 0.25 * e.value});
                messy.nonzeros.push_back({e.row, e.column, 0.75 * e.value});
                if (i % 3 == 0) messy.nonzeros.push_back({e.row, (e.column + 5) % n, 0.0});
            }
            auto dev_data = gko::device_matrix_data<double, int>::create_from_host(exec, messy);
            dev_data.remove_zeros();
            dev_data.sum_duplicates();
            auto C = csr::create(exec);
            C->read(std::move(dev_data));
            C->apply(x.get(), y2.get());
            std::cout << "assembly_nnz " << C->get_num_stored_elements() << " of " << A->get_num_stored_elements() << " diff "
                      << diff_norm(exec, y.get(), y2.get()) << "\n";
        }

        // error behaviour: dimension mismatch is caught before the boundary
        try {
            auto bad = vec::create(exec, gko::dim<2>(n + 1, 1));
            A->apply(bad.get(), y.get());
            std::cout << "dimension_check missing\n";
        } catch (const gko::DimensionMismatch&) {
            std::cout << "dimension_check ok\n";
        }
    } catch (const gko::Error& e) {
        std::cerr << "gko::Error: " << e.what() << std::endl;
        return 3;
    }
    return 0;
}
