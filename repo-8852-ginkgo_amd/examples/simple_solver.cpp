// Minimal CG solve through the gko:: host mirror, the flow of the reference's
// examples/simple-solver: read A, b, x0 from data/, CG with
// Combined(Iteration(20), ResidualNorm(1e-7)), print x and ||b - A x||.
// Usage: simple_solver [reference|omp|hip]   (only hip has kernels)
#include <ginkgo/ginkgo.hpp>

#include <fstream>
#include <iostream>
#include <string>

int main(int argc, char* argv[])
{
    using vec = gko::matrix::Dense<double>;
    using mtx = gko::matrix::Csr<double, int>;
    using cg = gko::solver::Cg<double>;
    const std::string which = argc >= 2 ? argv[1] : "hip";
    try {
        std::shared_ptr<gko::Executor> exec;
        if (which == "hip") {
            exec = gko::HipExecutor::create(0, gko::OmpExecutor::create(), true);
        } else if (which == "omp") {
            exec = gko::OmpExecutor::create();
        } else {
            exec = gko::ReferenceExecutor::create();
        }
        std::cout << gko::version_info::get() << std::endl;
        auto A = gko::share(gko::read<mtx>(std::ifstream("data/A.mtx"), exec));
        auto b = gko::read<vec>(std::ifstream("data/b.mtx"), exec);
        auto x = gko::read<vec>(std::ifstream("data/x0.mtx"), exec);
        auto solver = cg::build()
                          .with_criteria(gko::stop::Iteration::build().with_max_iters(20u).on(exec),
                                         gko::stop::ResidualNorm<double>::build().with_reduction_factor(1e-7).on(exec))
                          .on(exec)
                          ->generate(A);
        solver->apply(gko::lend(b), gko::lend(x));
        std::cout << "Solution (x):\n";
        gko::write(std::cout, gko::lend(x));
        auto one = gko::initialize<vec>({1.0}, exec);
        auto neg_one = gko::initialize<vec>({-1.0}, exec);
        auto res = gko::initialize<vec>({0.0}, exec);
        A->apply(gko::lend(one), gko::lend(x), gko::lend(neg_one), gko::lend(b));
        b->compute_norm2(gko::lend(res));
        std::cout << "Residual norm sqrt(r^T r):\n";
        gko::write(std::cout, gko::lend(res));
        std::cout << "iterations: " << solver->get_last_iteration_count() << "\n";
    } catch (const gko::Error& e) {
        std::cerr << "gko::Error: " << e.what() << std::endl;
        return 3;
    }
    return 0;
}
