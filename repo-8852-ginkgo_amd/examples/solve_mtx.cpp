// solve_mtx: solve A x = b for MatrixMarket files with the solvers of the hot path, through the gko:: host
// mirror over libgkomi.so.
//
//   solve_mtx --dir data [--executor hip|reference|omp] [--solver cg|fcg|gmres|bicgstab|cgs]
//             [--precond none|jacobi|ilu] [--max-iters N] [--reduction R] [--restart K] [--quiet]
//
// reads <dir>/A.mtx, <dir>/b.mtx and, when present, <dir>/x0.mtx (else x0 = 0), prints the solution (unless
// --quiet), the TRUE residual norm ||b - A x||_2 computed by a separate apply, and the iteration count.
// Host executors carry no kernels in this backend: they end in gko::NotCompiled (exit code 3).
#include <ginkgo/ginkgo.hpp>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <string>

namespace {

using dense = gko::matrix::Dense<double>;
using csr = gko::matrix::Csr<double, int>;

struct options {
    std::string dir = "data", executor = "hip", solver = "cg", precond = "none";
    unsigned max_iters = 20;
    double reduction = 1e-7;
    unsigned restart = 30;
    bool quiet = false;
};

options parse(int argc, char** argv)
{
    options o;
    std::map<std::string, std::string*> text{{"--dir", &o.dir}, {"--executor", &o.executor}, {"--solver", &o.solver}, {"--precond", &o.precond}};
    for (int i = 1; i < argc; ++i) {
        const std::string key = argv[i];
        if (key == "--quiet") {
            o.quiet = true;
        } else if (i + 1 < argc && text.count(key)) {
            *text[key] = argv[++i];
        } else if (i + 1 < argc && key == "--max-iters") {
            o.max_iters = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        } else if (i + 1 < argc && key == "--reduction") {
            o.reduction = std::strtod(argv[++i], nullptr);
        } else if (i + 1 < argc && key == "--restart") {
            o.restart = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        } else {
            throw std::runtime_error("unknown or incomplete option " + key);
        }
    }
    return o;
}

std::shared_ptr<gko::Executor> executor_named(const std::string& name)
{
    if (name == "hip") return gko::HipExecutor::create(0, gko::OmpExecutor::create(), true);
    if (name == "omp") return gko::OmpExecutor::create();
    if (name == "reference") return gko::ReferenceExecutor::create();
    throw std::runtime_error("no executor called " + name);
}

std::shared_ptr<const gko::LinOp> preconditioner_for(const options& o, std::shared_ptr<gko::Executor> exec, std::shared_ptr<const csr> A)
{
    if (o.precond == "jacobi") return gko::preconditioner::Jacobi<double, int>::build().with_max_block_size(8u).on(exec)->generate(A);
    if (o.precond == "ilu") return gko::preconditioner::Ilu<>::build().with_factorization_iterations(5u).on(exec)->generate(A);
    if (o.precond != "none") throw std::runtime_error("no preconditioner called " + o.precond);
    return nullptr;
}

// every solver of the path takes the same factory parameters: one template serves them all
template <typename Solver, typename... Extra>
std::unique_ptr<gko::LinOp> build(const options& o, std::shared_ptr<gko::Executor> exec, std::shared_ptr<const csr> A,
                                  std::shared_ptr<const gko::LinOp> precond, long long* iterations, Extra&&...)
{
    auto factory = Solver::build().with_criteria(gko::stop::Iteration::build().with_max_iters(o.max_iters).on(exec),
                                                 gko::stop::ResidualNorm<double>::build().with_reduction_factor(o.reduction).on(exec));
    auto solver = precond ? factory.with_generated_preconditioner(precond).on(exec)->generate(A) : factory.on(exec)->generate(A);
    struct counted : gko::LinOp {
        std::unique_ptr<Solver> inner;
        long long* out;
        counted(std::unique_ptr<Solver> s, long long* o_) : gko::LinOp(s->get_executor(), s->get_size()), inner(std::move(s)), out(o_) {}
        void apply_impl(const gko::LinOp* b, gko::LinOp* x) const override
        {
            inner->apply(b, x);
            *out = inner->get_last_iteration_count();
        }
        void apply_impl(const gko::LinOp* alpha, const gko::LinOp* b, const gko::LinOp* beta, gko::LinOp* x) const override
        {
            inner->apply(alpha, b, beta, x);
            *out = inner->get_last_iteration_count();
        }
    };
    return std::unique_ptr<gko::LinOp>(new counted(std::move(solver), iterations));
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        const options o = parse(argc, argv);
        auto exec = executor_named(o.executor);
        auto A = gko::share(gko::read<csr>(std::ifstream(o.dir + "/A.mtx"), exec));
        auto b = gko::read<dense>(std::ifstream(o.dir + "/b.mtx"), exec);
        std::unique_ptr<dense> x;
        if (std::ifstream guess(o.dir + "/x0.mtx"); guess.good()) {
            x = gko::read<dense>(std::move(guess), exec);
        } else {
            x = dense::create(exec, b->get_size());
            x->fill(0.0);
        }
        auto precond = preconditioner_for(o, exec, A);
        long long iterations = -1;
        std::unique_ptr<gko::LinOp> solver;
        if (o.solver == "cg") solver = build<gko::solver::Cg<double>>(o, exec, A, precond, &iterations);
        else if (o.solver == "fcg") solver = build<gko::solver::Fcg<double>>(o, exec, A, precond, &iterations);
        else if (o.solver == "bicgstab") solver = build<gko::solver::Bicgstab<double>>(o, exec, A, precond, &iterations);
        else if (o.solver == "cgs") solver = build<gko::solver::Cgs<double>>(o, exec, A, precond, &iterations);
        else if (o.solver == "gmres") {
            auto factory = gko::solver::Gmres<double>::build()
                               .with_krylov_dim(o.restart)
                               .with_criteria(gko::stop::Iteration::build().with_max_iters(o.max_iters).on(exec),
                                              gko::stop::ResidualNorm<double>::build().with_reduction_factor(o.reduction).on(exec));
            auto g = precond ? factory.with_generated_preconditioner(precond).on(exec)->generate(A) : factory.on(exec)->generate(A);
            g->apply(gko::lend(b), gko::lend(x));
            iterations = g->get_last_iteration_count();
        } else {
            throw std::runtime_error("no solver called " + o.solver);
        }
        if (solver) solver->apply(gko::lend(b), gko::lend(x));
        // the true residual, by an apply of its own: r = b - A x
        auto r = b->clone();
        auto plus = gko::initialize<dense>({1.0}, exec), minus = gko::initialize<dense>({-1.0}, exec);
        A->apply(gko::lend(minus), gko::lend(x), gko::lend(plus), gko::lend(r));
        auto norm = gko::initialize<dense>({0.0}, exec);
        r->compute_norm2(gko::lend(norm));
        std::cout << "executor: " << o.executor << "\nsolver: " << o.solver << "\npreconditioner: " << o.precond << "\nrows: " << A->get_size()[0]
                  << "\niterations: " << iterations << "\ntrue residual norm: " << exec->copy_val_to_host(norm->get_const_values()) << "\n";
        if (!o.quiet) {
            std::cout << "x:\n";
            gko::write(std::cout, gko::lend(x));
        }
    } catch (const gko::NotCompiled& e) {
        std::cerr << "gko::NotCompiled: " << e.what() << std::endl;
        return 3;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
