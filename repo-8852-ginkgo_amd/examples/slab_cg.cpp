// slab_cg: row-partitioned CG on a 1-D / 2-D / 3-D Poisson problem, one process per GPU, through the
// gko::experimental::{mpi, distributed} part of the host mirror (RCCL underneath; no MPI in this build).
//
//   [RANK=r WORLD_SIZE=p LOCAL_RANK=r MASTER_ADDR=127.0.0.1 MASTER_PORT=port] slab_cg <hip|reference> <n> [dims]
//
// The global system has n^dims rows (dims = 1: the 3-point stencil, 2: 5-point, 3: 7-point), split into
// contiguous row slabs by Partition::build_from_global_size_uniform; every rank assembles ONLY the rows of
// its slab, straight into device memory (Matrix::read_distributed on the device executor builds the local
// and the non-local block and runs the two setup exchanges).  Right-hand side b_i = sin(0.01 i), x0 = 0,
// stop at ||r|| < 1e-8 (absolute) or 20 * rows iterations.  Prints one "key: value" line per figure.
#include <ginkgo/ginkgo.hpp>

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>

namespace {

namespace dist = gko::experimental::distributed;
using global_index = gko::int64;
using entries = gko::matrix_data<double, global_index>;

// rows [first, last) of the (2 dims + 1)-point Laplacian on an n^dims grid, row = lexicographic index
void assemble_slab(global_index n, int dims, global_index first, global_index last, entries& A, entries& b)
{
    global_index stride[3] = {1, n, n * n};
    for (global_index row = first; row < last; ++row) {
        global_index coordinate[3];
        for (int d = 0; d < dims; ++d) coordinate[d] = (row / stride[d]) % n;
        for (int d = dims - 1; d >= 0; --d) {
            if (coordinate[d] > 0) A.nonzeros.emplace_back(row, row - stride[d], -1.0);
        }
        A.nonzeros.emplace_back(row, row, 2.0 * dims);
        for (int d = 0; d < dims; ++d) {
            if (coordinate[d] + 1 < n) A.nonzeros.emplace_back(row, row + stride[d], -1.0);
        }
        b.nonzeros.emplace_back(row, 0, std::sin(0.01 * static_cast<double>(row)));
    }
}

double global_norm(const dist::Vector<double>* v, std::shared_ptr<gko::Executor> exec)
{
    auto out = gko::initialize<gko::matrix::Dense<double>>({0.0}, exec->get_master());
    v->compute_norm2(gko::lend(out));
    return *out->get_values();
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        const std::string where = argc > 1 ? argv[1] : "hip";
        const global_index n = argc > 2 ? std::atoll(argv[2]) : 100;
        const int dims = argc > 3 ? std::atoi(argv[3]) : 1;
        if (n < 2 || dims < 1 || dims > 3) throw std::runtime_error("usage: slab_cg <hip|reference> <n >= 2> [dims 1..3]");
        global_index rows = 1;
        for (int d = 0; d < dims; ++d) rows *= n;

        const gko::experimental::mpi::environment env(argc, argv);
        const gko::experimental::mpi::communicator comm{MPI_COMM_WORLD};
        std::shared_ptr<gko::Executor> exec = gko::ReferenceExecutor::create();
        if (where == "hip") {
            const int device = gko::experimental::mpi::map_rank_to_device_id(MPI_COMM_WORLD, gko::HipExecutor::get_num_devices());
            exec = gko::HipExecutor::create(device, gko::ReferenceExecutor::create(), true);
        }
        auto slabs = gko::share(dist::Partition<gko::int32, global_index>::build_from_global_size_uniform(exec->get_master(), comm.size(), rows));
        const global_index first = slabs->get_range_bounds()[comm.rank()], last = slabs->get_range_bounds()[comm.rank() + 1];

        entries A_entries, b_entries;
        A_entries.size = {static_cast<gko::size_type>(rows), static_cast<gko::size_type>(rows)};
        b_entries.size = {static_cast<gko::size_type>(rows), 1};
        assemble_slab(n, dims, first, last, A_entries, b_entries);

        auto A = gko::share(dist::Matrix<double, gko::int32, global_index>::create(exec, comm));
        auto b = dist::Vector<double>::create(exec, comm);
        auto x = dist::Vector<double>::create(exec, comm);
        A->read_distributed(A_entries, slabs.get());
        b->read_distributed(b_entries, slabs.get());
        x->read_distributed(entries{b_entries.size}, slabs.get());  // no entries: x0 = 0

        auto cg = gko::solver::Cg<double>::build()
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(static_cast<gko::size_type>(20 * rows)).on(exec),
                                     gko::stop::ResidualNorm<double>::build().with_baseline(gko::stop::mode::absolute).with_reduction_factor(1e-8).on(exec))
                      .on(exec)
                      ->generate(A);
        comm.synchronize();
        const double started = gko::experimental::mpi::get_walltime();
        cg->apply(gko::lend(b), gko::lend(x));
        comm.synchronize();
        const double seconds = gko::experimental::mpi::get_walltime() - started;

        // the true residual r = b - A x with a distributed apply of its own, and the norm of the solution
        auto r = b->clone();
        auto plus = gko::initialize<gko::matrix::Dense<double>>({1.0}, exec), minus = gko::initialize<gko::matrix::Dense<double>>({-1.0}, exec);
        A->apply(gko::lend(minus), gko::lend(x), gko::lend(plus), gko::lend(r));
        const double residual = global_norm(r.get(), exec), solution = global_norm(x.get(), exec);
        if (comm.rank() == 0) {
            std::cout << "global rows: " << rows << "\nstencil points: " << 2 * dims + 1 << "\nranks: " << comm.size()
                      << "\nrank 0 rows: " << A->get_num_local_rows() << "\nrank 0 halo in: " << A->get_num_halo_entries()
                      << "\nrank 0 halo out: " << A->get_num_send_entries() << "\niterations: " << cg->get_last_iteration_count()
                      << "\nconverged: " << (cg->has_converged() ? "yes" : "no") << "\ntrue residual norm: " << residual
                      << "\nsolution norm: " << solution << "\nsolve seconds: " << seconds << std::endl;
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
