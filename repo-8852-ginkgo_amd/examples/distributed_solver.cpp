// The flow of the reference's examples/distributed-solver (row-partitioned CG on a
// 1-D 3-pt stencil) against the host mirror, with the final residual computed on the
// device (the mirror's host executors carry no kernels).  One process per GPU:
//   RANK=r WORLD_SIZE=n LOCAL_RANK=r MASTER_ADDR=127.0.0.1 MASTER_PORT=p ./distributed_solver hip <rows>
// or, alone, ./distributed_solver hip <rows>.
#include <ginkgo/ginkgo.hpp>

#include <cmath>
#include <iostream>
#include <string>

int main(int argc, char* argv[])
{
    using GlobalIndexType = gko::int64;
    using LocalIndexType = gko::int32;
    using ValueType = double;
    using dist_vec = gko::experimental::distributed::Vector<ValueType>;
    using dist_mtx = gko::experimental::distributed::Matrix<ValueType, LocalIndexType, GlobalIndexType>;
    using vec = gko::matrix::Dense<ValueType>;
    using part_type = gko::experimental::distributed::Partition<LocalIndexType, GlobalIndexType>;
    using solver = gko::solver::Cg<ValueType>;
    try {
        const gko::experimental::mpi::environment env(argc, argv);
        const gko::experimental::mpi::communicator comm{MPI_COMM_WORLD};
        const auto rank = comm.rank();
        const std::string executor_string = argc >= 2 ? argv[1] : "hip";
        const auto num_rows = static_cast<gko::size_type>(argc >= 3 ? std::atoi(argv[2]) : 100);
        std::shared_ptr<gko::Executor> exec;
        if (executor_string == "hip") {
            exec = gko::HipExecutor::create(gko::experimental::mpi::map_rank_to_device_id(MPI_COMM_WORLD, gko::HipExecutor::get_num_devices()),
                                            gko::ReferenceExecutor::create(), true);
        } else {
            exec = gko::ReferenceExecutor::create();
        }
        auto partition = gko::share(part_type::build_from_global_size_uniform(exec->get_master(), comm.size(), static_cast<GlobalIndexType>(num_rows)));
        gko::matrix_data<ValueType, GlobalIndexType> A_data, b_data, x_data;
        A_data.size = {num_rows, num_rows};
        b_data.size = {num_rows, 1};
        x_data.size = {num_rows, 1};
        const auto range_start = partition->get_range_bounds()[rank];
        const auto range_end = partition->get_range_bounds()[rank + 1];
        for (GlobalIndexType i = range_start; i < range_end; i++) {
            if (i > 0) A_data.nonzeros.emplace_back(i, i - 1, -1);
            A_data.nonzeros.emplace_back(i, i, 2);
            if (i < static_cast<GlobalIndexType>(num_rows) - 1) A_data.nonzeros.emplace_back(i, i + 1, -1);
            b_data.nonzeros.emplace_back(i, 0, std::sin(i * 0.01));
            x_data.nonzeros.emplace_back(i, 0, gko::zero<ValueType>());
        }
        auto A_host = gko::share(dist_mtx::create(exec->get_master(), comm));
        auto x_host = dist_vec::create(exec->get_master(), comm);
        auto b_host = dist_vec::create(exec->get_master(), comm);
        A_host->read_distributed(A_data, partition.get());
        b_host->read_distributed(b_data, partition.get());
        x_host->read_distributed(x_data, partition.get());
        auto A = gko::share(dist_mtx::create(exec, comm));
        auto x = dist_vec::create(exec, comm);
        auto b = dist_vec::create(exec, comm);
        A->copy_from(A_host.get());
        b->copy_from(b_host.get());
        x->copy_from(x_host.get());
        comm.synchronize();
        const double t0 = gko::experimental::mpi::get_walltime();
        auto Ainv = solver::build()
                        .with_criteria(gko::stop::Iteration::build().with_max_iters(static_cast<gko::size_type>(20 * num_rows)).on(exec),
                                       gko::stop::ResidualNorm<ValueType>::build().with_baseline(gko::stop::mode::absolute).with_reduction_factor(1e-8).on(exec))
                        .on(exec)
                        ->generate(A);
        Ainv->apply(gko::lend(b), gko::lend(x));
        comm.synchronize();
        const double t1 = gko::experimental::mpi::get_walltime();
        // residual b - A x on the device
        auto one = gko::initialize<vec>({1.0}, exec);
        auto minus_one = gko::initialize<vec>({-1.0}, exec);
        A->apply(gko::lend(minus_one), gko::lend(x), gko::lend(one), gko::lend(b));
        auto res_norm = gko::initialize<vec>({0.0}, exec->get_master());
        b->compute_norm2(gko::lend(res_norm));
        auto x_norm = gko::initialize<vec>({0.0}, exec->get_master());
        x->compute_norm2(gko::lend(x_norm));
        if (rank == 0) {
            std::cout << "Num rows in matrix: " << num_rows << "\nNum ranks: " << comm.size()
                      << "\nLocal rows / halo in / halo out on rank 0: " << A->get_num_local_rows() << " " << A->get_num_halo_entries() << " "
                      << A->get_num_send_entries() << "\nIterations: " << Ainv->get_last_iteration_count()
                      << "\nConverged: " << (Ainv->has_converged() ? 1 : 0) << "\nFinal Res norm: " << *res_norm->get_values()
                      << "\nSolution norm: " << *x_norm->get_values() << "\nSolver apply time: " << t1 - t0 << std::endl;
        }
    } catch (const gko::NotCompiled& e) {
        std::cerr << e.what() << std::endl;
        return 3;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
