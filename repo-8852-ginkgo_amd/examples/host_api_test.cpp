// CPU-only checks of the gko:: mirror: memory spaces, arrays, MatrixMarket
// reading, dimension validation, and that kernels on host executors raise
// gko::NotCompiled (a build without the reference/omp modules).
#include <ginkgo/ginkgo.hpp>

#include <iostream>
#include <sstream>

#define CHECK(cond) do { if (!(cond)) { std::cerr << "FAILED: " #cond " at line " << __LINE__ << "\n"; return 1; } } while (0)

int main()
{
    using vec = gko::matrix::Dense<double>;
    using csr = gko::matrix::Csr<double, int>;
    auto ref = gko::ReferenceExecutor::create();
    auto omp = gko::OmpExecutor::create();
    CHECK(ref->get_master() == ref);
    gko::array<int> a(ref, {1, 2, 3});
    gko::array<int> c(omp, a);
    CHECK(c.get_num_elems() == 3 && c.get_const_data()[2] == 3);
    // set_executor moves the data into the other memory space and relabels it
    gko::array<int> moved(ref, {7, 8});
    moved.set_executor(omp);
    CHECK(moved.get_executor() == omp && moved.get_const_data()[1] == 8);
    std::istringstream mm("%%MatrixMarket matrix coordinate real symmetric\n% c\n3 3 4\n1 1 2.0\n2 1 -1.0\n2 2 2.0\n3 3 5.0\n");
    auto A = gko::share(gko::read<csr>(mm, ref));
    CHECK(A->get_size() == gko::dim<2>(3, 3) && A->get_num_stored_elements() == 5);
    CHECK(A->get_const_row_ptrs()[1] == 2 && A->get_const_col_idxs()[1] == 1 && A->get_const_values()[1] == -1.0);
    CHECK(A->get_strategy()->get_name() == "automatical");
    std::istringstream arr("%%MatrixMarket matrix array real general\n3 1\n1\n2\n3\n");
    auto b = gko::read<vec>(arr, ref);
    CHECK(b->get_size() == gko::dim<2>(3, 1) && b->at(2) == 3.0);
    auto x = vec::create(ref, gko::dim<2>(3, 1));
    bool not_compiled = false;
    try { A->apply(b.get(), x.get()); } catch (const gko::NotCompiled&) { not_compiled = true; }
    CHECK(not_compiled);
    bool mismatch = false;
    try { auto bad = vec::create(ref, gko::dim<2>(4, 1)); A->apply(bad.get(), x.get()); } catch (const gko::DimensionMismatch&) { mismatch = true; }
    CHECK(mismatch);
    bool cuda_missing = false;
    try { gko::CudaExecutor::create(0, omp); } catch (const gko::NotCompiled&) { cuda_missing = true; }
    CHECK(cuda_missing);
    std::ostringstream os;
    gko::write(os, b.get());
    CHECK(os.str().find("3 1") != std::string::npos);
    auto solver = gko::solver::Cg<double>::build()
                      .with_criteria(gko::stop::Iteration::build().with_max_iters(5u).on(ref))
                      .on(ref)->generate(A);
    bool solver_not_compiled = false;
    try { solver->apply(b.get(), x.get()); } catch (const gko::NotCompiled&) { solver_not_compiled = true; }
    CHECK(solver_not_compiled);
    // matrix_data helpers and device_matrix_data on a host memory space
    gko::matrix_data<double, int> md;
    md.size = gko::dim<2>(3, 3);
    md.nonzeros = {{2, 1, 1.0}, {0, 0, 2.0}, {2, 1, 3.0}, {1, 1, 0.0}};
    auto dd = gko::device_matrix_data<double, int>::create_from_host(ref, md);
    CHECK(dd.get_num_elems() == 4 && dd.get_const_row_idxs()[0] == 2 && dd.copy_to_host().nonzeros[2].value == 3.0);
    bool sort_not_compiled = false;
    try { dd.sort_row_major(); } catch (const gko::NotCompiled&) { sort_not_compiled = true; }
    CHECK(sort_not_compiled);
    md.remove_zeros();
    md.sum_duplicates();
    CHECK(md.nonzeros.size() == 2 && md.nonzeros[0].row == 0 && md.nonzeros[1].value == 4.0);
    // binary matrix format (core/base/mtx_io.cpp:768-960): round trip, header layout, generic reader
    {
        std::ostringstream bin;
        gko::write_binary(bin, A.get());
        const std::string bytes = bin.str();
        CHECK(bytes.size() == 32 + 16 * A->get_num_stored_elements() && bytes.substr(0, 8) == "GINKGODI");
        std::istringstream in(bytes);
        auto A2 = gko::read_binary<csr>(in, ref);
        CHECK(A2->get_num_stored_elements() == A->get_num_stored_elements());
        for (gko::size_type k = 0; k < A->get_num_stored_elements(); ++k)
            CHECK(A2->get_const_col_idxs()[k] == A->get_const_col_idxs()[k] && A2->get_const_values()[k] == A->get_const_values()[k]);
        std::istringstream in2(bytes);
        auto A3 = gko::read_generic<csr>(in2, ref);
        CHECK(A3->get_size() == A->get_size());
        std::istringstream mm2("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 2 3.5\n");
        auto A4 = gko::read_generic<csr>(mm2, ref);
        CHECK(A4->get_num_stored_elements() == 1 && A4->get_const_values()[0] == 3.5);
        bool bad_magic = false;
        std::istringstream junk(std::string(64, 'x'));
        try { gko::read_binary<csr>(junk, ref); } catch (const gko::StreamError&) { bad_magic = true; }
        CHECK(bad_magic);
        // a float / int64 file converts on the way in
        std::string f;
        const char magic[8] = {'G', 'I', 'N', 'K', 'G', 'O', 'S', 'L'};
        f.append(magic, 8);
        const std::uint64_t hdr[3] = {2, 2, 1};
        f.append(reinterpret_cast<const char*>(hdr), 24);
        const std::int64_t rc[2] = {1, 0};
        const float fv = 2.5f;
        f.append(reinterpret_cast<const char*>(rc), 16);
        f.append(reinterpret_cast<const char*>(&fv), 4);
        std::istringstream fin(f);
        auto A5 = gko::read_binary<csr>(fin, ref);
        CHECK(A5->get_num_stored_elements() == 1 && A5->get_const_row_ptrs()[1] == 0 && A5->get_const_values()[0] == 2.5);
    }
    {
        // log::Logger on an executor: operation_launched / completed around every C-ABI operation
        // (executor.hpp:1153-1158); stop::Combined; the distributed Partition (host metadata)
        struct counter : gko::log::Logger {
            mutable int launched = 0, completed = 0;
            mutable std::string last;
            void on_operation_launched(const gko::Executor*, const char* op) const override { ++launched; last = op; }
            void on_operation_completed(const gko::Executor*, const char*) const override { ++completed; }
        };
        auto logger = std::make_shared<counter>();
        ref->add_logger(logger);
        using part = gko::experimental::distributed::Partition<gko::int32, gko::int64>;
        auto p = part::build_from_global_size_uniform(ref, 3, 10);
        CHECK(logger->launched == 3 && logger->completed == 3);
        CHECK(logger->last == "gkomi_partition_build_starting_indices");
        ref->remove_logger(logger.get());
        auto p2 = part::build_from_global_size_uniform(ref, 2, 7);
        CHECK(logger->launched == 3);
        CHECK(p->get_num_parts() == 3 && p->get_size() == 10);
        CHECK(p->get_range_bounds()[0] == 0 && p->get_range_bounds()[1] == 4 && p->get_range_bounds()[2] == 7 && p->get_range_bounds()[3] == 10);
        CHECK(p->get_part_size(0) == 4 && p->get_part_size(2) == 3 && p2->get_part_size(1) == 3);
        // reference/test/distributed/partition_kernels.cpp:88-135, 226-295: mapping with empty parts, connected / ordered
        {
            auto pm = part::build_from_mapping(ref, gko::array<int>(ref, {3, 3, 0, 1, 1, 3, 0, 0, 1, 0, 1, 1, 1, 3, 3, 0}), 5);
            CHECK(pm->get_num_ranges() == 10 && pm->get_num_parts() == 5 && pm->get_num_empty_parts() == 2 && pm->get_size() == 16);
            const gko::int32 sizes[5] = {5, 6, 0, 5, 0}, starts[10] = {0, 0, 0, 2, 1, 2, 3, 3, 3, 4};
            for (int i = 0; i < 5; ++i) CHECK(pm->get_part_sizes()[i] == sizes[i]);
            for (int i = 0; i < 10; ++i) CHECK(pm->get_range_starting_indices()[i] == starts[i]);
            CHECK(!pm->has_connected_parts() && !pm->has_ordered_parts());
            auto unordered = part::build_from_mapping(ref, gko::array<int>(ref, {1, 1, 0, 0, 2}), 3);
            CHECK(unordered->has_connected_parts() && !unordered->has_ordered_parts());
            auto ordered = part::build_from_mapping(ref, gko::array<int>(ref, {0, 2, 2, 5, 5}), 6);
            CHECK(ordered->has_connected_parts() && ordered->has_ordered_parts());
            auto ranges = part::build_from_contiguous(ref, std::vector<int64_t>{0, 5, 5, 7, 9, 10});
            // (an empty part still has its range here: 5 - 1 != 5 ranges, "not connected" by partition.cpp:120-124)
            CHECK(ranges->get_num_empty_parts() == 1 && ranges->get_part_size(2) == 2 && !ranges->has_connected_parts());
        }
        gko::stop::criterion_settings st;
        gko::stop::Combined::build()
            .with_criteria(gko::stop::Iteration::build().with_max_iters(17u).on(ref),
                           gko::stop::ImplicitResidualNorm<double>::build().with_reduction_factor(1e-7).on(ref))
            .on(ref)
            ->contribute(st);
        CHECK(st.max_iters == 17 && st.reduction_factor == 1e-7 && st.implicit);
        gko::preconditioner::block_interleaved_storage_scheme<gko::int32> sch{32, 64, 1};   // max_bs 32 at stride 64
        CHECK(sch.get_group_size() == 2 && sch.get_stride() == 64 && sch.compute_storage_space(5) == 3 * 64);
        CHECK(sch.get_global_block_offset(3) == 64 + 32);
    }
    std::cout << "host api ok\n";
    return 0;
}
