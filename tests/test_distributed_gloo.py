"""Multi-process (gloo, CPU) tests of the distributed layer's communication
logic: read_distributed's two setup exchanges, the halo all-to-all-v in apply,
the all-reduced dots/norms and a row-partitioned CG, world sizes 2 and 3 --
the reference's strategy (test/mpi/distributed/matrix.cpp runs 3 local ranks).
Compute runs on the ORACLE through dist_ops_cpu.OracleOps; the GPU kernels
behind the same calls are covered by tests/test_distributed_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, grid, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "repo-8852-ginkgo_amd")]
    import gkomi
    import gkomi.distributed as gd
    import matgen
    import oracle_lib
    from dist_ops_cpu import OracleOps
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gk = gkomi.lib()  # host-side Partition helpers only; no GPU call
        oracle = oracle_lib.load()
        rows, cols, vals, n_global = gd.poisson_slab_rows(grid, rank, world)
        part = gd.Partition.build_from_global_size_uniform(gk, world, n_global)
        A = gd.Matrix(OracleOps(oracle)).read_distributed(rows, cols, vals, part)
        n_local = A.num_local_rows
        lo = int(part.range_bounds[rank])
        xg = np.sin(0.01 * np.arange(n_global)).reshape(n_global, 1)
        x = torch.from_numpy(xg[lo:lo + n_local].copy())
        y = torch.zeros((n_local, 1), dtype=torch.float64)
        A.apply(x, y)
        # global reference on every rank
        ng, rp, ci, v = matgen.poisson_2d_5pt(grid * world, grid)
        assert ng == n_global
        ye = np.zeros((n_global, 1))
        oracle.ref_csr_spmv(n_global, 1, rp, ci, v, xg, 1, ye, 1)
        # rows whose stencil crosses the slab boundary add the halo term last
        # (local SpMV, then non-local advanced SpMV): equal to rounding there,
        # bit-identical in the interior
        got, exp = y.numpy(), ye[lo:lo + n_local]
        assert matgen.rel_err(got, exp) <= 1e-15, "distributed apply != global SpMV"
        assert np.array_equal(got[grid:-grid], exp[grid:-grid])
        # neighbours only: the communication plan of a slab partition
        expect_recv = [grid if abs(p - rank) == 1 else 0 for p in range(world)]
        assert A.recv_sizes == expect_recv and A.send_sizes == expect_recv
        # reductions
        vec = gd.VectorOps(A.ops, n_local, 1)
        d = torch.zeros(1, dtype=torch.float64)
        vec.dot(x, y, d)
        assert abs(d.item() - float((xg * ye).sum())) <= 1e-12 * abs(float((np.abs(xg * ye)).sum()))
        vec.norm2(y, d)
        assert abs(d.item() - np.linalg.norm(ye)) <= 1e-12 * np.linalg.norm(ye)
        # two right-hand sides through the same plan
        x2 = torch.from_numpy(np.concatenate([xg, 2 * xg + 1], axis=1)[lo:lo + n_local].copy())
        y2 = torch.zeros((n_local, 2), dtype=torch.float64)
        A.apply(x2, y2)
        assert matgen.rel_err(y2.numpy()[:, 0], ye[lo:lo + n_local, 0]) <= 1e-15
        # distributed CG == serial CG (same tolerances as the serial test,
        # test/mpi/solver/solver.cpp:492-560)
        bg = np.ones((n_global, 1))
        b = torch.from_numpy(bg[lo:lo + n_local].copy())
        xs = torch.zeros((n_local, 1), dtype=torch.float64)
        it, conv = gd.cg(A, b, xs, max_iters=2000, reduction=1e-10)
        xe = np.zeros(n_global)
        ite = oracle.ref_cg_solve(n_global, rp, ci, v, bg[:, 0].copy(), xe, 2000, 1e-10, 0, None, 0)
        assert conv and abs(it - ite) <= 1, (it, ite)
        assert matgen.rel_err(xs.numpy()[:, 0], xe[lo:lo + n_local]) <= 1e-6
        # the host may look at the criterion as rarely as it likes: same iterates, same count
        for every in (1, 5):
            xs2 = torch.zeros((n_local, 1), dtype=torch.float64)
            it2, conv2 = gd.cg(A, b, xs2, max_iters=2000, reduction=1e-10, check_every=every)
            assert (it2, conv2) == (it, conv) and torch.equal(xs2, xs)
        it3, conv3 = gd.cg(A, b, torch.zeros((n_local, 1), dtype=torch.float64), max_iters=3, reduction=1e-10)
        assert (it3, conv3) == (3, False)
        # the fused driver's communication schedule (rho and tau^2 in ONE all-reduce, beta in a
        # second one): the same solve, the same count for any polling interval
        xf = torch.zeros((n_local, 1), dtype=torch.float64)
        itf, convf = gd.cg_fused(A, b, xf, max_iters=2000, reduction=1e-10)
        assert convf and abs(itf - ite) <= 1, (itf, ite)
        assert matgen.rel_err(xf.numpy()[:, 0], xe[lo:lo + n_local]) <= 1e-6
        for every in (1, 7):
            xf2 = torch.zeros((n_local, 1), dtype=torch.float64)
            assert gd.cg_fused(A, b, xf2, max_iters=2000, reduction=1e-10, check_every=every) == (itf, convf)
            assert torch.equal(xf2, xf)
        assert gd.cg_fused(A, b, torch.zeros((n_local, 1), dtype=torch.float64), max_iters=3, reduction=1e-10) == (3, False)
        b2 = torch.from_numpy(np.concatenate([bg, np.cos(np.arange(n_global)).reshape(-1, 1)], axis=1)[lo:lo + n_local].copy())
        x2f, x2r = torch.zeros((n_local, 2), dtype=torch.float64), torch.zeros((n_local, 2), dtype=torch.float64)
        assert gd.cg_fused(A, b2, x2f, max_iters=2000, reduction=1e-10)[1] and gd.cg(A, b2, x2r, max_iters=2000, reduction=1e-10)[1]
        assert matgen.rel_err(x2f.numpy(), x2r.numpy()) <= 1e-8
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid", [(2, 12), (3, 7)])
def test_row_partitioned_apply_and_cg(tmp_path, world, grid):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, grid, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _worker_decoupled(rank, world, port, out_dir):
    """A block-diagonal matrix split so that the last rank has no off-rank coupling at all while
    the others do: every rank must still issue the same collectives (the halo
    exchange used to be skipped per rank, which hangs gloo)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "repo-8852-ginkgo_amd")]
    import gkomi
    import gkomi.distributed as gd
    import matgen
    import oracle_lib
    from dist_ops_cpu import OracleOps
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gk = gkomi.lib()
        oracle = oracle_lib.load()
        # global matrix = diag(P, Q): P = 1-D Poisson on the rows of ranks 0..world-2 (coupled across
        # their boundaries), Q = 1-D Poisson on the rows of the last rank alone
        m = 10
        n_global = m * world
        cut = m * (world - 1)
        rows, cols, vals = [], [], []
        for i in range(n_global):
            for j, val in ((i - 1, -1.0), (i, 2.0), (i + 1, -1.0)):
                if 0 <= j < n_global and (i < cut) == (j < cut):
                    rows.append(i), cols.append(j), vals.append(val)
        rows, cols, vals = np.array(rows, np.int64), np.array(cols, np.int64), np.array(vals)
        part = gd.Partition.build_from_global_size_uniform(gk, world, n_global)
        lo, hi = int(part.range_bounds[rank]), int(part.range_bounds[rank + 1])
        mine = (rows >= lo) & (rows < hi)
        A = gd.Matrix(OracleOps(oracle)).read_distributed(rows[mine], cols[mine], vals[mine], part)
        assert A._any_halo
        if rank == world - 1:
            assert A.send_count == 0 and A.recv_count == 0       # decoupled, yet it takes part
        xg = np.sin(0.3 * np.arange(n_global)).reshape(n_global, 1)
        y = torch.zeros((hi - lo, 1), dtype=torch.float64)
        A.apply(torch.from_numpy(xg[lo:hi].copy()), y)
        rp, ci, v = matgen.coo_to_csr(n_global, rows.astype(np.int32), cols.astype(np.int32), vals)
        ye = np.zeros((n_global, 1))
        oracle.ref_csr_spmv(n_global, 1, rp, ci, v, xg, 1, ye, 1)
        assert matgen.rel_err(y.numpy(), ye[lo:hi]) <= 1e-15
        b = torch.ones((hi - lo, 1), dtype=torch.float64)
        for solver in (gd.cg, gd.cg_fused):
            xs = torch.zeros((hi - lo, 1), dtype=torch.float64)
            it, conv = solver(A, b, xs, max_iters=500, reduction=1e-10)
            xe = np.zeros(n_global)
            ite = oracle.ref_cg_solve(n_global, rp, ci, v, np.ones(n_global), xe, 500, 1e-10, 0, None, 0)
            assert conv and abs(it - ite) <= 1
            assert matgen.rel_err(xs.numpy()[:, 0], xe[lo:hi]) <= 1e-6
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_rank_without_neighbours_still_takes_part(tmp_path):
    world = 3
    port = _free_port()
    mp.spawn(_worker_decoupled, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
