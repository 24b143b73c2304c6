"""GPU tests of the NATIVE distributed path (csrc/comm.hip, csrc/dist_cg.hip):
gkomi_dist_matrix_apply_f64 / gkomi_dist_cg_solve_f64 over a gkomi_comm.

  * RCCL transport, world size 1 (all a one-GPU box can do with RCCL);
  * world sizes 2, 3 and 8 with the ranks as THREADS of this process over the
    loopback communicator of tests/native (device-to-device copies + a host
    barrier): the halo plan, pack / unpack offsets, the two all-reduces per
    iteration and the iteration control run exactly as they do over RCCL, only
    the wire differs.  Reference shape: test/mpi/distributed/matrix.cpp:212-245,
    test/mpi/solver/solver.cpp:492-560 (distributed == serial at the serial tolerances).
"""
import ctypes
import os
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist

import gkomi.distributed as gd
import matgen
from gpu_util import dev, host

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def loopback_lib():
    path = os.path.join(HERE, "native", "libloopback_comm.so")
    assert os.path.exists(path), "build tests/native (make -C tests/native, or __graft_entry__.build())"
    lib = ctypes.CDLL(path)
    lib.loopback_world_create.restype = ctypes.c_void_p
    lib.loopback_world_create.argtypes = [ctypes.c_int]
    lib.loopback_world_destroy.argtypes = [ctypes.c_void_p]
    lib.loopback_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.loopback_comm_free.argtypes = [ctypes.c_void_p]
    return lib


class LoopbackComm:
    def __init__(self, lib, world_handle, rank, size):
        self.lib = lib
        self.struct = gd.CommStruct()
        assert lib.loopback_comm_init(world_handle, rank, ctypes.addressof(self.struct)) == 0
        self.ptr = ctypes.addressof(self.struct)
        self.rank, self.size = rank, size

    def close(self):
        self.lib.loopback_comm_free(self.ptr)


def build_parts(gk, world, n_global, rp, ci, v):
    """What read_distributed produces on every rank, assembled in one process: the device
    build_local_nonlocal per part, then the two setup exchanges of matrix.cpp:198-224 by hand."""
    ops = gd.GpuOps(gk, "cuda:0")
    part = gd.Partition.build_from_global_size_uniform(gk, world, n_global)
    outs = []
    for r in range(world):
        lo, hi = int(part.range_bounds[r]), int(part.range_bounds[r + 1])
        a, e = int(rp[lo]), int(rp[hi])
        rows = np.repeat(np.arange(lo, hi, dtype=np.int64), np.diff(rp[lo:hi + 1]))
        outs.append(ops.build_local_nonlocal(ops.tensor(rows), ops.tensor(ci[a:e].astype(np.int64)),
                                             ops.tensor(v[a:e]), part, part, r))
    parts = []
    for r in range(world):
        o = outs[r]
        n_loc = int(part.part_sizes[r])
        nl, nn, nu = o["num_local"], o["num_non_local"], o["num_unique"]
        local = (n_loc, n_loc, nl, ops.coo_to_csr(n_loc, o["l_rows"], nl), o["l_cols"], o["l_vals"])
        non_local = (n_loc, nu, nn, ops.coo_to_csr(n_loc, o["nl_rows"], nn), o["nl_cols"], o["nl_vals"])
        recv_sizes = [int(x) for x in host(o["recv_sizes"])]
        # what rank r must SEND to p = what p receives from r: p's gather indices of owner r
        send_sizes, gathers = [], []
        for p in range(world):
            rs = [int(x) for x in host(outs[p]["recv_sizes"])]
            off = sum(rs[:r])
            send_sizes.append(rs[r])
            gathers.append(host(outs[p]["gather_idxs"])[off:off + rs[r]])
        gi = ops.tensor(np.concatenate(gathers).astype(np.int32) if sum(send_sizes) else np.zeros(1, np.int32))
        parts.append((ops, n_loc, local, non_local, send_sizes, recv_sizes, gi))
    return part, parts


def run_ranks(world, fn):
    """fn(rank) on `world` threads, each on its own HIP stream; re-raises the first failure."""
    errors = [None] * world

    def body(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                fn(r)
                torch.cuda.synchronize()
        except BaseException as e:  # noqa: BLE001 - reported below
            errors[r] = e
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    for e in errors:
        if e is not None:
            raise e


@pytest.mark.parametrize("world,shape", [(2, (40, 30)), (3, (33, 17)), (8, (64, 20))])
def test_native_apply_and_cg_over_loopback_ranks(gk, oracle, world, shape):
    n, rp, ci, v = matgen.poisson_2d_5pt(*shape)
    xg = np.sin(0.01 * np.arange(n)).reshape(n, 1)
    ye = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, xg, 1, ye, 1)
    bg = np.ones(n)
    xe = np.zeros(n)
    ite = oracle.ref_cg_solve(n, rp, ci, v, bg, xe, 3000, 1e-10, 0, None, 0)
    part, parts = build_parts(gk, world, n, rp, ci, v)
    lib = loopback_lib()
    wh = lib.loopback_world_create(world)
    results = [None] * world

    def rank_body(r):
        comm = LoopbackComm(lib, wh, r, world)
        A = gd.NativeMatrix.from_parts(*parts[r])
        lo, hi = int(part.range_bounds[r]), int(part.range_bounds[r + 1])
        y = torch.full((hi - lo, 1), float("nan"), dtype=torch.float64, device="cuda:0")
        A.apply(comm, dev(xg[lo:hi]), y)
        torch.cuda.current_stream().synchronize()
        xs = torch.zeros((hi - lo, 1), dtype=torch.float64, device="cuda:0")
        res = A.cg(comm, dev(bg[lo:hi].reshape(-1, 1)), xs, max_iters=3000, reduction=1e-10, check_every=4)
        few = A.cg(comm, dev(bg[lo:hi].reshape(-1, 1)), torch.zeros_like(xs), max_iters=3, reduction=1e-10)
        results[r] = (host(y), host(xs), res, few)
        A.close()
        comm.close()

    try:
        run_ranks(world, rank_body)
    finally:
        lib.loopback_world_destroy(wh)
    width = shape[1]
    for r in range(world):
        lo, hi = int(part.range_bounds[r]), int(part.range_bounds[r + 1])
        y, xs, res, few = results[r]
        assert matgen.rel_err(y, ye[lo:hi]) <= 1e-15                       # apply == global SpMV
        inner = slice(width, hi - lo - width)
        assert np.array_equal(y[inner], ye[lo:hi][inner])                  # bit-identical away from the cuts
        assert res["converged"] and abs(res["iterations"] - ite) <= 1, (res, ite)
        assert res == results[0][2]                                        # every rank reports the same solve
        assert matgen.rel_err(xs[:, 0], xe[lo:hi]) <= 1e-6
        assert few["iterations"] == 3 and not few["converged"]


def test_native_loopback_rank_without_neighbours(gk, oracle):
    """diag(P, Q) over 3 ranks: the last rank has nothing to exchange, the plan still works."""
    world, m = 3, 50
    n = world * m
    cut = m * (world - 1)
    rows, cols, vals = [], [], []
    for i in range(n):
        for j, val in ((i - 1, -1.0), (i, 2.0), (i + 1, -1.0)):
            if 0 <= j < n and (i < cut) == (j < cut):
                rows.append(i), cols.append(j), vals.append(val)
    rp, ci, v = matgen.coo_to_csr(n, np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals))
    xe = np.zeros(n)
    ite = oracle.ref_cg_solve(n, rp, ci, v, np.ones(n), xe, 2000, 1e-10, 0, None, 0)
    part, parts = build_parts(gk, world, n, rp, ci, v)
    assert sum(parts[2][4]) == 0 and sum(parts[2][5]) == 0
    lib = loopback_lib()
    wh = lib.loopback_world_create(world)
    out = [None] * world

    def rank_body(r):
        comm = LoopbackComm(lib, wh, r, world)
        A = gd.NativeMatrix.from_parts(*parts[r])
        xs = torch.zeros((m, 1), dtype=torch.float64, device="cuda:0")
        res = A.cg(comm, torch.ones((m, 1), dtype=torch.float64, device="cuda:0"), xs, max_iters=2000, reduction=1e-10)
        out[r] = (host(xs), res)
        A.close()
        comm.close()

    try:
        run_ranks(world, rank_body)
    finally:
        lib.loopback_world_destroy(wh)
    for r in range(world):
        assert out[r][1]["converged"] and abs(out[r][1]["iterations"] - ite) <= 1
        assert matgen.rel_err(out[r][0][:, 0], xe[r * m:(r + 1) * m]) <= 1e-6


def test_native_path_over_rccl_world_size_one(gk, oracle):
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29579")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        grid = 96
        M = gd.poisson_slab_matrix(gk, grid, 0, 1, "cuda:0")
        comm = gd.RcclComm(gk, "cuda:0")
        A = gd.NativeMatrix(M)
        n = M.num_local_rows
        ng, rp, ci, v = matgen.poisson_2d_5pt(grid)
        x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
        ye = np.zeros((n, 1))
        oracle.ref_csr_spmv(n, 1, rp, ci, v, x, 1, ye, 1)
        y = torch.full((n, 1), float("nan"), dtype=torch.float64, device="cuda:0")
        A.apply(comm, dev(x), y)
        torch.cuda.synchronize()
        assert np.array_equal(host(y), ye)
        xs = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
        res = A.cg(comm, dev(np.ones((n, 1))), xs, max_iters=3000, reduction=1e-10)
        xe = np.zeros(n)
        ite = oracle.ref_cg_solve(n, rp, ci, v, np.ones(n), xe, 3000, 1e-10, 0, None, 0)
        assert res["converged"] and abs(res["iterations"] - ite) <= 1
        assert matgen.rel_err(host(xs)[:, 0], xe) <= 1e-6
        # the Python-orchestrated schedule (gloo-tested) and the native driver agree
        xp = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
        itp, convp = gd.cg_fused(M, dev(np.ones((n, 1))), xp, max_iters=3000, reduction=1e-10)
        assert convp and abs(itp - res["iterations"]) <= 1 and matgen.rel_err(host(xp), host(xs)) <= 1e-8
        A.close()
        comm.close()
    finally:
        dist.destroy_process_group()
