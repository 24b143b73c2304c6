"""GPU parity tests of the distributed layer's device side: the device
build_local_nonlocal (bit-exact against the oracle, on the reference's
known answers and on random partitions), a multi-part apply emulated in one
process (local SpMV + non-local SpMV over gathered halos), and the full
gkomi.distributed.Matrix / cg path over RCCL with world_size 1."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import gkomi.distributed as gd
import matgen
from gpu_util import dev, host
from test_oracle_distributed import oracle_build, oracle_partition_from_mapping

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distributed.json")))


def gpu_build(gk, rows, cols, vals, partition, part, col_partition=None):
    ops = gd.GpuOps(gk, "cuda:0")
    return ops.build_local_nonlocal(ops.tensor(np.array(rows, np.int64)), ops.tensor(np.array(cols, np.int64)),
                                    ops.tensor(np.array(vals, np.float64)), partition,
                                    partition if col_partition is None else col_partition, part)


def check_against(o, sz, g):
    nl, nn, nu = int(sz[0]), int(sz[1]), int(sz[2])
    assert (g["num_local"], g["num_non_local"], g["num_unique"]) == (nl, nn, nu)
    for a, b, n in (("l_rows", "l_rows", nl), ("l_cols", "l_cols", nl), ("l_vals", "l_vals", nl),
                    ("nl_rows", "nl_rows", nn), ("nl_cols", "nl_cols", nn), ("nl_vals", "nl_vals", nn),
                    ("gather", "gather_idxs", nu), ("n2g", "non_local_to_global", nu)):
        assert np.array_equal(o[a][:n], host(g[b])[:n]), a
    assert np.array_equal(o["recv"], host(g["recv_sizes"]))


@pytest.mark.parametrize("case", G["build_local_nonlocal"], ids=lambda c: c["name"])
def test_build_local_nonlocal_known_answers(gk, oracle, case):
    part = gd.Partition.build_from_mapping(gk, case["mapping"], case["num_parts"])
    meta = oracle_partition_from_mapping(oracle, case["mapping"], case["num_parts"])
    cpart = cmeta = None
    if "col_mapping" in case:   # a column partition of its own (matrix_kernels.cpp:333-513)
        cpart = gd.Partition.build_from_mapping(gk, case["col_mapping"], case["num_parts"])
        cmeta = oracle_partition_from_mapping(oracle, case["col_mapping"], case["num_parts"])
    for p in range(case["num_parts"]):
        g = gpu_build(gk, case["rows"], case["cols"], case["vals"], part, p, cpart)
        o, sz = oracle_build(oracle, case["rows"], case["cols"], case["vals"], meta, case["num_parts"], p, cmeta)
        check_against(o, sz, g)
        for key, field in (("local", "l_"), ("non_local", "nl_")):
            n = int(sz[0] if key == "local" else sz[1])
            for what in ("rows", "cols", "vals"):
                assert list(host(g[field + what])[:n]) == case[key][p][what], (key, what)
        assert list(host(g["gather_idxs"])[:sz[2]]) == case["gather_idxs"][p]
        assert list(host(g["recv_sizes"])) == case["recv_sizes"][p]


@pytest.mark.parametrize("case", G["ghost_maps"], ids=lambda c: c["name"])
def test_ghost_maps(gk, case):
    part = gd.Partition.build_from_mapping(gk, case["mapping"], case["num_parts"])
    for p in range(case["num_parts"]):
        g = gpu_build(gk, case["rows"], case["cols"], case["vals"], part, p)
        assert list(host(g["non_local_to_global"])[:g["num_unique"]]) == case["non_local_to_global"][p]


@pytest.mark.parametrize("case", G["vector_build_local"], ids=lambda c: c["name"])
def test_vector_build_local_known_answers(gk, case):
    """distributed_vector::build_local (reference/test/distributed/vector_kernels.cpp:105-152)"""
    ops = gd.GpuOps(gk, "cuda:0")
    part = gd.Partition.build_from_mapping(gk, case["mapping"], case["num_parts"])
    rows, cols = ops.tensor(np.array(case["rows"], np.int64)), ops.tensor(np.array(case["cols"], np.int64))
    vals = ops.tensor(np.array(case["vals"], np.float64))
    for p in range(case["num_parts"]):
        got = host(ops.vector_build_local(rows, cols, vals, part, p, case["size"][1]))
        assert np.array_equal(got, np.array(case["local"][p], np.float64).reshape(got.shape))


@pytest.mark.parametrize("seed,n,ncols,nparts", [(1, 300, 3, 4), (2, 20000, 5, 7)])
def test_vector_build_local_random_partitions(gk, oracle, seed, n, ncols, nparts):
    from test_oracle_distributed import oracle_vector_build_local
    rng = np.random.default_rng(seed)
    mapping = np.repeat(rng.integers(0, nparts, size=n // 5 + 1), 5)[:n].astype(np.int32)
    # distinct positions in random order (duplicates: any of them may win on the device)
    flat = rng.permutation(n * ncols)[: (n * ncols) // 2]
    case = dict(size=[n, ncols], rows=(flat // ncols).tolist(), cols=(flat % ncols).tolist(),
                vals=rng.standard_normal(len(flat)).tolist())
    ops = gd.GpuOps(gk, "cuda:0")
    part = gd.Partition.build_from_mapping(gk, mapping, nparts)
    meta = oracle_partition_from_mapping(oracle, mapping, nparts)
    rows, cols = ops.tensor(np.array(case["rows"], np.int64)), ops.tensor(np.array(case["cols"], np.int64))
    vals = ops.tensor(np.array(case["vals"], np.float64))
    for p in range(nparts):
        got = host(ops.vector_build_local(rows, cols, vals, part, p, ncols))
        assert np.array_equal(got, oracle_vector_build_local(oracle, case, meta, p))


@pytest.mark.parametrize("seed,n,nparts", [(1, 200, 3), (2, 5000, 8), (3, 777, 5)])
def test_build_local_nonlocal_random_partitions(gk, oracle, seed, n, nparts):
    rng = np.random.default_rng(seed)
    # non-contiguous ownership: runs of random length assigned to random parts
    mapping = np.repeat(rng.integers(0, nparts, size=n // 7 + 1), 7)[:n].astype(np.int32)
    rp, ci, v = matgen.random_csr(n, n, 0, 12, seed=seed)
    rows = np.repeat(np.arange(n), np.diff(rp)).astype(np.int64)
    cols = ci.astype(np.int64)
    part = gd.Partition.build_from_mapping(gk, mapping, nparts)
    meta = oracle_partition_from_mapping(oracle, mapping, nparts)
    for p in range(nparts):
        g = gpu_build(gk, rows, cols, v, part, p)
        o, sz = oracle_build(oracle, rows, cols, v, meta, nparts, p)
        check_against(o, sz, g)


def test_multi_part_apply_in_one_process(gk, oracle):
    """Every part's local + non-local SpMV over explicitly gathered halos
    reproduces the global SpMV: what Matrix::apply computes, minus the wire."""
    grid, world = 60, 4
    ops = gd.GpuOps(gk, "cuda:0")
    ng, rp, ci, v = matgen.poisson_2d_5pt(grid * world, grid)
    xg = np.sin(0.01 * np.arange(ng)).reshape(ng, 1)
    ye = np.zeros((ng, 1))
    oracle.ref_csr_spmv(ng, 1, rp, ci, v, xg, 1, ye, 1)
    part = gd.Partition.build_from_global_size_uniform(gk, world, ng)
    one = ops.tensor(np.ones(1))
    for r in range(world):
        rows, cols, vals, n_global = gd.poisson_slab_rows(grid, r, world)
        assert n_global == ng
        o = ops.build_local_nonlocal(ops.tensor(rows), ops.tensor(cols), ops.tensor(vals), part, part, r)
        n_loc = int(part.part_sizes[r])
        lo = int(part.range_bounds[r])
        nl, nn, nu = o["num_local"], o["num_non_local"], o["num_unique"]
        local = (n_loc, n_loc, nl, ops.coo_to_csr(n_loc, o["l_rows"], nl), o["l_cols"], o["l_vals"])
        nonlocal_ = (n_loc, nu, nn, ops.coo_to_csr(n_loc, o["nl_rows"], nn), o["nl_cols"], o["nl_vals"])
        x = ops.tensor(xg[lo:lo + n_loc])
        y = ops.empty((n_loc, 1), torch.float64)
        ops.spmv(local, x, y)
        halo = ops.tensor(xg[host(o["non_local_to_global"])[:nu]])  # what the neighbours would send
        if nn:
            ops.spmv(nonlocal_, halo, y, one, one)
        assert matgen.rel_err(host(y), ye[lo:lo + n_loc]) <= 1e-15
        assert np.array_equal(host(y)[grid:-grid], ye[lo + grid:lo + n_loc - grid])
        assert list(host(o["recv_sizes"])) == [grid if abs(p - r) == 1 else 0 for p in range(world)]


def test_matrix_and_cg_over_rccl_world_size_one(gk, oracle):
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        grid = 64
        A = gd.poisson_slab_matrix(gk, grid, 0, 1, "cuda:0")
        n = A.num_local_rows
        ng, rp, ci, v = matgen.poisson_2d_5pt(grid)
        x = np.sin(0.01 * np.arange(n)).reshape(n, 1)
        ye = np.zeros((n, 1))
        oracle.ref_csr_spmv(n, 1, rp, ci, v, x, 1, ye, 1)
        y = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
        A.apply(dev(x), y)
        torch.cuda.synchronize()
        assert np.array_equal(host(y), ye)
        xs = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
        it, conv = gd.cg(A, dev(np.ones((n, 1))), xs, max_iters=2000, reduction=1e-10)
        xe = np.zeros(n)
        ite = oracle.ref_cg_solve(n, rp, ci, v, np.ones(n), xe, 2000, 1e-10, 0, None, 0)
        assert conv and abs(it - ite) <= 1
        assert matgen.rel_err(host(xs)[:, 0], xe) <= 1e-6
    finally:
        dist.destroy_process_group()
