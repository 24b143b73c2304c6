"""Row-partitioned distributed matrix / vector / CG over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" on CPUs for the logic
tests).  Mirrors gko::experimental::distributed::{Partition, Matrix, Vector}
(core/distributed/{partition,matrix,vector}.cpp) for the hot path:

  read_distributed : build_local_nonlocal on the device, then the two setup
                     exchanges of matrix.cpp:198-224 (counts, gather indices);
  apply            : row_gather pack -> asynchronous all-to-all-v of the halo
                     values on a side stream || local SpMV -> non-local
                     advanced SpMV (matrix.cpp:263-335);
  dot / norm2      : local two-stage reduction + all-reduce of the device
                     scalar (vector.cpp:317-409), no host round trip.

All compute goes through an `ops` object; the product one (GpuOps) is the C ABI
on device tensors and refuses to run without the HIP library.  Tests inject
their own ops to exercise the communication logic on CPUs.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist


class Partition:
    """Host-side partition metadata (O(#ranges)), as
    experimental::distributed::Partition keeps it."""

    def __init__(self, range_bounds, part_ids, starts, part_sizes, gk=None, num_empty_parts=None):
        self.range_bounds = np.ascontiguousarray(range_bounds, np.int64)
        self.part_ids = np.ascontiguousarray(part_ids, np.int32)
        self.starts = np.ascontiguousarray(starts, np.int32)
        self.part_sizes = np.ascontiguousarray(part_sizes, np.int32)
        self.num_parts = len(self.part_sizes)
        self.num_ranges = len(self.part_ids)
        self.size = int(self.range_bounds[-1])
        self.num_empty_parts = int(np.count_nonzero(self.part_sizes == 0)) if num_empty_parts is None else int(num_empty_parts)
        self._gk = gk

    def has_connected_parts(self):
        """every part is one range (core/distributed/partition.cpp:120-124)"""
        return self.num_parts - self.num_empty_parts == self.num_ranges

    def has_ordered_parts(self):
        """connected, and the parts follow each other in ascending order (partition.cpp:128-138 ->
        partition::has_ordered_parts, reference/distributed/partition_kernels.cpp:139-155)"""
        if not self.has_connected_parts():
            return False
        out = ctypes.c_int64(0)
        self._gk.partition_has_ordered_parts(self.part_ids, self.num_ranges, ctypes.addressof(out))
        return bool(out.value)

    @staticmethod
    def _finish(gk, bounds, ids, num_parts):
        nr = len(ids)
        starts = np.zeros(max(nr, 1), np.int32)
        sizes = np.zeros(max(num_parts, 1), np.int32)
        empty = ctypes.c_int64(0)
        gk.partition_build_starting_indices(bounds, ids, nr, num_parts, starts, sizes, ctypes.addressof(empty))
        return Partition(bounds, ids, starts[:nr], sizes[:num_parts], gk, empty.value)

    @staticmethod
    def build_from_contiguous(gk, ranges):
        """one range per part: ranges[p] .. ranges[p + 1] (Partition::build_from_contiguous, partition.cpp:76-95)"""
        ranges = np.ascontiguousarray(ranges, np.int64)
        num_parts = len(ranges) - 1
        bounds = np.zeros(num_parts + 1, np.int64)
        ids = np.zeros(max(num_parts, 1), np.int32)
        gk.partition_build_from_contiguous(num_parts, ranges, bounds, ids)
        return Partition._finish(gk, bounds, ids[:num_parts], num_parts)

    @staticmethod
    def build_from_global_size_uniform(gk, num_parts, global_size):
        ranges = np.zeros(num_parts + 1, np.int64)
        gk.partition_build_ranges_from_global_size(num_parts, global_size, ranges)
        bounds = np.zeros(num_parts + 1, np.int64)
        ids = np.zeros(num_parts, np.int32)
        gk.partition_build_from_contiguous(num_parts, ranges, bounds, ids)
        return Partition._finish(gk, bounds, ids, num_parts)

    @staticmethod
    def build_from_mapping(gk, mapping, num_parts):
        mapping = np.ascontiguousarray(mapping, np.int32)
        n = len(mapping)
        bounds = np.zeros(n + 1, np.int64)
        ids = np.zeros(max(n, 1), np.int32)
        nr = ctypes.c_int64(0)
        gk.partition_build_from_mapping(n, mapping, bounds, ids, ctypes.addressof(nr))
        return Partition._finish(gk, bounds[:nr.value + 1].copy(), ids[:nr.value].copy(), num_parts)


class GpuOps:
    """The product compute path: C ABI kernels on device tensors."""

    def __init__(self, gk, device):
        self.gk = gk
        self.device = torch.device(device)
        assert self.device.type == "cuda", "GpuOps needs a GPU; there is no CPU fallback"

    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def tensor(self, a, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(a)).to(self.device)
        return t if dtype is None else t.to(dtype)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def build_local_nonlocal(self, rows, cols, vals, row_part, col_part, local_part):
        gk, s = self.gk, self.stream()
        nnz = int(rows.numel())
        nb = gk.dist_build_workspace_bytes(nnz)
        ws = self.empty(nb, torch.uint8)
        rb, rid, rst = self.tensor(row_part.range_bounds), self.tensor(row_part.part_ids), self.tensor(row_part.starts)
        cb, cid, cst = self.tensor(col_part.range_bounds), self.tensor(col_part.part_ids), self.tensor(col_part.starts)
        sizes = (ctypes.c_int64 * 3)()
        gk.dist_build_local_nonlocal_sizes(s, nnz, rows, cols, rb, rid, rst, row_part.num_ranges, cb, cid, cst,
                                           col_part.num_ranges, local_part, ws, nb, ctypes.addressof(sizes))
        nl, nn, nu = (int(v) for v in sizes)
        i32, f64 = torch.int32, torch.float64
        out = dict(l_rows=self.empty(max(nl, 1), i32), l_cols=self.empty(max(nl, 1), i32), l_vals=self.empty(max(nl, 1), f64),
                   nl_rows=self.empty(max(nn, 1), i32), nl_cols=self.empty(max(nn, 1), i32), nl_vals=self.empty(max(nn, 1), f64),
                   gather_idxs=self.empty(max(nu, 1), i32), recv_sizes=self.empty(row_part.num_parts, i32),
                   non_local_to_global=self.empty(max(nu, 1), torch.int64), num_local=nl, num_non_local=nn, num_unique=nu)
        gk.dist_build_local_nonlocal_fill(s, nnz, rows, cols, vals, rb, rid, rst, row_part.num_ranges, cb, cid, cst,
                                          col_part.num_ranges, row_part.num_parts, ws, nu, out["l_rows"], out["l_cols"],
                                          out["l_vals"], out["nl_rows"], out["nl_cols"], out["nl_vals"],
                                          out["gather_idxs"], out["recv_sizes"], out["non_local_to_global"])
        return out

    def vector_build_local(self, rows, cols, vals, partition, local_part, ncols):
        """distributed_vector::build_local (what Vector::read_distributed runs): the dense local block of
        `local_part`, zero where the input has no entry"""
        nloc = int(partition.part_sizes[local_part])
        local = torch.zeros((nloc, ncols), dtype=torch.float64, device=self.device)
        nnz = int(rows.numel())
        if nnz:
            rb, rid, rst = self.tensor(partition.range_bounds), self.tensor(partition.part_ids), self.tensor(partition.starts)
            self.gk.dist_vector_build_local_f64(self.stream(), nnz, rows, cols, vals, rb, rid, rst, partition.num_ranges,
                                                local_part, local, ncols)
        return local

    def coo_to_csr(self, nrows, row_idxs, nnz):
        # Csr::read(device_matrix_data): convert_idxs_to_ptrs (core/matrix/csr.cpp:453-470)
        ptrs = self.empty(nrows + 1, torch.int32)
        nb = self.gk.prefix_sum_workspace_bytes(nrows + 1)
        ws = self.empty(max(nb, 8), torch.uint8)
        self.gk.convert_idxs_to_ptrs_i32(self.stream(), row_idxs, nnz, nrows, ptrs, ws, nb)
        return ptrs

    def spmv(self, csr, b, x, alpha=None, beta=None):
        nrows, ncols, nnz, rp, ci, v = csr
        self.gk.csr_spmv_f64_i32(self.stream(), nrows, ncols, b.shape[1], nnz, rp, ci, v, b, b.stride(0), x,
                                 x.stride(0), alpha, beta, 0, -1)

    def row_gather(self, idxs, count, src, out):
        self.gk.dense_row_gather_f64_i32(self.stream(), count, src.shape[1], idxs, src, src.stride(0), out, out.stride(0))

    def local_dot(self, x, y, result, ws):
        self.gk.dense_compute_dot_f64(self.stream(), x.shape[0], x.shape[1], x, x.stride(0), y, y.stride(0), result,
                                      ws, ws.numel())

    def local_squared_norm2(self, x, result, ws):
        self.gk.dense_compute_squared_norm2_f64(self.stream(), x.shape[0], x.shape[1], x, x.stride(0), result, ws,
                                                ws.numel())

    def reduction_workspace(self, nrows, ncols):
        return self.empty(max(self.gk.dense_reduction_workspace_bytes(nrows, ncols), 8), torch.uint8)

    def sqrt_(self, t):
        self.gk.dense_compute_sqrt_f64(self.stream(), 1, t.numel(), t, t.numel())

    def cg_initialize(self, b, r, z, p, q, prev_rho, rho, stop):
        n, k = b.shape
        self.gk.cg_initialize_f64(self.stream(), n, k, b, k, r, k, z, k, p, k, q, k, prev_rho, rho, stop)

    def cg_step_1(self, p, z, rho, prev_rho, stop):
        n, k = p.shape
        self.gk.cg_step_1_f64(self.stream(), n, k, p, k, z, k, rho, prev_rho, stop)

    def cg_step_2(self, x, r, p, q, beta, rho, stop):
        n, k = x.shape
        self.gk.cg_step_2_f64(self.stream(), n, k, x, k, r, k, p, k, q, k, beta, rho, stop)

    def copy(self, src, dst):
        n, k = src.shape
        self.gk.dense_copy_f64(self.stream(), n, k, src, src.stride(0), dst, dst.stride(0))

    def sub_scaled(self, alpha, x, y):
        n, k = x.shape
        self.gk.dense_sub_scaled_f64(self.stream(), n, k, alpha, alpha.numel(), x, x.stride(0), y, y.stride(0))

    def fill(self, x, value):
        n, k = x.shape
        self.gk.dense_fill_f64(self.stream(), n, k, x, x.stride(0), value)

    def copy_scalar(self, src, dst):
        self.gk.dense_copy_f64(self.stream(), 1, src.numel(), src, src.numel(), dst, dst.numel())

    def residual_check(self, tau, orig_tau, reduction, stop, flags):
        """ResidualNorm::check_impl: returns all_converged (blocking 2-byte copy)."""
        host = np.zeros(2, np.uint8)
        self.gk.residual_norm_f64(self.stream(), tau.numel(), tau, orig_tau, reduction, 2, 1, stop, flags, host)
        return bool(host[0])

    def residual_check_device(self, tau, orig_tau, reduction, stop, flags):
        """the same kernel without the host copy: flags = {all_converged, one_changed} on the device"""
        self.gk.residual_norm_f64(self.stream(), tau.numel(), tau, orig_tau, reduction, 2, 1, stop, flags, None)


class Matrix:
    """experimental::distributed::Matrix for the hot path."""

    def __init__(self, ops, group=None):
        self.ops = ops
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def read_distributed(self, rows, cols, vals, row_partition, col_partition=None):
        """rows/cols: int64 global indices of this rank's entries (any rows may
        be passed; entries of other parts are dropped like the reference does),
        in row-major order; vals float64."""
        ops = self.ops
        col_partition = col_partition or row_partition
        assert row_partition.num_parts == self.world
        rows, cols, vals = ops.tensor(rows, torch.int64), ops.tensor(cols, torch.int64), ops.tensor(vals, torch.float64)
        o = ops.build_local_nonlocal(rows, cols, vals, row_partition, col_partition, self.rank)
        self.num_local_rows = int(row_partition.part_sizes[self.rank])
        self.num_local_cols = int(col_partition.part_sizes[self.rank])
        nl, nn, nu = o["num_local"], o["num_non_local"], o["num_unique"]
        self.local = (self.num_local_rows, self.num_local_cols, nl,
                      ops.coo_to_csr(self.num_local_rows, o["l_rows"], nl), o["l_cols"], o["l_vals"])
        self.non_local = (self.num_local_rows, nu, nn,
                          ops.coo_to_csr(self.num_local_rows, o["nl_rows"], nn), o["nl_cols"], o["nl_vals"])
        self.non_local_to_global = o["non_local_to_global"][:nu]
        # exchange step 1: counts (matrix.cpp:198-209)
        recv_sizes = o["recv_sizes"].to(torch.int64)
        send_sizes = torch.empty_like(recv_sizes)
        dist.all_to_all_single(send_sizes, recv_sizes, group=self.group)
        self.recv_sizes = [int(v) for v in recv_sizes.cpu()]
        self.send_sizes = [int(v) for v in send_sizes.cpu()]
        # exchange step 2: receivers tell senders which local rows they need (:211-224)
        recv_gather = o["gather_idxs"][:nu].contiguous()
        self.gather_idxs = ops.empty(max(sum(self.send_sizes), 1), torch.int32)[:sum(self.send_sizes)]
        dist.all_to_all_single(self.gather_idxs, recv_gather, self.send_sizes, self.recv_sizes, group=self.group)
        self.send_count, self.recv_count = sum(self.send_sizes), sum(self.recv_sizes)
        # whether the apply exchanges anything is decided ONCE and for ALL ranks (a rank
        # without neighbours still takes part in the collective, matrix.cpp:263-303)
        any_halo = torch.tensor([self.send_count + self.recv_count], dtype=torch.int64, device=recv_sizes.device)
        dist.all_reduce(any_halo, op=dist.ReduceOp.MAX, group=self.group)
        self._any_halo = int(any_halo.item()) > 0
        self._bufs = {}
        self._one = ops.tensor(np.array([1.0]))
        return self

    def _buffers(self, nrhs):
        if nrhs not in self._bufs:
            self._bufs[nrhs] = (self.ops.empty((max(self.send_count, 1), nrhs), torch.float64),
                                self.ops.empty((max(self.recv_count, 1), nrhs), torch.float64))
        return self._bufs[nrhs]

    def apply(self, b, x):
        """x = A b on the local rows.  b, x: (num_local_*, nrhs) tensors."""
        ops = self.ops
        nrhs = b.shape[1]
        send, recv = self._buffers(nrhs)
        ops.row_gather(self.gather_idxs, self.send_count, b, send)
        if not self._any_halo:
            ops.spmv(self.local, b, x)  # no rank has neighbours: nothing to exchange anywhere
        else:
            # the collective runs on the backend's own stream (RCCL) behind the
            # pack kernel; the local SpMV is launched meanwhile and the wait only
            # fences the non-local part: halo exchange overlapped with compute
            work = dist.all_to_all_single(recv[:self.recv_count], send[:self.send_count], self.recv_sizes,
                                          self.send_sizes, group=self.group, async_op=True)
            ops.spmv(self.local, b, x)
            work.wait()
        if self.non_local[2] > 0:
            ops.spmv(self.non_local, recv, x, self._one, self._one)
        return x


class VectorOps:
    """experimental::distributed::Vector reductions (vector.cpp:317-409)."""

    def __init__(self, ops, nrows, nrhs, group=None):
        self.ops, self.group = ops, group
        self.ws = ops.reduction_workspace(nrows, nrhs)

    def dot(self, x, y, result):
        self.ops.local_dot(x, y, result, self.ws)
        dist.all_reduce(result, op=dist.ReduceOp.SUM, group=self.group)
        return result

    def norm2(self, x, result):
        self.ops.local_squared_norm2(x, result, self.ws)
        dist.all_reduce(result, op=dist.ReduceOp.SUM, group=self.group)
        self.ops.sqrt_(result)
        return result


def cg(matrix, b, x, max_iters=1000, reduction=1e-10, check_every=8):
    """Cg::apply_dense_impl on distributed vectors (core/solver/cg.cpp:107-193
    with detail::get_local for the step kernels), Identity preconditioner,
    Combined(Iteration, ResidualNorm(rhs_norm)).  Returns (iterations, converged).

    The criterion is evaluated on the device every iteration, where the
    reference evaluates it (the statuses stop the step kernels at once); the
    host reads the outcome every `check_every` iterations only -- the iterates
    and the returned count (= the checks that did not end the solve, counted on
    the device) do not depend on check_every."""
    ops = matrix.ops
    n, k = b.shape
    r, z, p, q = (ops.empty((n, k), torch.float64) for _ in range(4))
    f = lambda: ops.empty((k,), torch.float64)
    prev_rho, rho, beta, tau, orig = f(), f(), f(), f(), f()
    stop = ops.empty((k,), torch.uint8)
    vec = VectorOps(ops, n, k, matrix.group)
    flags = ops.empty((2,), torch.uint8)
    running = ops.tensor(np.zeros(1, np.int64))   # checks that found unconverged columns
    one = ops.tensor(np.ones(1))
    ops.cg_initialize(b, r, z, p, q, prev_rho, rho, stop)
    matrix.apply(x, q)                       # r = b - A x
    ops.sub_scaled(one, q, r)
    ops.fill(q, 0.0)
    vec.norm2(b, orig)
    check_every = max(1, int(check_every))
    it = -1
    while True:
        ops.copy(r, z)
        vec.dot(r, z, rho)
        it += 1
        if it >= max_iters:
            done = int(running.item())
            return (done, True) if done < it else (it, False)
        vec.norm2(r, tau)
        ops.residual_check_device(tau, orig, reduction, stop, flags)
        running += (flags[0:1] == 0)
        if (it + 1) % check_every == 0 and int(flags[0].item()):
            return int(running.item()), True
        ops.cg_step_1(p, z, rho, prev_rho, stop)
        matrix.apply(p, q)
        vec.dot(p, q, beta)
        ops.cg_step_2(x, r, p, q, beta, rho, stop)
        prev_rho, rho = rho, prev_rho


def cg_fused(matrix, b, x, max_iters=1000, reduction=1e-10, check_every=8):
    """The communication schedule of the native fused driver (csrc/dist_cg.hip) over
    any `ops`: per iteration ONE two-element all-reduce for rho = r.z and tau^2 = r.r,
    one for beta = p.q, the halo exchange overlapped with the local SpMV, the criterion
    on the device, the host looking every `check_every` iterations.  Same iterates as
    cg() up to the order of the reductions; Identity preconditioner, one or more
    right-hand sides.  Returns (iterations, converged)."""
    ops = matrix.ops
    n, k = b.shape
    r, z, p, q = (ops.empty((n, k), torch.float64) for _ in range(4))
    f = lambda m=k: ops.empty((m,), torch.float64)
    prev_rho, rho, beta, tau, orig = f(), f(), f(), f(), f()
    pair = f(2 * k)                      # {r.z per column, r.r per column}: one all-reduce
    stop = ops.empty((k,), torch.uint8)
    ws = ops.reduction_workspace(n, k)
    flags = ops.empty((2,), torch.uint8)
    running = ops.tensor(np.zeros(1, np.int64))
    one = ops.tensor(np.ones(1))
    ops.cg_initialize(b, r, z, p, q, prev_rho, rho, stop)
    matrix.apply(x, q)
    ops.sub_scaled(one, q, r)
    ops.fill(q, 0.0)
    ops.local_squared_norm2(b, orig, ws)
    dist.all_reduce(orig, op=dist.ReduceOp.SUM, group=matrix.group)
    ops.sqrt_(orig)
    check_every = max(1, int(check_every))
    it = -1
    while True:
        ops.copy(r, z)                   # Identity preconditioner
        ops.local_dot(r, z, pair[:k], ws)
        ops.local_squared_norm2(r, pair[k:], ws)
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=matrix.group)
        ops.copy_scalar(pair[:k], rho)
        ops.copy_scalar(pair[k:], tau)
        ops.sqrt_(tau)
        it += 1
        if it >= max_iters:
            done = int(running.item())
            return (done, True) if done < it else (it, False)
        ops.residual_check_device(tau, orig, reduction, stop, flags)
        running += (flags[0:1] == 0)
        if (it + 1) % check_every == 0 and int(flags[0].item()):
            return int(running.item()), True
        ops.cg_step_1(p, z, rho, prev_rho, stop)
        matrix.apply(p, q)
        ops.local_dot(p, q, beta, ws)
        dist.all_reduce(beta, op=dist.ReduceOp.SUM, group=matrix.group)
        ops.cg_step_2(x, r, p, q, beta, rho, stop)
        prev_rho, rho = rho, prev_rho


# ---- native path: C-ABI communicator + distributed matrix + fused CG (csrc/comm.hip, dist_cg.hip) ----

class CommStruct(ctypes.Structure):
    """gkomi_comm (include/gkomi.h)."""
    _fields_ = [("self", ctypes.c_void_p), ("rank", ctypes.c_int), ("size", ctypes.c_int),
                ("allreduce_sum_f64", ctypes.c_void_p), ("alltoallv", ctypes.c_void_p)]


class DistMatrixStruct(ctypes.Structure):
    """gkomi_dist_matrix (include/gkomi.h)."""
    _fields_ = [("n_local", ctypes.c_int64), ("n_halo", ctypes.c_int64), ("l_nnz", ctypes.c_int64),
                ("l_row_ptrs", ctypes.c_void_p), ("l_col_idxs", ctypes.c_void_p), ("l_vals", ctypes.c_void_p),
                ("l_max_row_nnz", ctypes.c_int64), ("l_srow", ctypes.c_void_p), ("l_srow_tile", ctypes.c_int64),
                ("nl_rows", ctypes.c_int64), ("nl_nnz", ctypes.c_int64), ("nl_row_idxs", ctypes.c_void_p),
                ("nl_row_ptrs", ctypes.c_void_p), ("nl_col_idxs", ctypes.c_void_p), ("nl_vals", ctypes.c_void_p),
                ("send_total", ctypes.c_int64), ("gather_idxs", ctypes.c_void_p),
                ("send_counts", ctypes.c_void_p), ("send_offsets", ctypes.c_void_p),
                ("recv_counts", ctypes.c_void_p), ("recv_offsets", ctypes.c_void_p),
                ("send_buf", ctypes.c_void_p), ("recv_buf", ctypes.c_void_p)]


class RcclComm:
    """gkomi_comm over RCCL, bootstrapped through the torch.distributed group that
    launched the ranks (the unique id is the only thing that travels there)."""

    def __init__(self, gk, device, group=None):
        self.gk = gk
        if not gk.comm_rccl_available():
            raise RuntimeError("RCCL could not be opened: the native distributed path needs it")
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        nb = int(gk.comm_unique_id_bytes())
        ident = np.zeros(nb, np.uint8)
        if rank == 0:
            gk.comm_rccl_unique_id(ident)
        t = torch.from_numpy(ident)
        if dist.get_backend(group) == "nccl":
            t = t.to(device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = t.cpu().numpy().copy()
        self.struct = CommStruct()
        gk.comm_rccl_create(ident, rank, world, ctypes.addressof(self.struct))
        self.ptr = ctypes.addressof(self.struct)
        self.rank, self.size = rank, world

    def query(self):
        """(ranks, this rank) as RCCL itself reports them (ncclCommCount / ncclCommUserRank)"""
        count, me = ctypes.c_int(0), ctypes.c_int(0)
        self.gk.comm_rccl_query(self.ptr, ctypes.addressof(count), ctypes.addressof(me))
        return int(count.value), int(me.value)

    def close(self):
        if self.struct.self:
            self.gk.comm_rccl_destroy(self.ptr)


class NativeMatrix:
    """gkomi_dist_matrix of a Matrix that went through read_distributed (GpuOps), plus the
    side stream / events of its overlapped exchange."""

    def __init__(self, matrix):
        self._build(matrix.ops, matrix.num_local_rows, matrix.local, matrix.non_local, matrix.send_sizes,
                    matrix.recv_sizes, matrix.gather_idxs)

    @classmethod
    def from_parts(cls, ops, n_local, local, non_local, send_sizes, recv_sizes, gather_idxs):
        """local / non_local = (nrows, ncols, nnz, row_ptrs, col_idxs, vals) device CSR blocks,
        send/recv sizes per peer, gather_idxs (device int32) grouped by receiver."""
        self = cls.__new__(cls)
        self._build(ops, n_local, local, non_local, send_sizes, recv_sizes, gather_idxs)
        return self

    def _build(self, ops, n_loc, local, non_local, send_sizes, recv_sizes, gather_idxs):
        gk = ops.gk
        self.gk, self.ops, self.num_local_rows = gk, ops, n_loc
        self._keep = (local, non_local, gather_idxs)
        s = ops.stream()
        send_count = int(sum(send_sizes))
        _, _, l_nnz, l_rp, l_ci, l_v = local
        _, n_halo, nl_nnz, nl_rp, nl_ci, nl_v = non_local
        # the rows of the non-local block that have entries
        nb = gk.dist_nonlocal_rows_workspace_bytes(n_loc)
        ws = ops.empty(max(nb, 8), torch.uint8)
        self.nl_row_idxs = ops.empty(n_loc + 1, torch.int32)
        self.nl_row_ptrs = ops.empty(n_loc + 1, torch.int32)
        cnt = ctypes.c_int64(0)
        gk.dist_nonlocal_rows_i32(s, n_loc, nl_rp, self.nl_row_idxs, self.nl_row_ptrs, ws, nb, ctypes.addressof(cnt))
        self.nl_rows = int(cnt.value)
        # srow of the local block (Csr::make_srow)
        tile = int(gk.csr_srow_tile_for(l_nnz))
        self.max_row_nnz = -1
        if n_loc > 0:
            mx = ops.empty(1, torch.int32)
            gk.csr_max_row_nnz_i32(s, n_loc, l_rp, mx)
            self.max_row_nnz = int(mx.item())
        self.srow = ops.empty(max(int(gk.csr_srow_entries(l_nnz, tile)), 1), torch.int32)
        if l_nnz >= 2:
            gk.csr_make_srow_i32(s, n_loc, l_nnz, l_rp, tile, self.srow, self.srow.numel())
        self.send_counts = np.array(send_sizes, np.int64)
        self.recv_counts = np.array(recv_sizes, np.int64)
        self.send_offsets = np.concatenate([[0], np.cumsum(self.send_counts)[:-1]]).astype(np.int64)
        self.recv_offsets = np.concatenate([[0], np.cumsum(self.recv_counts)[:-1]]).astype(np.int64)
        assert len(self.send_counts) == len(self.recv_counts) and int(self.recv_counts.sum()) == n_halo
        self.send_buf = ops.empty(max(send_count, 1), torch.float64)
        self.recv_buf = ops.empty(max(n_halo, 1), torch.float64)
        self.struct = DistMatrixStruct(
            n_loc, n_halo, l_nnz, l_rp.data_ptr(), l_ci.data_ptr(), l_v.data_ptr(), self.max_row_nnz,
            self.srow.data_ptr() if l_nnz >= 2 else None, tile, self.nl_rows, nl_nnz, self.nl_row_idxs.data_ptr(),
            self.nl_row_ptrs.data_ptr(), nl_ci.data_ptr(), nl_v.data_ptr(), send_count,
            gather_idxs.data_ptr(), self.send_counts.ctypes.data, self.send_offsets.ctypes.data,
            self.recv_counts.ctypes.data, self.recv_offsets.ctypes.data, self.send_buf.data_ptr(),
            self.recv_buf.data_ptr())
        self.ptr = ctypes.addressof(self.struct)
        ctx = ctypes.c_void_p(0)
        gk.dist_ctx_create(ctypes.addressof(ctx))
        self.ctx = ctx.value

    def close(self):
        if self.ctx:
            self.gk.dist_ctx_destroy(self.ctx)
            self.ctx = None

    def apply(self, comm, b, x):
        """Matrix::apply_impl, one right-hand side (gkomi_dist_matrix_apply_f64)."""
        self.gk.dist_matrix_apply_f64(self.ops.stream(), comm.ptr, self.ctx, self.ptr, b, x)
        return x

    def cg(self, comm, b, x, max_iters=1000, reduction=1e-10, baseline="rhs_norm", check_every=16, precond=None):
        """gkomi_dist_cg_solve_f64.  Returns dict(iterations, converged, residual_norm, baseline_norm)."""
        ops = self.ops
        nb = self.gk.dist_cg_workspace_bytes(self.num_local_rows, self.nl_rows)
        if getattr(self, "_ws", None) is None or self._ws.numel() < nb:
            self._ws = ops.empty(nb, torch.uint8)
        info = np.zeros(4, np.float64)
        self.gk.dist_cg_solve_f64(ops.stream(), comm.ptr, self.ctx, self.ptr, precond.fn if precond is not None else None,
                                  precond.ctx_ptr if precond is not None else None, b, x, max_iters, reduction,
                                  {"rhs_norm": 0, "initial_resnorm": 1, "absolute": 2}[baseline], check_every,
                                  self._ws, nb, info)
        return {"iterations": int(info[0]), "converged": bool(info[1]), "residual_norm": float(info[2]),
                "baseline_norm": float(info[3])}


def poisson_slab_rows(grid, rank, world):
    """COO entries (global int64 indices, row-major) of rank's rows of the 5-pt
    Poisson matrix on a (grid*world) x grid mesh, row = i*grid + j: the bench's
    weak-scaling workload (one 1000 x 1000 slab per GPU)."""
    nx, ny = grid * world, grid
    i0 = rank * grid
    i, j = np.meshgrid(np.arange(i0, i0 + grid, dtype=np.int64), np.arange(ny, dtype=np.int64), indexing="ij")
    i, j = i.ravel(), j.ravel()
    row = i * ny + j
    cols = np.stack([row - ny, row - 1, row, row + 1, row + ny], axis=1)
    valid = np.stack([i > 0, j > 0, np.ones_like(i, bool), j < ny - 1, i < nx - 1], axis=1)
    vals = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), cols.shape)
    rows = np.broadcast_to(row[:, None], cols.shape)
    return rows[valid], cols[valid], np.ascontiguousarray(vals[valid]), nx * ny


def poisson3d_rows(g, lo, hi):
    """COO entries (global int64 indices, row-major, ascending columns) of rows [lo, hi) of the
    7-pt Poisson matrix on a g^3 grid, row = (i*g + j)*g + k: BASELINE config 5 (g = 256) cut into
    contiguous row slabs; every rank generates only its own rows."""
    idx = np.arange(lo, hi, dtype=np.int64)
    k = idx % g
    j = (idx // g) % g
    i = idx // (g * g)
    cols = np.stack([idx - g * g, idx - g, idx - 1, idx, idx + 1, idx + g, idx + g * g], axis=1)
    valid = np.stack([i > 0, j > 0, k > 0, np.ones(len(idx), bool), k < g - 1, j < g - 1, i < g - 1], axis=1)
    vals = np.broadcast_to(np.array([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0]), cols.shape)
    rows = np.broadcast_to(idx[:, None], cols.shape)
    return rows[valid], cols[valid], np.ascontiguousarray(vals[valid])


def poisson_slab_matrix(gk, grid, rank, world, device, group=None):
    rows, cols, vals, n_global = poisson_slab_rows(grid, rank, world)
    part = Partition.build_from_global_size_uniform(gk, world, n_global)
    m = Matrix(GpuOps(gk, device), group).read_distributed(rows, cols, vals, part)
    m.global_nnz_local_rows = len(vals)
    return m
