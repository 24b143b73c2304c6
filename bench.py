#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): CSR SpMV GFLOP/s + achieved HBM GB/s, CG
iterations/s to 1e-10, next to the reference's omp/ path (restated in oracle/,
kind "port") on the host cores of the same box.

    python bench.py --gpus N --steps K --warmup W

N = 1  (configs[1], SURVEY.md 8(d) "P2"): a step = one `Csr::apply` (y = A x) on
       the 1M-row 5-pt Poisson matrix.  Inputs are resident in HBM before the
       timed region.  `value` is measured COLD: steps rotate over 8 independent
       copies of (A, srow, x, y) -- 640 MB, so the 256 MiB Infinity Cache cannot
       serve the 80 MB problem on-die -- with the automatic strategy and no
       caller flag (what Csr::apply through the reference interface passes): the
       library's residency tracker sees the other copies go by and selects the
       nontemporal matrix streams itself.  The warm figure (same matrix
       every step = the reference's benchmark/spmv methodology, automatic
       strategy) is reported under "warm", CG on P2 under "cg", and the one-GPU
       anchor of the multi-GPU curve (P3: 256^3 7-pt Poisson, SpMV + CG to
       1e-10) under "p3".
N > 1  (configs[4], "P3", one process per GPU under torch.distributed): STRONG
       scaling of the row-partitioned 16.7M-row 256^3 7-pt Poisson problem:
       a step = one distributed apply (pack -> halo exchange over RCCL || local
       SpMV -> non-local rows), `value` = 2 nnz(global) / max-over-ranks time;
       "cg" = row-partitioned CG to 1e-10 without an iteration cap (native fused
       driver, csrc/dist_cg.hip, over its own RCCL communicator).
       GKOMI_BENCH_FORCE_DIST=1 runs that code path with a world of one rank.
       Without WORLD_SIZE in the environment `--gpus N` starts the N rank processes
       itself (python -m torch.distributed.run, as child processes, before this
       process makes any GPU call) and exits with their code; with fewer than N
       devices it says so and exits 2.

Timing: every timed region is EXACTLY K steps between barrier + synchronize on both
sides (max over ranks).  The line's `value` is the MEDIAN region of as many regions as it
takes to cover >= 0.05 s of measured steps (the reference's own floor,
benchmark/utils/general.hpp:96-117; at least 3, at most 400 regions); best and all-region
statistics are reported next to it.
"""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "repo-8852-ginkgo_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec, MI355X_MICROARCH.md "HBM3E peak BW"
GRID = 1000            # P2: 1000 x 1000 grid
T_START = time.perf_counter()
TRIALS = 3
GKOMI_CSR_STREAMING = 1 << 24


def algorithmic_bytes(nrows, ncols, nnz):
    """CSR SpMV fp64/i32, 1 rhs."""
    return 12 * nnz + 4 * (nrows + 1) + 8 * ncols + 8 * nrows


def traffic_from_profiles():
    """Per-launch HBM bytes of the dominant kernel from the committed PMC
    passes (profiles/r*_traffic.json, newest round); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        return int(json.load(open(files[-1]))["traffic_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--strategy", type=int, default=0, help="C-ABI strategy word of the cold leg (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cg", action="store_true")
    ap.add_argument("--no-p3", action="store_true")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--p3-grid", type=int, default=256, help="grid of the 3-D problem (256 = BASELINE config 5)")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-irregular", action="store_true",
                    help="skip the scattered-column classes (uniform random / power-law rows, 1 M rows: SuiteSparse's graph-like class)")
    ap.add_argument("--min-region-seconds", type=float, default=0.05,
                    help="floor on the total length of the timed regions of K steps each (more regions, never more steps)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-n1-anchor", action="store_true",
                    help="N > 1: skip rank 0's one-GPU measurement of the same matrix (n1_same_run)")
    ap.add_argument("--rehearse-line", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, initialise the process group, report the rank count, exit (launch rehearsal; "
                         "GKOMI_BENCH_BACKEND=gloo runs it without GPUs)")
    return ap.parse_args()


# ---- CPU baseline (child processes: the OpenMP binding is fixed when the runtime starts) ----

def cpu_baseline_child(seconds):
    """Reference omp/ CSR SpMV + CG (oracle port, omp/matrix/csr_kernels.cpp:76-99,
    core/solver/cg.cpp:107-193 on omp kernels) on this host's cores, first-touch
    placed; prints one JSON line."""
    allowed, share = len(os.sched_getaffinity(0)), cpu_share()   # before the OpenMP runtime binds this thread
    import matgen
    import oracle_lib
    orc = oracle_lib.load()
    n, rp, ci, v = matgen.poisson_2d_5pt(GRID)
    x = np.sin(0.01 * np.arange(n))
    nnz = int(rp[-1])
    h = orc.omp_bench_create(n, rp, ci, v, x)
    orc.omp_bench_spmv(h, 3)
    t0 = time.perf_counter()
    orc.omp_bench_spmv(h, 20)
    per = (time.perf_counter() - t0) / 20
    reps = max(20, int(seconds / max(per, 1e-6)))
    t0 = time.perf_counter()
    orc.omp_bench_spmv(h, reps)
    el = time.perf_counter() - t0
    y = np.zeros(n)
    orc.omp_bench_get_y(h, y)
    s = np.sin(np.arange(n, dtype=np.float64))
    s /= np.linalg.norm(s)
    b = np.zeros((n, 1))
    orc.ref_csr_spmv(n, 1, rp, ci, v, s.reshape(n, 1), 1, b, 1)
    xs, rel = np.zeros(n), np.zeros(1)
    t0 = time.perf_counter()
    its = int(orc.omp_bench_cg(h, b[:, 0].copy(), xs, 20000, 1e-10, rel))
    el_cg = time.perf_counter() - t0
    orc.omp_bench_destroy(h)
    numa = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]) \
        if os.path.isdir("/sys/devices/system/node") else 1
    out = {"value": round(2.0 * nnz * reps / el / 1e9, 3), "unit": "GFLOP/s",
           "cores": int(orc.omp_bench_threads()), "threads": int(orc.omp_bench_threads()),
           "host_cpus": os.cpu_count(), "cpus_allowed": allowed, "cpu_share": share,
           "numa_nodes": numa, "kind": "port",
           "omp_num_threads": os.environ.get("OMP_NUM_THREADS", ""),
           "omp_proc_bind": os.environ.get("OMP_PROC_BIND", ""), "omp_places": os.environ.get("OMP_PLACES", ""),
           "sample": f"{reps} x omp csr::spmv on the same 1M-row 5-pt Poisson matrix ({el:.1f} s), "
                     "arrays first-touched by the threads that stream them",
           "gbs": round(algorithmic_bytes(n, n, nnz) * reps / el / 1e9, 2),
           "cg": {"iterations": its, "seconds": round(el_cg, 4), "iters_per_sec": round(its / el_cg, 1),
                  "final_residual_norm_rel": float(rel[0]),
                  "sample": "one omp CG solve to 1e-10 (Cg::apply_dense_impl on omp kernels), sinus rhs: the GPU cg entry's system"},
           "y_checksum": float(np.sum(y))}
    print(json.dumps(out))


def cpu_share():
    """CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota
    (the GPU box shows 256 CPUs in the mask and a quota of 16: 256 OpenMP threads on that
    quota spend their time being throttled -- the first version of this leg did just that)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def progress(msg):
    sys.stderr.write(f"[bench {time.perf_counter() - T_START:7.1f}s] {msg}\n")
    sys.stderr.flush()


def cpu_baseline(seconds):
    """The omp/ path on the CPUs this process may use (cpu_share), in child processes (the
    OpenMP runtime fixes its binding when it starts): unbound, OMP_PROC_BIND=true
    and OMP_PROC_BIND=true + OMP_PLACES=cores; the fastest SpMV is the headline, all are reported."""
    runs = {}
    ncpu = str(cpu_share())
    for name, env in (("unbound", {"OMP_PROC_BIND": "false"}), ("bound", {"OMP_PROC_BIND": "true"}),
                      ("bound_cores", {"OMP_PROC_BIND": "true", "OMP_PLACES": "cores"})):
        e = {k: v for k, v in os.environ.items() if not k.startswith("OMP_") and not k.startswith("GOMP_")}
        e.update(env)
        e["OMP_NUM_THREADS"] = ncpu
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-seconds", str(seconds)],
                               env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
            progress(f"cpu baseline {name}: {runs[name]['value']} GFLOP/s with {runs[name]['threads']} threads")
        except Exception as ex:  # noqa: BLE001 - reported in the line
            runs[name] = {"error": repr(ex)[:200]}
    good = {k: v for k, v in runs.items() if "value" in v}
    if not good:
        return {"error": runs}
    best = max(good, key=lambda k: good[k]["value"])
    out = dict(good[best])
    out["binding"] = best
    out["runs"] = {k: ({"gflops": v["value"], "gbs": v["gbs"], "threads": v["threads"],
                        "cg_iters_per_sec": v["cg"]["iters_per_sec"]} if "value" in v else v) for k, v in runs.items()}
    return out


# ---- timing helpers ----

def time_loop(torch, fn, steps, barrier):
    """Exactly `steps` calls bracketed by barrier + synchronize; returns
    (wall seconds, HIP-event seconds on the launch stream)."""
    barrier()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, e0.elapsed_time(e1) * 1e-3


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE: start N rank processes (one per GPU) as CHILDREN of this
    process -- which has made no GPU call and never will -- and exit with their code.  A process that has
    touched the GPU must not exec another program on this pool, hence children, never os.exec*."""
    import socket
    backend = os.environ.get("GKOMI_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch  # importing torch and counting devices initialises no GPU context
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible on this node: "
                             f"nothing was started.  Use --gpus {max(have, 1)} or a node with {args.gpus} GPUs.\n")
            sys.exit(2)
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    progress(f"starting {args.gpus} ranks: {' '.join(cmd[1:8])} bench.py ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def rendezvous_only(args, world, rank, local_rank):
    """The launch path without the benchmark: process group up, every rank counted, one JSON line."""
    import torch
    import torch.distributed as dist
    backend = os.environ.get("GKOMI_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        t = torch.ones(1, device=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
        t = torch.ones(1)
    dist.all_reduce(t)
    ranks = sorted(int(r) for r in _gather_ranks(dist, rank, world))
    if rank == 0:
        print(json.dumps({"rendezvous": int(t.item()), "n_gpus": args.gpus, "world_size": dist.get_world_size(),
                          "backend": backend, "ranks": ranks}), flush=True)
    dist.destroy_process_group()


def _gather_ranks(dist, rank, world):
    got = [None] * world
    dist.all_gather_object(got, rank)
    return got


def p3_metric(g):
    """the metric string of every N > 1 line: the same for all N (the driver compares values across N)"""
    return f"distributed CSR SpMV GFLOP/s (fp64, {g}^3 7-pt Poisson, row-partitioned, one rank per GPU)"


def attach_anchor(out, anchor, world):
    """`n1_same_run` = the same matrix on ONE GPU, measured by rank 0 on its own device before the ranks'
    timed regions of THIS run; `parallel_efficiency` = (value(N) / value(1)) / N for the SpMV and the same
    ratio of CG iterations/s (strong scaling: total work fixed).  Pure function of measured numbers
    (tests/test_bench_launch.py rehearses it over gloo)."""
    if anchor is None:
        out["n1_same_run"] = None
        out["parallel_efficiency"] = None
        return out
    n1 = {"workload": anchor.get("workload"), "gflops": anchor["spmv_gflops"], "spmv_us": anchor.get("spmv_us"),
          "spmv_frac_of_8tbs": anchor.get("spmv_frac_of_8tbs"),
          "cg_iters_per_sec": (anchor.get("cg") or {}).get("iters_per_sec"),
          "cg_iterations": (anchor.get("cg") or {}).get("iterations"),
          "measured_on": "rank 0's GPU, before the distributed timed regions of this run"}
    out["n1_same_run"] = n1
    eff = {"spmv": round(out["value"] / (world * n1["gflops"]), 4) if n1["gflops"] else None, "n_gpus": world,
           "definition": "(value(N) / value(1)) / N, value(1) = n1_same_run (strong scaling)"}
    cg = out.get("cg") or {}
    if n1["cg_iters_per_sec"] and cg.get("iters_per_sec"):
        eff["cg"] = round(cg["iters_per_sec"] / (world * n1["cg_iters_per_sec"]), 4)
    out["parallel_efficiency"] = eff
    return out


def rehearse_line(args, world, rank):
    """The N > 1 line's flow without a GPU (GKOMI_BENCH_BACKEND=gloo --rendezvous-only --rehearse-line): rank 0
    'measures' the one-GPU anchor while the others wait at the barrier, every rank contributes a region time,
    the max over ranks makes the value, rank 0 assembles the line with the functions the real run uses.  The
    numbers are placeholders (marked "rehearsal"); the structure is the real one."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("GKOMI_BENCH_BACKEND", "gloo"))
    g = args.p3_grid
    nnz_global = 7 * g ** 3 - 6 * g * g
    anchor = None
    if rank == 0:
        time.sleep(0.5)   # the other ranks are held at the barrier meanwhile
        anchor = {"workload": f"{g}^3 7-pt Poisson on one GPU (rehearsal)", "spmv_gflops": 1000.0, "spmv_us": 2e-3 * nnz_global,
                  "spmv_frac_of_8tbs": 0.5, "cg": {"iters_per_sec": 100.0, "iterations": 500}}
    dist.barrier()
    steps = max(10, args.steps // 4)
    t = torch.tensor([steps * 2e-9 * nnz_global / (world * 1000.0) * (1.0 + 0.1 * rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    out = {"metric": p3_metric(g), "value": round(2.0 * nnz_global * steps / wall / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world,
           "steps": steps, "warmup": max(3, args.warmup // 4), "ms_per_step": round(wall / steps * 1e3, 5),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "rehearsal": True, "cg": {"iters_per_sec": 100.0 * world * 0.8, "iterations": 500}}
    if rank == 0:
        print(json.dumps(attach_anchor(out, anchor, world)), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.cpu_baseline_only:
        cpu_baseline_child(args.cpu_seconds)
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)   # does not return
    if args.rendezvous_only and args.rehearse_line:
        rehearse_line(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))
        return
    if args.rendezvous_only:
        rendezvous_only(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
                        int(os.environ.get("LOCAL_RANK", "0")))
        return
    import torch
    # stdout carries exactly one JSON line: whatever libraries print on fd 1
    # (RCCL's version banner, ...) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("GKOMI_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist
        if not dist.is_initialized():
            if world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=device)
        barrier = lambda: dist.barrier(device_ids=[local_rank])
    else:
        barrier = lambda: None

    import gkomi
    import matgen
    gk = gkomi.lib()
    stream = torch.cuda.current_stream().cuda_stream
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)

    def timed_region(step_fn, steps, local=False):
        """Timed regions of EXACTLY `steps` steps each (barrier + synchronize on both sides, max over
        ranks per region): at least TRIALS of them, and as many more as it takes for the measured steps
        to cover --min-region-seconds (a 20-step region of an 18-us kernel is 0.4 ms: one host stall --
        ~10 ms every few hundred ms on this pool, tools/stall_probe.py -- or one clock step decides
        it).  Returns the MEDIAN region (what the line's value is computed from) and all of them."""
        regions = []
        budget = max(args.min_region_seconds, 0.0)
        while len(regions) < TRIALS or (sum(w for w, _ in regions) < budget and len(regions) < 400):
            wall, ev = time_loop(torch, step_fn, steps, (lambda: None) if local else barrier)
            if distributed and not local:
                t = torch.tensor([wall, ev], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall, ev = float(t[0].item()), float(t[1].item())
            regions.append((wall, ev))
        ordered = sorted(regions)
        return ordered[(len(ordered) - 1) // 2], regions

    def spread(regions, steps):
        us = sorted(w / steps * 1e6 for w, _ in regions)
        ev = sorted(e / steps * 1e6 for _, e in regions)
        q = lambda a, f: round(a[min(len(a) - 1, int(f * len(a)))], 3)
        return {"regions": len(us), "measured_seconds": round(sum(w for w, _ in regions), 4),
                "best_us_per_step": round(us[0], 3), "median_us_per_step": round(statistics.median(us), 3),
                "p90_us_per_step": q(us, 0.9), "worst_us_per_step": round(us[-1], 3),
                "median_event_us_per_step": round(statistics.median(ev), 3), "best_event_us_per_step": round(ev[0], 3)}

    def make_srow(rp_d, n, nnz):
        tile = int(gk.csr_srow_tile_for(nnz))
        t = torch.empty(int(gk.csr_srow_entries(nnz, tile)), dtype=torch.int32, device=device)
        gk.csr_make_srow_i32(stream, n, nnz, rp_d, tile, t, t.numel())
        return t, tile

    import gkomi.solvers as solvers

    import gkomi.formats as formats

    def as_csr(nn, a):
        """gko::matrix::Csr on the device: carries its srow and its longest row like the C++ mirror's Csr;
        the solver drivers get it as a gkomi_csr_ctx record -> the nonzero-split kernel (with the
        dot-product epilogue in the fused iterations), the kernel `value` is measured on"""
        return formats.Csr(gk, nn, nn, a[0], a[1], a[2])

    def timed_solves(fn):
        fn()
        runs = []
        for _ in range(3):  # median of 3 whole solves, all of them reported
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0, res))
        runs.sort(key=lambda r: r[0])
        el, res = runs[1]
        return res, el, [round(r[0], 5) for r in runs]

    def timed_cg(A, rhs, check_every=32, single_launch=True, precond=None):
        """single_launch=False: gkomi_cg_persistent_enable(0) -- the three-launch iteration (what every
        system beyond ~1M rows / 7 nonzeros per row, and every preconditioned solve, runs)"""
        gk.cg_persistent_enable(1 if single_launch else 0)
        try:
            return timed_solves(lambda: solvers.solve_op(gk, "cg", A, rhs, max_iters=50000, reduction=1e-10,
                                                         check_every=check_every, fused=True, precond=precond))
        finally:
            gk.cg_persistent_enable(1)

    def sinus_system(A):
        nn = A.nrows
        s = np.sin(np.arange(nn, dtype=np.float64))
        s /= np.linalg.norm(s)
        sb = dev(s.reshape(nn, 1))
        b = torch.empty((nn, 1), dtype=torch.float64, device=device)
        A.apply(sb, b)
        return sb, b

    def cg_entry(res, el, all_s):
        return {"iterations": res["iterations"], "seconds": round(el, 5), "all_seconds": all_s,
                "iters_per_sec": round(res["iterations"] / el, 1), "us_per_iteration": round(el / max(res["iterations"], 1) * 1e6, 2),
                "final_residual_norm_rel": res["rel_residual"], "converged": bool(res["converged"])}


    def p3_one_gpu(local=False):
        """BASELINE config 5's matrix (256^3 7-pt Poisson) on ONE GPU: SpMV + CG to 1e-10.  The N = 1 point of
        the strong-scaling curve: an entry of the N = 1 line ("p3") and, with local=True (no collective inside
        the timed regions), what rank 0 measures on its own device at the start of every N > 1 run
        ("n1_same_run"), so that each line carries its own anchor."""
        progress(f"P3: building the {args.p3_grid}^3 matrix on one GPU")
        g = args.p3_grid
        # the matrix is written on the device (gkomi_diag_poisson3d_7pt_f64_i32: row_ptrs in closed form; the same arrays
        # as tests/matgen.py poisson_3d_7pt, tests/test_csr_i64_gpu.py::test_device_generated_poisson_matrix_equals_the_host_one)
        n3, nnz3 = g ** 3, 7 * g ** 3 - 6 * g * g
        a3 = [torch.empty(n3 + 1, dtype=torch.int32, device=device), torch.empty(nnz3, dtype=torch.int32, device=device),
              torch.empty(nnz3, dtype=torch.float64, device=device)]
        gk.diag_poisson3d_7pt_f64_i32(stream, g, a3[0], a3[1], a3[2])
        x3 = dev(np.sin(0.01 * np.arange(n3)).reshape(n3, 1))
        y3 = torch.empty((n3, 1), dtype=torch.float64, device=device)
        srow3, tile3 = make_srow(a3[0], n3, nnz3)
        step3 = lambda i: gk.csr_spmv_srow_f64_i32(stream, n3, n3, 1, nnz3, a3[0], a3[1], a3[2], x3, 1, y3, 1,
                                                   None, None, 0, 7, srow3, tile3)
        for i in range(5):
            step3(i)
        steps3 = max(10, args.steps // 10)
        (w3, e3), r3 = timed_region(step3, steps3, local=local)
        b3 = algorithmic_bytes(n3, n3, nnz3)
        p3 = {"workload": f"{g}^3 7-pt Poisson (n={n3}, nnz={nnz3}) on one GPU: the N=1 point of the N>1 lines",
              "spmv_gflops": round(2.0 * nnz3 * steps3 / w3 / 1e9, 2), "spmv_us": round(e3 / steps3 * 1e6, 2),
              "spmv_gbs": round(b3 * steps3 / e3 / 1e9, 1), "spmv_frac_of_8tbs": round(b3 * steps3 / e3 / 1e9 / HBM_PEAK_GBS, 4),
              "timing_spread": spread(r3, steps3)}
        if not args.no_cg:
            A3 = as_csr(n3, a3)
            sb3, bb3 = sinus_system(A3)
            res, el, all_s = timed_cg(A3, bb3)
            # per iteration (three launches): K1 3n + K2 (matrix + 2n) + K3 6n values
            cg3_bytes = 11 * 8 * n3 + 12 * nnz3 + 4 * (n3 + 1)
            p3["cg"] = dict(cg_entry(res, el, all_s), achieved_gbs=round(cg3_bytes * res["iterations"] / el / 1e9, 1),
                            frac_of_8tbs=round(cg3_bytes * res["iterations"] / el / 1e9 / HBM_PEAK_GBS, 4),
                            bytes_per_iteration=cg3_bytes,
                            solution_rel_err=float(torch.linalg.norm(res["x"] - sb3) / torch.linalg.norm(sb3)))
            p3["cg_rhs_ones"] = cg_entry(*timed_cg(A3, torch.ones((n3, 1), dtype=torch.float64, device=device)))
            del A3
        del a3, x3, y3, srow3
        torch.cuda.empty_cache()
        return p3

    out = {}
    if not distributed:
        # ---------------- N = 1: P2, configs[1] ----------------
        progress("P2: building the 1M-row matrix, 8 copies")
        n, rp, ci, v = matgen.poisson_2d_5pt(GRID)
        nnz = int(rp[-1])
        x_host = np.sin(0.01 * np.arange(n)).reshape(n, 1)
        bytes_per_launch = algorithmic_bytes(n, n, nnz)
        flops_per_launch = 2.0 * nnz
        ncopies = 8  # a copy's lines are evicted from the 256 MiB Infinity Cache before its next use
        copies = []
        for _ in range(ncopies):
            c = [dev(rp), dev(ci), dev(v), dev(x_host), torch.empty((n, 1), dtype=torch.float64, device=device)]
            c += list(make_srow(c[0], n, nnz))   # Csr::make_srow: part of the matrix, like its row_ptrs
            copies.append(c)
        # automatic strategy, no caller flag: what Csr::apply through the reference's interface (shims/hip/matrix/
        # csr_kernels.hip.cpp) passes.  The library's own residency tracker (csr_probably_evicted) sees 560 MB of other
        # matrices go by between two applies of a copy and reads the matrix with nontemporal loads by itself.
        cold_strategy = args.strategy if args.strategy else 0

        def launch(c, strategy):
            gk.csr_spmv_srow_f64_i32(stream, n, n, 1, nnz, c[0], c[1], c[2], c[3], 1, c[4], 1, None, None,
                                     strategy, 5, c[5], c[6])

        step_cold = lambda i: launch(copies[i % ncopies], cold_strategy)
        step_warm = lambda i: launch(copies[0], 0)
        progress("P2: timing the cold SpMV")
        for i in range(args.warmup):
            step_cold(i)
        (wall, ev), regions = timed_region(step_cold, args.steps)
        out = {
            "metric": "CSR SpMV GFLOP/s (fp64, 1M-row 5-pt Poisson per GPU)",
            "value": round(flops_per_launch * args.steps / wall / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 5),
            "timing": f"median of {len(regions)} timed regions of {args.steps} steps each "
                      f"(>= {args.min_region_seconds} s of measured steps)", "timing_spread": spread(regions, args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "benchmark/spmv: CSR fp64/int32 y=Ax on 1000x1000 5-pt Poisson "
                                   "(n=1e6, nnz=4996000), x=sin(0.01 i)",
                       "cache_state": f"cold: rotating over {ncopies} copies (> 256 MiB Infinity Cache)",
                       "partition": "one GPU", "strategy": cold_strategy,
                       "strategy_note": "automatic (0), as Csr::apply through the reference interface passes it: no caller "
                                        "flag; the library's residency tracker finds > 256 MiB of other CSR applies between "
                                        "two applies of a copy and selects the nontemporal streams itself; the matrix "
                                        "carries its srow (Csr::make_srow)"},
        }
        kern_s = ev / args.steps
        achieved = bytes_per_launch / kern_s / 1e9
        traffic, traffic_file = traffic_from_profiles()
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_source": (f"{traffic_file}: rocprofv3 --pmc passes of this command "
                                              "(tools/profile.sh), not re-measured in this run") if traffic_file else None,
                           "kernel": "csr_split_kernel", "bytes_per_launch": bytes_per_launch,
                           "us_per_launch": round(kern_s * 1e6, 3),
                           "clock": "HIP events on the launch stream around the K steps of the median region "
                                    "(achieved, frac, us_per_launch); value / ms_per_step use the host wall clock of "
                                    "the same region",
                           "frac_by_wall_clock": round(bytes_per_launch * args.steps / wall / 1e9 / HBM_PEAK_GBS, 4)}
        # the practical ceiling of THIS box for the same bytes: a pure 16-B streaming kernel over the same
        # rotating copies (no gather, no LDS, no dependent access), best grid of four, same region policy
        progress("P2: streaming ceiling for the same byte mix")
        ceiling = {}
        for blocks in (1024, 2048, 4096, 8192):
            stepm = lambda i, blocks=blocks: gk.diag_stream_csr_bytes(stream, blocks, n, nnz, copies[i % ncopies][0],
                                                                      copies[i % ncopies][1], copies[i % ncopies][2],
                                                                      copies[i % ncopies][3], copies[i % ncopies][4])
            for i in range(ncopies):
                stepm(i)
            (_, mev), mreg = timed_region(stepm, args.steps)
            ceiling[blocks] = mev / args.steps * 1e6
        best_blocks = min(ceiling, key=ceiling.get)
        out["roofline"]["practical_ceiling_us"] = round(ceiling[best_blocks], 3)
        out["roofline"]["practical_ceiling_gbs"] = round(bytes_per_launch / ceiling[best_blocks] / 1e3, 1)
        out["roofline"]["frac_of_practical_ceiling"] = round(ceiling[best_blocks] / (kern_s * 1e6), 4)
        out["roofline"]["practical_ceiling_note"] = (
            f"gkomi_diag_stream_csr_bytes ({best_blocks} workgroups; all grids: "
            + ", ".join(f"{b}: {t:.2f} us" for b, t in ceiling.items())
            + "): the same 79 952 004 B as pure coalesced 16-B streams, cold over the same copies, same run")
        for c in copies:   # the probe overwrote y
            launch(c, cold_strategy)
        for i in range(args.warmup):
            step_warm(i)
        (wwall, wev), wregions = timed_region(step_warm, args.steps)
        out["warm"] = {"gflops": round(flops_per_launch * args.steps / wwall / 1e9, 2),
                       "gbs": round(bytes_per_launch * args.steps / wev / 1e9, 1),
                       "us_per_launch": round(wev / args.steps * 1e6, 3), "timing_spread": spread(wregions, args.steps),
                       "note": "same matrix every step (benchmark/spmv methodology), automatic strategy; the 80 MB "
                               "working set is Infinity-Cache resident"}

        progress("P2: CG solves")
        if not args.no_cg:
            A2 = as_csr(n, copies[0])
            sb, b = sinus_system(A2)
            cg_bytes = 11 * 8 * n + (12 * nnz + 4 * (n + 1))  # per iteration: 11 n values + matrix
            before = gk.cg_persistent_solves()
            res, el, all_s = timed_cg(A2, b)
            single_launch = gk.cg_persistent_solves() > before
            res3, el3, all_s3 = timed_cg(A2, b, single_launch=False)
            # steady state of the three-launch iteration: 400 iterations that cannot converge (reduction 1e-30), so that
            # set-up (r = b - A x, baseline norm) and the launches issued after the stop do not weigh on the figure
            gk.cg_persistent_enable(0)
            try:
                r400, el400, _ = timed_solves(lambda: solvers.solve_op(gk, "cg", A2, b, max_iters=400, reduction=1e-30,
                                                                       check_every=32, fused=True))
            finally:
                gk.cg_persistent_enable(1)
            out["cg"] = {"metric": "CG iters/sec to 1e-10 (fused driver, Identity preconditioner, sinus rhs "
                                   "b = A s/|s|, benchmark/solver default)",
                         "driver": ("single launch: x, r, p and the matrix (rows <= 5 nonzeros) stay in the register "
                                    "files, three device-wide meetings per iteration (csrc/cg_persistent.hpp)")
                         if single_launch else "three launches per iteration",
                         "three_launch_driver": dict(cg_entry(res3, el3, all_s3),
                                                     steady_state_us_per_iteration=round(el400 / max(r400["iterations"], 1) * 1e6, 2),
                                                     steady_state_note="400 iterations with reduction 1e-30: no set-up share, no launches after the stop",
                                                     spmv_kernel="csr_split_kernel<Dot> over the matrix's srow",
                                                     achieved_gbs=round(cg_bytes * res3["iterations"] / el3 / 1e9, 1)),
                         "timing": "median of 3 solves",
                         # the byte model (11 n values + matrix per iteration) belongs to the three-launch
                         # iteration; the single-launch one moves 2 n values (p out, p gathered) per iteration
                         "achieved_gbs": None if single_launch else round(cg_bytes * res["iterations"] / el / 1e9, 1),
                         "solution_rel_err": float(torch.linalg.norm(res["x"] - sb) / torch.linalg.norm(sb))}
            out["cg"].update(cg_entry(res, el, all_s))
            ones = torch.ones((n, 1), dtype=torch.float64, device=device)
            out["cg_rhs_ones"] = cg_entry(*timed_cg(A2, ones))
            launch(copies[0], cold_strategy)
            torch.cuda.synchronize()
        got_p2 = copies[0][4].cpu().numpy().copy()

        if not args.no_p3:
            del copies[1:]
            out["p3"] = p3_one_gpu()

        if not args.no_config4:
            # BASELINE config 4's shape (GMRES(30) + ParILU on a 1.26M-row 7-point convection-diffusion system):
            # an extra entry, never allowed to take the headline line down with it
            try:
                progress("config 4: GMRES(30) with and without ParILU")
                import gkomi.solvers as solvers
                n4, rp4, ci4, v4 = matgen.at_like(108)
                a4 = [dev(rp4), dev(ci4), dev(v4)]
                A4 = as_csr(n4, a4)
                b4 = dev(np.cos(0.3 * np.arange(n4)).reshape(n4, 1))
                pre4 = solvers.par_ilu_generate(gk, n4, a4[0].clone(), a4[1], a4[2], iterations=0)   # warm-up (first call
                del pre4                                                                        # pays one-time set-up)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                # iterations=0: the reference's default, 10 sweeps on this backend (hip/factorization/par_ilu_kernels.hip.cpp:72)
                pre4 = solvers.par_ilu_generate(gk, n4, a4[0].clone(), a4[1], a4[2], iterations=0)
                torch.cuda.synchronize()
                gen4 = time.perf_counter() - t0
                c4 = {"workload": f"AT-like 108^3 7-pt convection-diffusion (n={n4}), rhs cos(0.3 i), reduction 1e-10",
                      "parilu_generate_incl_trs_analysis_ms": round(gen4 * 1e3, 1), "parilu_sweeps": "ParIlu factory default (iterations = 0 -> 10 sweeps, hip/factorization/par_ilu_kernels.hip.cpp:72)",
                      "trs_plan": ["bricks" if p is not None else "levels" for p in (pre4.l_bricks, pre4.u_bricks)]}
                for name, pc in (("gmres30", None), ("gmres30_parilu", pre4)):
                    best = None
                    for _ in range(2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        r4 = solvers.solve_op(gk, "gmres", A4, b4, krylov_dim=30, max_iters=3000, reduction=1e-10,
                                              precond=pc)
                        torch.cuda.synchronize()
                        el = time.perf_counter() - t0
                        best = el if best is None else min(best, el)
                    c4[name] = {"iterations": r4["iterations"], "ms": round(best * 1e3, 2), "converged": bool(r4["converged"]),
                                "us_per_iteration": round(best / max(r4["iterations"], 1) * 1e6, 1)}
                # benchmark/utils/preconditioners.hpp:58 --parilu_iterations defaults to 5: the asynchronous sweeps have
                # not converged then and the iteration count varies from run to run (90-112)
                try:
                    pre5 = solvers.par_ilu_generate(gk, n4, a4[0].clone(), a4[1], a4[2], iterations=5)
                    runs5 = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        r5 = solvers.solve_op(gk, "gmres", A4, b4, krylov_dim=30, max_iters=3000, reduction=1e-10, precond=pre5)
                        torch.cuda.synchronize()
                        runs5.append((time.perf_counter() - t0, r5["iterations"]))
                    el5, it5 = min(runs5)
                    c4["gmres30_parilu_5_sweeps"] = {"iterations": it5, "ms": round(el5 * 1e3, 2), "converged": bool(r5["converged"]),
                                                     "note": "the benchmark's flag default; best of 3 solves on one factorisation"}
                    del pre5
                except Exception as e5:  # noqa: BLE001
                    c4["gmres30_parilu_5_sweeps"] = {"error": repr(e5)}
                if pre4.l_bricks is not None and pre4.u_bricks is not None:
                    y4 = torch.zeros_like(b4)
                    z4 = torch.zeros_like(b4)
                    for _ in range(3):
                        pre4.l_bricks.solve(b4, y4)
                        pre4.u_bricks.solve(y4, z4)
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    e0.record()
                    for _ in range(20):
                        pre4.l_bricks.solve(b4, y4)
                    e1.record()
                    for _ in range(20):
                        pre4.u_bricks.solve(y4, z4)
                    e2.record()
                    torch.cuda.synchronize()
                    c4["trs_us"] = {"lower": round(e0.elapsed_time(e1) * 1e3 / 20, 1), "upper": round(e1.elapsed_time(e2) * 1e3 / 20, 1)}
                out["config4"] = c4
                del a4, b4, pre4, A4
            except Exception as e:  # noqa: BLE001
                out["config4"] = {"error": repr(e)}

        if not args.no_config3:
            # BASELINE config 3 (benchmark/solver: CG + block-Jacobi(32) on thermal2, 1.2 M rows) on SURVEY 8(d)'s
            # stand-ins; what benchmark/solver/solver.cpp:488-534 reports: generate time, apply time, iterations --
            # plus the apply's roofline (8 x stored block elements + 4 (#blocks + 1) + 16 n bytes)
            try:
                import gkomi.solvers as solvers
                c3 = {"preconditioner": "preconditioner::Jacobi, max_block_size 32 (benchmark default), fp64 blocks",
                      "rhs": "b = 1, x0 = 0, reduction 1e-10 (rhs_norm)"}
                for name, gen in (("t2_like_permuted_1108", lambda: matgen.t2_like_permuted(1108)),
                                  ("diffusion_patch_ordered_1104", lambda: matgen.diffusion_2d_patch_ordered(1104))):
                    progress(f"config 3: {name}")
                    n5, rp5, ci5, v5 = gen()
                    a5 = [dev(rp5), dev(ci5), dev(v5)]
                    nnz5 = int(rp5[-1])
                    del rp5, ci5, v5
                    A5 = as_csr(n5, a5)
                    solvers.jacobi_generate(gk, n5, a5[0], a5[1], a5[2], max_block_size=32)   # first call: one-time set-up
                    gens = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        pre5 = solvers.jacobi_generate(gk, n5, a5[0], a5[1], a5[2], max_block_size=32)
                        torch.cuda.synchronize()
                        gens.append(time.perf_counter() - t0)
                    sizes = torch.diff(pre5.block_ptrs[:pre5.num_blocks + 1]).to(torch.int64)
                    stored = int(torch.sum(sizes * sizes).item())
                    jac_bytes = 8 * stored + 4 * (pre5.num_blocks + 1) + 16 * n5
                    b5 = torch.ones((n5, 1), dtype=torch.float64, device=device)
                    z5 = torch.empty_like(b5)
                    stepj = lambda i: pre5.apply(b5, z5)
                    for i in range(5):
                        stepj(i)
                    (_, jev), jreg = timed_region(stepj, 50)
                    e = {"n": n5, "nnz": nnz5, "num_blocks": pre5.num_blocks,
                         "stored_block_elements": stored,
                         "generate_ms": round(statistics.median(gens) * 1e3, 3), "generate_all_ms": [round(g * 1e3, 3) for g in gens],
                         "apply": {"us": round(jev / 50 * 1e6, 2), "bytes": jac_bytes, "gbs": round(jac_bytes / (jev / 50) / 1e9, 1),
                                   "frac_of_8tbs": round(jac_bytes / (jev / 50) / 1e9 / HBM_PEAK_GBS, 4),
                                   "kernel": "jacobi_apply_kernel", "timing_spread": spread(jreg, 50)}}
                    res, el, all_s = timed_cg(A5, b5, precond=pre5)
                    e["cg_jacobi"] = dict(cg_entry(res, el, all_s), time_to_1e_10_ms=round(el * 1e3, 2))
                    gk.cg_persistent_enable(0)   # plain CG on the same footing: three launches per iteration
                    try:
                        res, el, all_s = timed_solves(lambda: solvers.solve_op(gk, "cg", A5, b5, max_iters=20000, reduction=1e-10,
                                                                               check_every=32, fused=True))
                    finally:
                        gk.cg_persistent_enable(1)
                    e["cg_plain"] = dict(cg_entry(res, el, all_s), time_to_1e_10_ms=round(el * 1e3, 2) if res["converged"] else None,
                                         note=None if res["converged"] else "not converged within 20000 iterations")
                    c3[name] = e
                    del a5, A5, pre5, b5, z5
                out["config3"] = c3
            except Exception as e:  # noqa: BLE001 - an extra entry, never allowed to take the headline down
                out["config3"] = {"error": repr(e)}

        if not args.no_irregular:
            # north_star names SuiteSparse; its graph-like matrices are this class (no files offline: stand-ins of
            # tools/spmv_classes.json): the automatic CSR strategy and the opt-in column-partitioned copy (csrp)
            try:
                progress("scattered-column classes (uniform random, power-law rows)")
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import benchmark_spmv as bs
                irr = {"note": "1 M x 1 M, same matrix every step (benchmark/spmv methodology); bytes = 12 nnz + 4 (n + 1) + 16 n; "
                               "csrp = gkomi_csr_colpart_* (opt-in analysis-based strategy, tolerance parity)"}
                for key, case in (("uniform_random_16_per_row", {"random": "uniform", "rows": 1000000, "nnz_per_row": 16}),
                                  ("powerlaw_rows", {"random": "powerlaw", "rows": 1000000, "nnz_per_row": 8})):
                    Mi = bs.random_matrix(gk, case, 42)
                    bi = dev(np.cos(0.001 * np.arange(Mi.ncols)).reshape(-1, 1))
                    yi = torch.zeros((Mi.nrows, 1), dtype=torch.float64, device=device)
                    ent = {"nnz": Mi.nnz}
                    ref_y = None
                    for fmt in ("csr", "csrp"):
                        Mf = Mi.to(fmt)
                        Mf.apply(bi, yi)
                        if fmt == "csr":
                            ref_y = yi.clone()
                        else:
                            ent["csrp_blocks"] = None
                            if Mf.colpart() is not None:
                                info = (ctypes.c_int64 * 8)()
                                gk.csr_colpart_info(Mf._colpart[0], ctypes.addressof(info))
                                ent["csrp_blocks"] = int(info[0])
                                ent["csrp_virtual_matrix_kernel"] = {0: "automatic (nonzero-split / load-balanced)", 1: "row-cut stream"}.get(int(info[4]), int(info[4]))
                            ent["csrp_max_rel_diff_vs_csr"] = float(((yi - ref_y).abs().max() / ref_y.abs().max()).item())
                        (_, evi), _ = timed_region(lambda i: Mf.apply(bi, yi), 40)
                        byts = 12 * Mi.nnz + 4 * (Mi.nrows + 1) + 16 * Mi.nrows
                        ent[fmt] = {"us": round(evi / 40 * 1e6, 1), "gbs": round(byts * 40 / evi / 1e9, 1),
                                    "frac_of_8tbs": round(byts * 40 / evi / 1e9 / HBM_PEAK_GBS, 4)}
                        del Mf
                    irr[key] = ent
                    del Mi, bi, yi, ref_y
                    torch.cuda.empty_cache()
                out["irregular"] = irr
            except Exception as e:  # noqa: BLE001 - an extra entry, never allowed to take the headline down
                out["irregular"] = {"error": repr(e)}

        if not args.no_cpu_baseline:
            progress("CPU baseline (3 child processes)")
            base = cpu_baseline(args.cpu_seconds)
            out["cpu_baseline"] = base
            # end-of-run parity check of what was timed, against the oracle (reference/ SpMV)
            import oracle_lib
            orc = oracle_lib.load()
            y_ref = np.empty((n, 1))
            orc.ref_csr_spmv(n, 1, rp, ci, v, x_host, 1, y_ref, 1)
            out["parity_vs_oracle"] = "bit-exact" if np.array_equal(got_p2, y_ref) else \
                f"rel err {matgen.rel_err(got_p2, y_ref):.3e}"
    else:
        # ---------------- N > 1: P3, configs[4], strong scaling ----------------
        import gkomi.distributed as gd
        g = args.p3_grid
        n_global = g ** 3
        # the curve's own N = 1 point: the whole matrix on rank 0's GPU (1.7 GB), SpMV + CG, before anything
        # distributed is built; the other ranks wait at the barrier
        anchor = None
        if rank == 0 and not args.no_n1_anchor:
            try:
                anchor = p3_one_gpu(local=True)
            except Exception as ex:  # noqa: BLE001 - the line says so instead of dying
                anchor = None
                progress(f"one-GPU anchor failed: {ex!r}")
        barrier()
        part = gd.Partition.build_from_global_size_uniform(gk, world, n_global)
        lo, hi = int(part.range_bounds[rank]), int(part.range_bounds[rank + 1])
        rows, cols, vals = gd.poisson3d_rows(g, lo, hi)
        nnz_local = len(vals)
        nnz_global = 7 * n_global - 6 * g * g
        if rank == 0:
            progress(f"P3 {g}^3 over {world} ranks: building the distributed matrix")
        M = gd.Matrix(gd.GpuOps(gk, device)).read_distributed(rows, cols, vals, part)
        del rows, cols, vals
        n_loc = M.num_local_rows
        driver = "python: torch.distributed collectives (no RCCL handle for the native driver)"
        comm = A = None
        rccl_ranks = rccl_rank = None
        try:
            comm = gd.RcclComm(gk, device)
            A = gd.NativeMatrix(M)
            rccl_ranks, rccl_rank = comm.query()
            driver = "native: csrc/dist_cg.hip over its own RCCL communicator"
        except Exception as ex:  # noqa: BLE001 - the portable path takes over, and says so
            driver += f" [{repr(ex)[:120]}]"
            comm = A = None
        x = dev(np.sin(0.01 * np.arange(lo, hi)).reshape(n_loc, 1))
        y = torch.empty((n_loc, 1), dtype=torch.float64, device=device)
        if A is not None:
            step = lambda i: A.apply(comm, x, y)
        else:
            step = lambda i: M.apply(x, y)
        for i in range(max(3, args.warmup // 4)):
            step(i)
        steps = max(10, args.steps // 4)
        (wall, ev), regions = timed_region(step, steps)
        local_bytes = algorithmic_bytes(n_loc, n_loc + M.non_local[1], nnz_local)
        out = {
            "metric": p3_metric(g),
            "value": round(2.0 * nnz_global * steps / wall / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": steps, "warmup": max(3, args.warmup // 4), "ms_per_step": round(wall / steps * 1e3, 5),
            "timing": f"median of {len(regions)} timed regions of {steps} steps each", "timing_spread": spread(regions, steps),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"distributed-solver: row-partitioned y=Ax and CG on {g}^3 7-pt Poisson "
                                   f"(n={n_global}, nnz={nnz_global}), {world} contiguous row slabs, RCCL halo exchange",
                       "partition": f"{world} row slabs of {n_global // world} rows; halo "
                                    f"{M.recv_count} doubles in / {M.send_count} out on rank {rank}",
                       "driver": driver, "rccl_ranks": rccl_ranks, "rccl_rank_of_rank0": rccl_rank,
                       "torch_distributed_world_size": dist.get_world_size()},
            "roofline": {"bound": "hbm", "achieved": round(local_bytes * steps / ev / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(local_bytes * steps / ev / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "distributed apply (rank 0's share: local block + non-local rows)",
                         "bytes_per_launch": local_bytes, "us_per_launch": round(ev / steps * 1e6, 3)},
        }
        if not args.no_cg:
            # sinus right-hand side of benchmark/solver: b = A s / |s|, s_i = sin(i)
            s_loc = np.sin(np.arange(lo, hi, dtype=np.float64))
            nrm = torch.tensor([float(np.sum(s_loc * s_loc))], dtype=torch.float64, device=device)
            dist.all_reduce(nrm)
            sd = dev((s_loc / np.sqrt(float(nrm.item()))).reshape(n_loc, 1))
            b = torch.empty((n_loc, 1), dtype=torch.float64, device=device)
            (A.apply(comm, sd, b) if A is not None else M.apply(sd, b))
            torch.cuda.synchronize()

            if rank == 0:
                progress("distributed CG solves")
            def solve(rhs):
                xs = torch.zeros((n_loc, 1), dtype=torch.float64, device=device)
                if A is not None:
                    r = A.cg(comm, rhs, xs, max_iters=100000, reduction=1e-10, check_every=32)
                    return xs, r["iterations"], r["converged"], r["residual_norm"] / max(r["baseline_norm"], 1e-300)
                it, conv = gd.cg_fused(M, rhs, xs, max_iters=100000, reduction=1e-10, check_every=32)
                return xs, it, conv, None

            def timed_solve(rhs):
                solve(rhs)
                runs = []
                for _ in range(3):
                    barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    xs, its, conv, rel = solve(rhs)
                    torch.cuda.synchronize()
                    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    runs.append(float(t.item()))
                el = min(runs)
                return xs, {"iterations": int(its), "converged": bool(conv), "seconds": round(el, 5),
                            "all_seconds": [round(r, 5) for r in runs], "timing": "best of 3 solves, max over ranks",
                            "iters_per_sec": round(its / el, 1), "final_residual_norm_rel": rel, "global_rows": n_global}

            xs, res = timed_solve(b)
            err = torch.stack([torch.sum((xs - sd) ** 2), torch.sum(sd ** 2)])
            dist.all_reduce(err)
            res["metric"] = "row-partitioned CG to 1e-10 (no iteration cap), sinus rhs b = A s/|s| (benchmark/solver default)"
            res["solution_rel_err"] = float(torch.sqrt(err[0] / err[1]).item())
            out["cg"] = res
            _, res1 = timed_solve(torch.ones((n_loc, 1), dtype=torch.float64, device=device))
            res1["metric"] = "row-partitioned CG to 1e-10 (no iteration cap), b = 1"
            out["cg_rhs_ones"] = res1
        if A is not None:
            A.close()
            comm.close()
        attach_anchor(out, anchor, world)

    if rank == 0:
        progress("done")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
