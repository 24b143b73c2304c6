"""bench.py --gpus N must start its own ranks when no launcher did: child
processes through torch.distributed.run, before any GPU call, one JSON line from rank 0, a clear
non-zero exit when the node has fewer GPUs.  Rehearsed here over gloo without touching a GPU
(reference flow: examples/distributed-solver/distributed-solver.cpp -- one rank per device)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks_and_they_rendezvous():
    r = _run(["--gpus", "2", "--rendezvous-only"], {"GKOMI_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rendezvous"] == 2 and out["world_size"] == 2 and out["n_gpus"] == 2 and out["ranks"] == [0, 1]


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--rendezvous-only"], {})
    assert r.returncode == 2
    assert f"--gpus {have + 2} asked for, {have} GPU(s) visible" in r.stderr


def test_bench_refuses_a_launcher_with_the_wrong_world_size():
    # an outer launcher that started 3 ranks for --gpus 2: say so instead of running something else
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_the_multi_gpu_line_carries_its_own_one_gpu_anchor():
    """every N > 1 line has `n1_same_run` (rank 0's one-GPU measurement of the same
    matrix, taken while the other ranks wait) and `parallel_efficiency`; `metric` and `scaling` are the same
    strings for every N.  The flow is rehearsed over gloo with placeholder numbers; attach_anchor is the
    function the real run uses."""
    lines = {}
    for n in (2, 3):
        r = _run(["--gpus", str(n), "--rendezvous-only", "--rehearse-line", "--p3-grid", "64"], {"GKOMI_BENCH_BACKEND": "gloo"})
        assert r.returncode == 0, r.stderr[-2000:]
        out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(out) == 1, r.stdout
        lines[n] = out[0]
    for n, out in lines.items():
        assert out["n_gpus"] == n and out["scaling"] == "strong" and out["rehearsal"] is True
        assert out["n1_same_run"]["gflops"] == 1000.0 and out["n1_same_run"]["cg_iters_per_sec"] == 100.0
        eff = out["parallel_efficiency"]
        # the slowest rank (max over ranks) is 1 + 0.1 (n - 1) slower than ideal
        assert abs(eff["spmv"] - 1.0 / (1.0 + 0.1 * (n - 1))) < 1e-3 and abs(eff["cg"] - 0.8) < 1e-9
    assert lines[2]["metric"] == lines[3]["metric"] and lines[2]["unit"] == lines[3]["unit"]


def test_attach_anchor_without_an_anchor_says_so():
    sys.path.insert(0, ROOT)
    import bench
    out = bench.attach_anchor({"value": 5.0}, None, 4)
    assert out["n1_same_run"] is None and out["parallel_efficiency"] is None
