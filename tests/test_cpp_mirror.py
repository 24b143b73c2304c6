"""The C++ host mirror of the reference interface (repo-8852-ginkgo_amd/include/
ginkgo/ginkgo.hpp) over the C ABI: API compatibility with the reference's own
example source (compiled here when /root/reference is mounted; never copied),
host-side behaviour on CPU, and on the GPU the documented simple-solver result
plus a tour of formats, preconditioners and solvers."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "repo-8852-ginkgo_amd")
EX = os.path.join(PKG, "examples")
REF_EXAMPLE = "/root/reference/examples/simple-solver/simple-solver.cpp"


@pytest.fixture(scope="module")
def bins():
    r = subprocess.run(["make", "-C", EX], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(EX, "bin")


def _data_dir(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    for name in ("A", "b", "x0"):
        shutil.copy(os.path.join(HERE, "golden", f"simple_solver_{name}.mtx"), d / f"{name}.mtx")
    return tmp_path


def test_host_api(bins):
    r = subprocess.run([os.path.join(bins, "host_api_test")], capture_output=True, text=True)
    assert r.returncode == 0 and "host api ok" in r.stdout, r.stdout + r.stderr


def test_host_executor_raises_not_compiled(bins, tmp_path):
    r = subprocess.run([os.path.join(bins, "simple_solver"), "reference"], cwd=_data_dir(tmp_path), capture_output=True, text=True)
    assert r.returncode == 3 and "NotCompiled" in r.stderr


@pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="reference tree not mounted")
def test_reference_simple_solver_source_compiles_unchanged(tmp_path):
    """The reference's examples/simple-solver/simple-solver.cpp, read where it
    lies, compiles and links against the mirror + libgkomi.so."""
    out = tmp_path / "ref_simple_solver"
    r = subprocess.run(["g++", "-std=c++14", "-Wall", f"-I{PKG}/include", REF_EXAMPLE, "-o", str(out),
                        f"-L{PKG}/lib", "-lgkomi", f"-Wl,-rpath,{PKG}/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(out)


@pytest.mark.gpu
def test_simple_solver_on_hip_matches_documented_result(bins, tmp_path):
    g = json.load(open(os.path.join(HERE, "golden", "cg.json")))["simple_solver"]
    r = subprocess.run([os.path.join(bins, "simple_solver"), "hip"], cwd=_data_dir(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    i = lines.index("Solution (x):")
    assert lines[i + 1].startswith("%%MatrixMarket matrix array real general") and lines[i + 2].split() == ["19", "1"]
    x = np.array([float(t) for t in lines[i + 3:i + 22]])
    # results.dox prints 6 significant digits (operator<< default), and so does the mirror
    assert np.array_equal(x, np.array(g["expect_x"]))
    j = lines.index("Residual norm sqrt(r^T r):")
    res = float(lines[j + 3])
    assert res < 1e-13  # documented: 2.10788e-15 (depends on the reduction order at this level)
    assert int(lines[-1].split()[-1]) <= g["max_iters"]


@pytest.mark.gpu
def test_hot_path_tour(bins):
    r = subprocess.run([os.path.join(bins, "hot_path_tour")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    kv = {}
    for line in r.stdout.splitlines():
        t = line.split()
        kv[t[0]] = t[1:]
    assert float(kv["ell_diff"][0]) == 0.0 and float(kv["sellp_diff"][0]) == 0.0
    assert float(kv["coo_diff"][0]) < 1e-12 and float(kv["hybrid_diff"][0]) < 1e-12
    assert int(kv["hybrid_diff"][2]) > 0
    assert kv["cg_jacobi_iters"][2] == "1" and float(kv["cg_jacobi_iters"][4]) < 1e-9
    # adaptive block storage: some blocks reduced, same convergence within a few iterations
    assert kv["cg_adaptive_jacobi_iters"][2] == "1" and int(kv["cg_adaptive_jacobi_iters"][4]) > 0
    assert abs(int(kv["cg_adaptive_jacobi_iters"][0]) - int(kv["cg_jacobi_iters"][0])) <= 5
    assert kv["gmres_ilu_iters"][2] == "1" and float(kv["gmres_ilu_iters"][4]) < 1e-9
    assert int(kv["gmres_ilu_iters"][0]) < 200
    for name in ("gmres_ilu_sellp_iters", "gmres_ilu_ell_iters"):   # config 4: other formats as system matrix
        assert kv[name][2] == "1" and float(kv[name][4]) < 1e-9
    for name in ("fcg_jacobi_iters", "bicgstab_ilu_iters", "cgs_ilu_iters"):
        assert kv[name][2] == "1" and float(kv[name][4]) < 1e-8, (name, kv[name])
    assert kv["fcg_jacobi_iters"][0] == kv["cg_jacobi_iters"][0]   # FCG = CG in exact arithmetic on an SPD matrix
    assert kv["cg_ic_iters"][2] == "1" and float(kv["cg_ic_iters"][4]) < 1e-8
    # (asynchronous sweeps: the factor, hence the count, varies a little from run to run)
    assert int(kv["cg_ic_iters"][0]) < 0.7 * int(kv["cg_jacobi_iters"][0])
    assert kv["bicg_iters"][2] == "1" and float(kv["bicg_iters"][4]) < 1e-8
    # Jacobi::transpose gives BiCG its transposed preconditioner: fewer iterations
    assert kv["bicg_jacobi_iters"][2] == "1" and float(kv["bicg_jacobi_iters"][4]) < 1e-8
    assert int(kv["bicg_jacobi_iters"][0]) < int(kv["bicg_iters"][0])
    assert kv["bicg_ilu_iters"][2] == "1" and float(kv["bicg_ilu_iters"][4]) < 1e-8
    assert int(kv["bicg_ilu_iters"][0]) < int(kv["bicg_jacobi_iters"][0])
    # IR stops at its iteration limit or at the loose goal; either way the residual it reports is the true one
    assert float(kv["ir_jacobi_iters"][4]) < 1.0 and (kv["ir_jacobi_iters"][2] == "1") == (float(kv["ir_jacobi_iters"][4]) < 1e-2)
    # device assembly: duplicates summed, explicit zeros dropped, Csr::read on the device
    assert kv["assembly_nnz"][0] == kv["assembly_nnz"][2] and float(kv["assembly_nnz"][4]) == 0.0
    assert kv["dimension_check"] == ["ok"]
