"""The benchmark harness (tools/benchmark_{spmv,solver}.py, the reference's
benchmark/spmv + benchmark/solver JSON layout) runs end to end on the GPU: the
reference's own example matrix from a MatrixMarket file, a generated stencil."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MTX = os.path.join(ROOT, "tests", "golden", "simple_solver_A.mtx")


def run(tool, cases, *flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *flags], input=json.dumps(cases),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)


def test_benchmark_spmv_layout_and_agreement():
    out = run("benchmark_spmv.py", [{"filename": MTX}, {"stencil": "5pt", "size": 300}, {"filename": "/nonexistent.mtx"}],
              "--min_runtime", "0.01")
    assert out[0]["problem"] == {"rows": 19, "cols": 19, "nonzeros": 147}
    for case in out[:2]:
        for fmt in ("csr", "coo", "ell", "sellp", "hybrid"):
            e = case["spmv"][fmt]
            assert e["completed"] and e["time"] > 0 and e["repetitions"] >= 10 and e["storage"] > 0
            assert e["max_relative_norm2"] < 1e-14
        assert case["optimal"]["spmv"] in case["spmv"]
    assert "error" in out[2] and "spmv" not in out[2]


def test_benchmark_solver_layout_and_convergence():
    out = run("benchmark_solver.py", [{"stencil": "5pt", "size": 64}], "--solvers", "cg,bicgstab,fcg,gmres",
              "--preconditioners", "none,jacobi,paric", "--rel_res_goal", "1e-8", "--jacobi_max_block_size", "8")
    s = out[0]["solver"]
    for key in ("cg", "cg-jacobi", "cg-paric", "bicgstab", "fcg-jacobi", "gmres", "gmres-jacobi"):
        e = s[key]
        assert e["completed"] and e["converged"], (key, e)
        assert e["residual_norm"] <= 2e-8 * e["rhs_norm"] and e["apply"]["iterations"] > 0
    assert s["cg-paric"]["apply"]["iterations"] < s["cg"]["apply"]["iterations"]
