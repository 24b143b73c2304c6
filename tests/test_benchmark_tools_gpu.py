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


def test_binary_matrix_files_are_read_like_text_ones(tmp_path):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "repo-8852-ginkgo_amd"))
    import gkomi
    from gkomi import formats
    gk = gkomi.lib()
    A = formats.read_mtx(gk, MTX)
    rows = A.row_idxs().cpu().numpy()[:A.nnz]
    rec = np.zeros(A.nnz, dtype=[("row", np.int32), ("col", np.int32), ("val", np.float64)])
    rec["row"], rec["col"], rec["val"] = rows, A.col_idxs.cpu().numpy(), A.vals.cpu().numpy()
    rec = rec[np.random.default_rng(0).permutation(A.nnz)]      # the reader sorts
    path = tmp_path / "A.bin"
    with open(path, "wb") as f:
        f.write(b"GINKGODI" + np.array([A.nrows, A.ncols, A.nnz], np.uint64).tobytes() + rec.tobytes())
    B = formats.read_matrix(gk, str(path))
    for a, b in ((A.row_ptrs, B.row_ptrs), (A.col_idxs, B.col_idxs), (A.vals, B.vals)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    out = run("benchmark_spmv.py", [{"filename": str(path)}], "--formats", "csr", "--min_runtime", "0.01")
    assert out[0]["problem"]["nonzeros"] == 147 and out[0]["spmv"]["csr"]["completed"]


def test_benchmark_solver_layout_and_convergence():
    out = run("benchmark_solver.py", [{"stencil": "5pt", "size": 64}], "--solvers", "cg,bicgstab,fcg,gmres",
              "--preconditioners", "none,jacobi,paric", "--rel_res_goal", "1e-8", "--jacobi_max_block_size", "8")
    s = out[0]["solver"]
    for key in ("cg", "cg-jacobi", "cg-paric", "bicgstab", "fcg-jacobi", "gmres", "gmres-jacobi"):
        e = s[key]
        assert e["completed"] and e["converged"], (key, e)
        assert e["residual_norm"] <= 2e-8 * e["rhs_norm"] and e["apply"]["iterations"] > 0
    assert s["cg-paric"]["apply"]["iterations"] < s["cg"]["apply"]["iterations"]
