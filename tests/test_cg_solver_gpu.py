"""GPU parity tests of the native CG driver (C ABI gkomi_cg_solve_f64_i32)
against the oracle's restatement of Cg::apply_dense_impl and the reference's
known-answer solves (reference/test/solver/cg_kernels.cpp:255-477,
examples/simple-solver/doc/results.dox).  Tolerance on solutions: the
north-star's 1e-6 relative; iteration counts must match the oracle's within
+-1 (fused reductions use a different summation order)."""
import json
import os

import numpy as np
import pytest
import torch

import gkomi.solvers as solvers
import matgen
from gpu_util import dev, host

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(G, "cg.json")))


def _solve(gk, rp, ci, v, b, x0, mode, **kw):
    n = len(b)
    return solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), dev(np.asarray(b, np.float64)),
                            x=dev(np.asarray(x0, np.float64)), mode=mode, **kw)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("case", CASES["solve_cases"], ids=lambda c: c["name"])
def test_known_answer_solves(gk, case, mode):
    rp, ci, v = matgen.dense_to_csr(case["A"])
    res = _solve(gk, rp, ci, v, case["b"], case["x0"], mode, max_iters=case["max_iters"],
                 reduction=case["reduction"], check_every=3)
    assert res["converged"] and res["iterations"] < case["max_iters"]
    # the reference asserts r<T>*1e2 on its own executor; a different reduction
    # order costs a few ulps more on these ill-conditioned 6x6 systems
    assert matgen.rel_err(host(res["x"]), case["expect_x"]) <= max(case["tol"], 1e-11)


@pytest.mark.parametrize("mode", [0, 1])
def test_multiple_rhs_stencil_system(gk, mode):
    # SolvesMultipleStencilSystems (cg_kernels.cpp:325-341); mode 1 falls back to mode 0 for nrhs > 1
    rp, ci, v = matgen.dense_to_csr([[2, -1, 0], [-1, 2, -1], [0, -1, 2]])
    b = np.array([[-1.0, 1.0], [3.0, 0.0], [1.0, 1.0]])
    res = solvers.cg_solve(gk, 3, dev(rp), dev(ci), dev(v), dev(b), x=dev(np.zeros((3, 2))),
                           mode=mode, max_iters=400, reduction=2.2e-15)
    assert matgen.rel_err(host(res["x"]), [[1.0, 1.0], [3.0, 1.0], [2.0, 1.0]]) <= 1e-14


@pytest.mark.parametrize("mode", [0, 1])
def test_simple_solver_example(gk, oracle, mode):
    g = CASES["simple_solver"]
    _, n, _, rows, cols, vals = matgen.read_mtx(os.path.join(G, "simple_solver_A.mtx"))
    rp, ci, v = matgen.coo_to_csr(n, rows, cols, vals)
    b = np.ones(n)
    res = _solve(gk, rp, ci, v, b, np.zeros(n), mode, max_iters=g["max_iters"], reduction=g["reduction"])
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b, xe, g["max_iters"], g["reduction"], 0, None, 0)
    assert res["iterations"] == it
    x = host(res["x"])
    assert matgen.rel_err(x, xe) <= 1e-12
    printed = np.array([float(f"{t:.6g}") for t in x])
    assert np.array_equal(printed, np.array(g["expect_x"]))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("grid", [(17, 23), (64, 64), (200, 150)])
def test_poisson_matches_oracle(gk, oracle, mode, grid):
    n, rp, ci, v = matgen.poisson_2d_5pt(*grid)
    s = np.sin(np.arange(n, dtype=np.float64))
    s /= np.linalg.norm(s)
    b = np.empty((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, s.reshape(n, 1), 1, b, 1)
    b = b[:, 0].copy()
    xe = np.zeros(n)
    hist = np.zeros(5000)
    it = oracle.ref_cg_solve(n, rp, ci, v, b, xe, 5000, 1e-10, 0, hist, 5000)
    res = _solve(gk, rp, ci, v, b, np.zeros(n), mode, max_iters=5000, reduction=1e-10, check_every=7)
    assert res["converged"]
    assert abs(res["iterations"] - it) <= 1, (res["iterations"], it)
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-6
    assert matgen.rel_err(host(res["x"]), s) <= 1e-6
    # the reported residual is the recurrence residual at the stopping iteration
    assert res["rel_residual"] < 1e-10
    if abs(res["iterations"] - it) == 0:
        assert abs(res["residual_norm"][0] - hist[it]) <= 1e-6 * hist[it]


@pytest.mark.parametrize("mode", [0, 1])
def test_iteration_limit_stops_exactly(gk, oracle, mode):
    n, rp, ci, v = matgen.poisson_2d_5pt(50, 50)
    b = np.ones(n)
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b, xe, 13, 1e-30, 0, None, 0)
    assert it == 13
    res = _solve(gk, rp, ci, v, b, np.zeros(n), mode, max_iters=13, reduction=1e-30, check_every=5)
    assert res["iterations"] == 13 and not res["converged"]
    # x after exactly 13 updates equals the oracle's up to reduction order
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-10


@pytest.mark.parametrize("mode", [0, 1])
def test_baselines_and_nonzero_initial_guess(gk, oracle, mode):
    n, rp, ci, v = matgen.poisson_2d_5pt(40, 30)
    rng = np.random.default_rng(4)
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    for name, code in (("rhs_norm", 0), ("initial_resnorm", 1), ("absolute", 2)):
        xe = x0.copy()
        it = oracle.ref_cg_solve(n, rp, ci, v, b, xe, 2000, 1e-8, code, None, 0)
        res = _solve(gk, rp, ci, v, b, x0, mode, max_iters=2000, reduction=1e-8, baseline=name)
        assert abs(res["iterations"] - it) <= 1, name
        assert matgen.rel_err(host(res["x"]), xe) <= 1e-6, name


def test_already_converged_rhs_zero_iterations(gk):
    # x0 is the exact solution: the first check (iter 0) fires, x untouched
    n, rp, ci, v = matgen.poisson_2d_5pt(10, 10)
    for mode in (0, 1):
        res = _solve(gk, rp, ci, v, np.zeros(n) + 1e-300, np.zeros(n), mode, max_iters=10, reduction=1e-3,
                     baseline="absolute")
        assert res["iterations"] == 0 and res["converged"]
        assert not host(res["x"]).any()


def test_workspace_too_small(gk):
    import gkomi
    n, rp, ci, v = matgen.poisson_2d_5pt(10, 10)
    info = np.zeros(4)
    ws = torch.empty(16, dtype=torch.uint8, device="cuda:0")
    b = dev(np.ones((n, 1)))
    with pytest.raises(gkomi.GkomiError) as e:
        gk.cg_solve_f64_i32(None, n, 1, int(rp[-1]), dev(rp), dev(ci), dev(v), 0, -1, None, None, b,
                            torch.zeros_like(b), 10, 1e-6, 0, 0, 1, ws, 16, info)
    assert e.value.code == -4


def test_fused_mode_on_views_at_odd_offsets(gk, oracle):
    """Cg on Dense views / offset arrays (the reference works on any alignment): x 8 bytes
    off a 16-B boundary runs the reference sequence instead of mode 1; CSR value / column
    arrays at odd offsets keep the fused loop (SpMV through the generic apply + a partials
    kernel).  Same iteration count and solution either way."""
    n, rp, ci, v = matgen.poisson_2d_5pt(37, 41)
    b = np.sin(0.1 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b, xe, 2000, 1e-10, 0, None, 0)
    pad_v = torch.zeros(len(v) + 1, dtype=torch.float64, device="cuda:0")
    pad_c = torch.zeros(len(ci) + 1, dtype=torch.int32, device="cuda:0")
    pad_v[1:] = dev(v)
    pad_c[1:] = dev(ci)
    xbuf = torch.zeros(n + 1, dtype=torch.float64, device="cuda:0")
    for vals, cols, x in ((pad_v[1:], pad_c[1:], None), (dev(v), dev(ci), xbuf[1:]), (pad_v[1:], pad_c[1:], xbuf[1:])):
        if x is not None:
            x.zero_()
            assert x.data_ptr() % 16 == 8
        res = solvers.cg_solve(gk, n, dev(rp), cols, vals, dev(b), x=x, max_iters=2000, reduction=1e-10, mode=1)
        assert res["converged"] and abs(res["iterations"] - it) <= 1
        assert matgen.rel_err(host(res["x"]), xe) <= 1e-8


def test_preconditioner_generated_for_another_column_count_is_refused(gk):
    n, rp, ci, v = matgen.poisson_2d_5pt(10, 10)
    pc = solvers.jacobi_generate(gk, n, dev(rp), dev(ci), dev(v), max_block_size=4)      # nrhs = 1
    b = dev(np.ones((n, 3)))
    with pytest.raises(ValueError, match="nrhs"):
        solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), b, precond=pc, mode=0)
    pc3 = solvers.jacobi_generate(gk, n, dev(rp), dev(ci), dev(v), max_block_size=4, nrhs=3)
    res = solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), b, precond=pc3, mode=0, max_iters=500, reduction=1e-10)
    assert res["converged"]


# ---- the single-launch ("persistent") CG: vectors and matrix in the register files ----------
def _persistent_solve(gk, n, rp, ci, v, b, hint, **kw):
    before = gk.cg_persistent_solves()
    res = solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), mode=1, max_row_nnz=hint, **kw)
    return res, gk.cg_persistent_solves() - before


@pytest.mark.parametrize("grid", [(300, 270), (600, 500), (1000, 1000)], ids=["2_rows_per_thread", "4", "8"])
def test_persistent_cg_poisson_like_the_oracle(gk, oracle, grid):
    """mode 1 with max_row_nnz_hint = 5 on >= 16 k rows: the whole solve is one launch.  Same
    iteration (the partial sums are grouped per workgroup instead of per 1024 rows: +-1 at most),
    same solution as the oracle's loop, and the same bits on every run (no atomics anywhere)."""
    n, rp, ci, v = matgen.poisson_2d_5pt(*grid)
    s = np.sin(np.arange(n, dtype=np.float64))
    b = np.zeros((n, 1))
    oracle.ref_csr_spmv(n, 1, rp, ci, v, (s / np.linalg.norm(s)).reshape(n, 1), 1, b, 1)
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b[:, 0].copy(), xe, 5000, 1e-10, 0, None, 0)
    res, took = _persistent_solve(gk, n, rp, ci, v, b[:, 0].copy(), 5, max_iters=5000, reduction=1e-10)
    assert took == 1, "the persistent kernel did not run (or gave up)"
    assert res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-7
    again, took = _persistent_solve(gk, n, rp, ci, v, b[:, 0].copy(), 5, max_iters=5000, reduction=1e-10)
    assert took == 1 and again["iterations"] == res["iterations"]
    assert host(again["x"]).tobytes() == host(res["x"]).tobytes()
    # the three-launch iteration (no hint) agrees to rounding
    plain = solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b[:, 0].copy()), mode=1, max_iters=5000,
                             reduction=1e-10)
    assert abs(plain["iterations"] - res["iterations"]) <= 1
    assert matgen.rel_err(host(plain["x"]), host(res["x"])) <= 1e-8


@pytest.mark.parametrize("g3", [48, 80, 100], ids=["2_rows_per_thread", "4", "8_x_in_lds"])
def test_persistent_cg_seven_point_stencil(gk, oracle, g3):
    n, rp, ci, v = matgen.poisson_3d_7pt(g3)
    b = np.cos(0.01 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b.copy(), xe, 3000, 1e-9, 0, None, 0)
    res, took = _persistent_solve(gk, n, rp, ci, v, b, 7, max_iters=3000, reduction=1e-9)
    assert took == 1 and res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-7


def test_persistent_cg_stops_like_the_other_paths(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(200, 150)
    b = np.ones(n)
    # iteration limit: exactly max_iters iterations, not converged
    res, took = _persistent_solve(gk, n, rp, ci, v, b, 5, max_iters=7, reduction=1e-30)
    assert took == 1 and res["iterations"] == 7 and not res["converged"]
    xe = np.zeros(n)
    oracle.ref_cg_solve(n, rp, ci, v, b.copy(), xe, 7, 1e-30, 0, None, 0)
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-12
    # nonzero initial guess, every baseline
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal(n)
    for name, code in (("rhs_norm", 0), ("initial_resnorm", 1), ("absolute", 2)):
        xe = x0.copy()
        it = oracle.ref_cg_solve(n, rp, ci, v, b.copy(), xe, 2000, 1e-8, code, None, 0)
        before = gk.cg_persistent_solves()
        res = solvers.cg_solve(gk, n, dev(rp), dev(ci), dev(v), dev(b), x=dev(x0.copy()), mode=1, max_row_nnz=5,
                               max_iters=2000, reduction=1e-8, baseline=name)
        assert gk.cg_persistent_solves() == before + 1
        assert abs(res["iterations"] - it) <= 1 and matgen.rel_err(host(res["x"]), xe) <= 1e-6, name
    # already converged: zero iterations, x untouched
    res, took = _persistent_solve(gk, n, rp, ci, v, np.zeros(n) + 1e-300, 5, max_iters=10, reduction=1e-3,
                                  baseline="absolute")
    assert took == 1 and res["iterations"] == 0 and res["converged"] and not host(res["x"]).any()


def test_persistent_cg_with_a_hint_that_is_too_small_falls_back(gk, oracle):
    """rows of 7 nonzeros announced as 5: the kernel (which keeps 5 per row) notices before its
    first iteration and the three-launch iteration does the solve."""
    n, rp, ci, v = matgen.poisson_3d_7pt(40)
    b = np.sin(0.1 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b.copy(), xe, 2000, 1e-10, 0, None, 0)
    res, took = _persistent_solve(gk, n, rp, ci, v, b, 5, max_iters=2000, reduction=1e-10)
    assert took == 0 and res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-8


def test_persistent_cg_on_an_ell_system_matrix(gk, oracle):
    """Ell behind the library's callback: same single launch (the rows are loaded from the ELL
    arrays, padding skipped), same bits as the CSR solve -- the arithmetic is the same."""
    from gkomi import formats
    n, rp, ci, v = matgen.poisson_2d_5pt(300, 270)
    A = formats.Csr.from_host(gk, n, n, rp, ci, v)
    E = A.to("ell")
    b = dev(np.sin(0.1 * np.arange(n)))
    before = gk.cg_persistent_solves()
    r_csr = solvers.solve_op(gk, "cg", A, b, max_iters=3000, reduction=1e-10, fused=True)
    r_ell = solvers.solve_op(gk, "cg", E, b, max_iters=3000, reduction=1e-10, fused=True)
    assert gk.cg_persistent_solves() == before + 2
    assert r_ell["converged"] and r_ell["iterations"] == r_csr["iterations"]
    assert host(r_ell["x"]).tobytes() == host(r_csr["x"]).tobytes()
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, host(b), xe, 3000, 1e-10, 0, None, 0)
    assert abs(r_ell["iterations"] - it) <= 1 and matgen.rel_err(host(r_ell["x"]), xe) <= 1e-7


def test_persistent_cg_rows_of_mixed_length(gk, oracle):
    """A shifted graph Laplacian with 1..7 entries per row (rows without neighbours included), rows
    of unsorted columns: the register-resident rows keep storage order and skip what a row lacks."""
    rng = np.random.default_rng(12)
    n = 70001
    deg = rng.integers(0, 4, size=n)          # partners drawn per row; the symmetric closure gives <= 6
    src = np.repeat(np.arange(n), deg)
    dst = (src + rng.integers(1, 400, size=len(src))) % n
    keep = src != dst
    a, b_ = np.concatenate([src[keep], dst[keep]]), np.concatenate([dst[keep], src[keep]])
    pairs = np.unique(np.stack([a, b_], 1), axis=0)
    counts = np.bincount(pairs[:, 0], minlength=n)
    ok = counts[pairs[:, 0]] <= 6
    # drop edges of over-full rows symmetrically until every row has at most 6 neighbours
    while not ok.all():
        bad_rows = set(np.flatnonzero(counts > 6).tolist())
        mask = np.array([(p in bad_rows) or (q in bad_rows) for p, q in pairs])
        drop = mask & (rng.random(len(pairs)) < 0.5)
        dropset = set(map(tuple, pairs[drop])) | set(map(tuple, pairs[drop][:, ::-1]))
        pairs = np.array([pq for pq in map(tuple, pairs) if pq not in dropset])
        counts = np.bincount(pairs[:, 0], minlength=n)
        ok = counts[pairs[:, 0]] <= 6
    rows = np.concatenate([pairs[:, 0], np.arange(n)])
    cols = np.concatenate([pairs[:, 1], np.arange(n)])
    vals = np.concatenate([-np.ones(len(pairs)), counts + 0.5])
    order = np.lexsort((rng.random(len(rows)), rows))       # rows grouped, columns in random order
    rp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rp[1:])
    ci, v = cols[order].astype(np.int32), vals[order]
    assert np.diff(rp).max() <= 7 and np.diff(rp).min() == 1
    b = np.cos(0.01 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b.copy(), xe, 3000, 1e-10, 0, None, 0)
    res, took = _persistent_solve(gk, n, rp, ci, v, b, int(np.diff(rp).max()), max_iters=3000, reduction=1e-10)
    assert took == 1 and res["converged"] and abs(res["iterations"] - it) <= 1
    assert matgen.rel_err(host(res["x"]), xe) <= 1e-8


def test_concurrent_solves_share_the_gpu(gk, oracle):
    """Two host threads, two streams, the same kind of solve at the same time: at most one of them
    may run the single-launch kernel (its workgroups must all be resident), the other takes the
    three-launch iteration -- both finish with the right answer."""
    import threading
    n, rp, ci, v = matgen.poisson_2d_5pt(400, 400)
    b = np.sin(0.1 * np.arange(n))
    xe = np.zeros(n)
    it = oracle.ref_cg_solve(n, rp, ci, v, b.copy(), xe, 5000, 1e-10, 0, None, 0)
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    results, errors = [None, None], []

    def work(slot):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                bd = dev(b)
                for _ in range(20):
                    results[slot] = solvers.cg_solve(gk, n, rpd, cid, vd, bd, mode=1, max_iters=5000, reduction=1e-10,
                                                     max_row_nnz=5)
                stream.synchronize()
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    before = gk.cg_persistent_solves()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    took = gk.cg_persistent_solves() - before
    assert 1 <= took <= 40
    for r in results:
        assert r["converged"] and abs(r["iterations"] - it) <= 1
        assert matgen.rel_err(host(r["x"]), xe) <= 1e-7


def test_fused_cg_block_jacobi_compresses_the_partials_of_a_large_system(gk, oracle):
    """The block-Jacobi apply of the fused CG leaves one partial of r.z / r.r per workgroup (about n / 256).
    Beyond 8192 of them (n > 2.1 M rows at block size 32) they are compressed to 4096 before every workgroup of
    the update kernel re-adds them (the rule of the SpMV's partials).  1500^2 5-pt Poisson + I
    (2.25 M rows, 8790 partials): the fused driver and the reference kernel sequence stop at the same iteration
    with the same solution, and the true residual (oracle SpMV) is where the criterion says."""
    n, rp, ci, v = matgen.poisson_2d_5pt(1500)
    v = v.copy()
    v[v == 4.0] = 5.0
    assert n // 256 > 8192
    rpd, cid, vd = dev(rp), dev(ci), dev(v)
    pre = solvers.jacobi_generate(gk, n, rpd, cid, vd, max_block_size=32)
    b = np.cos(0.001 * np.arange(n))
    bd = dev(b)
    fused = solvers.cg_solve(gk, n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, mode=1, check_every=8, precond=pre)
    seq = solvers.cg_solve(gk, n, rpd, cid, vd, bd, max_iters=2000, reduction=1e-10, mode=0, precond=pre)
    assert fused["converged"] and seq["converged"]
    assert abs(fused["iterations"] - seq["iterations"]) <= 1, (fused["iterations"], seq["iterations"])
    assert matgen.rel_err(host(fused["x"]), host(seq["x"])) <= 1e-9
    r = b.reshape(n, 1).copy()
    oracle.ref_csr_advanced_spmv(n, 1, -1.0, rp, ci, v, np.ascontiguousarray(host(fused["x"])).reshape(n, 1), 1, 1.0, r, 1)
    assert np.linalg.norm(r) / np.linalg.norm(b) <= 2e-10
