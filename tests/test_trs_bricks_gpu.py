"""GPU parity tests (C ABI) of the brick plan of the triangular solves (csrc/trs_bricks.hip,
gkomi_trs_bricks_*) against the oracle: bit-exact like every triangular solve here (per row the
subtractions run in storage order, one division: reference/solver/lower_trs_kernels.cpp:90-120,
upper_trs_kernels.cpp:90-123).  The analysis itself is covered on the CPU
(test_trs_bricks_analysis.py)."""
import numpy as np
import pytest
import torch

import gkomi
import gkomi.solvers as solvers
import ilu_util
import matgen
from gpu_util import dev, host
from test_trs_bricks_analysis import triangle

pytestmark = pytest.mark.gpu


def oracle_solve(oracle, n, rp, ci, v, lower, unit, b):
    e = np.zeros_like(b)
    (oracle.ref_lower_trs_solve if lower else oracle.ref_upper_trs_solve)(
        n, b.shape[1], rp, ci, v, int(unit), b.copy(), b.shape[1], e, b.shape[1])
    return e


def brick_solve(gk, n, rp, ci, v, lower, unit, b, brick_rows=0, threads=0, mode=0):
    bk = solvers.TrsBricks(gk, n, dev(rp), dev(ci), dev(v), lower, brick_rows, threads, mode)
    x = torch.full(b.shape, 777.0, dtype=torch.float64, device="cuda:0")
    bk.solve(dev(b), x, unit)
    assert not bk.overrun()
    return host(x), bk


GRIDS = [
    ("2d", lambda: matgen.poisson_2d_5pt(70, 45)),
    ("2d-line", lambda: matgen.poisson_2d_5pt(700, 3)),
    ("3d", lambda: matgen.poisson_3d_7pt(21, 17, 19)),
]


@pytest.mark.parametrize("lower", [True, False])
@pytest.mark.parametrize("unit", [False, True])
@pytest.mark.parametrize("name,make", GRIDS)
@pytest.mark.parametrize("brick_rows,threads,mode", [(0, 0, 0), (64, 64, 1), (300, 128, 1), (900, 256, 1), (0, 0, 1),
                                                     (64, 0, 2), (300, 0, 2), (2000, 0, 2)])
def test_bricks_bitexact_vs_oracle(gk, oracle, name, make, lower, unit, brick_rows, threads, mode):
    n, rp, ci, v = make()
    rng = np.random.default_rng(n + brick_rows)
    v = v * (1.0 + 0.3 * rng.random(len(v)))       # no two rows alike
    rp, ci, v = triangle(n, rp, ci, v, lower)
    for nrhs in (1, 3):
        b = rng.standard_normal((n, nrhs))
        x, bk = brick_solve(gk, n, rp, ci, v, lower, unit, b, brick_rows, threads, mode)
        assert bk.nbricks >= 1 and (threads == 0 or bk.threads == threads) and bk.mode == (mode or 2)
        assert np.array_equal(x, oracle_solve(oracle, n, rp, ci, v, lower, unit, b))


def test_bricks_entries_in_any_order_and_other_triangle_ignored(gk, oracle):
    """a full (not triangular) matrix with shuffled rows: the solve takes the lower / upper part, the
    subtractions in STORAGE order"""
    n, rp, ci, v = matgen.poisson_3d_7pt(14)
    rng = np.random.default_rng(11)
    ci, v = ci.copy(), v * (1.0 + rng.random(len(v)))
    for r in range(n):
        o = rng.permutation(rp[r + 1] - rp[r]) + rp[r]
        ci[rp[r]:rp[r + 1]], v[rp[r]:rp[r + 1]] = ci[o], v[o]
    b = rng.standard_normal((n, 1))
    for lower in (True, False):
        for mode in (1, 2):
            x, _ = brick_solve(gk, n, rp, ci, v, lower, False, b, 500, 0, mode)
            assert np.array_equal(x, oracle_solve(oracle, n, rp, ci, v, lower, False, b))


def test_bricks_division_matches_ieee_over_the_exponent_range(gk, oracle):
    """the solve finishes x = sum / d from a reciprocal prepared by the numeric phase when the
    exponents are moderate and divides otherwise: both must give the oracle's (IEEE) quotient --
    values from 2^-1000 to 2^1000, subnormals, zeros, infinities and the edges of the moderate box"""
    n, rp, ci, v = matgen.poisson_2d_5pt(96, 64)
    rp, ci, v = triangle(n, rp, ci, v, True)
    rng = np.random.default_rng(17)
    rows = np.repeat(np.arange(n), np.diff(rp))
    diag = ci == rows
    edges = np.array([-1023, -1000, -384, -383, -382, -1, 0, 1, 382, 383, 384, 700])
    with np.errstate(all="ignore"):
        for trial in range(4):
            v2 = v.copy()
            v2[~diag] = 0.0 if trial == 3 else rng.standard_normal((~diag).sum()) * 2.0 ** rng.integers(-30, 30, (~diag).sum())
            ex = rng.integers(-1000, 1000, n) if trial < 2 else rng.choice(edges, n)
            v2[diag] = (1.0 + rng.random(n)) * 2.0 ** ex * rng.choice([-1.0, 1.0], n)
            eb = rng.integers(-1000, 1000, n) if trial != 1 else rng.choice(edges, n)
            b = ((1.0 + rng.random(n)) * 2.0 ** eb * rng.choice([-1.0, 1.0], n)).reshape(n, 1)
            b[rng.integers(0, n, 50), 0] = 0.0
            b[rng.integers(0, n, 20), 0] = 5e-324
            b[rng.integers(0, n, 5), 0] = np.inf
            e = oracle_solve(oracle, n, rp, ci, v2, True, False, b)
            for mode in (1, 2):
                x, _ = brick_solve(gk, n, rp, ci, v2, True, False, b, 512, 0, mode)
                same = (x.view(np.int64) == e.view(np.int64)) | (np.isnan(x) & np.isnan(e))
                assert same.all(), (trial, mode, np.flatnonzero(~same.ravel())[:5])


def test_bricks_refresh_and_repeated_solves(gk, oracle):
    """new values through the numeric phase alone; the plan re-arms itself after every solve"""
    n, rp, ci, v = matgen.poisson_3d_7pt(20)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    lrp, lc, lv = f["L"]
    b = np.cos(0.01 * np.arange(n)).reshape(n, 1)
    for mode in (1, 2):
        bk = solvers.TrsBricks(gk, n, dev(lrp), dev(lc), dev(lv), True, 1000, 0, mode)
        x = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
        e = oracle_solve(oracle, n, lrp, lc, lv, True, False, b)
        for _ in range(5):
            x.fill_(3.0)
            bk.solve(dev(b), x)
            assert np.array_equal(host(x), e)
        lv2 = lv * 1.25
        bk.refresh(dev(lv2))
        bk.solve(dev(b), x)
        assert np.array_equal(host(x), oracle_solve(oracle, n, lrp, lc, lv2, True, False, b)) and not bk.overrun()


def test_bricks_numeric_phase_notices_a_plan_that_lost_its_index_arrays(gk, oracle):
    """the index arrays are uploaded once per plan buffer; a caller that reuses the address for something else in
    between (here: zeroes it) must still get a valid plan from the next numeric phase"""
    n, rp, ci, v = matgen.poisson_2d_5pt(60)
    rp, ci, v = triangle(n, rp, ci, v, True)
    b = np.cos(np.arange(n) * 0.2).reshape(n, 1)
    e = oracle_solve(oracle, n, rp, ci, v, True, False, b)
    bk = solvers.TrsBricks(gk, n, dev(rp), dev(ci), dev(v), True, 400)
    x = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    bk.solve(dev(b), x)
    assert np.array_equal(host(x), e)
    bk.plan.zero_()
    bk.refresh(dev(v))
    x.fill_(5.0)
    bk.solve(dev(b), x)
    assert np.array_equal(host(x), e) and not bk.overrun()


def test_bricks_solve_in_place(gk, oracle):
    n, rp, ci, v = matgen.poisson_2d_5pt(90)
    rp, ci, v = triangle(n, rp, ci, v, False)
    b = np.sin(np.arange(n) * 0.3).reshape(n, 1)
    bk = solvers.TrsBricks(gk, n, dev(rp), dev(ci), dev(v), False, 700, 0, 1)
    x = dev(b).clone()
    bk.solve(x, x)
    assert np.array_equal(host(x), oracle_solve(oracle, n, rp, ci, v, False, False, b))
    # the pipelined plan keeps its ready flags in x: an aliased solve takes the other kernel on the same plan
    bk2 = solvers.TrsBricks(gk, n, dev(rp), dev(ci), dev(v), False, 700, 0, 2)
    x2 = dev(b).clone()
    bk2.solve(x2, x2)
    assert np.array_equal(host(x2), host(x)) and not bk2.overrun()


def test_bricks_not_for_irregular_factors(gk):
    n, rp, ci, v = matgen.poisson_2d_5pt(40)
    rp, ci, v = matgen.permute_symmetric(n, rp, ci, v, seed=3)
    rp, ci, v = triangle(n, rp, ci, v, True)
    with pytest.raises(gkomi.GkomiError) as e:
        solvers.TrsBricks(gk, n, dev(rp), dev(ci), dev(v), True)
    assert "not supported" in str(e.value).lower()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("lower", [True, False])
def test_bricks_config4_factor_full_size(gk, oracle, lower, mode):
    """BASELINE config 4's factor shape at full size: 7-point 108^3, ILU(0) pattern"""
    n, rp, ci, v = matgen.at_like(108)
    rp, ci, v = triangle(n, rp, ci, v, lower)
    b = (np.sin(0.001 * np.arange(n)) + 1.1).reshape(n, 1)
    x, bk = brick_solve(gk, n, rp, ci, v, lower, False, b, 0, 0, mode)
    assert np.array_equal(x, oracle_solve(oracle, n, rp, ci, v, lower, False, b))
    assert bk.coarse_levels < 64


# ---- round 3: the analysis itself runs on the device ------------------------------------------------------------
def _host_and_device_analysis(gk, n, rp, ci, lower, brick_rows, mode):
    import ctypes
    from test_trs_bricks_analysis import Bricks
    hb = Bricks(gk, n, rp, ci, lower, brick_rows=brick_rows, mode=mode)
    handle = ctypes.c_void_p(0)
    s = torch.cuda.current_stream().cuda_stream
    gk.trs_bricks_create_i32(s, n, dev(np.ascontiguousarray(rp, np.int32)), dev(np.ascontiguousarray(ci, np.int32)), int(lower),
                             brick_rows, 0, mode, ctypes.addressof(handle))
    db = Bricks.__new__(Bricks)
    db.gk, db.handle = gk, handle
    info = (ctypes.c_int64 * 8)()
    gk.trs_bricks_info(handle.value, ctypes.addressof(info))
    (db.nbricks, db.coarse_levels, db.nsteps, db.critical_steps, db.max_lds, db.width, db.threads, db.mode) = (int(x) for x in info)
    return hb, db


@pytest.mark.parametrize("case", ["2d_61x47", "3d_20x17x23", "3d_unsorted_rows", "2d_upper", "odd_tail"])
@pytest.mark.parametrize("mode", [1, 2])
def test_device_analysis_equals_host_analysis(gk, case, mode):
    """gkomi_trs_bricks_create_i32 analyses the pattern on the device (one workgroup per brick); every array
    of the plan -- permutation, brick and step tables, inflow lists, LDS indices -- must be the one the host
    analysis (gkomi_trs_bricks_create_host_i32, pinned by tests/test_trs_bricks_analysis.py) produces.
    Predecessor lists are compared as sets per brick (the host lists them in the order it meets them)."""
    from test_trs_bricks_analysis import triangle
    lower = case != "2d_upper"
    if case in ("2d_61x47", "2d_upper"):
        n, rp, ci, v = matgen.poisson_2d_5pt(61, 47)
    elif case == "3d_20x17x23":
        n, rp, ci, v = matgen.poisson_3d_7pt(20, 17, 23)
    elif case == "3d_unsorted_rows":
        # the entries of every row in a random storage order: inflow lists follow storage order
        n, rp, ci, v = matgen.poisson_3d_7pt(14, 15, 16)
        rng = np.random.default_rng(5)
        ci, v = ci.copy(), v.copy()
        for r in range(n):
            q = rng.permutation(rp[r + 1] - rp[r])
            ci[rp[r]:rp[r + 1]] = ci[rp[r]:rp[r + 1]][q]
            v[rp[r]:rp[r + 1]] = v[rp[r]:rp[r + 1]][q]
    else:
        n, rp, ci, v = matgen.poisson_3d_7pt(12, 12, 12)
        keep = 12 * 12 * 11 + 77       # the last plane is incomplete
        rows = np.repeat(np.arange(n), np.diff(rp))
        m = (rows < keep) & (ci < keep)
        n = keep
        rp, ci, v = matgen.coo_to_csr(n, rows[m].astype(np.int32), ci[m].astype(np.int32), v[m])
    trp, tci, tv = triangle(n, rp, ci, v, lower)
    for brick_rows in (0, 300):
        hb, db = _host_and_device_analysis(gk, n, trp, tci, lower, brick_rows, mode)
        try:
            assert (hb.nbricks, hb.coarse_levels, hb.nsteps, hb.critical_steps, hb.max_lds, hb.width, hb.threads, hb.mode) == \
                (db.nbricks, db.coarse_levels, db.nsteps, db.critical_steps, db.max_lds, db.width, db.threads, db.mode)
            assert gk.trs_bricks_plan_bytes(hb.handle.value) == gk.trs_bricks_plan_bytes(db.handle.value)
            assert gk.trs_bricks_levels_estimate(hb.handle.value) == gk.trs_bricks_levels_estimate(db.handle.value)
            for which in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10):
                assert np.array_equal(hb.array(which), db.array(which)), (case, mode, brick_rows, which)
            hp, dp, ptr = hb.array(7), db.array(7), hb.array(6)
            for r in range(hb.nbricks):
                assert sorted(hp[ptr[r]:ptr[r + 1]]) == sorted(dp[ptr[r]:ptr[r + 1]])
        finally:
            hb.close()
            db.close()


def test_device_analysis_refuses_what_the_host_analysis_refuses(gk):
    """16 levels of 700 rows with random dependencies on the level before: no divisor chain of offsets, no grid in the
    graph, not thin -- GKOMI_ENOTSUPPORTED from the device entry (which asks the host analysis before it gives up)"""
    import ctypes
    import gkomi
    rng = np.random.default_rng(2)
    n, per = 16 * 700, 700
    r, c = [], []
    for row in range(n):
        lvl = row // per
        if lvl > 0:
            for col in rng.choice(np.arange((lvl - 1) * per, lvl * per), size=3, replace=False):
                r.append(row)
                c.append(int(col))
        r.append(row)
        c.append(row)
    order = np.lexsort((c, r))
    trp, tci, tv = matgen.coo_to_csr(n, np.array(r, np.int32)[order], np.array(c, np.int32)[order], np.ones(len(r)))
    h = ctypes.c_void_p(0)
    with pytest.raises(gkomi._lib.GkomiError) as e:
        gk.trs_bricks_create_i32(torch.cuda.current_stream().cuda_stream, n, dev(trp), dev(tci), 1, 0, 0, 2, ctypes.addressof(h))
    assert e.value.code == -2 and not h.value


@pytest.mark.parametrize("lower", [True, False], ids=["lower", "upper"])
@pytest.mark.parametrize("name", ["ani4", "tridiagonal", "narrow_band"])
def test_bricks_on_thin_factors_without_a_grid(gk, oracle, name, lower):
    """Round 4: pieces of the level order as bricks (the reference's ani4 factors, a chain, a narrow band with up to 7
    dependencies per row): device entry -> level count on the device -> host analysis -> the same solve kernels;
    bit-exact against the oracle, both modes."""
    from test_trs_bricks_analysis import THIN
    n, rp, ci, v = THIN[name]
    rp, ci, v = triangle(n, rp, ci, v, lower)
    rng = np.random.default_rng(len(name))
    b = rng.standard_normal((n, 2))
    for brick_rows, mode in ((0, 2), (400, 2), (400, 1)):
        x, bk = brick_solve(gk, n, rp, ci, v, lower, False, b, brick_rows, 0, mode)
        assert bk.nbricks >= 2
        assert np.array_equal(x, oracle_solve(oracle, n, rp, ci, v, lower, False, b)), (brick_rows, mode)


@pytest.mark.parametrize("lower", [True, False], ids=["lower", "upper"])
@pytest.mark.parametrize("name", ["patches_2d", "morton_2d", "patches_3d"])
def test_bricks_on_grids_numbered_in_patches_or_along_a_curve(gk, oracle, name, lower):
    """Round 4: a grid problem whose numbering is not lexicographic -- 4 x 8 patches (the
    thermal2 stand-in's numbering), Morton order, 2 x 3 x 4 patches in three dimensions -- has no divisor chain of offsets;
    gkomi_trs_bricks_create_i32 then reads the grid coordinates off the dependency graph on the host and cuts bricks from
    them.  Same solve kernels, bit-exact against the oracle, both modes, two right-hand sides."""
    from test_trs_bricks_analysis import NUMBERINGS
    n, rp, ci, v = NUMBERINGS[name]
    rng = np.random.default_rng(len(name))
    rp, ci, v = triangle(n, rp, ci, v * (1.0 + 0.3 * rng.random(len(v))), lower)
    b = rng.standard_normal((n, 2))
    for brick_rows, mode in ((0, 2), (150, 2), (150, 1)):
        x, bk = brick_solve(gk, n, rp, ci, v, lower, False, b, brick_rows, 0, mode)
        assert bk.nbricks > 1 or brick_rows == 0
        assert np.array_equal(x, oracle_solve(oracle, n, rp, ci, v, lower, False, b)), (brick_rows, mode)


def test_ilu_of_the_patch_ordered_problem_takes_the_brick_plan(gk, oracle):
    """generate's choice (gkomi.solvers.ilu_from_factors, the rule of the shims and the mirror): the ILU(0) factors of the
    patch-ordered diffusion problem (config 3's second stand-in at 1/4 size: 280^2) get bricks, and Ilu::apply equals the
    oracle's two solves bit for bit."""
    n, rp, ci, v = matgen.diffusion_2d_patch_ordered(280)
    f = ilu_util.oracle_par_ilu(oracle, n, rp, ci, v)
    pre = solvers.ilu_from_factors(gk, n, tuple(dev(a) for a in f["L"]), tuple(dev(a) for a in f["U"]))
    assert pre.l_bricks is not None and pre.u_bricks is not None
    b = np.sin(0.1 * np.arange(n)).reshape(n, 1) + 2.0
    y, e = np.zeros_like(b), np.zeros_like(b)
    oracle.ref_lower_trs_solve(n, 1, *f["L"], 0, b, 1, y, 1)
    oracle.ref_upper_trs_solve(n, 1, *f["U"], 0, y, 1, e, 1)
    z = torch.zeros((n, 1), dtype=torch.float64, device="cuda:0")
    pre.apply(dev(b), z)
    assert np.array_equal(host(z), e)
