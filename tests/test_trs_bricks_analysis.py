"""CPU tests of the brick analysis of the triangular solves (csrc/trs_bricks.hip, host side of
gkomi_trs_bricks_create): the schedule it produces is LEGAL for the factor it was given -- every
dependency of a row is either in the same brick at an earlier level or in a brick that finishes before
the row's brick starts -- and replaying the schedule with the reference's per-row arithmetic gives the
oracle's bits (reference/solver/lower_trs_kernels.cpp:90-120, upper_trs_kernels.cpp:90-123).  No GPU:
only the host analysis runs here."""
import ctypes

import numpy as np
import pytest

import matgen

LEVEL_BIT = -(1 << 31)


def triangle(n, rp, ci, v, lower):
    """the triangle (with diagonal) of a CSR matrix, entries in storage order"""
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = (ci <= rows) if lower else (ci >= rows)
    nrp = np.zeros(n + 1, np.int32)
    np.add.at(nrp, rows[keep] + 1, 1)
    return np.cumsum(nrp).astype(np.int32), ci[keep].astype(np.int32), v[keep].copy()


class Bricks:
    def __init__(self, gk, n, rp, ci, lower, brick_rows=0, threads=0, mode=1):
        self.gk = gk
        self.handle = ctypes.c_void_p(0)
        rp = np.ascontiguousarray(rp, np.int32)
        ci = np.ascontiguousarray(ci, np.int32)
        gk.trs_bricks_create_host_i32(n, rp, ci, int(lower), brick_rows, threads, mode, ctypes.addressof(self.handle))
        info = (ctypes.c_int64 * 8)()
        gk.trs_bricks_info(self.handle.value, ctypes.addressof(info))
        (self.nbricks, self.coarse_levels, self.nsteps, self.critical_steps, self.max_lds, self.width,
         self.threads, self.mode) = (int(x) for x in info)

    def array(self, which):
        data = ctypes.POINTER(ctypes.c_int32)()
        count = ctypes.c_int64(0)
        self.gk.trs_bricks_host_array(self.handle.value, which, ctypes.addressof(data), ctypes.addressof(count))
        return np.ctypeslib.as_array(data, shape=(count.value,)).copy() if count.value else np.zeros(0, np.int32)

    def close(self):
        self.gk.trs_bricks_destroy(self.handle.value)


def replay(bk, n, rp, ci, v, lower, unit, b):
    """what the solve kernel does, brick by brick in ticket order, checking legality on the way"""
    perm, row_begin, step_ptr, step_begin = (bk.array(i) for i in range(4))
    ext_begin, ext_col, pred_ptr, pred_idx, row_rank, inv_local = (bk.array(i) for i in range(4, 10))
    assert sorted(perm.tolist()) == list(range(n))
    assert row_begin[0] == 0 and row_begin[-1] == n and len(step_begin) == bk.nsteps + 1
    x = np.full(n, np.nan)
    finished = np.zeros(bk.nbricks, bool)
    for r in range(bk.nbricks):
        preds = pred_idx[pred_ptr[r]:pred_ptr[r + 1]]
        assert np.all(preds < r) and finished[preds].all()
        r0, r1 = row_begin[r], row_begin[r + 1]
        nst = step_ptr[r + 1] - step_ptr[r]
        assert 8 * ((r1 - r0) * (3 + bk.width) + 1 + (ext_begin[r + 1] - ext_begin[r])) + 4 * ((r1 - r0) * (bk.width + 1) + nst + 4 + ext_begin[r + 1] - ext_begin[r]) <= bk.max_lds
        lds = np.full(r1 - r0 + ext_begin[r + 1] - ext_begin[r], np.nan)
        lds[:r1 - r0] = b[perm[r0:r1]]
        inflow = ext_col[ext_begin[r]:ext_begin[r + 1]]
        assert np.isin(row_rank[inflow], preds).all()
        lds[r1 - r0:r1 - r0 + len(inflow)] = x[inflow]
        ready = np.zeros(r1 - r0, bool)       # rows of the brick whose result is visible (level closed)
        pending = []
        next_inflow = 0
        pos = r0
        for s in range(step_ptr[r], step_ptr[r + 1]):
            begin = step_begin[s] & ~LEVEL_BIT
            end = min(step_begin[s + 1] & ~LEVEL_BIT, begin + bk.threads)
            assert begin == pos and end > begin
            pos = end
            if step_begin[s] < 0:
                ready[pending] = True
                pending = []
            for p in range(begin, end):
                row = perm[p]
                assert row_rank[row] == r and inv_local[row] == p - r0
                total, d, deps = lds[p - r0], 1.0, 0
                for k in range(rp[row], rp[row + 1]):
                    col = ci[k]
                    if col == row:
                        d = v[k]
                    if (col < row) if lower else (col > row):
                        deps += 1
                        if row_rank[col] == r:
                            assert ready[inv_local[col]], "dependency inside the brick is not ready"
                            total -= v[k] * lds[inv_local[col]]
                        else:
                            assert inflow[next_inflow] == col
                            total -= v[k] * lds[r1 - r0 + next_inflow]
                            next_inflow += 1
                assert deps <= bk.width
                lds[p - r0] = total if unit else total / d
                pending.append(p - r0)
        assert pos == r1 and next_inflow == len(inflow)
        x[perm[r0:r1]] = lds[:r1 - r0]
        finished[r] = True
    return x


def replay_pipelined(bk, n, rp, ci, v, lower, unit, b, resident):
    """mode 2: bricks run concurrently (at most `resident` of them, started in ticket order); a pump
    per brick moves inflow values from x to LDS 64 list entries at a time, in list order, once all of
    them are there; a step runs when the entries its rows need are in.  Every round must make progress."""
    perm, row_begin, step_ptr, step_begin = (bk.array(i) for i in range(4))
    ext_begin, ext_col, _, _, row_rank, inv_local = (bk.array(i) for i in range(4, 10))
    ext_row_off = bk.array(10)
    assert bk.threads == 64 and bk.mode == 2
    x = np.full(n, np.nan)

    class State:
        pass

    def start(r):
        st = State()
        st.r, st.r0, st.r1 = r, row_begin[r], row_begin[r + 1]
        st.inflow = ext_col[ext_begin[r]:ext_begin[r + 1]]
        st.lds = np.full(st.r1 - st.r0 + len(st.inflow), np.nan)
        st.lds[:st.r1 - st.r0] = b[perm[st.r0:st.r1]]
        st.ready, st.step = 0, step_ptr[r]
        return st

    live, next_ticket, done = [], 0, 0
    while done < bk.nbricks:
        while len(live) < resident and next_ticket < bk.nbricks:
            live.append(start(next_ticket))
            next_ticket += 1
        progress = False
        for st in list(live):
            if st.ready < len(st.inflow):     # the pump
                batch = st.inflow[st.ready:st.ready + 64]
                if not np.isnan(x[batch]).any():
                    k = st.r1 - st.r0 + st.ready
                    st.lds[k:k + len(batch)] = x[batch]
                    st.ready += len(batch)
                    progress = True
            if st.step < step_ptr[st.r + 1]:  # the compute wave
                s = st.step
                begin = step_begin[s] & ~LEVEL_BIT
                end = min(step_begin[s + 1] & ~LEVEL_BIT, begin + 64)
                need = (ext_row_off[end] if end < st.r1 else ext_begin[st.r + 1]) - ext_begin[st.r]
                if st.ready >= need:
                    out = []
                    for p in range(begin, end):
                        row = perm[p]
                        total, d = st.lds[p - st.r0], 1.0
                        nxt = ext_row_off[p] - ext_begin[st.r]
                        for k in range(rp[row], rp[row + 1]):
                            col = ci[k]
                            if col == row:
                                d = v[k]
                            if (col < row) if lower else (col > row):
                                if row_rank[col] == st.r:
                                    assert inv_local[col] < begin - st.r0, "same-level dependency inside a brick"
                                    total -= v[k] * st.lds[inv_local[col]]
                                else:
                                    assert nxt < need and st.inflow[nxt] == col
                                    total -= v[k] * st.lds[st.r1 - st.r0 + nxt]
                                    nxt += 1
                        out.append(total if unit else total / d)
                    st.lds[begin - st.r0:end - st.r0] = out
                    x[perm[begin:end]] = out
                    st.step += 1
                    progress = True
            if st.step == step_ptr[st.r + 1] and st.ready == len(st.inflow):
                live.remove(st)
                done += 1
                progress = True
        assert progress, "the pipelined schedule stalls"
    return x


CASES = [
    ("2d", lambda: matgen.poisson_2d_5pt(37, 23), 64),
    ("2d-wide", lambda: matgen.poisson_2d_5pt(130, 9), 256),
    ("3d", lambda: matgen.poisson_3d_7pt(13, 9, 11), 200),
    ("3d-cube", lambda: matgen.poisson_3d_7pt(12), 64),
    ("27pt", lambda: matgen.stencil_3d_27pt(7), 64),
]


@pytest.mark.parametrize("lower", [True, False])
@pytest.mark.parametrize("name,make,brick_rows", CASES)
def test_brick_schedule_is_legal_and_replays_to_the_oracle(gk, oracle, name, make, brick_rows, lower):
    n, rp, ci, v = make()
    rp, ci, v = triangle(n, rp, ci, v, lower)
    if name == "27pt":
        # 13 dependencies per row: more than the plan's 8 slots -> the level plan keeps the factor
        with pytest.raises(Exception) as e:
            Bricks(gk, n, rp, ci, lower, brick_rows)
        assert "not supported" in str(e.value).lower() or "ENOTSUPPORTED" in str(e.value)
        return
    bk = Bricks(gk, n, rp, ci, lower, brick_rows)
    try:
        assert bk.nbricks > 1 and bk.coarse_levels > 1 and bk.threads in (64, 128, 256)
        b = np.sin(0.37 * np.arange(n)) + 1.5
        for unit in (False, True):
            e = np.zeros((n, 1))
            (oracle.ref_lower_trs_solve if lower else oracle.ref_upper_trs_solve)(
                n, 1, rp, ci, v, int(unit), b.reshape(n, 1).copy(), 1, e, 1)
            assert np.array_equal(replay(bk, n, rp, ci, v, lower, unit, b), e[:, 0])
    finally:
        bk.close()
    # the pipelined schedule on the same factor, everything resident and a window of three bricks
    bk = Bricks(gk, n, rp, ci, lower, brick_rows, 0, 2)
    try:
        e = np.zeros((n, 1))
        (oracle.ref_lower_trs_solve if lower else oracle.ref_upper_trs_solve)(
            n, 1, rp, ci, v, 0, b.reshape(n, 1).copy(), 1, e, 1)
        for resident in (bk.nbricks, 3):
            assert np.array_equal(replay_pipelined(bk, n, rp, ci, v, lower, False, b, resident), e[:, 0])
    finally:
        bk.close()


def test_brick_analysis_refuses_what_it_cannot_schedule(gk, oracle):
    """No grid in the factor: since round 4 a THIN factor (many more levels than rows per level) is still cut into
    pieces of its level order, anything else is refused.  Whatever comes back must be a legal schedule."""
    rng = np.random.default_rng(3)
    # random lower triangle (more than 16 distinct offsets): ~20 levels of ~25 rows -- thin enough for level pieces
    n = 500
    rows = np.repeat(np.arange(n), 3)
    cols = (rows * rng.random(len(rows))).astype(np.int32)
    m = np.unique(np.stack([rows, cols], 1), axis=0)
    m = np.concatenate([m[m[:, 0] != m[:, 1]], np.stack([np.arange(n), np.arange(n)], 1)])
    m = m[np.lexsort((m[:, 1], m[:, 0]))]
    rp, ci, v = matgen.coo_to_csr(n, m[:, 0].astype(np.int32), m[:, 1].astype(np.int32),
                                  np.where(m[:, 0] == m[:, 1], 3.0, -0.4))
    b = np.cos(np.arange(n))
    e = np.zeros((n, 1))
    oracle.ref_lower_trs_solve(n, 1, rp, ci, v, 0, b.reshape(n, 1).copy(), 1, e, 1)
    bk = Bricks(gk, n, rp, ci, True, 128)
    try:
        assert np.array_equal(replay(bk, n, rp, ci, v, True, False, b), e[:, 0])
    finally:
        bk.close()
    # offsets that are no divisor chain (1, 7, 10): a band, thin -- level pieces, legal
    n = 300
    rp, ci = [0], []
    for r in range(n):
        ci += [c for c in (r - 10, r - 7, r - 1) if c >= 0] + [r]
        rp.append(len(ci))
    rp, ci = np.array(rp, np.int32), np.array(ci, np.int32)
    v = np.where(ci == np.repeat(np.arange(n), np.diff(rp)), 2.0, -0.3)
    b = np.cos(np.arange(n))
    e = np.zeros((n, 1))
    oracle.ref_lower_trs_solve(n, 1, rp, ci, v, 0, b.reshape(n, 1).copy(), 1, e, 1)
    bk = Bricks(gk, n, rp, ci, True, 64)
    try:
        assert np.array_equal(replay(bk, n, rp, ci, v, True, False, b), e[:, 0])
    finally:
        bk.close()
    # a diagonal matrix has nothing to schedule
    with pytest.raises(Exception):
        Bricks(gk, 100, np.arange(101, dtype=np.int32), np.arange(100, dtype=np.int32), True)
    # more than 8 dependencies per row: not for the bricks' record layout
    n = 200
    r = np.repeat(np.arange(n), 12)
    c = np.clip(r - np.tile(np.arange(12), n), 0, None)
    m = np.unique(np.stack([r, c], 1), axis=0)
    rp, ci, v = matgen.coo_to_csr(n, m[:, 0].astype(np.int32), m[:, 1].astype(np.int32), np.ones(len(m)))
    with pytest.raises(Exception):
        Bricks(gk, n, rp, ci, True)


def test_brick_analysis_wraparound_band_is_checked_on_the_entries(gk, oracle):
    """offsets 1 and 16 on rows that are NOT a 16-wide grid (the -1 neighbour exists across the line
    ends): the guessed geometry is wrong, but the brick graph is built from the entries -- either the
    analysis refuses (cycle) or the schedule it returns is legal"""
    n = 16 * 40
    rp, ci, v = [0], [], []
    for r in range(n):
        for c in (r - 16, r - 1):
            if c >= 0:
                ci.append(c)
                v.append(-0.3)
        ci.append(r)
        v.append(2.0)
        rp.append(len(ci))
    rp, ci, v = np.array(rp, np.int32), np.array(ci, np.int32), np.array(v)
    try:
        bk = Bricks(gk, n, rp, ci, True, 64)
    except Exception:
        return
    try:
        b = np.cos(np.arange(n))
        e = np.zeros((n, 1))
        oracle.ref_lower_trs_solve(n, 1, rp, ci, v, 0, b.reshape(n, 1).copy(), 1, e, 1)
        assert np.array_equal(replay(bk, n, rp, ci, v, True, False, b), e[:, 0])
    finally:
        bk.close()


@pytest.mark.parametrize("lower", [True, False])
def test_brick_analysis_is_the_same_on_any_number_of_host_threads(gk, monkeypatch, lower):
    """the symbolic analysis runs layer by layer on up to 8 host threads: the plan must not depend on how many"""
    n, rp, ci, v = matgen.poisson_3d_7pt(23, 19, 29)
    rp, ci, v = triangle(n, rp, ci, v, lower)
    plans = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("GKOMI_ANALYSIS_THREADS", threads)
        bk = Bricks(gk, n, rp, ci, lower, 300, 0, 2)
        plans.append(([bk.array(i) for i in range(11)], (bk.nbricks, bk.coarse_levels, bk.nsteps, bk.max_lds, bk.width)))
        bk.close()
    for arrays, info in plans[1:]:
        assert info == plans[0][1]
        for a, b in zip(arrays, plans[0][0]):
            assert np.array_equal(a, b)


def renumber(n, rp, ci, v, new_index):
    """P A P^T for the numbering new_index[old row], rows and columns sorted"""
    rows = np.repeat(np.arange(n), np.diff(rp))
    pr, pc = new_index[rows], new_index[ci]
    order = np.lexsort((pc, pr))
    return matgen.coo_to_csr(n, pr[order].astype(np.int32), pc[order].astype(np.int32), v[order])


def morton(i, j):
    code = np.zeros_like(i)
    for bit in range(12):
        code |= ((i >> bit) & 1) << (2 * bit + 1)
        code |= ((j >> bit) & 1) << (2 * bit)
    return code


def grid_numberings():
    out = {}
    n, rp, ci, v = matgen.diffusion_2d_patch_ordered(48, patch=(4, 8))          # 2-D, 4 x 8 patches (the thermal2 stand-in's numbering)
    out["patches_2d"] = (n, rp, ci, v)
    g = 32
    n, rp, ci, v = matgen.poisson_2d_5pt(g)
    i, j = np.divmod(np.arange(n), g)
    out["morton_2d"] = (n, *renumber(n, rp, ci, v + 0.01 * np.arange(len(v)) % 0.3, np.argsort(np.argsort(morton(i, j)))))
    g = 12
    n, rp, ci, v = matgen.poisson_3d_7pt(g)
    idx = np.arange(n)
    i, j, k = idx // (g * g), (idx // g) % g, idx % g
    pi, pj, pk = 2, 3, 4                                                         # 3-D, 2 x 3 x 4 patches
    new = (((i // pi) * (g // pj) + (j // pj)) * (g // pk) + (k // pk)) * (pi * pj * pk) + ((i % pi) * pj + (j % pj)) * pk + (k % pk)
    out["patches_3d"] = (n, *renumber(n, rp, ci, v, new))
    # a lexicographic numbering with its second axis reversed is NOT monotone: must be refused, not mis-scheduled
    g = 20
    n, rp, ci, v = matgen.poisson_2d_5pt(g)
    i, j = np.divmod(np.arange(n), g)
    out["reversed_axis"] = (n, *renumber(n, rp, ci, v, i * g + (g - 1 - j)))
    return out


NUMBERINGS = grid_numberings()


@pytest.mark.parametrize("lower", [True, False], ids=["lower", "upper"])
@pytest.mark.parametrize("name,brick_rows", [("patches_2d", 256), ("patches_2d", 0), ("morton_2d", 100), ("patches_3d", 216), ("reversed_axis", 64)])
def test_brick_analysis_recovers_the_grid_of_a_non_lexicographic_numbering(gk, oracle, name, brick_rows, lower):
    """Round 4: factors of grid problems numbered in patches / along a space-filling curve have no divisor chain of
    offsets; the host analysis reads the grid coordinates off the dependency graph (recover_grid_coordinates) and cuts
    bricks from them.  The schedule must be legal and replay to the oracle's bits, blocking and pipelined; a numbering
    that is monotone along both axes after a reflection (reversed axis) is still a grid walk from ANOTHER corner for one
    of the two factors -- whatever the analysis decides there (bricks or refusal), it must not mis-schedule."""
    n, rp, ci, v = NUMBERINGS[name]
    rp, ci, v = triangle(n, rp, ci, v, lower)
    b = np.sin(0.37 * np.arange(n)) + 1.5
    e = np.zeros((n, 1))
    (oracle.ref_lower_trs_solve if lower else oracle.ref_upper_trs_solve)(n, 1, rp, ci, v, 0, b.reshape(n, 1).copy(), 1, e, 1)
    try:
        bk = Bricks(gk, n, rp, ci, lower, brick_rows, 0, 1)
    except Exception:
        assert name == "reversed_axis", "a monotone numbering of a box grid must be recognised"
        return
    try:
        if name != "reversed_axis":
            assert bk.nbricks > 1 and bk.coarse_levels > 1
        assert np.array_equal(replay(bk, n, rp, ci, v, lower, False, b), e[:, 0])
    finally:
        bk.close()
    bk = Bricks(gk, n, rp, ci, lower, brick_rows, 0, 2)
    try:
        for resident in (bk.nbricks, 3):
            assert np.array_equal(replay_pipelined(bk, n, rp, ci, v, lower, False, b, resident), e[:, 0])
    finally:
        bk.close()


def test_grid_recovery_can_be_switched_off(gk, monkeypatch):
    # GKOMI_TRS_RECOVER_GRID is read once per process: only check that the default recognises the patch numbering
    n, rp, ci, v = NUMBERINGS["patches_2d"]
    rp, ci, v = triangle(n, rp, ci, v, True)
    bk = Bricks(gk, n, rp, ci, True, 256, 0, 2)
    assert bk.nbricks == 9 and bk.coarse_levels == 5      # 48 x 48 cells in 16 x 16 bricks
    bk.close()


def thin_factors():
    import os
    out = {}
    here = os.path.dirname(os.path.abspath(__file__))
    kind, n, nc, rows, cols, vals = matgen.read_mtx(os.path.join(here, "golden", "ani4.mtx"))
    rp, ci, v = matgen.coo_to_csr(n, rows, cols, vals)
    out["ani4"] = (n, rp, ci, v + 0.0)
    n = 5000                                                     # a chain: every row its own level
    rows = np.repeat(np.arange(n), 3)
    cols = rows + np.tile([-1, 0, 1], n)
    keep = (cols >= 0) & (cols < n)
    out["tridiagonal"] = (n, *matgen.coo_to_csr(n, rows[keep].astype(np.int32), cols[keep].astype(np.int32),
                                                np.where(cols[keep] == rows[keep], 2.5, -1.0)))
    rng = np.random.default_rng(8)                               # a narrow random band: up to 6 dependencies within 9 rows
    n = 4000
    r, c = [], []
    for row in range(n):
        near = np.arange(max(0, row - 9), min(n, row + 10))
        pick = rng.choice(near, size=min(len(near), 7), replace=False)
        for col in set(pick.tolist()) | {row}:
            r.append(row)
            c.append(col)
    order = np.lexsort((c, r))
    r, c = np.array(r, np.int32)[order], np.array(c, np.int32)[order]
    out["narrow_band"] = (n, *matgen.coo_to_csr(n, r, c, np.where(r == c, 4.0, 0.3 * np.cos(np.arange(len(r))))))
    return out


THIN = thin_factors()


@pytest.mark.parametrize("lower", [True, False], ids=["lower", "upper"])
@pytest.mark.parametrize("name,brick_rows", [("ani4", 0), ("ani4", 300), ("tridiagonal", 0), ("tridiagonal", 700), ("narrow_band", 0)])
def test_brick_analysis_cuts_thin_factors_into_pieces_of_the_level_order(gk, oracle, name, brick_rows, lower):
    """Round 4: a factor with no grid in it but far more levels than rows per level (the reference's ani4 factors: 183 levels
    of 17 rows; a chain; a narrow band) becomes bricks too -- consecutive pieces of its (level, row) order, acyclic by
    construction.  Legal schedule, the oracle's bits, blocking and pipelined."""
    n, rp, ci, v = THIN[name]
    rp, ci, v = triangle(n, rp, ci, v, lower)
    b = np.sin(0.37 * np.arange(n)) + 1.5
    e = np.zeros((n, 1))
    (oracle.ref_lower_trs_solve if lower else oracle.ref_upper_trs_solve)(n, 1, rp, ci, v, 0, b.reshape(n, 1).copy(), 1, e, 1)
    bk = Bricks(gk, n, rp, ci, lower, brick_rows, 0, 1)
    try:
        assert bk.nbricks >= 2 and bk.coarse_levels == bk.nbricks      # a chain of pieces
        assert np.array_equal(replay(bk, n, rp, ci, v, lower, False, b), e[:, 0])
    finally:
        bk.close()
    bk = Bricks(gk, n, rp, ci, lower, brick_rows, 0, 2)
    try:
        for resident in (bk.nbricks, 2):
            assert np.array_equal(replay_pipelined(bk, n, rp, ci, v, lower, False, b, resident), e[:, 0])
    finally:
        bk.close()


def test_brick_analysis_leaves_wide_factors_without_a_grid_to_the_level_plan(gk):
    # 16 levels of 700 rows, random dependencies on the level before: neither a grid nor thin
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
    rp, ci, v = matgen.coo_to_csr(n, np.array(r, np.int32)[order], np.array(c, np.int32)[order], np.ones(len(r)))
    with pytest.raises(Exception):
        Bricks(gk, n, rp, ci, True)
