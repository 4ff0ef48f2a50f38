"""Ordering + symbolic factorisation of the tile-sparse pose solve (csrc/pvlm_spd_plan.h, what pvlm_spd_solve_blocks uploads as per-block-column
tile lists), compiled for the host: the permutation is one, and the lists COVER a numeric factorisation — every nonzero of the Cholesky factor
of the permuted matrix lies in a row tile its block column lists, every tile a rank-32 update changes is listed as a pair.  A missing tile would
be a wrong answer on the GPU; this is where the planner is validated before it is run (the numeric solve itself: tests/test_linalg_gpu.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB, TILE = 32, 64


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("spd_plan") / "spd_plan_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "spd_plan_check.cpp")])
    lib = ctypes.CDLL(out)
    lib.chk_spd_plan.restype = ctypes.c_int
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def plan(lib, n, rows, cols):
    rows = np.ascontiguousarray(rows, np.int32); cols = np.ascontiguousarray(cols, np.int32)
    sizes = np.zeros(3, np.int64); frac = ctypes.c_double()
    lib.chk_spd_plan(n, len(rows) // 6, _p(rows, ctypes.c_int), _p(cols, ctypes.c_int), NB, None, None, None, None, None, _p(sizes, ctypes.c_longlong), ctypes.byref(frac))
    perm = np.zeros(n, np.int32); row_off = np.zeros(sizes[2] + 1, np.int32); pair_off = np.zeros(sizes[2] + 1, np.int32)
    row_tiles = np.zeros(max(sizes[0], 1), np.int32); pairs = np.zeros((max(sizes[1], 1), 2), np.int32)
    ordered = lib.chk_spd_plan(n, len(rows) // 6, _p(rows, ctypes.c_int), _p(cols, ctypes.c_int), NB, _p(perm, ctypes.c_int), _p(row_off, ctypes.c_int), _p(row_tiles, ctypes.c_int),
                               _p(pair_off, ctypes.c_int), _p(pairs, ctypes.c_int), _p(sizes, ctypes.c_longlong), ctypes.byref(frac))
    return dict(ordered=bool(ordered), perm=perm, row_off=row_off, row_tiles=row_tiles[:sizes[0]], pair_off=pair_off, pairs=pairs[:sizes[1]], fraction=frac.value)


def system(rng, P, pairs, constant=()):
    off = np.full((P, 6), -1, np.int64); n = 0
    for p in range(P):
        for half in range(2):
            if (p, half) in constant:
                continue
            off[p, 3 * half:3 * half + 3] = np.arange(n, n + 3); n += 3
    H = np.zeros((n, n))
    rows, cols = [], []
    for a, b in pairs:
        blk = rng.normal(size=(6, 6)) * (0.2 if a != b else 1.0)
        rows.append(off[a]); cols.append(off[b])
        va, vb = off[a] >= 0, off[b] >= 0
        H[np.ix_(off[a][va], off[b][vb])] += blk[np.ix_(va, vb)]
        if a != b:
            H[np.ix_(off[b][vb], off[a][va])] += blk[np.ix_(va, vb)].T
    H = H @ H.T * 0 + H + H.T                               # symmetric pattern
    M = H + np.eye(n) * (np.abs(H).sum(axis=1).max() + 1.0)  # diagonally dominant: positive definite, no cancellation to exact zeros
    return n, np.concatenate(rows), np.concatenate(cols), M


@pytest.mark.parametrize("shape", ["chain", "loops", "star", "grid"])
def test_lists_cover_the_numeric_factor(chk, shape):
    rng = np.random.default_rng(7)
    P = 160
    pairs = [(p, p) for p in range(P)]
    if shape == "chain":
        pairs += [(p, q) for p in range(P) for q in range(p + 1, min(p + 3, P))]
    elif shape == "loops":
        pairs += [(p, p + 1) for p in range(P - 1)] + [(int(a), int(b)) for a, b in zip(rng.integers(0, P, 80), rng.integers(0, P, 80)) if a < b]
    elif shape == "star":
        pairs += [(0, p) for p in range(1, P)] + [(p, p + 1) for p in range(1, P - 1, 7)]
    else:
        side = 12
        pairs = [(p, p) for p in range(side * side)] + [(r * side + c, r * side + c + 1) for r in range(side) for c in range(side - 1)] + \
                [(r * side + c, (r + 1) * side + c) for r in range(side - 1) for c in range(side)]
        P = side * side
    n, rows, cols, M = system(rng, P, pairs, constant={(0, 0), (3, 1)})
    pl = plan(chk, n, rows, cols)
    assert pl["ordered"] and sorted(pl["perm"].tolist()) == list(range(n))
    Mp = np.zeros_like(M); Mp[np.ix_(pl["perm"], pl["perm"])] = M          # new index of unknown i = perm[i]
    L = np.linalg.cholesky(Mp)
    C = (n + NB - 1) // NB
    assert len(pl["row_off"]) == C + 1
    listed_updates = set()
    for k in range(C):
        base = min(n, (k + 1) * NB)
        tiles = set(pl["row_tiles"][pl["row_off"][k]:pl["row_off"][k + 1]].tolist())
        nzrows = np.flatnonzero(np.abs(L[base:, k * NB:base]).max(axis=1) > 1e-13) + base
        assert set((nzrows // TILE).tolist()) <= tiles, (shape, k)
        for a, b in pl["pairs"][pl["pair_off"][k]:pl["pair_off"][k + 1]]:
            assert a >= b and a in tiles and b in tiles
            listed_updates.add((k, int(a), int(b)))
        # every pair of nonzero panel rows is an update the kernel must perform
        t = sorted(set((nzrows // TILE).tolist()))
        for i, a in enumerate(t):
            for b in t[:i + 1]:
                assert (k, a, b) in listed_updates, (shape, k, a, b)
    assert 0 < pl["fraction"] <= 1.2
    if shape in ("chain", "grid"):
        assert pl["fraction"] < 0.5


def test_degenerate_structures(chk):
    rng = np.random.default_rng(1)
    n, rows, cols, _ = system(rng, 1, [(0, 0)])
    pl = plan(chk, n, rows, cols)
    assert sorted(pl["perm"].tolist()) == list(range(n)) and len(pl["pairs"]) == 0
    # unknowns no block mentions stay where a permutation puts them; blocks with every index constant are ignored
    rows = np.array([0, 1, 2, 3, 4, 5, -1, -1, -1, -1, -1, -1], np.int32); cols = np.array([0, 1, 2, 3, 4, 5, -1, -1, -1, -1, -1, -1], np.int32)
    pl = plan(chk, 9, rows, cols)
    assert sorted(pl["perm"].tolist()) == list(range(9))


# ---- nested dissection + level schedule (pvlm_spd::plan_levels) ----------------------------------------------------------------------------------
def levels_plan(lib, n, rows, cols, leaf=8):
    rows = np.ascontiguousarray(rows, np.int32); cols = np.ascontiguousarray(cols, np.int32)
    sizes = np.zeros(10, np.int64); frac = ctypes.c_double()
    args = (n, len(rows) // 6, _p(rows, ctypes.c_int), _p(cols, ctypes.c_int), NB, leaf, _p(sizes, ctypes.c_longlong), ctypes.byref(frac))
    lib.chk_spd_levels(*args, *([None] * 13))
    n_pad, L, C, nrt, npw, ntg, nsrc, nft, nfs, _ = [int(v) for v in sizes]
    A = dict(new_of_old=np.zeros(n, np.int32), col_off=np.zeros(L + 1, np.int32), cols=np.zeros(C, np.int32), row_off=np.zeros(C + 1, np.int32), row_tiles=np.zeros(max(nrt, 1), np.int32),
             pwg_off=np.zeros(L + 1, np.int32), pwg=np.zeros((max(npw, 1), 2), np.int32), upd_off=np.zeros(L + 1, np.int32), targets=np.zeros((max(ntg, 1), 6), np.int32),
             sources=np.zeros(max(nsrc, 1), np.int32), fwd_off=np.zeros(L + 1, np.int32), ftargets=np.zeros((max(nft, 1), 4), np.int32), fsources=np.zeros(max(nfs, 1), np.int32))
    order = ("new_of_old", "col_off", "cols", "row_off", "row_tiles", "pwg_off", "pwg", "upd_off", "targets", "sources", "fwd_off", "ftargets", "fsources")
    ok = lib.chk_spd_levels(*args, *[_p(A[k], ctypes.c_int) for k in order])
    A.update(n_pad=n_pad, levels=L, block_cols=C, tile_updates=int(sizes[9]), fraction=frac.value, ordered=bool(ok))
    return A


def solve_by_levels(P, M, rhs, tail=None):
    """The arithmetic of chol_factor_solve_levels (csrc/pvlm_linalg.hip) in numpy, with EXACTLY the plan's lists and the launch structure of the device: every
    launch reads a snapshot of what the launches before it left (a job that needed a result of its own launch would read stale data here, as it would race there),
    touches only the tiles the lists name, masks as the kernels mask."""
    n = len(rhs); n_pad = P["n_pad"]; nw = P["new_of_old"]
    A = np.eye(n_pad); b = np.zeros(n_pad)
    A[np.ix_(nw, nw)] = M; b[nw] = rhs
    A = np.tril(A)
    Linv = {}; y = np.zeros(n_pad)
    n_main = P["levels"] if tail is None else tail[1]
    for l in range(n_main):
        snap = A.copy(); bsnap = b.copy()
        seen = set()
        for k, g in P["pwg"][P["pwg_off"][l]:P["pwg_off"][l + 1]]:
            k0, base = k * NB, (k + 1) * NB
            rt = P["row_tiles"][P["row_off"][k]:P["row_off"][k + 1]]
            assert g < max(1, 8 * len(rt))
            if k not in seen:
                seen.add(k)
                D = snap[k0:base, k0:base]; D = np.tril(D) + np.tril(D, -1).T
                Lkk = np.linalg.cholesky(D); Linv[k] = np.linalg.inv(Lkk)
                y[k0:base] = Linv[k] @ bsnap[k0:base]
            if len(rt):
                r0 = rt[g // 8] * TILE + (g % 8) * 8
                rows = np.arange(r0, r0 + 8); rows = rows[(rows >= base) & (rows < n_pad)]
                A[np.ix_(rows, np.arange(k0, base))] = snap[np.ix_(rows, np.arange(k0, base))] @ Linv[k].T
        assert seen == set(P["cols"][P["col_off"][l]:P["col_off"][l + 1]].tolist())
        snap = A.copy()
        written = set()
        for ti, tj, so, ns, col_min, _ in P["targets"][P["upd_off"][l]:P["upd_off"][l + 1]]:
            assert (ti, tj) not in written and ti >= tj
            written.add((ti, tj))
            r = np.arange(ti * TILE, (ti + 1) * TILE); c = np.arange(tj * TILE, (tj + 1) * TILE)
            acc = np.zeros((TILE, TILE))
            for k in P["sources"][so:so + ns]:
                k0, base = k * NB, (k + 1) * NB
                a = snap[np.ix_(r, np.arange(k0, base))] * (r >= base)[:, None]
                bb = snap[np.ix_(c, np.arange(k0, base))] * (c >= base)[:, None]
                acc += a @ bb.T
            mask = (c[None, :] <= r[:, None]) & (c[None, :] >= col_min)
            A[np.ix_(r, c)] = np.where(mask, snap[np.ix_(r, c)] - acc, snap[np.ix_(r, c)])
        for tile, so, ns, _ in P["ftargets"][P["fwd_off"][l]:P["fwd_off"][l + 1]]:
            r = np.arange(tile * TILE, (tile + 1) * TILE)
            for k in P["fsources"][so:so + ns]:
                k0, base = k * NB, (k + 1) * NB
                b[r] -= (snap[np.ix_(r, np.arange(k0, base))] @ y[k0:base]) * (r >= base)
    x = b.copy()
    if tail is not None:
        # k_nd_tail / k_nd_tail_bwd: the rows from the tail's first column on are ONE dense block — tile Cholesky in 64 x 64 tiles, column by column, the forward
        # substitution with it, then the backward substitution by tile columns in descending order
        r0 = tail[0] * NB; T = (n_pad - r0) // TILE
        assert r0 % TILE == 0 and T >= 1
        inv = {}
        sl = lambda q: slice(r0 + q * TILE, r0 + (q + 1) * TILE)
        for j in range(T):
            for i in range(j, T):
                C = A[sl(i), sl(j)].copy()
                for k in range(j):
                    C -= A[sl(i), sl(k)] @ A[sl(j), sl(k)].T
                if i == j:
                    C = np.tril(C) + np.tril(C, -1).T
                    inv[j] = np.linalg.inv(np.linalg.cholesky(C))
                    acc = np.zeros(TILE)
                    for k in range(j):
                        acc += A[sl(j), sl(k)] @ y[sl(k)]
                    y[sl(j)] = inv[j] @ (b[sl(j)] - acc)
                else:
                    A[sl(i), sl(j)] = C @ inv[j].T
        for j in range(T - 1, -1, -1):
            v = y[sl(j)].copy()
            for i in range(T - 1, j, -1):
                v -= A[sl(i), sl(j)].T @ x[sl(i)]
            x[sl(j)] = inv[j].T @ v
    for l in range(n_main - 1, -1, -1):
        xs = x.copy()
        for k in P["cols"][P["col_off"][l]:P["col_off"][l + 1]]:
            k0, base = k * NB, (k + 1) * NB
            v = y[k0:base].copy()
            for t in P["row_tiles"][P["row_off"][k]:P["row_off"][k + 1]]:
                r = np.arange(t * TILE, (t + 1) * TILE)
                v -= A[np.ix_(r, np.arange(k0, base))].T @ (xs[r] * (r >= base))
            x[k0:base] = Linv[k].T @ v
    return x[nw], A


def flow_plan(lib, n, rows, cols, leaf=8):
    rows = np.ascontiguousarray(rows, np.int32); cols = np.ascontiguousarray(cols, np.int32)
    sizes = np.zeros(6, np.int64)
    args = (n, len(rows) // 6, _p(rows, ctypes.c_int), _p(cols, ctypes.c_int), NB, leaf, _p(sizes, ctypes.c_longlong))
    lib.chk_spd_flow.restype = ctypes.c_int
    lib.chk_spd_flow(*args, None, None, None, None, None)
    TC, nt, ns, nb, depth, finals = [int(v) for v in sizes]
    F = dict(finals=finals, tasks=np.zeros((max(nt, 1), 8), np.int32), sources=np.zeros((max(ns, 1), 4), np.int32), col_order=np.zeros(max(TC, 1), np.int32), below_off=np.zeros(TC + 1, np.int32),
             below=np.zeros(max(nb, 1), np.int32))
    ok = lib.chk_spd_flow(*args, *[_p(F[k], ctypes.c_int) for k in ("tasks", "sources", "col_order", "below_off", "below")])
    F.update(tile_cols=TC, n_tasks=nt, n_sources=ns, depth=depth, ready=bool(ok))
    return F


def solve_by_flow(P, F, M, rhs):
    """The arithmetic of k_nd_flow / k_nd_flow_bwd (csrc/pvlm_linalg.hip) in numpy with EXACTLY the task lists: tasks run in list order and may only read tiles that an
    EARLIER task has published (a list that named a later task would deadlock the launch: asserted), sources are added in list order, nothing outside the lists is touched."""
    n_pad = P["n_pad"]; nw = P["new_of_old"]
    A = np.eye(n_pad); b = np.zeros(n_pad)
    A[np.ix_(nw, nw)] = M; b[nw] = rhs
    A = np.tril(A)
    sl = lambda q: slice(q * TILE, (q + 1) * TILE)
    published = np.zeros(F["n_tasks"], bool); inv = {}; y = np.zeros(n_pad)
    diag_done = set(); final_of = {}
    for tid in range(F["n_tasks"]):
        I, J, so, ns, prev, final = [int(v) for v in F["tasks"][tid][:6]]
        assert I >= J
        if prev >= 0:                                             # the chunk before this task has left its partial sums in the tile
            assert prev < tid and published[prev] and tuple(F["tasks"][prev][:2]) == (I, J) and F["tasks"][prev][5] == 0
        acc = np.zeros((TILE, TILE)); acc_y = np.zeros(TILE)
        for K, ta, tb, _ in F["sources"][so:so + ns]:
            assert ta < tid and tb < tid and published[ta] and published[tb], (tid, K)
            assert tuple(F["tasks"][ta][:2]) == (I, K) and tuple(F["tasks"][tb][:2]) == (J, K) and F["tasks"][ta][5] == 1 and F["tasks"][tb][5] == 1
            acc += A[sl(I), sl(K)] @ A[sl(J), sl(K)].T
            if I == J:
                assert K in diag_done
                acc_y += A[sl(J), sl(K)] @ y[sl(K)]
        C = A[sl(I), sl(J)] - acc
        if not final:
            A[sl(I), sl(J)] = C
            if I == J: b[sl(J)] -= acc_y
        elif I == J:
            assert (I, J) not in final_of
            C = np.tril(C) + np.tril(C, -1).T
            inv[J] = np.linalg.inv(np.linalg.cholesky(C))
            y[sl(J)] = inv[J] @ (b[sl(J)] - acc_y)
            diag_done.add(J)
        else:
            assert J in diag_done and (I, J) not in final_of
            A[sl(I), sl(J)] = C @ inv[J].T
        if final: final_of[(I, J)] = tid
        published[tid] = True
    assert len(final_of) == F["finals"]
    x = np.zeros(n_pad); solved = set()
    for J in F["col_order"][::-1]:
        v = y[sl(J)].copy()
        for I in F["below"][F["below_off"][J]:F["below_off"][J + 1]][::-1]:
            assert I in solved and I > J
            v -= A[sl(I), sl(J)].T @ x[sl(I)]
        x[sl(J)] = inv[J].T @ v
        solved.add(int(J))
    return x[nw], A


def proximity_pairs(rng, P, degree):
    """Poses scattered over a floor, every pose tied to its `degree` nearest ones (what FindNeighbors produces on a trajectory that revisits a room), + self blocks."""
    xy = rng.uniform(0, 1, size=(P, 2))
    d = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    pairs = [(p, p) for p in range(P)]
    seen = set()
    for p in range(P):
        for q in np.argsort(d[p])[1:degree + 1]:
            e = (min(p, int(q)), max(p, int(q)))
            if e not in seen:
                seen.add(e); pairs.append(e)
    return pairs


@pytest.mark.parametrize("shape", ["chain", "chain_with_loops", "proximity", "two_components", "tiny"])
def test_level_schedule_solves_the_system_with_its_own_lists(chk, shape):
    chk.chk_spd_levels.restype = ctypes.c_int
    rng = np.random.default_rng(11)
    if shape == "chain":
        P = 90; pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(P, p + 4))]
    elif shape == "chain_with_loops":
        P = 120; pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(P, p + 3))] + [(3, 97), (10, 60), (11, 61), (40, 118)]
    elif shape == "proximity":
        P = 260; pairs = proximity_pairs(rng, P, 7)
    elif shape == "two_components":
        P = 80; pairs = [(p, p) for p in range(P)] + [(p, p + 1) for p in range(39)] + [(p, p + 1) for p in range(40, 79)] + [(p, p + 2) for p in range(40, 78)]
    else:
        P = 3; pairs = [(0, 0), (1, 1), (2, 2), (0, 1)]
    n, rows, cols, M = system(rng, P, pairs, constant={(0, 0), (0, 1)})
    plan_ = levels_plan(chk, n, rows, cols, leaf=6)
    assert plan_["ordered"] and plan_["n_pad"] % TILE == 0 and plan_["n_pad"] >= n
    assert len(set(plan_["new_of_old"].tolist())) == n and plan_["new_of_old"].max() < plan_["n_pad"]
    assert sorted(plan_["cols"].tolist()) == list(range(plan_["block_cols"]))
    rhs = rng.normal(size=n)
    x, _ = solve_by_levels(plan_, M, rhs)
    want = np.linalg.solve(M, rhs)
    assert np.abs(x - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    if shape in ("chain", "proximity", "chain_with_loops"):
        assert plan_["levels"] < 0.75 * plan_["block_cols"], (plan_["levels"], plan_["block_cols"])      # the schedule is shorter than the column-by-column chain
    # the dense tail (pvlm_spd::plan_tail): the last levels hold one column each, the last columns of the matrix in ascending order from a 64-row tile on; the
    # same system solved with those levels replaced by the tile Cholesky of the trailing block
    out = np.zeros(4, np.int32)
    r32 = np.ascontiguousarray(rows, np.int32); c32 = np.ascontiguousarray(cols, np.int32)
    chk.chk_spd_tail(n, len(r32) // 6, _p(r32, ctypes.c_int), _p(c32, ctypes.c_int), NB, 6, 2, _p(out, ctypes.c_int))
    col0, n_main = int(out[0]), int(out[1])
    C_, L_ = plan_["block_cols"], plan_["levels"]
    assert 0 <= col0 <= C_ and (C_ - col0) == (L_ - n_main)
    if out[2] < C_:
        assert C_ - out[2] >= 8 and out[2] >= col0                   # what plan_levels keeps: the same run, from eight columns on
    if col0 < C_:
        assert col0 % (TILE // NB) == 0
        for q, l in enumerate(range(n_main, L_)):
            assert plan_["cols"][plan_["col_off"][l]:plan_["col_off"][l + 1]].tolist() == [col0 + q]
        assert all(plan_["cols"][q] < col0 for q in range(plan_["col_off"][n_main]))
        x_t, _ = solve_by_levels(plan_, M, rhs, tail=(col0, n_main))
        assert np.abs(x_t - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    if shape in ("proximity", "chain_with_loops"):
        assert col0 < C_, "a separator of several block columns at the top of the dissection"
    # the same factorisation as tasks of ONE launch (pvlm_spd::plan_flow): every tile of the factor a task, tasks in an order in which each depends only on earlier ones
    F = flow_plan(chk, n, rows, cols, leaf=6)
    assert F["ready"] and F["tile_cols"] == plan_["n_pad"] // TILE and sorted(F["col_order"].tolist()) == list(range(F["tile_cols"]))
    assert F["depth"] <= F["tile_cols"] and F["n_tasks"] >= F["finals"] >= F["tile_cols"]
    if shape == "proximity": assert F["n_tasks"] > F["finals"], "a graph with a top separator has tiles with early sources: chunk tasks"

    x_f, A_f = solve_by_flow(plan_, F, M, rhs)
    assert np.abs(x_f - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    # every nonzero tile of a numeric factor of the permuted, padded matrix is a task
    nw_ = plan_["new_of_old"]; Ap = np.eye(plan_["n_pad"]); Ap[np.ix_(nw_, nw_)] = M
    Lnum = np.linalg.cholesky(Ap)
    have = {(int(t[0]), int(t[1])) for t in F["tasks"][:F["n_tasks"]] if t[5] == 1}
    T_ = plan_["n_pad"] // TILE
    for I in range(T_):
        for J in range(I + 1):
            if np.abs(Lnum[I * TILE:(I + 1) * TILE, J * TILE:(J + 1) * TILE]).max() > 1e-13: assert (I, J) in have, (I, J)
    if shape in ("chain", "proximity", "chain_with_loops"):
        assert F["depth"] < 0.75 * F["tile_cols"], (F["depth"], F["tile_cols"])
