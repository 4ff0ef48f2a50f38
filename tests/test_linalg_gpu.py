"""GPU dense / block-sparse SPD solve of the LM driver (pvlm_spd_solve*: hand-written blocked Cholesky) against numpy,
and the host mirror's Solve with the GPU factorisation forced on against the CPU twin."""
import os

import numpy as np
import pytest

from tests import host_io, lm_twin, synth

pytestmark = pytest.mark.gpu


# driven through the C++ test driver (the real caller of these entry points)
def _run_spd(tmp_path, A, B):
    import struct
    n = A.shape[0]; B2 = np.asarray(B, np.float64).reshape(n, -1)
    path = os.path.join(str(tmp_path), "spd.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, B2.shape[1])); f.write(np.ascontiguousarray(A, np.float64).tobytes()); f.write(np.asfortranarray(B2).tobytes(order="F"))
    out = host_io.run("spd", path)
    info = int(out[0].split()[1])
    x = np.array([float.fromhex(l.split()[1]) for l in out[1:]])
    return (x.reshape(B2.shape[1], n).T.reshape(np.shape(B)) if info == 0 else None), info


@pytest.mark.parametrize("n", [7, 32, 33, 300, 1800])
def test_dense_spd_solve(tmp_path, n):
    rng = np.random.default_rng(n)
    Q = rng.normal(size=(n, n))
    A = Q @ Q.T + n * np.eye(n)
    B = rng.normal(size=(n, 3))
    X, info = _run_spd(tmp_path, A, B)
    assert info == 0
    assert np.allclose(A @ X, B, rtol=0, atol=1e-9 * np.abs(B).max() * n)
    assert np.allclose(X, np.linalg.solve(A, B), rtol=1e-9, atol=1e-12)
    # not positive definite: reported, not an error
    A2 = A.copy(); A2[n // 2, n // 2] = -1.0
    _, info = _run_spd(tmp_path, A2, B)
    assert info == n // 2 + 1


def _block_system(rng, P, pairs, constant=()):
    """Poses with 6 unknowns each (those in `constant`: (pose, half) pairs are fixed), one 6x6 block per (a, b) of `pairs`."""
    off = np.full((P, 6), -1, np.int64); n = 0
    for p in range(P):
        for half in range(2):
            if (p, half) in constant:
                continue
            off[p, 3 * half:3 * half + 3] = np.arange(n, n + 3); n += 3
    rows, cols, mirror, blocks = [], [], [], []
    H = np.zeros((n, n))
    for (a, b) in pairs:
        if a == b:
            J = rng.normal(size=(9, 6)); blk = J.T @ J
        else:
            blk = rng.normal(size=(6, 6)) * 0.2
        rows.append(off[a]); cols.append(off[b]); mirror.append(int(a != b)); blocks.append(blk.reshape(-1))
        ia, ib = off[a], off[b]
        va, vb = ia >= 0, ib >= 0
        H[np.ix_(ia[va], ib[vb])] += blk[np.ix_(va, vb)]
        if a != b:
            H[np.ix_(ib[vb], ia[va])] += blk[np.ix_(va, vb)].T
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0)))
    diag = rng.uniform(0.5, 1.0, size=n) + 10.0
    rhs = rng.normal(size=n)
    M = H * scale[:, None] * scale[None, :] + np.diag(diag)
    return n, rows, cols, mirror, blocks, scale, diag, rhs, M


def _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag, rhs, repeat=1):
    import struct
    path = os.path.join(str(tmp_path), "blk.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, len(blocks)))
        f.write(np.array(rows, np.int32).tobytes()); f.write(np.array(cols, np.int32).tobytes()); f.write(np.array(mirror, np.int32).tobytes())
        f.write(np.array(blocks, np.float64).tobytes()); f.write(scale.tobytes()); f.write(diag.tobytes()); f.write(rhs.tobytes())
    out = host_io.run("spdblocks", path, repeat)
    info = int(out[0].split()[1])
    x = np.array([float.fromhex(l.split()[1]) for l in out if l.startswith("x ")])
    return info, x, out


def test_block_sparse_assembly_and_solve(tmp_path):
    """M = D (sum of 6x6 blocks) D + diag: pose-pair blocks mirrored, constant blocks (-1) dropped, repeated blocks summed."""
    rng = np.random.default_rng(5)
    P = 40                                   # poses; pose 0 fully constant, pose 1 has a constant rotation block
    pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(p + 4, P))] + [(2, 3), (5, 5)]   # two repeated
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1), (1, 0)})
    info, x, _ = _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0
    assert np.allclose(x, np.linalg.solve(M, rhs), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("P,links", [(300, "chain"), (400, "loops"), (260, "dense")])
def test_tile_sparse_factorisation(tmp_path, monkeypatch, P, links):
    """The tile-sparse plan of pvlm_spd_solve_blocks (minimum-degree + elimination-tree postorder of the pose graph, symbolic
    factorisation on 64-row x 32-column cells, per block column only the tiles that hold a nonzero): same solution as a dense
    solve — for a temporal chain, for a chain with long loop closures (fill-in), and for a graph too dense to pay (dense kernels)."""
    monkeypatch.setenv("PVLM_SPD_SPARSE_MIN", "0")
    rng = np.random.default_rng(P)
    pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(p + 3, P))]
    if links == "loops":
        pairs += [(int(a), int(b)) for a, b in zip(rng.integers(0, P, 150), rng.integers(0, P, 150)) if a < b]
    if links == "dense":
        pairs += [(p, q) for p in range(P) for q in range(p + 1, P) if rng.uniform() < 0.3]
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1), (7, 1)})
    info, x, out = _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag, rhs, repeat=2)      # the second solve reuses the cached plan
    want = np.linalg.solve(M, rhs)
    assert info == 0
    assert np.allclose(x, want, rtol=1e-9, atol=1e-12)
    plan = [l for l in out if l.startswith("plan")][0].split()
    print(" ".join(plan))
    assert (plan[1] == "tile-sparse") == (links != "dense")
    if links == "chain":
        assert float(plan[3]) < 0.15        # fraction of the dense factorisation's tile updates
    # not positive definite -> reported through info, as the dense path does
    diag2 = diag.copy(); diag2[n // 2] = -1e6
    info, _, _ = _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag2, rhs)
    assert info != 0


def test_solve_with_gpu_cholesky_matches_twin(oracle, tmp_path, monkeypatch):
    """The bundle-adjustment solve (reprojection sets + Schur complement) with the reduced camera system factorised on
    the GPU instead of the host skyline: same poses / points / step counts as the twin."""
    from tests.test_host_gpu import _bundle_scene
    monkeypatch.setenv("PVLM_GPU_CHOLESKY_MIN", "1")
    rng = np.random.default_rng(91)
    frames, tracks, _ = _bundle_scene(rng)
    fpath, spath = os.path.join(str(tmp_path), "bf.bin"), os.path.join(str(tmp_path), "bs.bin")
    host_io.write_frames(fpath, np.eye(4), frames)
    host_io.write_structure(spath, frames, tracks)
    out = host_io.run("bundle", fpath, spath, 1.5, 1, 12)
    summ = [l.split() for l in out if l.startswith("summary")][0]
    cams = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("cam ")])
    pts = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("point ")])
    res, aa, t, X = lm_twin.bundle_adjust(oracle, [dict(f) for f in frames], tracks, 1.5, True, 12)
    assert abs(float(summ[6]) - res["final_cost"]) <= 1e-6 * res["final_cost"]
    assert int(summ[8]) == res["successful"] and int(summ[10]) == res["unsuccessful"]
    assert np.abs(cams[:, :3] - aa).max() <= 1e-6 and np.abs(cams[:, 3:] - t).max() <= 1e-6
    assert np.abs(pts - X).max() <= 1e-6 * max(1.0, np.abs(X).max())


def _proximity_pairs(rng, P, degree):
    xy = rng.uniform(0, 1, size=(P, 2))
    d = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    pairs, seen = [(p, p) for p in range(P)], set()
    for p in range(P):
        for q in np.argsort(d[p])[1:degree + 1]:
            e = (min(p, int(q)), max(p, int(q)))
            if e not in seen:
                seen.add(e); pairs.append(e)
    return pairs


@pytest.mark.parametrize("shape,P", [("proximity", 700), ("chain_with_loops", 600)])
def test_nested_dissection_level_schedule(shape, P):
    """pvlm_spd_solve_blocks from 1500 unknowns on: nested dissection of the pose graph + the level schedule (csrc/pvlm_spd_plan.h: plan_levels; kernels k_nd_panel /
    k_nd_update / k_nd_bwd) — the graph FindNeighbors builds on a trajectory that keeps coming back to the same rooms (every pose tied to its nearest ones) and a
    temporal chain with loop closures.  Same solution as a dense solve and as the column-by-column plan of round 5 (PVLM_SPD_LEVELS=0, in a child process); far fewer
    dependent steps than block columns; two solves give the same bits (a workgroup per target tile, no atomics); a matrix that is not positive definite is reported.
    Upstream this is Ceres' SPARSE_SCHUR (util/Optimization.cpp:638-666)."""
    import panovlm_amd as pv
    rng = np.random.default_rng(P)
    if shape == "proximity":
        pairs = _proximity_pairs(rng, P, 9)
    else:
        pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(p + 4, P))] + \
                [(int(a), int(b)) for a, b in zip(rng.integers(0, P, 60), rng.integers(0, P, 60)) if a < b]
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1), (9, 1)})
    assert n >= 1500
    ctx = pv.Context(0)
    x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    plan = ctx.spd_plan()
    want = np.linalg.solve(M, rhs)
    assert info == 0 and np.allclose(x, want, rtol=1e-9, atol=1e-12)
    assert plan["tile_sparse"] and 0 < plan["levels"] <= 0.67 * plan["block_columns"] and plan["padded_rows"] % 64 == 0 and plan["padded_rows"] >= n, plan
    x2, _ = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert np.array_equal(x, x2)
    diag_bad = diag.copy(); diag_bad[n // 3] = -1e6
    _, info_bad = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag_bad, rhs)
    assert info_bad != 0
    ctx.close()
    # the round-5 plan on the same system, in a child process (the switch is read once per process)
    import pickle, subprocess, sys, tempfile
    with tempfile.TemporaryDirectory() as d:
        pickle.dump((n, rows, cols, mirror, blocks, scale, diag, rhs), open(os.path.join(d, "sys.pkl"), "wb"))
        code = ("import pickle, sys, numpy as np; sys.path.insert(0, %r); import panovlm_amd as pv; a = pickle.load(open(%r, 'rb')); ctx = pv.Context(0); "
                "x, info = ctx.spd_solve_blocks(*a); p = ctx.spd_plan(); pickle.dump((x, info, p), open(%r, 'wb'))") % (host_io.ROOT, os.path.join(d, "sys.pkl"), os.path.join(d, "out.pkl"))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PVLM_SPD_LEVELS="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        x_old, info_old, plan_old = pickle.load(open(os.path.join(d, "out.pkl"), "rb"))
    assert info_old == 0 and plan_old["levels"] == 0
    assert np.abs(x - x_old).max() <= 1e-11 * max(1.0, np.abs(x_old).max())


def _solve_in_child(args, env):
    import pickle, subprocess, sys, tempfile
    with tempfile.TemporaryDirectory() as d:
        pickle.dump(args, open(os.path.join(d, "sys.pkl"), "wb"))
        code = ("import pickle, sys, numpy as np; sys.path.insert(0, %r); import panovlm_amd as pv; a = pickle.load(open(%r, 'rb')); ctx = pv.Context(0); "
                "x, info = ctx.spd_solve_blocks(*a); x2, _ = ctx.spd_solve_blocks(*a); p = ctx.spd_plan(); pickle.dump((x, info, p, bool(np.array_equal(x, x2))), open(%r, 'wb'))") % (
                    host_io.ROOT, os.path.join(d, "sys.pkl"), os.path.join(d, "out.pkl"))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return pickle.load(open(os.path.join(d, "out.pkl"), "rb"))


@pytest.mark.parametrize("P,degree", [(700, 9), (1100, 14)])
def test_factorisation_in_one_launch_and_the_dense_tail(P, degree):
    """k_nd_flow / k_nd_flow_bwd (the default from 1500 unknowns on): the whole tile-sparse factorisation as the tasks of ONE launch (pvlm_spd::plan_flow — a task per
    tile of the factor, workgroups take them in dependency order by a ticket and hand finished tiles to each other inside the launch), the backward substitution as a
    second.  PVLM_SPD_FLOW=0: the level launches with the dense TAIL (k_nd_tail: the top separator, a level per block column, in one launch); PVLM_SPD_TAIL=0 as well:
    every level by its launches.  Same solution from all three and numpy; repeated solves give the same bits whatever order the workgroups ran in; a pivot that fails
    anywhere — the last unknowns of the order, inside the tail, included — is reported and nobody hangs waiting for its tile."""
    import panovlm_amd as pv
    rng = np.random.default_rng(P + degree)
    pairs = _proximity_pairs(rng, P, degree)
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1), (9, 1)})
    ctx = pv.Context(0)
    x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    plan = ctx.spd_plan()
    assert plan["levels"] > 0 and plan["tail_block_columns"] == plan["block_columns"] and plan["launched_levels"] == 0, plan
    want = np.linalg.solve(M, rhs)
    assert info == 0 and np.allclose(x, want, rtol=1e-9, atol=1e-12)
    for _ in range(5):
        x2, info2 = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
        assert info2 == 0 and np.array_equal(x, x2)
    last = 0
    for q in list(range(0, n, max(1, n // 12))) + [n - 1]:
        diag_bad = diag.copy(); diag_bad[q] = -1e7
        _, info_bad = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag_bad, rhs)
        assert info_bad != 0, q
        last = max(last, info_bad)
    assert last > plan["padded_rows"] // 2, "none of the broken diagonals fell into the second half of the order"
    x3, info3 = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)      # and the context is as good as before
    assert info3 == 0 and np.array_equal(x, x3)
    # pvlm_spd_one_launch: the switch of a context (what ranks sharing a GPU use); the plan of the same structure is made again for the other form; no solve here had
    # to be redone
    assert ctx.spd_one_launch(False) == 0
    x4, info4 = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    plan4 = ctx.spd_plan()
    assert info4 == 0 and plan4["launched_levels"] > 0 and plan4["launched_levels"] + plan4["tail_block_columns"] == plan4["levels"], plan4
    assert np.abs(x4 - x).max() <= 1e-11 * max(1.0, np.abs(x).max())
    assert ctx.spd_one_launch(True) == 0
    x5, info5 = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info5 == 0 and np.array_equal(x5, x) and ctx.spd_plan()["launched_levels"] == 0
    ctx.close()
    args = (n, rows, cols, mirror, blocks, scale, diag, rhs)
    x_t, info_t, plan_t, same_t = _solve_in_child(args, {"PVLM_SPD_FLOW": "0"})
    assert info_t == 0 and same_t and plan_t["tail_block_columns"] >= 8 and plan_t["tail_block_columns"] % 2 == 0, plan_t
    assert plan_t["launched_levels"] + plan_t["tail_block_columns"] == plan_t["levels"] == plan["levels"], plan_t
    x_l, info_l, plan_l, same_l = _solve_in_child(args, {"PVLM_SPD_FLOW": "0", "PVLM_SPD_TAIL": "0"})
    assert info_l == 0 and same_l and plan_l["tail_block_columns"] == 0 and plan_l["launched_levels"] == plan_l["levels"] == plan["levels"]
    scale_x = max(1.0, np.abs(x_l).max())
    assert np.abs(x - x_l).max() <= 1e-11 * scale_x and np.abs(x_t - x_l).max() <= 1e-11 * scale_x
    # a broken diagonal inside the tail, on the path with the tail
    diag_bad = diag.copy(); diag_bad[n - 1] = -1e7
    _, info_bt, _, _ = _solve_in_child((n, rows, cols, mirror, blocks, scale, diag_bad, rhs), {"PVLM_SPD_FLOW": "0"})
    assert info_bt != 0


def test_a_one_launch_solve_that_does_not_get_through_is_redone_with_the_level_launches():
    """The recovery path of the one-launch factorisation: a task that never publishes its tile (the test hook of pvlm_spd_one_launch — what starvation on a GPU shared
    by many processes looks like) makes every workgroup that needs it run into the 2 s limit; the library then redoes the SAME solve with the level launches, keeps
    them for the context and counts the event.  The caller sees a correct solution and info 0."""
    import time
    import panovlm_amd as pv
    rng = np.random.default_rng(77)
    P = 700
    pairs = _proximity_pairs(rng, P, 9)
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1)})
    ctx = pv.Context(0)
    x_flow, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0 and ctx.spd_plan()["launched_levels"] == 0 and ctx.spd_one_launch() == 0
    ctx.spd_one_launch(2 + 40)                        # task 40: a tile of a leaf column that others need
    t0 = time.perf_counter()
    x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    waited = time.perf_counter() - t0
    plan = ctx.spd_plan()
    assert info == 0 and ctx.spd_one_launch() == 1 and plan["launched_levels"] > 0, (info, plan)
    assert 1.5 < waited < 30.0, waited
    want = np.linalg.solve(M, rhs)
    assert np.allclose(x, want, rtol=1e-9, atol=1e-12) and np.abs(x - x_flow).max() <= 1e-11 * max(1.0, np.abs(x_flow).max())
    x2, info2 = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)            # the context stays with the level launches
    assert info2 == 0 and np.array_equal(x, x2) and ctx.spd_one_launch() == 1
    assert ctx.spd_one_launch(True) == 1                                                      # and back, by hand
    x3, info3 = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info3 == 0 and np.array_equal(x3, x_flow) and ctx.spd_plan()["launched_levels"] == 0
    ctx.close()


def test_plan_prefetch_is_a_hint_never_a_change_of_result():
    """pvlm_spd_plan_prefetch: the host half of the plan made on a thread of the library ahead of the solve.  Same lists: the solve takes it (hit counted) and returns the
    bits of a solve that planned for itself; other lists: the prefetch is dropped, the solve plans for itself; a second prefetch replaces the first; a context may end
    with a prefetch in flight."""
    import panovlm_amd as pv
    rng = np.random.default_rng(23)
    F = 330
    pairs = [(p, p) for p in range(F)] + [(p, q) for p in range(F) for q in range(p + 1, min(F, p + 4))] + [(5, 300), (40, 222)]
    n = 6 * (F - 1)
    off = np.arange(-6, n).reshape(F, 6); off[0] = -1
    rows = np.array([off[a] for a, b in pairs], np.int32); cols = np.array([off[b] for a, b in pairs], np.int32)
    mirror = np.array([int(a != b) for a, b in pairs], np.int32)
    blocks = np.empty((len(pairs), 36))
    for k, (a, b) in enumerate(pairs):
        if a == b:
            J = rng.normal(size=(9, 6)); blocks[k] = (J.T @ J + 30 * np.eye(6)).reshape(-1)
        else:
            blocks[k] = (rng.normal(size=(6, 6)) * 0.2).reshape(-1)
    scale = np.full(n, 0.3); diag = np.full(n, 1.0); rhs = rng.normal(size=n)
    a = pv.Context(0)
    want, info = a.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0 and a.spd_plan()["tile_sparse"] and a.spd_plan_prefetch_hits() == 0
    b = pv.Context(0)
    b.spd_plan_prefetch(n, rows, cols, mirror)
    got, info = b.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0 and b.spd_plan_prefetch_hits() == 1 and np.array_equal(got, want)
    assert b.spd_plan() == a.spd_plan()
    b.spd_plan_prefetch(n, rows, cols, mirror)                     # the context holds this plan already: nothing is started, nothing is taken
    got, _ = b.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert b.spd_plan_prefetch_hits() == 1 and np.array_equal(got, want)
    # another structure prefetched (the loop closures left out): dropped by the solve, which plans for itself
    c = pv.Context(0)
    c.spd_plan_prefetch(n, rows[:-2], cols[:-2], mirror[:-2])
    got, info = c.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0 and c.spd_plan_prefetch_hits() == 0 and np.array_equal(got, want)
    c.spd_plan_prefetch(n, rows[:-2], cols[:-2], mirror[:-2])      # replaced by the next one before any solve ...
    c.spd_plan_prefetch(n, rows[:-1], cols[:-1], mirror[:-1])      # ... and this one is still in flight when the context ends
    del c
