"""GPU parity of the association kernels (voxel-hash k-NN, plane fit + accept tests, ordered
compaction, line votes) against the CPU oracle through the C ABI.  Bar: correspondence indices
bit-exact in both modes of the plane fit; fp64 records bit-exact with PVLM_FLAG_ASSOC_EXACT_FIT (the
reference's QR restated, no FMA contraction), and within 1e-6 relative in the default mode, where the
plane comes from the certified fast fit (csrc/pvlm_assoc_core.h: form_plane_fast — its bound keeps the
record within 5e-7 of the QR's; observed ~1e-11) and only the accept / reject DECISION is the QR's."""
import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import synth

pytestmark = pytest.mark.gpu
KEEP = 0x100
EXACT = 0x200


def same_planes(got, want, exact):
    """rows[:, 3:7] of a residual set against the oracle's planes: bit for bit in exact mode, 1e-6 of the record's size otherwise (observed 1e-11)."""
    if exact:
        return np.array_equal(got, want, equal_nan=True)
    scale = np.maximum(1.0, np.abs(want).max(axis=1, keepdims=True)) if len(want) else 1.0
    return bool(np.all(np.abs(got - want) <= 1e-6 * scale))


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


def _check_knn(ctx, oracle, tgt, q, k, max_dist):
    import panovlm_amd as pv
    scan = pv.Scan(ctx, dict(id=0, less_xyz=tgt, corner_xyz=tgt))
    for which in (0, 1):
        idx, sqd = ctx.knn(scan, q, k, max_dist, which=which)
        oi, od = oracle.knn(tgt, q, k)
        thr2 = np.float32(max_dist) * np.float32(max_dist)
        valid = od <= thr2
        exp_i = np.where(valid, oi, -1)
        exp_d = np.where(valid, od, np.float32(np.inf))
        assert np.array_equal(idx, exp_i), (np.argwhere(idx != exp_i)[:5], idx[idx != exp_i][:5], exp_i[idx != exp_i][:5])
        assert np.array_equal(sqd, exp_d)
    scan.close()


def test_knn_random_cloud(ctx, oracle):
    rng = np.random.default_rng(11)
    tgt = (rng.normal(size=(5000, 3)) * 2).astype(np.float32)
    q = (rng.normal(size=(3000, 3)) * 2.2).astype(np.float32)
    _check_knn(ctx, oracle, tgt, q, 10, 1.0)
    _check_knn(ctx, oracle, tgt, q, 5, 0.3)


def test_knn_vlp_geometry_and_ties(ctx, oracle):
    a = sy.make_scan(2, cols=1024)["flat_xyz"]
    b = sy.make_scan(3, cols=1024)["flat_xyz"]
    _check_knn(ctx, oracle, a, b, 10, 1.0)
    # duplicated targets -> exact distance ties, resolved by ascending index on both sides
    dup = np.concatenate([a[:2000], a[:2000]])
    _check_knn(ctx, oracle, dup, b[:1500], 10, 1.0)
    # queries identical to targets (distance 0)
    _check_knn(ctx, oracle, a[:3000], a[:3000], 5, 0.5)


def _compare_assoc(ctx, oracle, scans, pairs, tol, thr):
    return [_compare_assoc_mode(ctx, oracle, scans, pairs, tol, thr, exact) for exact in (True, False)][0]


def _compare_assoc_mode(ctx, oracle, scans, pairs, tol, thr, exact):
    import panovlm_amd as pv
    dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}
    rs = ctx.assoc_point2plane([dev[r] for r, _ in pairs], [dev[n] for _, n in pairs], tol, thr,
                               kind=pv.POINT2PLANE_ANGLE, flags=pv.FLAG_NORMALIZE_DISTANCE | KEEP | (EXACT if exact else 0))
    assert exact or rs.assoc_exact_fits() <= 0.02 * max(rs.n, 50), (rs.assoc_exact_fits(), rs.n)      # the fast fit answers almost always
    assert not exact or rs.assoc_exact_fits() == 0
    off, ref, nei, rows = rs.download()
    qidx, nn = rs.assoc_debug()
    assert list(ref) == [scans[r]["id"] for r, _ in pairs] and list(nei) == [scans[n]["id"] for _, n in pairs]
    total = 0
    for p, (r, n) in enumerate(pairs):
        o = oracle.assoc_point2plane(scans[r], scans[n], tol, thr)
        s, e = off[p], off[p + 1]
        assert e - s == len(o["qidx"]), (p, e - s, len(o["qidx"]))
        assert np.array_equal(qidx[s:e], o["qidx"])
        assert np.array_equal(nn[s:e], o["nn"])
        assert np.array_equal(rows[s:e, 0:3], o["point"]), np.abs(rows[s:e, 0:3] - o["point"]).max()
        assert same_planes(rows[s:e, 3:7], o["plane"], exact), np.abs(rows[s:e, 3:7] - o["plane"]).max()
        total += e - s
    assert rs.n == total
    rs.close()
    for d in dev.values():
        d.close()
    return total


def test_point2plane_dense_synthetic(ctx, oracle):
    scans = {k: sy.make_scan(k, cols=512) for k in (0, 1, 2)}
    n = _compare_assoc(ctx, oracle, scans, [(0, 1), (1, 0), (1, 2), (2, 0)], 0.05, 1.0)
    assert n > 1000
    _compare_assoc(ctx, oracle, scans, [(0, 1), (2, 1)], 0.01, 1.0)   # Floor tolerance


def test_point2plane_downsampled_targets_and_small_threshold(ctx, oracle):
    scans = {k: sy.make_scan(k, cols=1024, downsample_targets=0.2) for k in (5, 6)}
    n = _compare_assoc(ctx, oracle, scans, [(5, 6), (6, 5)], 0.05, 1.0)
    assert n > 500
    _compare_assoc(ctx, oracle, scans, [(5, 6)], 0.05, 0.35)        # many queries fail the 10th-neighbour test


def test_point2plane_edge_cases(ctx, oracle):
    rng = np.random.default_rng(3)
    base = sy.make_scan(1, cols=256)
    few = dict(base); few["id"] = 7; few["less_xyz"] = base["less_xyz"][:9]; few["less_tag"] = base["less_tag"][:9]   # < 10 targets
    empty_q = dict(base); empty_q["id"] = 8; empty_q["flat_xyz"] = np.zeros((0, 3), np.float32); empty_q["flat_tag"] = np.zeros(0, np.float32)
    mixed = dict(sy.make_scan(2, cols=256)); mixed["id"] = 9
    tags = np.where(rng.uniform(size=len(mixed["less_xyz"])) < 0.2, 16.0, 1.0).astype(np.float32)  # POINT_GROUND sprinkled in
    mixed["less_tag"] = tags
    mixed["flat_tag"] = np.where(rng.uniform(size=len(mixed["flat_xyz"])) < 0.5, 16.0, 1.0).astype(np.float32)
    far = dict(sy.make_scan(3, cols=256)); far["id"] = 10
    far["flat_xyz"] = far["flat_xyz"] + np.float32(50.0)                    # queries far outside the target grid
    scans = {1: base, 7: few, 8: empty_q, 9: mixed, 10: far}
    _compare_assoc(ctx, oracle, scans, [(7, 1), (1, 8), (9, 1), (1, 9), (9, 9), (1, 10), (1, 1)], 0.05, 1.0)


def test_point2plane_residual_set_is_evaluable(ctx, oracle):
    """The association output feeds the evaluation kernels directly (device-resident hand-off)."""
    import panovlm_amd as pv
    scans = {k: sy.make_scan(k, cols=256) for k in (0, 1)}
    dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}
    rs = ctx.assoc_point2plane([dev[0], dev[1]], [dev[1], dev[0]], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE, flags=1)
    aa, t = zip(*[sy.pose_params(scans[k]["R_wl"], scans[k]["t_wl"]) for k in (0, 1)])
    aa, t = np.array(aa), np.array(t)
    ctx.set_poses(aa, t)
    r, J = rs.eval()
    off, ref, nei, rows = rs.download()
    rid, nid = synth.expand_ids(off, ref, nei)
    ro, Jo = oracle.evaluate(1, synth.oracle_rows(1, rows), rid, nid, aa, t, normalize=True)
    assert np.allclose(r, ro, rtol=1e-6, atol=1e-12) and np.allclose(J, Jo, rtol=1e-6, atol=1e-9)
    assert np.median(r[r > 0]) < 0.05   # radians: scans are ~cm-aligned
    with pytest.raises(pv.PvlmError):
        rs.assoc_debug()                # indices were not requested
    rs.close()


def test_line2line_votes(ctx, oracle):
    import panovlm_amd as pv
    rng = np.random.default_rng(21)
    lines = synth.random_world_lines(rng, 14)
    Ra, ta = sy.estimated_pose(3); Rb, tb = sy.estimated_pose(4)
    a = synth.make_line_scan(rng, 3, Ra, ta, lines[:12])
    b = synth.make_line_scan(rng, 4, Rb, tb, lines[2:])
    da, db = pv.Scan(ctx, a), pv.Scan(ctx, b)
    for thr in (0.3, 0.4, 0.05):
        v = ctx.line2line_votes(da, db, thr)
        o = oracle.assoc_line2line(a, b, thr)
        assert v.shape == o["votes"].shape and np.array_equal(v, o["votes"])
        v2 = ctx.line2line_votes(db, da, thr)
        assert np.array_equal(v2, oracle.assoc_line2line(b, a, thr)["votes"])
    assert v.sum() > 0
    # batched form (one launch for many pairs, incl. a scan without segments and a repeated pair) == pair by pair
    empty = dict(a); empty.update(p2s=[[] for _ in a["p2s"]], seg_size=np.zeros(0, np.int32), seg_coeffs=np.zeros((0, 6)), end_points=np.zeros((0, 6)))
    de = pv.Scan(ctx, empty)
    refs, neis = [da, db, de, da, db], [db, da, da, de, da]
    got = ctx.line2line_votes_batch(refs, neis, 0.3)
    for r, n, g in zip(refs, neis, got):
        assert g.shape == (n.n_segments, r.n_segments) and np.array_equal(g, ctx.line2line_votes(r, n, 0.3))
    assert ctx.line2line_votes_batch([], [], 0.3) == []
    # the row maxima of the blocks (pvlm_line2line_best_batch; since round 6 its votes come from the thread-per-point kernel, which decides dist > thr on the squared
    # distance against the largest double whose root does not exceed thr): the arg-max of the blocks above, first of equals, for thresholds on both sides of the data
    for thr in (0.3, 0.05, 0.4, 0.0, 2.5, float("inf")):
        blocks = ctx.line2line_votes_batch(refs, neis, thr)
        roff, col, cnt = ctx.line2line_best_batch(refs, neis, thr)
        at = 0
        for p, (r, nb, g) in enumerate(zip(refs, neis, blocks)):
            assert roff[p] == at
            if r.n_segments == 0:
                continue
            for row in range(nb.n_segments):
                assert col[at] == int(np.argmax(g[row])) and cnt[at] == int(g[row].max()), (thr, p, row)
                at += 1
        assert roff[-1] == at
    da.close(); db.close(); de.close()


def test_batched_scan_upload_equals_scan_by_scan(ctx, oracle):
    """pvlm_scan_upload_batch (one staging copy, one slab, one grid build for all scans) hands out scans that behave exactly
    like individually uploaded ones: k-NN in both target clouds, the association's residual set and the vote matrices are
    bit-identical; scans of a batch can be destroyed in any order; a staging window smaller than the batch (several
    copies) gives the same device contents."""
    import os
    import panovlm_amd as pv
    rng = np.random.default_rng(31)
    lines = synth.random_world_lines(rng, 12)
    host = []
    for k in range(7):
        s = sy.make_scan(k, cols=512, downsample_targets=0.2 if k % 2 else 0.0)
        R, t = sy.estimated_pose(k)
        ls = synth.make_line_scan(rng, k, R, t, lines[k % 3:k % 3 + 8])
        s.update(corner_xyz=ls["corner_xyz"], p2s=ls["p2s"], seg_size=ls["seg_size"], seg_coeffs=ls["seg_coeffs"], end_points=ls["end_points"])
        host.append(s)
    host[5] = dict(id=5, R_wl=host[5]["R_wl"], t_wl=host[5]["t_wl"])                      # a scan without any cloud
    host[6]["less_xyz"] = (rng.normal(size=(3000, 3)) * [40, 40, 0.01]).astype(np.float32)  # flat and wide: hashed table
    host[6]["less_tag"] = np.ones(3000, np.float32)
    single = [pv.Scan(ctx, s) for s in host]

    def compare(batch):
        q = (rng.normal(size=(500, 3)) * 3).astype(np.float32)
        for a, b, h in zip(single, batch, host):
            assert (a.n_flat, a.n_less, a.n_corner, a.n_segments) == (b.n_flat, b.n_less, b.n_corner, b.n_segments)
            for which, n in ((0, a.n_less), (1, a.n_corner)):
                if n == 0:
                    continue
                ia, da = ctx.knn(a, q, 10, 2.0, which=which); ib, db = ctx.knn(b, q, 10, 2.0, which=which)
                assert np.array_equal(ia, ib) and np.array_equal(da, db)
        pairs = [(0, 1), (1, 0), (2, 3), (3, 4), (4, 6), (6, 2), (5, 1), (1, 5)]
        ra = ctx.assoc_point2plane([single[r] for r, _ in pairs], [single[n] for _, n in pairs], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE, flags=pv.FLAG_NORMALIZE_DISTANCE)
        rb = ctx.assoc_point2plane([batch[r] for r, _ in pairs], [batch[n] for _, n in pairs], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE, flags=pv.FLAG_NORMALIZE_DISTANCE)
        xa, xb = ra.download(), rb.download()
        assert ra.n == rb.n > 1000 and np.array_equal(xa[0], xb[0]) and np.array_equal(xa[1], xb[1])
        ra.close(); rb.close()
        va = ctx.line2line_votes_batch([single[0], single[2]], [single[1], single[3]], 0.4)
        vb = ctx.line2line_votes_batch([batch[0], batch[2]], [batch[1], batch[3]], 0.4)
        for x, y in zip(va, vb):
            assert np.array_equal(x, y)
        assert sum(int(x.sum()) for x in va) > 0

    batch = pv.Scan.upload_batch(ctx, host)
    assert len(batch) == len(host)
    compare(batch)
    # the slab lives until the last scan of the batch goes: destroy some, the others keep working
    for k in (0, 3, 5):
        batch[k].close()
    i1, d1 = ctx.knn(batch[1], host[0]["flat_xyz"][:200], 10, 1.0)
    i2, d2 = ctx.knn(single[1], host[0]["flat_xyz"][:200], 10, 1.0)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    for b in batch:
        b.close()
    # staging window of 64 KiB: the front of the slab goes over in many pieces
    os.environ["PVLM_UPLOAD_STAGE_MB"] = "0.0625"
    try:
        c2 = pv.Context(0)                                # a fresh context: its staging buffer is sized by the small window
        try:
            small = pv.Scan.upload_batch(c2, host)
            q = host[0]["flat_xyz"][:300]
            for a, b in zip(single, small):
                if a.n_less:
                    ia, da = ctx.knn(a, q, 10, 1.0); ib, db = c2.knn(b, q, 10, 1.0)
                    assert np.array_equal(ia, ib) and np.array_equal(da, db)
            for b in small:
                b.close()
        finally:
            c2.close()
    finally:
        del os.environ["PVLM_UPLOAD_STAGE_MB"]
    assert pv.Scan.upload_batch(ctx, []) == []
    # all or nothing: one bad scan (NaN) fails the whole batch loudly
    bad = dict(host[1]); bad["less_xyz"] = host[1]["less_xyz"].copy(); bad["less_xyz"][7, 1] = np.nan
    with pytest.raises(pv.PvlmError):
        pv.Scan.upload_batch(ctx, [host[0], bad])
    for a in single:
        a.close()


def test_knn_results_larger_than_the_staging_arena(ctx, oracle):
    """Host<->device copies go through a 32 MB pinned arena in pieces of at most 16 MB; results are delivered to the caller's
    buffers when the call synchronises.  900 k queries x 10 neighbours = 36 MB of indices + 36 MB of distances + an 11 MB
    query upload: several wraps of the arena inside one call.  Checked against the oracle on a slice and against two half
    calls everywhere."""
    import panovlm_amd as pv
    rng = np.random.default_rng(41)
    tgt = (rng.normal(size=(4000, 3)) * 2).astype(np.float32)
    q = (rng.normal(size=(900_000, 3)) * 2.1).astype(np.float32)
    scan = pv.Scan(ctx, dict(id=0, less_xyz=tgt))
    idx, sqd = ctx.knn(scan, q, 10, 1.0)
    h = len(q) // 2
    i1, d1 = ctx.knn(scan, q[:h], 10, 1.0); i2, d2 = ctx.knn(scan, q[h:], 10, 1.0)
    assert np.array_equal(idx, np.concatenate([i1, i2])) and np.array_equal(sqd, np.concatenate([d1, d2]))
    sel = slice(450_000 - 1500, 450_000 + 1500)                     # across the boundary of the copy pieces
    oi, od = oracle.knn(tgt, q[sel], 10)
    valid = od <= np.float32(1.0)
    assert np.array_equal(idx[sel], np.where(valid, oi, -1)) and np.array_equal(sqd[sel], np.where(valid, od, np.float32(np.inf)))
    assert (idx >= 0).mean() > 0.3
    # a single read-back above 64 MB bypasses the arena (the runtime's own pageable path): 1.8 M queries x 10 x 4 B = 72 MB
    big = np.concatenate([q, q])
    ib, db = ctx.knn(scan, big, 10, 1.0)
    assert np.array_equal(ib[:len(q)], idx) and np.array_equal(ib[len(q):], idx) and np.array_equal(db[:len(q)], sqd)
    scan.close()


@pytest.mark.parametrize("n", [1, 2, 37, 1024, 1593, 2500])
def test_centre_orders_are_the_sorted_distance_keys_of_find_neighbors(n):
    """K28 pvlm_centre_orders — the two searches of FindNeighbors (lidar_mapping/LidarFeatureAssociate.cpp:19-111: nearestKSearch + radiusSearch over the scan centres)
    for all scans at once: row i = the positions j ordered by the key (bits of the float32 squared distance ((dx*dx)+dy*dy)+dz*dz from centre i) << 32 | j.  Bit-exact against the
    same float chain in numpy, ties (coincident centres) in index order; the row's first key is the scan itself."""
    import panovlm_amd as pv
    rng = np.random.default_rng(n)
    xyz = rng.uniform(-15, 15, size=(n, 3)).astype(np.float32)
    if n > 40:
        xyz[7] = xyz[3]; xyz[n - 1] = xyz[3]          # coincident centres: equal distances, the position decides
    ctx = pv.Context(0)
    keys = ctx.centre_orders(xyz)
    assert keys.shape == (n, n) and keys.dtype == np.uint16
    for i in sorted(set([0, 3 % n, 7 % n, n // 2, n - 1])):
        d = xyz[i][None, :] - xyz
        s = (d[:, 0] * d[:, 0]).astype(np.float32)
        s = (s + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
        s = (s + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
        want = np.sort((s.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64))
        assert np.array_equal(keys[i], (want & np.uint64(0xffff)).astype(np.uint16)), i
        assert s[int(keys[i][0])] == 0                                    # distance zero first ...
    again = ctx.centre_orders(xyz)
    assert np.array_equal(keys, again)
    with pytest.raises(Exception):
        ctx.centre_orders(np.zeros((5000, 3), np.float32))                # beyond the kernel's bound: the caller keeps its own search
    ctx.close()
