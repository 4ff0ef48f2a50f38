"""BASELINE.json's full size — 64 k-point VLP-16 scans (16 x 4096), every point both query and target — where the oracle
would take minutes: size-independent properties of the association and evaluation outputs, plus the oracle itself on
every 32nd query against the full 64 k target cloud (bit-exact records)."""
import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import synth

KEEP = 0x100


def check_association_properties(ref, nei, qidx, nn, point, plane, tol, thr, rng, spot=256):
    """Properties every AssociatePoint2Plane output must have (lidar_mapping/LidarFeatureAssociate.cpp:550-630), whatever the size.
    ref / nei: scan dicts (world-frame float32 clouds + pose); qidx / nn / point / plane: the accepted records of the pair."""
    tgt, q = ref["less_xyz"], nei["flat_xyz"]
    assert np.all(np.diff(qidx) > 0) and qidx.min() >= 0 and qidx.max() < len(q)              # query order, each query at most once
    assert nn.min() >= 0 and nn.max() < len(tgt)
    assert np.all(np.sort(nn, axis=1)[:, 1:] != np.sort(nn, axis=1)[:, :-1])                   # ten distinct targets
    d = tgt[nn] - q[qidx][:, None, :]                                                          # float32, x then y then z (FLANN L2_Simple)
    sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert np.all(np.diff(sq, axis=1) >= 0)                                                    # sorted ascending
    assert np.all(sq[:, -1] <= np.float32(thr) * np.float32(thr))                              # 10th neighbour within the threshold
    assert np.all(ref["less_tag"][nn] == nei["flat_tag"][qidx][:, None])                       # same class tag
    # the plane: unit normal, all ten neighbours (ref-local, fp64) within the tolerance; the query point in nei-local coordinates
    assert np.abs(np.linalg.norm(plane[:, :3], axis=1) - 1).max() < 1e-12
    loc = (tgt[nn].astype(np.float64) - ref["t_wl"]) @ ref["R_wl"]                             # R^T (p - t)
    off = np.abs(np.einsum("nkc,nc->nk", loc, plane[:, :3]) + plane[:, 3:4])
    assert off.max() <= tol + 1e-9
    qloc = (q[qidx].astype(np.float64) - nei["t_wl"]) @ nei["R_wl"]
    assert np.abs(qloc - point).max() < 1e-12
    # spot check of the k-NN itself against brute force: the ten smallest distances
    for i in rng.choice(len(qidx), min(spot, len(qidx)), replace=False):
        dd = tgt - q[qidx[i]]
        s = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
        assert np.array_equal(np.sort(s)[:10], sq[i])
    return len(qidx)


def test_property_checker_on_the_oracle(oracle):
    """The checker itself, on oracle output at a size the oracle handles (CPU, no GPU)."""
    scans = {k: sy.make_scan(k, cols=256) for k in (0, 1)}
    o = oracle.assoc_point2plane(scans[0], scans[1], 0.05, 1.0)
    n = check_association_properties(scans[0], scans[1], o["qidx"], o["nn"], o["point"], o["plane"], 0.05, 1.0, np.random.default_rng(0))
    assert n > 500
    bad = o["nn"].copy(); bad[7, 9] = bad[7, 0]
    with pytest.raises(AssertionError):
        check_association_properties(scans[0], scans[1], o["qidx"], bad, o["point"], o["plane"], 0.05, 1.0, np.random.default_rng(0))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False], ids=["exact_qr", "certified_fast_fit"])
def test_full_size_association_and_evaluation(oracle, exact):
    import panovlm_amd as pv
    from tests.test_assoc_gpu import same_planes
    mode = 0x200 if exact else 0
    ctx = pv.Context(0)
    rng = np.random.default_rng(4)
    scans = {k: sy.make_scan(k, cols=4096) for k in (0, 1)}
    assert len(scans[0]["flat_xyz"]) == 65536
    dev = {k: pv.Scan(ctx, s) for k, s in scans.items()}
    pairs = [(0, 1), (1, 0)]
    for tol in (0.05, 0.01):                                                                    # Room and Floor tolerances
        rs = ctx.assoc_point2plane([dev[r] for r, _ in pairs], [dev[n] for _, n in pairs], tol, 1.0, kind=pv.POINT2PLANE_ANGLE,
                                   flags=pv.FLAG_NORMALIZE_DISTANCE | KEEP | mode)
        off, ref, nei, rows = rs.download()
        qidx, nn = rs.assoc_debug()
        for p, (r, n) in enumerate(pairs):
            s, e = off[p], off[p + 1]
            cnt = check_association_properties(scans[r], scans[n], qidx[s:e], nn[s:e], rows[s:e, 0:3], rows[s:e, 3:7], tol, 1.0, rng)
            assert cnt > 3000          # ~12 % of 65536: at 0.09 degree azimuth steps most 10-NN sets are one ring segment (collinear, rejected :594-596)
        if tol == 0.05:
            full = (off, qidx, nn, rows)
            # ---- evaluation at this size: fused per-pair blocks == J^T J / J^T r / cost accumulated from the materialised rows
            aa, t = zip(*[sy.pose_params(scans[k]["R_wl"], scans[k]["t_wl"]) for k in (0, 1)])
            ctx.set_poses(np.array(aa), np.array(t))
            r, J = rs.eval()
            blocks = rs.pair_blocks(pv.LOSS_HUBER, np.deg2rad(2.0))
            want = synth.pair_blocks_from_jacobian(r, J, off, pv.LOSS_HUBER, np.deg2rad(2.0))
            assert np.allclose(blocks, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
            assert np.isfinite(r).all() and np.isfinite(J).all() and r.min() >= 0 and np.median(r) < 0.05
        rs.close()
    # ---- the oracle on every 32nd query against the full target cloud: bit-exact records, and the same records the
    # full run produced for those queries
    sub = dict(scans[1]); sub["id"] = 5
    sub["flat_xyz"] = scans[1]["flat_xyz"][::32]; sub["flat_tag"] = scans[1]["flat_tag"][::32]
    dsub = pv.Scan(ctx, sub)
    rs = ctx.assoc_point2plane([dev[0]], [dsub], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE, flags=pv.FLAG_NORMALIZE_DISTANCE | KEEP | mode)
    _, _, _, rows = rs.download()
    q, nn = rs.assoc_debug()
    o = oracle.assoc_point2plane(scans[0], sub, 0.05, 1.0)
    assert len(o["qidx"]) > 100 and np.array_equal(q, o["qidx"]) and np.array_equal(nn, o["nn"])
    assert np.array_equal(rows[:, 0:3], o["point"]) and same_planes(rows[:, 3:7], o["plane"], exact)
    off, fq, fnn, frows = full
    sel = np.nonzero(fq[off[0]:off[1]] % 32 == 0)[0]
    assert np.array_equal(fq[off[0]:off[1]][sel] // 32, q) and np.array_equal(fnn[off[0]:off[1]][sel], nn)
    assert np.array_equal(frows[off[0]:off[1]][sel], rows)
    rs.close(); dsub.close()
    for d in dev.values():
        d.close()
    ctx.close()
