"""Edge cases of the C ABI on the GPU: empty / degenerate inputs must neither crash nor diverge from the
reference's behaviour (empty result, error status), and state errors must be loud."""
import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


def test_empty_residual_set_and_zero_pairs(ctx):
    import panovlm_amd as pv
    rs = pv.ResidualSet.upload(ctx, 1, np.zeros((0, 7)), [0], [], [], flags=1)
    ctx.set_poses(np.zeros((1, 3)), np.zeros((1, 3)))
    r, J = rs.eval()
    assert r.shape == (0,) and J.shape == (0, 12)
    assert rs.pair_blocks(1, 0.03).shape == (0, 121)
    rs2 = pv.ResidualSet.upload(ctx, 0, np.zeros((0, 7)), [0, 0, 0], [0, 1], [1, 0])   # segments without rows
    ctx.set_poses(np.zeros((2, 3)), np.zeros((2, 3)))
    b = rs2.pair_blocks(0, 0.0)
    assert b.shape == (2, 121) and np.all(b == 0)
    neq = pv.NormalEq(ctx, 2, [0], [1])
    assert np.all(neq.accumulate(rs2) == 0)
    res = ctx.assoc_point2plane([], [], 0.05, 1.0)
    assert res.n == 0 and res.n_pairs == 0


def test_eval_requires_poses_and_valid_ids():
    import panovlm_amd as pv
    c = pv.Context(0)
    rows = np.array([[0, 0, 1.0, 0, 0, 1, -0.9]])
    rs = pv.ResidualSet.upload(c, 0, rows, [0, 1], [0], [1])
    with pytest.raises(pv.PvlmError, match="set_poses"):
        rs.eval()
    c.set_poses(np.zeros((2, 3)), np.zeros((2, 3)))
    r, J = rs.eval()
    assert np.isclose(r[0], 0.1) and np.allclose(J[0, 3:6], [0, 0, 1]) and np.allclose(J[0, 9:12], [0, 0, -1])
    with pytest.raises(pv.PvlmError):
        pv.ResidualSet.upload(c, 7, rows, [0, 1], [0], [1])            # unknown functor
    with pytest.raises(pv.PvlmError):
        pv.NormalEq(c, 2, [1], [0])                                      # upair must be i < j
    neq = pv.NormalEq(c, 3, [0], [2])
    with pytest.raises(pv.PvlmError, match="missing"):
        neq.accumulate(rs)                                               # pose pair (0,1) not in the structure
    rs.close(); c.close()


def test_knn_small_and_sparse_clouds(ctx, oracle):
    import panovlm_amd as pv
    rng = np.random.default_rng(2)
    tgt = (rng.normal(size=(40, 3)) * 5).astype(np.float32)
    q = (rng.normal(size=(100, 3)) * 5).astype(np.float32)
    scan = pv.Scan(ctx, dict(id=0, less_xyz=tgt, corner_xyz=tgt))
    for k, md in [(5, 2.0), (10, 6.0), (10, 0.05)]:
        idx, sqd = ctx.knn(scan, q, k, md)
        oi, od = oracle.knn(tgt, q, k)
        valid = od <= np.float32(md) * np.float32(md)
        assert np.array_equal(idx, np.where(valid, oi, -1)) and np.array_equal(sqd, np.where(valid, od, np.float32(np.inf)))
    with pytest.raises(pv.PvlmError):
        ctx.knn(scan, q, 7, 1.0)
    # a cloud spanning kilometres (sparse): falls back to the hashed cell table, same answers
    far = (rng.uniform(-3000, 3000, size=(5000, 3))).astype(np.float32)
    qs = far[:200] + rng.normal(size=(200, 3)).astype(np.float32)
    s2 = pv.Scan(ctx, dict(id=1, less_xyz=far))
    idx, sqd = ctx.knn(s2, qs, 5, 60.0)
    oi, od = oracle.knn(far, qs, 5)
    valid = od <= np.float32(60.0) ** 2
    assert np.array_equal(idx, np.where(valid, oi, -1))


def test_association_with_coincident_and_collinear_targets(ctx, oracle):
    """Degenerate neighbourhoods: all ten neighbours identical (rank-1 system) or exactly collinear — the
    QR / eigen code must take the same accept/reject decisions as the oracle, without NaNs leaking out."""
    import panovlm_amd as pv
    base = sy.make_scan(1, cols=128)
    tg = base["less_xyz"].copy()
    tg[:40] = tg[0]                                   # 40 coincident points
    line = np.linspace(0, 1, 60)[:, None] * np.array([[0.3, 0.0, 0.0]]) + tg[100]
    tg[100:160] = line.astype(np.float32)             # 60 exactly collinear points
    a = dict(base); a["less_xyz"] = tg; a["id"] = 0
    b = dict(sy.make_scan(2, cols=128)); b["id"] = 1
    qs = np.concatenate([b["flat_xyz"], tg[:5] + np.float32(0.001), line[10:20].astype(np.float32) + np.float32(0.002)])
    b["flat_xyz"] = qs; b["flat_tag"] = np.ones(len(qs), np.float32)
    da, db = pv.Scan(ctx, a), pv.Scan(ctx, b)
    from tests.test_assoc_gpu import same_planes
    o = oracle.assoc_point2plane(a, b, 0.05, 1.0)
    for exact in (True, False):           # the reference's QR for every query / the certified fast fit (which must refuse the rank-deficient systems)
        rs = ctx.assoc_point2plane([da], [db], 0.05, 1.0, flags=0x101 | (0x200 if exact else 0))
        off, _, _, rows = rs.download()
        qidx, nn = rs.assoc_debug()
        assert np.array_equal(qidx, o["qidx"]) and np.array_equal(nn, o["nn"])
        assert np.array_equal(rows[:, :3], o["point"]) and same_planes(rows[:, 3:], o["plane"], exact)
        assert np.all(np.isfinite(rows))
        rs.close()


def test_votes_with_empty_inputs(ctx):
    import panovlm_amd as pv
    rng = np.random.default_rng(5)
    lines = synth.random_world_lines(rng, 3)
    s = synth.make_line_scan(rng, 0, np.eye(3), np.zeros(3), lines)
    empty = dict(id=1, R_wl=np.eye(3), t_wl=np.zeros(3))
    ds, de = pv.Scan(ctx, s), pv.Scan(ctx, empty)
    assert ctx.line2line_votes(ds, de, 0.3).shape == (0, 3)
    assert ctx.line2line_votes(de, ds, 0.3).shape == (3, 0)
    local = dict(s); local["corner_xyz"] = s["corner_local"]
    dl = pv.Scan(ctx, local)
    assert ctx.cam_lidar_votes(2880, 5760, np.zeros((0, 4), np.float32), dl, np.eye(4)).shape == (0, 3)
    assert ctx.cam_to_image(720, 1440, np.zeros((0, 3), np.float32)).shape == (0, 2)
