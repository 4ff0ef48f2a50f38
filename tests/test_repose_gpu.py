"""K26 — Velodyne::Transform2LidarWorld / Transform2Local on the RESIDENT clouds (sensors/Velodyne.cpp:1773-1848; called around every solve by
lidar_mapping/LidarOdometry.cpp:17-21, :100-110) through pvlm_scan_transform_batch: the float clouds a scan keeps on the device follow the
reference's World <-> Local round trips bit for bit (pcl::transformPointCloud arithmetic: float(((m0 x + m1 y) + m2 z) + m3)), the voxel grids
rebuilt from device-side bounding boxes equal the grids an upload of the same floats builds (plan, cell membership, stored records), and the
association on re-posed scans equals the association on freshly uploaded ones — over three outer iterations, at small, Room (454) and Floor
(1593) batch sizes.  Bar: bit-exact."""
import os

import numpy as np
import pytest

from panovlm_amd import synthetic as sy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


def _rot(rng, scale):
    from tests.synth import rodrigues
    return rodrigues(rng.normal(size=3) * scale)


def _pose_chain(rng, n_scans, n_iter):
    """poses[it][k] = (R_wl, t_wl): a trajectory + a fresh perturbation per outer iteration (what a solve does to the poses)."""
    base = [sy.estimated_pose(k) for k in range(n_scans)]
    chain = []
    for it in range(n_iter):
        chain.append([(_rot(rng, 0.01 * it) @ R, t + rng.normal(size=3) * 0.02 * it) for R, t in base])
    return chain


def _T(R, t):
    return np.concatenate([R, t[:, None]], axis=1)


def _T_inv(R, t):
    Rt = R.T.copy()
    rt = np.array([(Rt[r, 0] * t[0] + Rt[r, 1] * t[1]) + Rt[r, 2] * t[2] for r in range(3)])
    return np.concatenate([Rt, -rt[:, None]], axis=1)


def _apply(T, xyz):
    return sy.to_world_f32(np.ascontiguousarray(xyz, np.float32), T[:, :3], T[:, 3]) if len(xyz) else np.zeros((0, 3), np.float32)


CLOUDS = ("flat_xyz", "less_xyz", "corner_xyz", "seg_points_xyz")


def _local_scan(rng, k, cols, lines, big_extent=False, empty=False):
    """A scan dict in the LOCAL frame: planar clouds from the synthetic room, corner points + segments sampled from lines."""
    local = sy.raycast_local(k, cols=cols)
    less = local[::3].copy()
    if big_extent:                                  # a far outlier blows the bounding box up: the grid falls back to the hashed table
        less = np.concatenate([less, np.array([[900.0, -700.0, 40.0]], np.float32)])
    d = dict(id=k, flat_xyz=local, flat_tag=np.ones(len(local), np.float32), less_xyz=less, less_tag=np.ones(len(less), np.float32))
    if empty:
        d = dict(id=k)
    if lines:
        ls = sy.make_line_scan(rng, k, np.eye(3), np.zeros(3), sy.random_world_lines(rng, lines))
        seg_pts = []
        for s in range(len(ls["seg_size"])):
            members = [i for i, l in enumerate(ls["p2s"]) if s in l]
            assert len(members) == ls["seg_size"][s]
            seg_pts.append(ls["corner_local"][members])
        d.update(corner_xyz=ls["corner_local"], p2s=ls["p2s"], seg_size=ls["seg_size"], seg_coeffs=ls["seg_coeffs"], end_points=ls["end_points"],
                 seg_points_xyz=np.concatenate(seg_pts) if seg_pts else np.zeros((0, 3), np.float32))
    return d


def _posed(scan, T, pose):
    out = dict(scan)
    for c in CLOUDS:
        if c in scan:
            out[c] = _apply(T, scan[c])
    out["R_wl"], out["t_wl"] = pose
    return out


def _same_grid(a, b):
    assert (a is None) == (b is None)
    if a is None:
        return
    assert a["plan"] == b["plan"], (a["plan"], b["plan"])
    for f in ("cell_key", "index", "points"):
        assert a[f].tobytes() == b[f].tobytes(), f


def _check_against_fresh_upload(ctx, dev, host_scans, grids=True):
    import panovlm_amd as pv
    fresh = pv.Scan.upload_batch(ctx, host_scans)
    for d, f, h in zip(dev, fresh, host_scans):
        for which, name in enumerate(CLOUDS):
            exp = np.ascontiguousarray(h.get(name, np.zeros((0, 3))), np.float32).reshape(-1, 3)
            if which in (1, 2) and grids:
                xyz, g = d.fetch_cloud(which, grid=True)
                _, gf = f.fetch_cloud(which, grid=True)
                _same_grid(g, gf)
            else:
                xyz = d.fetch_cloud(which)
            assert xyz.tobytes() == exp.tobytes(), (h["id"], name)
    return fresh


def _round_trips(ctx, rng, locals_, n_iter=3, assoc_pairs=()):
    import panovlm_amd as pv
    n = len(locals_)
    chain = _pose_chain(rng, n, n_iter)
    host = [_posed(s, _T(*chain[0][k]), chain[0][k]) for k, s in enumerate(locals_)]
    dev = pv.Scan.upload_batch(ctx, host)
    for it in range(n_iter):
        if it > 0:
            # Transform2Local with the poses the clouds were posed with, then the setters, then Transform2LidarWorld (LidarOdometry.cpp:100-110, :17-21)
            T_lw = [_T_inv(*chain[it - 1][k]) for k in range(n)]
            pv.Scan.transform_batch(ctx, dev, T_lw, rebuild_grids=False)
            host = [_posed(h, T_lw[k], chain[it - 1][k]) for k, h in enumerate(host)]
            _check_against_fresh_upload(ctx, dev, host, grids=False)
            assert all(d.cloud_info(1).stale == (d.cloud_info(1).n > 0) for d in dev)
            for k, d in enumerate(dev):
                d.set_pose(*chain[it][k])
            T_wl = [_T(*chain[it][k]) for k in range(n)]
            pv.Scan.transform_batch(ctx, dev, T_wl, rebuild_grids=True)
            host = [_posed(h, T_wl[k], chain[it][k]) for k, h in enumerate(host)]
        fresh = _check_against_fresh_upload(ctx, dev, host)
        if assoc_pairs:
            out = []
            for scans in (dev, fresh):
                rs = ctx.assoc_point2plane([scans[r] for r, _ in assoc_pairs], [scans[q] for _, q in assoc_pairs], 0.05, 1.0, kind=pv.POINT2PLANE_ANGLE,
                                           flags=pv.FLAG_NORMALIZE_DISTANCE | 0x100)
                off, ref, nei, rows = rs.download()
                qidx, nn = rs.assoc_debug()
                out.append((off.tobytes(), ref.tobytes(), nei.tobytes(), rows.tobytes(), qidx.tobytes(), nn.tobytes(), rs.n))
                rs.close()
            assert out[0] == out[1] and out[0][-1] > 1000
            votes = [ctx.line2line_votes_batch([s[r] for r, _ in assoc_pairs], [s[q] for _, q in assoc_pairs], 0.3) for s in (dev, fresh)]
            assert len(votes[0]) == len(assoc_pairs) and all(a.shape == b.shape and a.tobytes() == b.tobytes() for a, b in zip(*votes))
            assert sum(int(v.sum()) for v in votes[0]) > 0
        for f in fresh:
            f.close()
    for d in dev:
        d.close()


def test_three_outer_iterations_small_batch_with_every_cloud_shape(ctx):
    rng = np.random.default_rng(26)
    locals_ = [_local_scan(rng, 0, 512, 6), _local_scan(rng, 1, 512, 9), _local_scan(rng, 2, 256, 0), _local_scan(rng, 3, 512, 4, big_extent=True),
               _local_scan(rng, 4, 128, 3, empty=True), _local_scan(rng, 5, 64, 0, empty=True), _local_scan(rng, 6, 1024, 12)]
    _round_trips(ctx, rng, locals_, n_iter=3, assoc_pairs=[(0, 1), (1, 0), (1, 6), (6, 0), (3, 1)])


@pytest.mark.parametrize("n_scans", [454, 1593])
def test_three_outer_iterations_at_room_and_floor_batch_size(ctx, n_scans):
    rng = np.random.default_rng(n_scans)
    base = [_local_scan(rng, k, 128, 5) for k in range(16)]
    locals_ = []
    for k in range(n_scans):
        s = dict(base[k % 16]); s["id"] = k
        locals_.append(s)
    _round_trips(ctx, rng, locals_, n_iter=3)


def test_argument_and_state_errors(ctx):
    import panovlm_amd as pv
    rng = np.random.default_rng(5)
    s = [_posed(_local_scan(rng, k, 256, 3), _T(np.eye(3), np.zeros(3)), (np.eye(3), np.zeros(3))) for k in range(2)]
    dev = pv.Scan.upload_batch(ctx, s)
    I = _T(np.eye(3), np.zeros(3))
    with pytest.raises(pv.PvlmError):
        pv.Scan.transform_batch(ctx, [dev[0], dev[0]], [I, I])                       # a scan listed twice
    bad = I.copy(); bad[1, 3] = np.nan
    with pytest.raises(pv.PvlmError):
        pv.Scan.transform_batch(ctx, dev, [I, bad])
    pv.Scan.transform_batch(ctx, dev, [I, I], rebuild_grids=False)                   # identity: the floats are unchanged, the grids are declared stale
    assert dev[0].fetch_cloud(1).tobytes() == np.ascontiguousarray(s[0]["less_xyz"], np.float32).tobytes()
    with pytest.raises(pv.PvlmError, match="rebuild"):
        ctx.assoc_point2plane([dev[0]], [dev[1]], 0.05, 1.0)
    with pytest.raises(pv.PvlmError, match="rebuild"):
        ctx.knn(dev[0], s[1]["flat_xyz"][:10], 10, 1.0)
    pv.Scan.transform_batch(ctx, dev, [I, I], rebuild_grids=True)
    rs = ctx.assoc_point2plane([dev[0]], [dev[1]], 0.05, 1.0)
    assert rs.n > 0
    rs.close()
    huge = I.copy(); huge[0, 3] = 1e39                                              # leaves the float range: reported, not silently gridded
    with pytest.raises(pv.PvlmError, match="non-finite"):
        pv.Scan.transform_batch(ctx, dev, [huge, I])
    for d in dev:
        d.close()


def test_upload_from_point_records_equals_packed_upload(ctx):
    """pvlm_scan_desc::point_stride_floats = 4 — the clouds handed over as pcl::PointXYZI records where they lie (what the host mirror does since round 6: no flattening
    pass) — builds the same resident scan as the packed arrays: floats, tags (through the association: class test), grids."""
    import panovlm_amd as pv
    rng = np.random.default_rng(9)
    scans = []
    for k in range(5):
        s = _posed(_local_scan(rng, k, 256, 4, big_extent=(k == 3)), _T(*sy.estimated_pose(k)), sy.estimated_pose(k))
        s["less_tag"] = np.where(rng.uniform(size=len(s["less_xyz"])) < 0.1, 16.0, 1.0).astype(np.float32)
        s["flat_tag"] = np.where(rng.uniform(size=len(s["flat_xyz"])) < 0.3, 16.0, 1.0).astype(np.float32)
        scans.append(s)
    packed = pv.Scan.upload_batch(ctx, scans)
    records = pv.Scan.upload_batch(ctx, [dict(s, point_records=True) for s in scans])
    for a, b in zip(packed, records):
        for which in range(4):
            if which in (1, 2):
                xa, ga = a.fetch_cloud(which, grid=True); xb, gb = b.fetch_cloud(which, grid=True)
                _same_grid(ga, gb)
            else:
                xa, xb = a.fetch_cloud(which), b.fetch_cloud(which)
            assert xa.tobytes() == xb.tobytes()
    pairs = [(0, 1), (1, 2), (3, 4), (4, 0)]
    out = []
    for dev in (packed, records):
        rs = ctx.assoc_point2plane([dev[r] for r, _ in pairs], [dev[q] for _, q in pairs], 0.05, 1.0, flags=pv.FLAG_NORMALIZE_DISTANCE | 0x100)
        off, _, _, rows = rs.download(); qidx, nn = rs.assoc_debug()
        out.append((off.tobytes(), rows.tobytes(), qidx.tobytes(), nn.tobytes())); assert rs.n > 500
        rs.close()
    assert out[0] == out[1]
    for d in packed + records:
        d.close()


def test_upload_from_point_records_with_a_staging_window_that_cuts_records(tmp_path):
    """The same equality when the pinned staging window is far smaller than the upload and not a multiple of a record (PVLM_UPLOAD_STAGE_MB: window boundaries fall
    inside 12-byte coordinate records — the gather copies the cut records byte by byte and the whole ones in a fixed-size loop).  In a process of its own: the window
    size is read per upload, the context's pinned buffer is grow-only."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, panovlm_amd as pv
        from panovlm_amd import synthetic as sy
        from tests.test_repose_gpu import _posed, _local_scan, _T
        rng = np.random.default_rng(9)
        scans = []
        for k in range(6):
            s = _posed(_local_scan(rng, k, 256, 4), _T(*sy.estimated_pose(k)), sy.estimated_pose(k))
            s["less_tag"] = np.where(rng.uniform(size=len(s["less_xyz"])) < 0.1, 16.0, 1.0).astype(np.float32)
            s["flat_tag"] = np.where(rng.uniform(size=len(s["flat_xyz"])) < 0.3, 16.0, 1.0).astype(np.float32)
            scans.append(s)
        ctx = pv.Context(0)
        packed = pv.Scan.upload_batch(ctx, scans)
        records = pv.Scan.upload_batch(ctx, [dict(s, point_records=True) for s in scans])
        n = 0
        for a, b in zip(packed, records):
            for which in range(4):
                xa, xb = a.fetch_cloud(which), b.fetch_cloud(which)
                assert xa.tobytes() == xb.tobytes(), which
                n += xa.size
        print("same", n)
    """)
    env = dict(os.environ, PVLM_UPLOAD_STAGE_MB="0.004", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # 4 KiB windows
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("same") and int(out.stdout.split()[1]) > 10000
