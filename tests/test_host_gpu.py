"""The C++ host mirror (panovlm_amd/host, PanoVLM's own interface names) on the GPU against the CPU
oracle: single cost functions, AssociatePoint2Plane / AssociateLine2Line, line tracks, FindNeighbors and
the end-to-end LidarOdometry::EstimatePose loop against its CPU twin (tests/lm_twin.py)."""
import os
import re
import tempfile

import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import host_io, lm_twin, synth

pytestmark = pytest.mark.gpu


def _vlp(k, cols):
    s = sy.make_scan(k, cols=cols, downsample_targets=0.2)
    R, t = s["R_wl"], s["t_wl"]
    local = s["local_xyz"]
    Rl, tl = lm_twin.inv_pose(R, t)
    less_local = lm_twin.transform_f32(s["less_xyz"], Rl, tl)   # centroids back to the local frame
    return dict(id=k, R_wl=R, t_wl=t, flat_local=local, flat_tag=np.ones(len(local), np.float32), less_local=less_local,
                less_tag=np.ones(len(less_local), np.float32))


@pytest.fixture(scope="module")
def tmp():
    with tempfile.TemporaryDirectory() as d:
        yield d


def test_single_cost_function_api(oracle):
    rng = np.random.default_rng(3)
    for kind, norm in [(0, 0), (1, 1), (2, 0), (3, 1), (4, 0), (5, 0)]:
        aa, t = synth.random_poses(rng, 2)
        rows, off = synth.random_resset(rng, kind, aa, t, np.array([0]), np.array([1]), [1], offsets_scale=(0.05,))
        w = 1.25
        params = np.concatenate([aa[0], t[0], aa[1], t[1]])
        out = host_io.run("costfn", kind, norm, w, *["%.17g" % v for v in rows[0]], *["%.17g" % v for v in params])
        vals = out[0].split()
        ok, r = int(vals[1]), float(vals[3])
        J = np.array([float(v) for v in vals[5:14]])
        ro, Jo = oracle.evaluate(kind, synth.oracle_rows(kind, rows, w), [0], [1], aa, t, normalize=bool(norm))
        assert ok == 1 and abs(r - ro[0]) <= 1e-6 * abs(ro[0]) + 1e-12
        Jexp = np.concatenate([Jo[0, 0:3], Jo[0, 3:6], Jo[0, 9:12]])   # block 2 was passed as null
        assert np.allclose(J, Jexp, rtol=1e-6, atol=1e-9)
        assert abs(float(out[1].split()[1]) - r) == 0.0


def test_find_neighbors_and_point2plane(oracle, tmp):
    scans = [_vlp(k, 256) for k in range(5)]
    path = os.path.join(tmp, "scans.bin")
    host_io.write_scans(path, scans)
    nb = [[int(v) for v in l.split()[2:]] for l in host_io.run("neighbors", path, 6)]
    poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in scans])
    assert nb == oracle.find_neighbors(poses, np.ones(5, np.int32), 6)
    world = [dict(id=s["id"], R_wl=s["R_wl"], t_wl=s["t_wl"], flat_xyz=lm_twin.transform_f32(s["flat_local"], s["R_wl"], s["t_wl"]), flat_tag=s["flat_tag"],
                  less_xyz=lm_twin.transform_f32(s["less_local"], s["R_wl"], s["t_wl"]), less_tag=s["less_tag"]) for s in scans]
    for r, n in [(0, 1), (3, 2)]:
        got = np.array([[float(v) for v in l.split()[1:]] for l in host_io.run("p2plane", path, r, n, 0.05, 1.0)])
        o = oracle.assoc_point2plane(world[r], world[n], 0.05, 1.0)
        assert got.shape[0] == len(o["qidx"]) > 100
        assert np.array_equal(got[:, :3], o["point"]) and np.array_equal(got[:, 3:], o["plane"])


def test_find_neighbors_with_the_searches_on_the_gpu(oracle, tmp, monkeypatch):
    """FindNeighbors with its two searches (nearestKSearch + radiusSearch over the scan centres) on the GPU (K28, pvlm_centre_orders: what the mirror does from
    PVLM_NEIGHBORS_GPU_MIN scans on) against the host's own distance loop and sort and against the oracle: the same lists, for 6 and for 4 neighbours, with scans far
    apart (radius search cut off), scans on top of each other (equal distances: the position decides) and loop closures."""
    rng = np.random.default_rng(5)
    scans = []
    for k in range(260):
        s = _vlp(k % 7, 64)
        t = np.array([0.1 * k, 3.0 * np.sin(0.04 * k), 0.02 * k]) if k < 225 else np.array([0.1 * (k - 224) + 0.01, 3.0 * np.sin(0.04 * (k - 224)), 0.0])   # the last 35 revisit the start, more than loop_length = 200 scans later
        if k in (11, 12): t = np.array([4.0, 1.0, 0.5])                   # coincident centres
        if k == 40: t = np.array([500.0, 0.0, 0.0])                       # out of every radius
        s = dict(s, id=k, t_wl=t.astype(np.float64))
        scans.append(s)
    path = os.path.join(tmp, "scans_nb.bin")
    host_io.write_scans(path, scans)
    poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in scans])
    for k in (6, 4):
        want = oracle.find_neighbors(poses, np.ones(len(scans), np.int32), k)
        monkeypatch.setenv("PVLM_NEIGHBORS_GPU_MIN", "1")
        on_gpu = [[int(v) for v in l.split()[2:]] for l in host_io.run("neighbors", path, k)]
        monkeypatch.setenv("PVLM_NEIGHBORS_GPU_MIN", "100000")
        on_host = [[int(v) for v in l.split()[2:]] for l in host_io.run("neighbors", path, k)]
        assert on_gpu == on_host == want
        assert any(abs(v - i) > 200 for i, l in enumerate(want) for v in l), "no loop closure in the test trajectory"


def _line_scans(rng, n, pose_offset=2):
    lines = synth.random_world_lines(rng, 10)
    out = []
    for k in range(n):
        R, t = sy.estimated_pose(k + pose_offset)
        s = synth.make_line_scan(rng, k, R, t, lines, pts_per_line=(10, 24), extra_pts=12, noise=0.005)
        seg_points = [[i for i, l in enumerate(s["p2s"]) if sid in l] for sid in range(len(s["seg_size"]))]
        out.append(dict(id=k, R_wl=R, t_wl=t, corner_local=s["corner_local"], p2s=s["p2s"], seg_points=seg_points,
                        seg_coeffs=s["seg_coeffs"], end_points=s["end_points"], _oracle=s))
    return out


def test_line2line_and_tracks(oracle, tmp):
    rng = np.random.default_rng(17)
    scans = _line_scans(rng, 5)
    path = os.path.join(tmp, "lines.bin")
    host_io.write_scans(path, scans)
    for r, n, thr in [(0, 1, 0.3), (2, 1, 0.3), (1, 3, 0.4)]:
        got = [l.split()[1:] for l in host_io.run("line2line", path, r, n, thr)]
        o = oracle.assoc_line2line(scans[r]["_oracle"], scans[n]["_oracle"], thr)
        assert [int(g[0]) for g in got] == o["nei_idx"].tolist() and [int(g[1]) for g in got] == o["ref_idx"].tolist()
        assert np.allclose(np.array([[float(v) for v in g[2:5]] for g in got]), o["p1"], atol=1e-13)
        assert len(got) >= 5
    # tracks: union-find over the pairwise associations (util/Tracks.cpp) -> python reimplementation here
    tracks = [set(tuple(int(x) for x in p.split(":")) for p in l.split()[2:]) for l in host_io.run("tracks", path, 4, 3)]
    poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in scans])
    nb = oracle.find_neighbors(poses, np.ones(5, np.int32), 4)
    parent = {}

    def find(a):
        while parent.setdefault(a, a) != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    for i in range(5):
        for j in nb[i]:
            o = oracle.assoc_line2line(scans[j]["_oracle"], scans[i]["_oracle"], 0.3)
            for a, b in zip(o["nei_idx"], o["ref_idx"]):
                ra, rb = find((i, int(a))), find((j, int(b)))
                if ra != rb:
                    parent[ra] = rb
    comps = {}
    for node in list(parent):
        comps.setdefault(find(node), set()).add(node)
    expect = [c for c in comps.values() if len({n[0] for n in c}) >= 3 and len(c) > 1]
    assert sorted(map(sorted, tracks)) == sorted(map(sorted, expect)) and len(tracks) >= 5


def test_camera_lidar_by_angle(oracle, tmp):
    rng = np.random.default_rng(8)
    rows, cols = 2880, 5760
    lw = synth.random_world_lines(rng, 9, extent=3.0)
    s = synth.make_line_scan(rng, 0, np.eye(3), np.zeros(3), lw, pts_per_line=(20, 50), extra_pts=40)
    seg_points = [[i for i, l in enumerate(s["p2s"]) if sid in l] for sid in range(len(s["seg_size"]))]
    scan = dict(id=0, R_wl=np.eye(3), t_wl=np.zeros(3), corner_local=s["corner_local"], p2s=s["p2s"], seg_points=seg_points,
                seg_coeffs=s["seg_coeffs"], end_points=s["end_points"])
    path = os.path.join(tmp, "cam.bin")
    host_io.write_scans(path, [scan], world=False)
    ang = np.deg2rad(rng.uniform(-2, 2, size=3))
    T = np.eye(4); T[:3, :3] = synth.rodrigues(ang); T[:3, 3] = rng.uniform(-0.05, 0.05, size=3)
    ends_cam = s["end_points"].reshape(-1, 3) @ T[:3, :3].T + T[:3, 3]
    px = oracle.cam_to_image(rows, cols, ends_cam).reshape(-1, 4).astype(np.float32)
    px += rng.normal(size=px.shape).astype(np.float32) * 2.0
    lines = np.concatenate([px, px[:3] + np.float32(6.0)])   # near-duplicates exercise UniqueLinePair
    lpath = os.path.join(tmp, "lines_T.bin")
    with open(lpath, "wb") as f:
        f.write(np.int32(len(lines)).tobytes()); f.write(lines.astype(np.float32).tobytes()); f.write(T.astype(np.float64).tobytes())
    local = dict(s); local["corner_xyz"] = s["corner_local"]
    for mult in (1, 0):
        got = [l.split()[1:] for l in host_io.run("byangle", path, lpath, rows, cols, mult)]
        o = oracle.assoc_by_angle(rows, cols, lines, local, T, multiple=bool(mult))
        assert [int(g[0]) for g in got] == o["image_line_id"].tolist() and [int(g[1]) for g in got] == o["lidar_line_id"].tolist()
        assert np.allclose([float(g[2]) for g in got], o["score"], rtol=1e-6)
        assert np.allclose(np.array([[float(v) for v in g[3:6]] for g in got]), o["start"], atol=1e-12)
    assert len(got) >= 5


def test_estimate_pose_matches_cpu_twin(oracle, tmp):
    """LidarOdometry::EstimatePose (GPU association + GPU normal equations + host LM) vs the same loop on the
    oracle: residual-block counts equal, costs and final poses within 1e-6 relative (north_star)."""
    scans = [_vlp(k, 256) for k in range(4)]
    path = os.path.join(tmp, "odo.bin")
    host_io.write_scans(path, scans, world=False)
    out = host_io.run("odometry", path, 3, 1, 1, 0, 1, 0.05, 1.0, 0.3)
    iters = [l.split() for l in out if l.startswith("iter")]
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    cfg = dict(angle=True, normalize=True, tol=0.05, thr=1.0)
    twin = [dict(s) for s in scans]
    log = lm_twin.estimate_pose(oracle, twin, cfg, 3)
    assert len(iters) == len(log)
    for it, lg in zip(iters, log):
        assert int(it[6]) == lg["blocks"]
        assert abs(float(it[2]) - lg["final_cost"]) <= 1e-6 * lg["final_cost"]
        assert int(it[4]) == lg["successful"]
    for k, s in enumerate(twin):
        R = poses[k][:9].reshape(3, 3); t = poses[k][9:]
        assert np.abs(R - s["R_wl"]).max() <= 1e-6 and np.abs(t - s["t_wl"]).max() <= 1e-6 * max(1.0, np.abs(s["t_wl"]).max())
    # the refinement actually moved the perturbed poses towards the truth
    err0 = np.mean([np.linalg.norm(scans[k]["t_wl"] - sy.true_pose(k)[1]) for k in range(1, 4)])
    err1 = np.mean([np.linalg.norm(poses[k][9:] - sy.true_pose(k)[1] - (poses[0][9:] - sy.true_pose(0)[1])) for k in range(1, 4)])
    assert err1 < err0


def test_camera_lidar_residual_blocks(oracle, tmp):
    """AddCameraLidarResidual (util/Optimization.cpp:564-607): Plane2Plane_Global + PlaneIOUResidual blocks built
    from the by-angle association; initial robust cost against the oracle, then a short solve."""
    rng = np.random.default_rng(12)
    rows, cols = 2880, 5760
    lw = synth.random_world_lines(rng, 9, extent=3.0)
    s = synth.make_line_scan(rng, 0, np.eye(3), np.zeros(3), lw, pts_per_line=(20, 50), extra_pts=40)
    seg_points = [[i for i, l in enumerate(s["p2s"]) if sid in l] for sid in range(len(s["seg_size"]))]
    scan = dict(id=0, R_wl=np.eye(3), t_wl=np.zeros(3), corner_local=s["corner_local"], p2s=s["p2s"], seg_points=seg_points,
                seg_coeffs=s["seg_coeffs"], end_points=s["end_points"])
    path = os.path.join(tmp, "cam2.bin")
    host_io.write_scans(path, [scan], world=False)
    ang = np.deg2rad(rng.uniform(-2, 2, size=3))
    T = np.eye(4); T[:3, :3] = synth.rodrigues(ang); T[:3, 3] = rng.uniform(-0.05, 0.05, size=3)
    ends_cam = s["end_points"].reshape(-1, 3) @ T[:3, :3].T + T[:3, 3]
    lines = oracle.cam_to_image(rows, cols, ends_cam).reshape(-1, 4).astype(np.float32)
    lines += rng.normal(size=lines.shape).astype(np.float32) * 3.0
    lpath = os.path.join(tmp, "lines_T2.bin")
    with open(lpath, "wb") as f:
        f.write(np.int32(len(lines)).tobytes()); f.write(lines.tobytes()); f.write(T.astype(np.float64).tobytes())
    w, a = 1.5, 3 * np.pi / 180
    out = host_io.run("camlidar", path, lpath, rows, cols, w, a, 8)
    v = out[0].split()
    blocks, initial, final = int(v[1]), float(v[3]), float(v[5])
    # oracle side
    local = dict(s); local["corner_xyz"] = s["corner_local"]
    o = oracle.assoc_by_angle(rows, cols, lines, local, T, multiple=True)
    assert blocks == 2 * len(o["image_line_id"]) > 8
    Rlc = T[:3, :3].T; tlc = -Rlc @ T[:3, 3]
    aa = np.stack([np.zeros(3), oracle.matrix_to_angle_axis(Rlc)]); t = np.stack([np.zeros(3), tlc])
    r4, r5 = [], []
    for k, li in enumerate(o["image_line_id"]):
        px = lines[li].astype(np.float64)
        p1 = oracle.image_to_cam(rows, cols, px[None, 0:2], 1.0)[0]; p2 = oracle.image_to_cam(rows, cols, px[None, 2:4], 1.0)[0]
        pl = np.cross(p2 - p1, -p1); plane = np.concatenate([pl, [-(pl @ p1)]])
        st, en = o["start"][k], o["end"][k]
        r4.append(np.concatenate([plane[:3], en, st, [1.0 * w]]))
        c = float(np.clip(p1 @ p2, -1, 1))
        r5.append(np.concatenate([plane, (en + st) / 2, (p1 + p2) / 2, [np.arccos(c)], [2.0 * w]]))
    n = len(r4)
    ra, _ = oracle.evaluate(4, np.array(r4), [0] * n, [1] * n, aa, t, jac=False)
    rb, _ = oracle.evaluate(5, np.array(r5), [0] * n, [1] * n, aa, t, jac=False)
    cost = synth.huber_weights(np.concatenate([ra, rb]), 1, a)[1].sum()
    assert abs(initial - cost) <= 1e-6 * cost
    assert final <= initial * (1 + 1e-12) and np.isfinite(final)


def _image_to_cam_f32(rows, cols, px, r):
    """Equirectangular::ImageToCam<float> (sensors/Equirectangular.h:102-103, :125-128) statement by statement in float32."""
    f = np.float32
    x, y = f(px[0]), f(px[1])
    sx = f(np.float64(f(f(f(2) * x) / f(cols)) - f(1)) * np.pi)
    sy = f((0.5 - np.float64(f(y / f(rows)))) * np.pi)
    cy = f(np.cos(np.float64(sy)))
    return np.array([f(f(r) * cy) * f(np.sin(np.float64(sx))), f(-f(r)) * f(np.sin(np.float64(sy))), f(f(r) * cy) * f(np.cos(np.float64(sx)))], np.float32)


def test_calibration_mode_optimize_matches_cpu_twin(oracle, tmp):
    """CameraLidarOptimizer::Optimize(line_pairs, T_cl), calibration mode (joint_optimization/CameraLidarOptimizer.cpp:32-87): one
    unknown transform, Plane2Plane_Relative (degrees, Huber) + PlaneRelativeIOUResidual (weight 2, half arc, no loss) per associated
    line pair.  The host mirror evaluates them on the GPU as kinds 4 / 5 with an identity second pose; the twin runs the restated
    trust-region policy on the oracle's Jet evaluation of the same blocks.  Same block count, step count, cost and transform; and the
    transform moves towards the one the pairs were generated with."""
    rng = np.random.default_rng(44)
    rows, cols = 2880, 5760
    n = 60
    R_true = synth.rodrigues(np.array([0.02, -0.03, 0.015])); t_true = np.array([0.05, -0.02, 0.08])
    mid = rng.normal(size=(n, 3)); mid *= (rng.uniform(2.0, 6.0, size=(n, 1)) / np.linalg.norm(mid, axis=1, keepdims=True))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    half = rng.uniform(0.3, 1.2, size=(n, 1))
    start, end = mid - half * d, mid + half * d
    ends_cam = np.stack([start @ R_true.T + t_true, end @ R_true.T + t_true], axis=1).reshape(-1, 3)
    lines = oracle.cam_to_image(rows, cols, ends_cam).reshape(-1, 4).astype(np.float32)
    lines += rng.normal(size=lines.shape).astype(np.float32) * 1.5
    keep = np.abs(lines[:, 0] - lines[:, 2]) < 0.5 * cols                       # no wrap-around lines
    lines, start, end = lines[keep], start[keep], end[keep]
    n = len(lines)
    T0 = np.eye(4); T0[:3, :3] = synth.rodrigues(np.array([0.05, -0.01, -0.02])); T0[:3, 3] = t_true + np.array([0.06, 0.04, -0.05])
    path = os.path.join(tmp, "calib.bin")
    with open(path, "wb") as f:
        f.write(np.int32(n).tobytes())
        for k in range(n):
            f.write(lines[k].tobytes()); f.write(start[k].astype(np.float64).tobytes()); f.write(end[k].astype(np.float64).tobytes())
        f.write(T0.astype(np.float64).tobytes())
    out = host_io.run("calib", path, rows, cols)
    v = out[0].split()
    blocks, final, steps = int(v[1]), float(v[3]), int(v[5])
    T = np.array([float(x) for x in out[1].split()[1:]]).reshape(4, 4)
    assert blocks == 2 * n
    # the twin: rows as the reference's statements build them (float points, float plane, float half arc), kinds 4 / 5, poses {T_cl, identity}
    r4, r5 = [], []
    for k in range(n):
        p1 = _image_to_cam_f32(rows, cols, lines[k, 0:2], 5.0); p2 = _image_to_cam_f32(rows, cols, lines[k, 2:4], 5.0)
        f32 = np.float32; z = f32(0)
        a = f32(f32(p2[1] - p1[1]) * f32(z - p1[2])) - f32(f32(p2[2] - p1[2]) * f32(z - p1[1]))
        b = f32(f32(p2[2] - p1[2]) * f32(z - p1[0])) - f32(f32(p2[0] - p1[0]) * f32(z - p1[2]))
        c = f32(f32(p2[0] - p1[0]) * f32(z - p1[1])) - f32(f32(p2[1] - p1[1]) * f32(z - p1[0]))
        plane = np.array([a, b, c], np.float64)
        r4.append(np.concatenate([plane, end[k], start[k], [180.0 / np.pi]]))
        cosang = f32(f32(f32(p1[0] * p2[0]) + f32(p1[1] * p2[1])) + f32(p1[2] * p2[2]))
        n1 = np.sqrt(f32(f32(f32(p1[0] * p1[0]) + f32(p1[1] * p1[1])) + f32(p1[2] * p1[2])), dtype=np.float32)
        n2 = np.sqrt(f32(f32(f32(p2[0] * p2[0]) + f32(p2[1] * p2[1])) + f32(p2[2] * p2[2])), dtype=np.float32)
        cosang = f32(cosang / f32(n1 * n2))
        full = f32(0) if cosang >= 1 else (f32(np.pi) if cosang <= -1 else np.arccos(cosang, dtype=np.float32))
        mid_i = np.array([f32(f32(p1[q] + p2[q]) / f32(2)) for q in range(3)], np.float64)
        r5.append(np.concatenate([plane, [0.0], (start[k] + end[k]) / 2.0, mid_i, [np.float64(f32(full / f32(2)))], [2.0]]))
    aa = np.stack([oracle.matrix_to_angle_axis(T0[:3, :3]), np.zeros(3)]); t = np.stack([T0[:3, 3].copy(), np.zeros(3)])
    ids0 = np.zeros(n, np.int32); ids1 = np.ones(n, np.int32)
    groups = [dict(kind=4, normalize=False, rows=np.array(r4), rid=ids0, nid=ids1, loss=1, a=2.0 * np.pi / 180.0),
              dict(kind=5, normalize=False, rows=np.array(r5), rid=ids0, nid=ids1, loss=0, a=0.0)]
    opt = lm_twin.Options(); opt.max_num_iterations = 50
    res = lm_twin.solve(oracle, groups, aa, t, {1}, opt)
    Rt = synth.rodrigues(aa[0])
    assert steps == res["successful"] and abs(final - res["final_cost"]) <= 1e-6 * res["final_cost"]
    assert np.abs(T[:3, :3] - Rt).max() <= 1e-6 and np.abs(T[:3, 3] - t[0]).max() <= 1e-6
    assert np.array_equal(T[3], [0, 0, 0, 1])
    # towards the truth: rotation and translation errors shrink
    ang = lambda R: np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    assert ang(T[:3, :3] @ R_true.T) < 0.3 * ang(T0[:3, :3] @ R_true.T)
    assert np.linalg.norm(T[:3, 3] - t_true) < 0.5 * np.linalg.norm(T0[:3, 3] - t_true)
    assert res["final_cost"] < 0.2 * res["initial_cost"]


def test_line_to_line_refine_matches_cpu_twin(oracle, tmp):
    """One RefinePose with only the line-to-line term (GenerateTracks + AddLidarLineToLineResidual2 + Solve) against
    the oracle twin: same residual-block count, costs and poses within 1e-6."""
    rng = np.random.default_rng(31)
    scans = _line_scans(rng, 5)
    path = os.path.join(tmp, "lodo.bin")
    host_io.write_scans(path, scans, world=False)
    out = host_io.run("odometry", path, 1, 1, 1, 1, 0, 0.05, 1.0, 0.3)
    it = [l.split() for l in out if l.startswith("iter")][0]
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    twin = [dict(s) for s in scans]
    for s in twin:
        s["corner_cur"] = np.asarray(s["corner_local"], np.float32)
    res = lm_twin.refine_pose_lines(oracle, twin, thr=0.3, normalize=True)
    assert int(it[6]) == res["blocks"] > 200
    assert abs(float(it[2]) - res["final_cost"]) <= 1e-6 * max(res["final_cost"], 1e-12)
    assert int(it[4]) == res["successful"]
    for k, s in enumerate(twin):
        R = poses[k][:9].reshape(3, 3); t = poses[k][9:]
        assert np.abs(R - s["R_wl"]).max() <= 1e-6 and np.abs(t - s["t_wl"]).max() <= 1e-6 * max(1.0, np.abs(s["t_wl"]).max())


def _joint_scene(rng, n):
    """n LiDAR scans (VLP plane clouds + line segments) and n panoramas looking at the same world lines."""
    rows, cols = 2880, 5760
    lw = synth.random_world_lines(rng, 10, extent=3.0)
    a = np.deg2rad(np.array([1.0, -2.0, 0.5])); T_cl = np.eye(4); T_cl[:3, :3] = synth.rodrigues(a); T_cl[:3, 3] = [0.03, -0.02, 0.05]
    lidars, frames = [], []
    for k in range(n):
        base = _vlp(k, 256)
        ls = synth.make_line_scan(rng, k, base["R_wl"], base["t_wl"], lw, pts_per_line=(20, 40), extra_pts=20, noise=0.004, shared_frac=0.0)
        seg_points = [[i for i, l in enumerate(ls["p2s"]) if sid in l] for sid in range(len(ls["seg_size"]))]
        base.update(corner_local=ls["corner_local"], p2s=ls["p2s"], seg_points=seg_points, seg_coeffs=ls["seg_coeffs"], end_points=ls["end_points"])
        lidars.append(base)
        R_true, t_true = sy.true_pose(k)
        T_wc = lm_twin.pose4(R_true, t_true) @ np.linalg.inv(T_cl)          # the camera sees the TRUE geometry
        T_cw = np.linalg.inv(T_wc)
        ends = np.array([np.concatenate([T_cw[:3, :3] @ p + T_cw[:3, 3], T_cw[:3, :3] @ q + T_cw[:3, 3]]) for p, q in lw]).reshape(-1, 3)
        px = oracle_cam_to_image(rows, cols, ends).reshape(-1, 4).astype(np.float32) + rng.normal(size=(len(lw), 4)).astype(np.float32) * 1.5
        # estimated camera pose = estimated LiDAR pose composed with the calibration (SetFramePose upstream)
        T_wc_est = lm_twin.pose4(base["R_wl"], base["t_wl"]) @ np.linalg.inv(T_cl)
        frames.append(dict(id=k, rows=rows, cols=cols, valid=1, R_wc=T_wc_est[:3, :3].copy(), t_wc=T_wc_est[:3, 3].copy(), lines=px))
    return lidars, frames, T_cl


def oracle_cam_to_image(rows, cols, pts):
    from oracle import oracle as orc
    return orc.cam_to_image(rows, cols, np.asarray(pts, np.float64))


def test_joint_optimize_matches_cpu_twin(oracle, tmp):
    """CameraLidarOptimizer::JointOptimize (mapping mode without the SfM term): AssociateLineMulti on the GPU voting
    kernel, camera-LiDAR + LiDAR-LiDAR point-to-plane blocks, host LM — against the oracle twin."""
    rng = np.random.default_rng(77)
    lidars, frames, T_cl = _joint_scene(rng, 3)
    lpath, fpath = os.path.join(tmp, "jl.bin"), os.path.join(tmp, "jf.bin")
    host_io.write_scans(lpath, lidars, world=False)
    host_io.write_frames(fpath, T_cl, frames)
    out = host_io.run("joint", lpath, fpath, 3, 2, 0, 1, 0.05, 1.0, 0.3, 1.0, 2.0)
    iters = [l.split() for l in out if l.startswith("iter")]
    lposes = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    fposes = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("frame")}
    tl = [dict(s) for s in lidars]; tf = [dict(f) for f in frames]
    cfg = dict(p2plane=True, tol=0.05, thr=1.0, lidar_weight=1.0, camera_lidar_weight=2.0)
    log = lm_twin.joint_optimize(oracle, tl, tf, T_cl, cfg, 3, 2)
    assert len(iters) == len(log) >= 1
    for it, lg in zip(iters, log):
        assert int(it[8]) == lg["pairs"] > 10
        assert int(it[6]) == lg["blocks"]
        assert abs(float(it[2]) - lg["final_cost"]) <= 1e-6 * lg["final_cost"]
        assert int(it[4]) == lg["successful"]
    for k in range(3):
        assert np.abs(lposes[k][:9].reshape(3, 3) - tl[k]["R_wl"]).max() <= 1e-6 and np.abs(lposes[k][9:] - tl[k]["t_wl"]).max() <= 1e-6
        assert np.abs(fposes[k][:9].reshape(3, 3) - tf[k]["R_wc"]).max() <= 1e-6 and np.abs(fposes[k][9:] - tf[k]["t_wc"]).max() <= 1e-6
    assert np.allclose(fposes[0][9:], frames[0]["t_wc"])    # camera 0 is the gauge


# ------------------------------------------------------------------------------------------------
# SfM reprojection term (AddCameraResidual) with GPU point elimination
# ------------------------------------------------------------------------------------------------
def _add_tracks(rng, frames, true_poses, n_points, extent, px_noise=0.3, pt_noise=0.03, min_views=4, center=(0.0, 0.0, 0.5)):
    """Keypoints = projections of random world points into the TRUE camera poses (+ pixel noise); tracks carry a
    perturbed triangulation.  true_poses: list of (R_wc, t_wc).  The one-angle residual gives ONE equation per
    observation, so a point needs >= 4 views to be well determined (fewer leave it to the LM damping)."""
    Xt = rng.normal(size=(n_points, 3)) * np.asarray(extent)
    lateral = np.maximum(np.hypot(Xt[:, 0], Xt[:, 1]), 1e-9)          # keep points off the (z) axis of motion: parallax
    Xt[:, :2] *= np.maximum(1.0, 0.8 / lateral)[:, None]
    Xt += np.asarray(center)
    for fr in frames:
        fr["keypoints"] = []
    tracks = []
    F = len(frames)
    for p in range(n_points):
        obs = []
        for fi in sorted(rng.choice(F, size=int(rng.integers(min(min_views, F), F + 1)), replace=False)):
            R, t = true_poses[fi]
            pc = R.T @ (Xt[p] - t)
            px = oracle_cam_to_image(frames[fi]["rows"], frames[fi]["cols"], pc[None])[0] + rng.normal(size=2) * px_noise
            frames[fi]["keypoints"].append(px.astype(np.float32)); obs.append((int(fi), len(frames[fi]["keypoints"]) - 1))
        tracks.append(dict(point=Xt[p] + rng.normal(size=3) * pt_noise, obs=obs))
    return tracks, Xt


def _bundle_scene(rng, F=5, M=80):
    rows, cols = 2880, 5760
    frames, true = [], []
    for i in range(F):
        R = synth.rodrigues(rng.normal(size=3) * 0.15); t = np.array([0.5 * i, 0.03 * i, 0.15 * i]) + rng.normal(size=3) * 0.03
        true.append((R, t))
        d = synth.rodrigues(rng.normal(size=3) * (0.004 if i else 0.0))
        frames.append(dict(id=i, rows=rows, cols=cols, valid=1, R_wc=R @ d, t_wc=t + (rng.normal(size=3) * 0.02 if i else 0.0),
                           lines=np.zeros((0, 4), np.float32)))
    tracks, Xt = _add_tracks(rng, frames, true, M, (3.0, 1.0, 3.0))
    return frames, tracks, Xt


@pytest.mark.parametrize("refine_structure", [1, 0])
def test_bundle_adjustment_matches_cpu_twin(oracle, tmp, refine_structure):
    """AddCameraResidual + SetOptionsSfM + Solve: reprojection blocks linearised on the GPU, 3-D points eliminated
    there (Schur), camera system solved on the host — against the twin's dense full-system LM on the oracle."""
    rng = np.random.default_rng(91)
    frames, tracks, _ = _bundle_scene(rng)
    fpath, spath = os.path.join(tmp, "bf.bin"), os.path.join(tmp, "bs.bin")
    host_io.write_frames(fpath, np.eye(4), frames)
    host_io.write_structure(spath, frames, tracks)
    max_iter = 12
    out = host_io.run("bundle", fpath, spath, 1.5, refine_structure, max_iter)
    summ = [l.split() for l in out if l.startswith("summary")][0]
    cams = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("cam ")])
    pts = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("point ")])
    res, aa, t, X = lm_twin.bundle_adjust(oracle, [dict(f) for f in frames], tracks, 1.5, bool(refine_structure), max_iter)
    assert int(summ[2]) == res["blocks"] == sum(len(tr["obs"]) for tr in tracks)
    assert abs(float(summ[4]) - res["initial_cost"]) <= 1e-9 * res["initial_cost"]
    assert res["final_cost"] < (0.2 if refine_structure else 0.8) * res["initial_cost"]   # the adjustment does something
    assert abs(float(summ[6]) - res["final_cost"]) <= 1e-6 * res["final_cost"]
    assert int(summ[8]) == res["successful"] and int(summ[10]) == res["unsuccessful"]
    assert np.abs(cams[:, :3] - aa).max() <= 1e-6 and np.abs(cams[:, 3:] - t).max() <= 1e-6
    assert np.abs(pts - X).max() <= 1e-6 * max(1.0, np.abs(X).max())
    if not refine_structure:
        assert np.array_equal(pts, np.array([tr["point"] for tr in tracks]))     # constant blocks are untouched
    # camera 0 is the gauge
    a0, t0 = lm_twin.frame_params(oracle, frames)
    assert np.array_equal(cams[0, :3], a0[0]) and np.array_equal(cams[0, 3:], t0[0])
    # the three-block functor alone (API parity): r and J against the oracle
    single = np.array([float(v) for v in [l for l in out if l.startswith("single")][0].split()[1:]])
    ro, Jo = oracle.evaluate_reproj(np.array([[0.3, -0.2, 1.0]]), 1.5, [0], [0], cams[1:2, :3], cams[1:2, 3:], pts[0:1])
    assert abs(single[0] - ro[0]) <= 1e-6 * abs(ro[0]) and np.abs(single[1:] - Jo[0]).max() <= 1e-6 * np.abs(Jo[0]).max()


def test_joint_optimize_with_sfm_term_matches_cpu_twin(oracle, tmp):
    """CameraLidarOptimizer::JointOptimize with all three terms of Optimize (CameraLidarOptimizer.cpp:387-548):
    camera-LiDAR line pairs, SfM reprojection with free 3-D points, LiDAR-LiDAR point-to-plane.
    The one-angle reprojection residual makes LM converge slowly (rank-one Gauss-Newton blocks), so the solve ends on
    Ceres' function tolerance |dcost| <= 1e-6 cost after 30-40 steps — a borderline test that the GPU run and the twin
    may pass one or a few steps apart.  Parity is therefore asserted step by step on the cost history (1e-6 per LM
    step on the common prefix), and on the final state to 1e-6 when both stop at the same step, 1e-4 otherwise."""
    rng = np.random.default_rng(78)
    NS = 5
    lidars, frames, T_cl = _joint_scene(rng, NS)
    true = []
    for k in range(NS):
        R_true, t_true = sy.true_pose(k)
        T_wc = lm_twin.pose4(R_true, t_true) @ np.linalg.inv(T_cl)
        true.append((T_wc[:3, :3].copy(), T_wc[:3, 3].copy()))
    tracks, _ = _add_tracks(rng, frames, true, 60, (1.0, 0.5, 0.8), center=(0.0, 0.0, -3.3))   # around the trajectory: parallax
    lpath, fpath, spath = os.path.join(tmp, "sl.bin"), os.path.join(tmp, "sf.bin"), os.path.join(tmp, "ss.bin")
    host_io.write_scans(lpath, lidars, world=False)
    host_io.write_frames(fpath, T_cl, frames)
    host_io.write_structure(spath, frames, tracks)
    out = host_io.run("joint", lpath, fpath, 3, 1, 0, 1, 0.05, 1.0, 0.3, 1.0, 2.0, 1.5, spath)
    it = [l.split() for l in out if l.startswith("iter")][0]
    hist = np.array([float(v) for v in [l for l in out if l.startswith("hist")][0].split()[1:]])
    lposes = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    fposes = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("frame")}
    pts = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("point ")])
    tl = [dict(s) for s in lidars]; tf = [dict(f) for f in frames]
    structure = dict(tracks=tracks, X=np.array([tr["point"] for tr in tracks], np.float64))
    cfg = dict(p2plane=True, tol=0.05, thr=1.0, lidar_weight=1.0, camera_lidar_weight=2.0, camera_weight=1.5)
    lg = lm_twin.joint_optimize(oracle, tl, tf, T_cl, cfg, 3, 1, structure)[0]
    n_reproj = sum(len(tr["obs"]) for tr in tracks)
    assert int(it[8]) == lg["pairs"] > 10
    assert int(it[6]) == lg["blocks"] > n_reproj
    th = np.array(lg["history"])
    n = min(len(hist), len(th))
    assert n >= 20 and abs(len(hist) - len(th)) <= 6
    assert np.all(np.abs(hist[:n] - th[:n]) <= 1e-6 * th[:n]), (np.abs(hist[:n] - th[:n]) / th[:n]).max()
    assert hist[-1] < 0.2 * hist[0]
    same = len(hist) == len(th)
    tol = 1e-6 if same else 5e-4          # measured with the GPU run 4 steps longer: poses 6e-5, points 1e-3, cost 2.7e-5
    assert int(it[4]) == len(hist) and lg["successful"] == len(th)
    assert abs(float(it[2]) - lg["final_cost"]) <= (1e-6 if same else 1e-3) * lg["final_cost"]
    for k in range(NS):
        assert np.abs(lposes[k][:9].reshape(3, 3) - tl[k]["R_wl"]).max() <= tol and np.abs(lposes[k][9:] - tl[k]["t_wl"]).max() <= tol
        assert np.abs(fposes[k][:9].reshape(3, 3) - tf[k]["R_wc"]).max() <= tol and np.abs(fposes[k][9:] - tf[k]["t_wc"]).max() <= tol
    moved = np.abs(structure["X"] - np.array([tr["point"] for tr in tracks])).max()
    # 1e-6 relative to the size of the coordinates (|X| up to 3.7 m here): a point's depth is its least constrained direction,
    # and over the 30+ LM steps of this solve the GPU / twin difference was observed between 3e-8 and 1.0e-6 absolute.  (Until
    # the reprojection blocks were summed without atomics — DESIGN.md §3 K9 — that figure also moved from run to run.)
    assert pts.shape == structure["X"].shape and np.abs(pts - structure["X"]).max() <= (1e-6 if same else 1e-2) * max(1.0, moved, np.abs(structure["X"]).max())
    assert moved > 1e-4      # the structure was refined


def test_knn_based_line_association_variants(oracle, tmp):
    """AssociatePoint2Line / ...SegmentKNN / ...Segment / AssociateLine2LineKNN (LidarFeatureAssociate.cpp:238-440,
    :478-548): 5-NN in the ref corner cloud on the GPU (pvlm_knn), PCA / segment counting on the host."""
    rng = np.random.default_rng(19)
    scans = _line_scans(rng, 3)
    path = os.path.join(tmp, "p2l.bin")
    host_io.write_scans(path, scans)
    for mode, name in ((0, "knn"), (1, "segment_knn"), (2, "segment")):
        for r, n, thr in [(0, 1, 0.5), (1, 2, 0.3)]:
            got = np.array([[float(v) for v in l.split()[1:]] for l in host_io.run("p2line", path, r, n, mode, thr)]).reshape(-1, 9)
            o = oracle.assoc_point2line(scans[r]["_oracle"], scans[n]["_oracle"], thr, mode=name)
            assert len(got) == len(o["point"]) > 10, (name, len(got), len(o["point"]))
            assert np.allclose(got[:, :3], o["point"], rtol=0, atol=1e-12)
            # the fitted direction's sign is the eigen solver's choice: (a, b) may come swapped
            same = np.abs(got[:, 3:6] - o["a"]).max(axis=1) <= 1e-9
            swapped = np.abs(got[:, 3:6] - o["b"]).max(axis=1) <= 1e-9
            assert np.all(same | swapped)
            assert np.all(np.where(same[:, None], np.abs(got[:, 6:9] - o["b"]), np.abs(got[:, 6:9] - o["a"])) <= 1e-9)
            if mode:
                assert same.all()          # segment coefficients are taken as they are
    for r, n, thr in [(0, 1, 0.5), (2, 1, 0.4)]:
        got = [l.split()[1:] for l in host_io.run("line2lineknn", path, r, n, thr)]
        o = oracle.assoc_line2line(scans[r]["_oracle"], scans[n]["_oracle"], thr, knn=True)
        assert [int(g[0]) for g in got] == o["nei_idx"].tolist() and [int(g[1]) for g in got] == o["ref_idx"].tolist()
        assert np.allclose(np.array([[float(v) for v in g[2:5]] for g in got]).reshape(-1, 3), o["p1"], atol=1e-13)
        assert len(got) >= 4


def test_raw_scans_to_refined_poses(oracle, tmp):
    """BASELINE config 0 plumbing: raw VLP-16 scans in firing order (~29 k points) -> ReOrderVLP -> ExtractFeatures (host,
    SURVEY.md §8 N3 planar branch) -> LidarOdometry::EstimatePose with the point-to-plane term (GPU association + normal
    equations) against the same chain on the oracle: feature counts, residual-block counts, costs and poses."""
    scans = []
    for k in range(3):
        R, t = sy.estimated_pose(k)
        scans.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=30)))
    path = os.path.join(tmp, "raw.bin")
    host_io.write_raw_scans(path, scans)
    out = host_io.run("rawodometry", path, 2, 1, 1, 0.05, 1.0, 1000.0, 5.0, 1)
    feats = [l.split() for l in out if l.startswith("features")]
    iters = [l.split() for l in out if l.startswith("iter")]
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    twin = []
    for s, ft in zip(scans, feats):
        f = oracle.ScanFeatures(s["raw"])
        assert (int(ft[5]), int(ft[7]), int(ft[9])) == (len(f.surfFlat), len(f.surfLessFlat), len(f.cornerLessSharp))
        assert 300 <= len(f.surfFlat) <= 384 and 2000 < len(f.surfLessFlat) < 8000                 # Nq <= 384, Nt = O(3-8 k)
        twin.append(dict(id=s["id"], R_wl=s["R_wl"], t_wl=s["t_wl"], flat_local=f.surfFlat[:, :3], flat_tag=f.surfFlat[:, 3],
                         less_local=f.surfLessFlat[:, :3], less_tag=f.surfLessFlat[:, 3]))
    log = lm_twin.estimate_pose(oracle, twin, dict(angle=True, normalize=True, tol=0.05, thr=1.0), 2)
    assert len(iters) == len(log)
    for it, lg in zip(iters, log):
        assert int(it[6]) == lg["blocks"] and lg["blocks"] > 500
        assert abs(float(it[2]) - lg["final_cost"]) <= 1e-6 * lg["final_cost"]
    for k, s in enumerate(twin):
        R = poses[k][:9].reshape(3, 3); t = poses[k][9:]
        assert np.abs(R - s["R_wl"]).max() <= 1e-6 and np.abs(t - s["t_wl"]).max() <= 1e-6 * max(1.0, np.abs(s["t_wl"]).max())
    err0 = np.mean([np.linalg.norm(scans[k]["t_wl"] - sy.true_pose(k)[1]) for k in range(1, 3)])
    err1 = np.mean([np.linalg.norm(poses[k][9:] - sy.true_pose(k)[1] - (poses[0][9:] - sy.true_pose(0)[1])) for k in range(1, 3)])
    assert err1 < err0



def test_feature_batch_takes_the_host_path_for_a_scan_the_device_refuses(oracle):
    """A scan with a non-finite coordinate: the device batch refuses it (upstream gives such a point ring -1 and skips it), ExtractFeaturesBatch runs the per-scan
    host path for that part of the call, and the result is the oracle's — clouds, per-point arrays and line segments."""
    raw = sy.raw_vlp16_scan(3, clutter=20).copy()
    raw[1234, 1] = np.nan; raw[20000, 0] = np.inf
    o = oracle.ScanFeatures(raw, edge_to_line=True)
    g = host_io.extract_features(raw, edge_to_line=True, on_gpu=True)
    assert o.valid and g["valid"]
    for name in ("cloud_scan", "cornerSharp", "cornerLessSharp", "surfFlat", "surfLessFlat", "rc", "scan_start", "scan_end", "curvature", "state", "sort_ind", "left", "right"):
        a, b = getattr(o, name), g[name]
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), name
    assert len(o.edge_segmented) == len(g["edge_segmented"]) and all(np.array_equal(a, b) for a, b in zip(o.edge_segmented, g["edge_segmented"]))


def _fnv(xyz):
    """FNV-1a over the 32-bit words of an n x 3 float32 array, as the driver's `features2` lines print it."""
    h = 1469598103934665603
    for w in np.ascontiguousarray(xyz, np.float32).view(np.uint32).reshape(-1).tolist():
        h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_two_passes_with_motion_compensation(oracle, tmp):
    """main.cpp:415-432: EstimatePose, UndistortLidars with the poses it found, ResetAllLidars, EstimatePose again — through the host mirror (K25 for the
    clouds, the GPU feature batch incl. K23 / K24 inside the second EstimatePose, GPU association + normal equations) against the same chain on the oracle,
    started from the poses the mirror's first pass printed: compensated clouds -> feature counts, residual-block counts, costs, poses."""
    scans = []
    for k in range(4):
        R, t = sy.estimated_pose(k)
        scans.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=30)))
    path = os.path.join(tmp, "raw2.bin")
    host_io.write_raw_scans(path, scans)
    gap = 0.05
    out = host_io.run("rawodometry", path, 2, 1, 1, 0.05, 1.0, 1000.0, 5.0, 1, 0, 0.3, gap)
    pose1 = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pass1pose")}
    feats2 = [l.split() for l in out if l.startswith("features2")]
    comp = {int(l.split()[1]): int(l.split()[3]) for l in out if l.startswith("compensated")}
    iters = [l.split() for l in out if l.startswith("iter")]
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose ")}
    assert len(pose1) == 4 and len(feats2) == 4 and len([l for l in out if l.startswith("pass1 ")]) == 2
    first = [(pose1[k][:9].reshape(3, 3), pose1[k][9:]) for k in range(4)]
    twin = []
    moved = 0
    for k, s in enumerate(scans):
        end = oracle.sweep_end_pose(first, [1] * 4, [1] * 4, k, gap)
        raw = s["raw"]
        if end is not None:
            done, raw = oracle.undistort_cloud(raw, *first[k], *end)
            assert done
            moved += int(not np.array_equal(raw, s["raw"]))
        f = oracle.ScanFeatures(raw)
        assert (int(feats2[k][5]), int(feats2[k][7])) == (len(f.surfFlat), len(f.surfLessFlat)), k
        assert comp[k] == _fnv(raw[:, :3]), ("compensated cloud", k)
        twin.append(dict(id=k, R_wl=first[k][0], t_wl=first[k][1], flat_local=f.surfFlat[:, :3], flat_tag=f.surfFlat[:, 3],
                         less_local=f.surfLessFlat[:, :3], less_tag=f.surfLessFlat[:, 3]))
    assert moved == 4
    log = lm_twin.estimate_pose(oracle, twin, dict(angle=True, normalize=True, tol=0.05, thr=1.0), 2)
    assert len(iters) == len(log)
    for it, lg in zip(iters, log):
        assert int(it[6]) == lg["blocks"] and lg["blocks"] > 500
        assert abs(float(it[2]) - lg["final_cost"]) <= 1e-6 * lg["final_cost"]
    for k, s in enumerate(twin):
        R = poses[k][:9].reshape(3, 3); t = poses[k][9:]
        assert np.abs(R - s["R_wl"]).max() <= 1e-6 and np.abs(t - s["t_wl"]).max() <= 1e-6 * max(1.0, np.abs(s["t_wl"]).max())


def _parse_odometry(out):
    iters = [l.split() for l in out if l.startswith("iter")]
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    return iters, poses


def test_sharded_estimate_pose_two_ranks_on_one_gpu(tmp):
    """LidarOdometry::EstimatePose sharded over two processes (Exchange; the loop that is sharded is
    util/Optimization.cpp:521-560 / :345-441): each rank associates and evaluates the blocks of its half of the
    reference scans on the GPU, the ranks sum [cost | g | 6x6 blocks] once per evaluation, everything else is
    replicated.  Both ranks must report the same log and poses, bit for bit, and agree with the one-process run
    (same blocks; 1e-9: the order of the sums differs)."""
    import subprocess
    rng = np.random.default_rng(77)
    # point-to-plane scans that also carry line segments: both sharded adders run
    scans = [_vlp(k, 256) for k in range(6)]
    lines = _line_scans(rng, 6, pose_offset=0)          # same estimated poses as the point-to-plane scans
    for s, l in zip(scans, lines):
        assert np.array_equal(s["R_wl"], l["R_wl"]) and np.array_equal(s["t_wl"], l["t_wl"])
        for key in ("corner_local", "p2s", "seg_points", "seg_coeffs", "end_points"):
            s[key] = l[key]
    path = os.path.join(tmp, "shard.bin")
    host_io.write_scans(path, scans, world=False)
    args = ["odometry", path, 2, 1, 1, 1, 1, 0.05, 1.0, 0.3]
    single = host_io.run(*args)
    xdir = os.path.join(tmp, "xchg"); os.makedirs(xdir, exist_ok=True)
    procs = [subprocess.Popen([host_io.driver()] + [str(a) for a in args] + ["2", str(r), "file:" + xdir], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-2000:]
        outs.append(o.splitlines())
    it0, po0 = _parse_odometry(outs[0]); it1, po1 = _parse_odometry(outs[1]); its, pos = _parse_odometry(single)
    assert [l[:7] for l in it0] == [l[:7] for l in it1] and len(it0) == len(its) >= 1          # identical decisions on both ranks
    for k in po0:
        assert np.array_equal(po0[k], po1[k])
    for a, b in zip(it0, its):
        assert int(a[6]) == int(b[6]) > 1000 and int(a[4]) == int(b[4])
        assert abs(float(a[2]) - float(b[2])) <= 1e-9 * float(b[2])
    for k in pos:
        assert np.abs(po0[k] - pos[k]).max() <= 1e-9 * max(1.0, np.abs(pos[k]).max())
    # each rank really did only its share: the association stage of a rank saw half of the reference scans
    assert any(l.startswith("stage") and "exchange of the normal equations" in l for l in outs[0])


def test_resident_reposing_equals_reupload(tmp, monkeypatch):
    """K26 under the call surface: LidarOdometry::EstimatePose (three outer iterations, both LiDAR terms) and CameraLidarOptimizer::JointOptimize with the
    scans' device copies re-posed in place (pvlm_scan_transform_batch inside Velodyne::TransformBatch — sensors/Velodyne.cpp:1773-1848 on the resident
    clouds) against the same calls with every scan dropped and uploaded again at each outer iteration (PVLM_HOST_REUPLOAD=1, the round-5 path):
    same costs, step counts, block counts and poses, bit for bit — and after the first iteration no scan upload at all."""
    rng = np.random.default_rng(78)
    scans = [_vlp(k, 256) for k in range(6)]
    lines = _line_scans(rng, 6, pose_offset=0)
    for s, l in zip(scans, lines):
        for key in ("corner_local", "p2s", "seg_points", "seg_coeffs", "end_points"):
            s[key] = l[key]
    path = os.path.join(tmp, "repose.bin")
    host_io.write_scans(path, scans, world=False)
    args = ["odometry", path, 3, 1, 1, 1, 1, 0.05, 1.0, 0.3]
    monkeypatch.delenv("PVLM_HOST_REUPLOAD", raising=False)
    resident = host_io.run(*args)
    monkeypatch.setenv("PVLM_HOST_REUPLOAD", "1")
    reupload = host_io.run(*args)
    monkeypatch.delenv("PVLM_HOST_REUPLOAD")
    keep = lambda out: [l for l in out if l.startswith(("iter", "pose"))]
    assert keep(resident) == keep(reupload) and len([l for l in resident if l.startswith("iter")]) >= 2
    uploads = lambda out: [int(re.search(r"\[(\d+) calls\]", l).group(1)) for l in out if l.startswith("stage") and "scan upload: host SoA staging + pvlm_scan_upload" in l]
    n_iter = len([l for l in resident if l.startswith("iter")])
    assert uploads(resident) == [1] and uploads(reupload)[0] >= n_iter, (uploads(resident), uploads(reupload))
    assert any(l.startswith("stage") and "pvlm_scan_transform_batch" in l for l in resident)
    # the joint problem: scans uploaded in the LOCAL frame by AssociateLineMulti, to the world frame and back around every solve
    rng = np.random.default_rng(77)
    lidars, frames, T_cl = _joint_scene(rng, 3)
    lpath, fpath = os.path.join(tmp, "jl2.bin"), os.path.join(tmp, "jf2.bin")
    host_io.write_scans(lpath, lidars, world=False)
    host_io.write_frames(fpath, T_cl, frames)
    jargs = ["joint", lpath, fpath, 3, 2, 0, 1, 0.05, 1.0, 0.3, 1.0, 2.0]
    resident = host_io.run(*jargs)
    monkeypatch.setenv("PVLM_HOST_REUPLOAD", "1")
    reupload = host_io.run(*jargs)
    keep = lambda out: [l for l in out if l.startswith(("iter", "pose", "frame"))]
    assert keep(resident) == keep(reupload) and len(keep(resident)) > 6


def test_raw_scans_with_line_segments_to_refined_poses(oracle, tmp):
    """SURVEY.md §8 N3, line branch: raw VLP-16 scans -> ReOrderVLP -> ExtractFeatures WITH EdgeToLine (host) -> segments ->
    LidarLineMatch tracks + line-to-line blocks + point-to-plane blocks in LidarOdometry::EstimatePose (GPU votes,
    association and normal equations) against the same chain on the oracle (oracle/lines.hpp feeds the twin)."""
    scans = []
    for k in range(4):
        R, t = sy.estimated_pose(k)
        scans.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=20)))
    path = os.path.join(tmp, "raw_lines.bin")
    host_io.write_raw_scans(path, scans)
    out = host_io.run("rawodometry", path, 2, 1, 1, 0.05, 1.0, 1000.0, 5.0, 1, 1, 0.3)
    feats = [l.split() for l in out if l.startswith("features")]
    iters, poses = _parse_odometry(out)
    twin = []
    for s, ft in zip(scans, feats):
        f = oracle.ScanFeatures(s["raw"], edge_to_line=True)
        assert (int(ft[5]), int(ft[7]), int(ft[9]), int(ft[11])) == (len(f.surfFlat), len(f.surfLessFlat), len(f.cornerLessSharp), len(f.edge_segmented))
        assert len(f.edge_segmented) >= 5
        twin.append(lm_twin.twin_scan_from_features(s["id"], s["R_wl"], s["t_wl"], f))
    log = lm_twin.estimate_pose_full(oracle, twin, dict(angle=True, normalize=True, tol=0.05, thr=1.0, line_thr=0.3), 2)
    assert len(iters) == len(log)
    for it, lg in zip(iters, log):
        assert int(it[6]) == lg["blocks"] and lg["line_blocks"] > 50          # the line-to-line term is really in the problem
        assert abs(float(it[2]) - lg["final_cost"]) <= 1e-6 * lg["final_cost"]
        assert int(it[4]) == lg["successful"]
    for k, s in enumerate(twin):
        R = poses[k][:9].reshape(3, 3); t = poses[k][9:]
        assert np.abs(R - s["R_wl"]).max() <= 1e-6 and np.abs(t - s["t_wl"]).max() <= 1e-6 * max(1.0, np.abs(s["t_wl"]).max())


def test_ceres_adapter_rows(tmp):
    """integration/pvlm_ceres.hpp (the reference-side binding that keeps Ceres as the outer solver), compiled against the
    interface-only Ceres stand-in of tests/cpp/ceres_double and run on the GPU: the residual and the 1 x 12 Jacobian row
    every CeresRow::Evaluate hands out (formed on the host from the 56-byte wrench row + its pair's tables, delivered
    through pinned memory) equal pvlm_eval's materialised rows; a second evaluation at a moved point is really re-run."""
    scans = [_vlp(k, 256) for k in range(4)]
    path = os.path.join(tmp, "ceres.bin")
    host_io.write_scans(path, scans, world=False)
    out = host_io.run("ceresadapter", path, 0.05, 1.0)
    f = [l.split() for l in out if l.startswith("blocks")][0]
    n, direct, dr, dJ, jmax, moved = int(f[1]), int(f[3]), float(f[5]), float(f[7]), float(f[9]), float(f[11])
    assert n == direct > 1000
    assert dr == 0.0                                  # the residual is copied, not recomputed
    assert dJ <= 1e-12 * jmax                         # the row is re-formed on the host: rounding only
    assert moved > 1e-7                               # PrepareForEvaluation(new_evaluation_point = true) re-evaluated



def test_ceres_adapter_full_optimize_problem(tmp):
    """integration/pvlm_ceres.hpp, all FOUR adders of CameraLidarOptimizer::Optimize (AddCameraLidarResidualGpu: Plane2Plane_Global +
    PlaneIOU, AddCameraResidualGpu: three-block reprojection rows from pvlm_ba_eval, AddLidarLineToLineResidual2Gpu: device-built
    Point2Line rows with a nullptr loss, AddLidarPointToPlaneResidualGpu) on ONE CeresBatch against the interface-only Ceres
    stand-in: block counts equal to the host mirror's adders, every row of one "Ceres evaluation" equal to pvlm_eval /
    pvlm_ba_eval of the same sets (1e-13 of the row's scale), the losses the reference passes, and a re-evaluation after the
    parameters — points included — moved."""
    rng = np.random.default_rng(78)
    NS = 5
    lidars, frames, T_cl = _joint_scene(rng, NS)
    true = []
    for k in range(NS):
        R_true, t_true = sy.true_pose(k)
        T_wc = lm_twin.pose4(R_true, t_true) @ np.linalg.inv(T_cl)
        true.append((T_wc[:3, :3].copy(), T_wc[:3, 3].copy()))
    tracks, _ = _add_tracks(rng, frames, true, 60, (1.0, 0.5, 0.8), center=(0.0, 0.0, -3.3))
    lpath, fpath, spath = os.path.join(tmp, "cl.bin"), os.path.join(tmp, "cf.bin"), os.path.join(tmp, "cs.bin")
    host_io.write_scans(lpath, lidars, world=False)
    host_io.write_frames(fpath, T_cl, frames)
    host_io.write_structure(spath, frames, tracks)
    out = host_io.run("ceresjoint", lpath, fpath, spath, 3, 0.05, 1.0, 0.3)
    c = [l.split() for l in out if l.startswith("counts")][0]
    adapter = [int(v) for v in c[2:6]]; mirror = [int(v) for v in c[7:11]]; total = int(c[12])
    assert adapter == mirror and total == sum(adapter), (adapter, mirror, total)
    n_cl, n_cam, n_l2l, n_p2p = adapter
    assert n_cl >= 20 and n_cam == sum(len(tr["obs"]) for tr in tracks) and n_l2l > 100 and n_p2p > 1000
    r = [l.split() for l in out if l.startswith("rows")][0]
    checked, of, dr, dJ, jmax, loss_errors, moved, sets = int(r[2]), int(r[4]), float(r[6]), float(r[8]), float(r[10]), int(r[12]), float(r[14]), int(r[16])
    assert checked == of == total and sets == 4
    assert dr == 0.0                                  # residuals are copied, not recomputed
    assert dJ <= 1e-13 * jmax                         # four-block rows are re-formed on the host from the wrench rows: rounding only
    assert loss_errors == 0                           # Huber(3 deg) camera-LiDAR, Huber(4 deg) reprojection, nullptr line-to-line (Angle), Huber point-to-plane
    assert moved > 1e-7

def test_host_visible_evaluation_matches_device_rows(oracle):
    """pvlm_eval_host_async / pvlm_eval_wrench_host_async / pvlm_eval_force_host_async into pinned memory, sliced through a tiny staging buffer."""
    import subprocess, sys
    code = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "import panovlm_amd as pv\nfrom tests import synth\n"
        "rng = np.random.default_rng(3); F, P = 6, 17\n"
        "aa, t = synth.random_poses(rng, F); ref, nei = synth.random_pairs(rng, F, P)\n"
        "rows, off = synth.random_resset(rng, 1, aa, t, ref, nei, rng.integers(0, 900, size=P))\n"
        "ctx = pv.Context(0); rs = pv.ResidualSet.upload(ctx, 1, rows, off, ref, nei, flags=1); ctx.set_poses(aa, t)\n"
        "r, J = rs.eval(jac=True); n = rs.n\n"
        "hr = ctx.host_alloc(n * 8); hJ = ctx.host_alloc(n * 96); hw = ctx.host_alloc(n * 56); ht = ctx.host_alloc(P * 33 * 8)\n"
        "rs.eval_host_async(hr, hJ); rs.eval_wrench_host_async(hw, ht); ctx.synchronize()\n"
        "w = hw[:n * 7].reshape(n, 7); T = ht[:P * 33].reshape(P, 33)\n"
        "pr = np.repeat(np.arange(P), np.diff(off))\n"
        "c, g = w[:, 1:4], w[:, 4:7]; R = T[pr, 0:9].reshape(n, 3, 3); Jl = T[pr, 15:24].reshape(n, 3, 3); Mn = T[pr, 24:33].reshape(n, 3, 3)\n"
        "Jw = np.concatenate([np.einsum('ni,nik->nk', c, Jl), g, np.einsum('ni,nik->nk', c, Mn), -np.einsum('ni,nik->nk', g, R)], axis=1)\n"
        "ok = np.array_equal(hr[:n], r) and np.array_equal(hJ[:n * 12].reshape(n, 12), J) and np.array_equal(w[:, 0], r)\n"
        "sc = np.abs(J).max(axis=1, keepdims=True); ok = ok and bool(np.all(np.abs(Jw - J) <= 1e-12 * sc))\n"
        "hr2 = np.zeros(n); rs.eval_host_async(hr2, None); ctx.synchronize(); ok = ok and np.array_equal(hr2, r)\n"
        "hf = ctx.host_alloc(n * 32); ht2 = ctx.host_alloc(P * 33 * 8); rs.eval_force_host_async(hf, ht2); ctx.synchronize()\n"       # [r | g]: 32 B per block
        "f = hf[:n * 4].reshape(n, 4); ok = ok and np.array_equal(f[:, 0], r) and np.array_equal(f[:, 1:4], g) and np.array_equal(ht2[:P * 33], ht[:P * 33])\n"
        "Pn = rows[:, 0:3]; Pr = np.einsum('nik,nk->ni', R, Pn) + T[pr, 9:12]; c2 = np.cross(Pr - T[pr, 12:15], g)\n"               # the moment the host rebuilds
        "ok = ok and bool(np.all(np.abs(c2 - c) <= 1e-12 * np.maximum(np.abs(c).max(axis=1, keepdims=True), 1e-300)))\n"
        "print('OK' if ok else 'MISMATCH', n)\n" % host_io.ROOT)
    for stage in ("700", None):                       # 700 rows: every pair its own slice; default: one slice
        env = dict(os.environ)
        if stage:
            env["PVLM_STAGE_ROWS"] = stage
        o = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert o.returncode == 0, o.stderr[-2000:]
        assert o.stdout.split()[0] == "OK" and int(o.stdout.split()[1]) > 1000, o.stdout


@pytest.mark.parametrize("case", [dict(k=3), dict(k=2, clutter=150, dropout=0.3), dict(k=4, jitter=0.6, skew=0.9), dict(k=5, start_deg=359.0, elevation_noise=0.6),
                                  dict(k=7, start_deg=180.0, dropout=0.9), dict(k=8, segment=False, max_curvature=5.0, angle_threshold=10.0),
                                  dict(k=9, cols=4096, clutter=40)],
                         ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
@pytest.mark.parametrize("picks", ["device", "host"])
def test_feature_extraction_with_gpu_stages_matches_oracle(oracle, case, picks, monkeypatch):
    """Velodyne::ExtractFeaturesBatch — ring / column order, range image, Segmentation, curvature, sector orders from the GPU
    (pvlm_ring_extract_batch[_picks]); the picks and the voxel grid from K24 ("device": AssemblePicks) or on the host from the device's arrays
    ("host": PickFeatures, the path a scan with an undecided ring takes); EdgeToLine on the host — leaves a scan exactly as the oracle's
    ReOrderVLP + ExtractFeatures do: every cloud, every per-point array, the line segments."""
    monkeypatch.setenv("PVLM_FEATURE_PICKS", picks)
    c = dict(case); k = c.pop("k")
    ext = {n: c.pop(n) for n in ("segment", "max_curvature", "angle_threshold") if n in c}
    cols = c.get("cols", 1800)
    raw = sy.raw_vlp16_scan(k, **c)
    o = oracle.ScanFeatures(raw, horizon=cols, max_curvature=ext.get("max_curvature", 1000.0), intersect_angle_threshold=ext.get("angle_threshold", 5.0),
                            segment=ext.get("segment", True), edge_to_line=True)
    g = host_io.extract_features(raw, horizon=cols, edge_to_line=True, on_gpu=True, **ext)
    assert o.valid == g["valid"]
    names = ("cloud_scan", "rc", "scan_start", "scan_end") if not o.valid else \
        ("cloud_scan", "cornerSharp", "cornerLessSharp", "surfFlat", "surfLessFlat", "rc", "scan_start", "scan_end", "range_image", "image_to_point_idx", "curvature",
         "state", "sort_ind", "left", "right")
    for name in names:
        a, b = getattr(o, name), g[name]
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), name
    if o.valid:
        assert len(o.edge_segmented) == len(g["edge_segmented"])
        for a, b in zip(o.edge_segmented, g["edge_segmented"]):
            assert np.array_equal(a, b)
        assert np.array_equal(o.segment_coeffs, g["segment_coeffs"], equal_nan=True) and np.array_equal(o.end_points, g["end_points"], equal_nan=True)
        assert o.point_to_segment == g["point_to_segment"] and np.array_equal(o.cornerBeforeFilter, g["cornerBeforeFilter"])
