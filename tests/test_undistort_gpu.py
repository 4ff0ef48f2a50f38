"""K25 (pvlm_undistort_batch: Velodyne::UndistortCloud for a batch) and the host mirror's LidarOdometry::UndistortLidars against the oracle
(oracle/undistort.hpp).  Floating point through double sines on two libms: tolerance 1e-6 relative (north_star), stated below; most points are identical."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import panovlm_amd as pv
from panovlm_amd import api
from panovlm_amd import synthetic as sy
from tests import host_io

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    c = pv.Context()
    yield c
    c.close()


def _pose(rng, angle=0.3, shift=1.0):
    return Rotation.from_rotvec(rng.normal(0, angle, 3)).as_matrix(), rng.normal(0, shift, 3)


def _close(got, want):
    scale = np.maximum(np.abs(want[:, :3]).max(axis=1, keepdims=True), 1.0)
    return np.all(np.abs(got[:, :3] - want[:, :3]) <= TOL * scale) and np.array_equal(got[:, 3], want[:, 3])


def test_batch_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(12)
    clouds, starts, ends = [], [], []
    for k in range(9):
        n = [0, 1, 2, 255, 256, 257, 5000, 28800, 40000][k]
        clouds.append(np.concatenate([rng.normal(0, 10, (n, 3)), rng.integers(0, 16, (n, 1))], axis=1).astype(np.float32))
        R_wl, t_wl = _pose(rng, 2.0, 20.0)
        dR, dt = _pose(rng, [0.05, 1e-10, 0.0, 0.3, 3.0, 0.01, 0.02, 0.03, 0.04][k], 0.3)          # small, vanishing, zero, large and near-pi rotations
        starts.append((R_wl, t_wl)); ends.append((R_wl @ dR, t_wl + R_wl @ dt))
    got = api.undistort_batch(ctx, clouds, starts, ends)
    same = total = 0
    for k in range(len(clouds)):
        done, want = oracle.undistort_cloud(clouds[k], *starts[k], *ends[k])
        assert done == (len(clouds[k]) > 0)
        assert _close(got[k], want), k
        same += int(np.sum(np.all(got[k].view(np.uint32) == want.view(np.uint32), axis=1))); total += len(want)
        if len(want):
            assert np.array_equal(got[k][0], clouds[k][0])                                             # ratio 0: the sweep's first point stays
    print("points identical to the oracle bit for bit: %d of %d" % (same, total))
    assert same >= 0.99 * total
    # batch == scan by scan
    for k in (3, 6):
        alone = api.undistort_batch(ctx, [clouds[k]], [starts[k]], [ends[k]])[0]
        assert np.array_equal(alone.view(np.uint32), got[k].view(np.uint32))


def test_undistort_lidars_matches_oracle(oracle, tmp_path):
    """LidarOdometry::UndistortLidars through the host mirror: which pose ends each sweep (next usable scan through SlerpPose; the last scan continues the motion;
    scans without a pose or flagged invalid stay as they are), then K25 on all scans at once — against the oracle's pose rule + per-point loop."""
    rng = np.random.default_rng(5)
    n_scans = 9
    scans = []
    t = np.zeros(3)
    for k in range(n_scans):
        R = Rotation.from_rotvec([0.0, 0.05 * k, 0.01 * k]).as_matrix()
        t = t + rng.normal(0.1, 0.02, 3)
        raw = sy.raw_vlp16_scan(k, cols=360, clutter=10)
        scans.append(dict(id=k, R_wl=R, t_wl=t.copy(), raw=raw))
    scans[3]["R_wl"] = np.zeros((3, 3))                         # no pose
    invalid = [6]                                               # flagged invalid (pose kept)
    path, out = str(tmp_path / "raw.bin"), str(tmp_path / "out.bin")
    host_io.write_raw_scans(path, scans)
    for gap in (0.0, 0.05):
        lines = host_io.run("undistort", path, gap, out, *invalid)
        assert any(l.startswith("undistort_seconds") for l in lines)
        blob = np.fromfile(out, np.float32)
        poses = [(s["R_wl"], s["t_wl"]) for s in scans]
        pose_ok = [0 if k == 3 else 1 for k in range(n_scans)]
        ok = [0 if k in invalid else 1 for k in range(n_scans)]
        at = 0
        moved = 0
        for k, s in enumerate(scans):
            m = len(s["raw"])
            got = blob[at:at + 4 * m].reshape(m, 4); at += 4 * m
            end = oracle.sweep_end_pose(poses, pose_ok, ok, k, gap)
            if end is None:
                assert np.array_equal(got, s["raw"]), k
                continue
            done, want = oracle.undistort_cloud(s["raw"], *poses[k], *end)
            assert done and _close(got, want), k
            differ = int(np.sum(np.any(got.view(np.uint32) != want.view(np.uint32), axis=1)))
            print("scan %d: %d of %d points differ from the oracle in some bit" % (k, differ, m))
            moved += 1
        assert at == len(blob) and moved == n_scans - 2
