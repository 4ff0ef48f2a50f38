"""Motion compensation between the two EstimatePose passes (LidarOdometry::UndistortLidars -> Velodyne::UndistortCloud, SlerpPose): the oracle's
statement-by-statement restatement (oracle/undistort.hpp, Eigen's quaternion routines recalled) against scipy's Rotation / Slerp, and the host mirror's
pose selection against the oracle.  No GPU."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation, Slerp

from tests import host_io


def _pose(rng, angle=0.3, shift=1.0):
    return Rotation.from_rotvec(rng.normal(0, angle, 3)).as_matrix(), rng.normal(0, shift, 3)


def _T(R, t):
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def test_slerp_pose_against_scipy(oracle):
    rng = np.random.default_rng(3)
    for trial in range(200):
        R1, t1 = _pose(rng, 1.5); R2, t2 = _pose(rng, 1.5 if trial % 3 else 1e-9)
        ratio = float(rng.uniform(-0.5, 1.5)) if trial % 5 else float(trial % 2)
        R, t = oracle.slerp_pose(R1, t1, R2, t2, ratio)
        T21 = np.linalg.inv(_T(R2, t2)) @ _T(R1, t1)
        key = Rotation.from_matrix([np.eye(3), T21[:3, :3]])
        rv = key[1].as_rotvec()
        Rs = Rotation.from_rotvec(rv * ratio).as_matrix()                      # slerp from the identity = a fraction of the rotation vector
        want = _T(R1, t1) @ np.linalg.inv(_T(Rs, T21[:3, 3] * ratio))
        assert np.allclose(R, want[:3, :3], atol=1e-12) and np.allclose(t, want[:3, 3], atol=1e-11), trial
    R1, t1 = _pose(rng); R2, t2 = _pose(rng)
    for ratio, (Rw, tw) in ((0.0, (R1, t1)), (1.0, (R2, t2))):
        R, t = oracle.slerp_pose(R1, t1, R2, t2, ratio)
        assert np.allclose(R, Rw, atol=1e-14) and np.allclose(t, tw, atol=1e-13)


def test_undistort_cloud_against_numpy(oracle):
    rng = np.random.default_rng(8)
    for trial in range(12):
        n = int(rng.integers(1, 4000))
        cloud = np.concatenate([rng.normal(0, 8, (n, 3)), rng.integers(0, 16, (n, 1))], axis=1).astype(np.float32)
        R_wl, t_wl = _pose(rng, 2.0, 5.0)
        dR, dt = _pose(rng, 0.05 if trial % 4 else 1e-10, 0.2)
        R_we, t_we = R_wl @ dR, t_wl + R_wl @ dt                                   # the sweep ends dR, dt away from where it started
        done, got = oracle.undistort_cloud(cloud, R_wl, t_wl, R_we, t_we)
        assert done
        ratio = (np.arange(n, dtype=np.float32) * np.float32(1.0) / np.float32(n)).astype(np.float64)
        rot = Rotation.from_rotvec(Rotation.from_matrix(dR).as_rotvec()[None, :] * ratio[:, None])
        want = rot.apply(cloud[:, :3].astype(np.float64)) + ratio[:, None] * dt[None, :]
        assert np.allclose(got[:, :3], want, rtol=2e-7, atol=2e-6) and np.array_equal(got[:, 3], cloud[:, 3])
        assert np.array_equal(got[0, :3], cloud[0, :3])                            # the first point of the sweep stays
    done, got = oracle.undistort_cloud(cloud, R_wl, t_wl, R_we, t_we, pose_valid=False)
    assert not done and np.array_equal(got, cloud)


def test_sweep_end_pose_rules(oracle):
    """Which pose ends a sweep (lidar_mapping/LidarOdometry.cpp:206-241): the next scan's, interpolated back by the sweep's share of the scan period; the last scan
    extrapolates from the one before; a scan without a pose, or the last one when only scan 0 is behind it, is left as it is."""
    rng = np.random.default_rng(1)
    poses = [_pose(rng, 0.2, 2.0) for _ in range(6)]
    ok = [1] * 6
    for gap in (0.0, 0.05):
        R, t = oracle.sweep_end_pose(poses, ok, ok, 2, gap)
        Rw, tw = oracle.slerp_pose(*poses[2], *poses[3], 0.1 / (1 * (0.1 + float(np.float32(gap)))))
        assert np.array_equal(R, Rw) and np.array_equal(t, tw)
    assert oracle.sweep_end_pose(poses, [1, 1, 0, 1, 1, 1], ok, 2, 0.0) is None        # no pose: left as it is
    assert oracle.sweep_end_pose(poses, ok, [1, 1, 0, 1, 1, 1], 2, 0.0) is None        # not valid: left as it is
    R, t = oracle.sweep_end_pose(poses, [1, 1, 1, 0, 1, 1], [1, 1, 1, 0, 1, 1], 2, 0.0)   # the next scan has neither: the one after, two periods away
    Rw, tw = oracle.slerp_pose(*poses[2], *poses[4], 0.1 / (2 * 0.1))
    assert np.array_equal(R, Rw) and np.array_equal(t, tw)
    R, t = oracle.sweep_end_pose(poses, ok, ok, 5, 0.0)                                 # the last scan: the motion of the period before, continued
    assert R is not None
    Rw, tw = oracle.slerp_pose(*poses[4], *poses[5], 2.0)                               # ratio 1 - 0.1 / (-1 * 0.1): SlerpPose run past its second pose
    assert np.allclose(R, Rw, atol=1e-13) and np.allclose(t, tw, atol=1e-12)           # (T_i (T_i^-1 X) = X up to rounding)
    step = np.linalg.inv(_T(*poses[4])) @ _T(*poses[5])
    assert np.allclose(R, (_T(*poses[5]) @ step)[:3, :3], atol=1e-12)                   # the rotation is the last period's, continued; SlerpPose scales the
                                                                                        # translation of T_21 linearly, which is not the screw motion's
    assert oracle.sweep_end_pose(poses[:2], [1, 1], [1, 1], 1, 0.0) is None             # idx <= 0 gives up (as written upstream)


def test_host_slerp_pose_equals_the_oracle(oracle):
    """pvlm::SlerpPose (host mirror, own code over csrc/pvlm_undistort_core.h) against the oracle's: the same doubles."""
    rng = np.random.default_rng(6)
    for trial in range(20):
        R1, t1 = _pose(rng, 1.0, 3.0); R2, t2 = _pose(rng, 1.0 if trial % 4 else 1e-10, 3.0)
        ratio = float(rng.uniform(-0.5, 2.0))
        args = [repr(float(x)) for x in np.concatenate([R1.reshape(9), t1, R2.reshape(9), t2])]
        line = [l for l in host_io.run("slerp_pose", repr(ratio), *args) if l.startswith("pose")][0]
        T = np.array([float(x) for x in line.split()[1:]]).reshape(4, 4)
        R, t = oracle.slerp_pose(R1, t1, R2, t2, ratio)
        assert np.array_equal(T[:3, :3], R) and np.array_equal(T[:3, 3], t), trial
