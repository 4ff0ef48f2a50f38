#!/usr/bin/env python3
"""Room-like end-to-end run of the mirrored CameraLidarOptimizer::JointOptimize (BASELINE.json configs[2] shape):
F LiDAR scans (16 x 1800, <= 384 surfFlat queries, voxel targets, room-edge line segments) + F panoramas (5760 x 2880,
image lines = projections of the room edges, SIFT-like keypoints of triangulated tracks).  All three terms of Optimize:
camera-LiDAR line pairs, SfM reprojection with free 3-D points, LiDAR-LiDAR point-to-plane.  Prints wall time and the
stage times of the C++ driver.   usage: python tools/room_like_joint.py [--frames 64] [--points 20000] [--iters 2]"""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panovlm_amd import synthetic as sy
from tests import host_io, lm_twin
from tools.room_like_odometry import room_edges, room_scan


def cam_to_image(rows, cols, p):
    """plain numpy equirect (true atan2): keypoints / image lines only need to be plausible here"""
    lon = np.arctan2(p[:, 0], p[:, 2]); lat = -np.arctan2(p[:, 1], np.hypot(p[:, 0], p[:, 2]))
    return np.stack([cols * (0.5 + lon / (2 * np.pi)), rows * (0.5 - lat / np.pi)], axis=1)


def _stage_line(l):
    """'stage <seconds> <label> [<n> calls]' of the driver -> seconds, label, ms per call"""
    import re
    m = re.match(r"stage (\S+) (.*) \[(\d+) calls\]$", l)
    if not m:
        return "   stage %8.3f s  %s" % (float(l.split()[1]), " ".join(l.split()[2:]))
    sec, label, n = float(m.group(1)), m.group(2), int(m.group(3))
    return "   stage %8.3f s  %s  [%d calls, %.3f ms each]" % (sec, label, n, sec / max(n, 1) * 1e3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--keep", default="", help="write the scene files here and print the driver command (e.g. to run it under rocprofv3)")
    ap.add_argument("--repeat", type=int, default=1, help="run the driver this many times on the same files and compare the results bit for bit")
    a = ap.parse_args()
    rng = np.random.default_rng(2)
    rows, cols = 2880, 5760
    edges = room_edges()
    ang = np.deg2rad(np.array([1.0, -2.0, 0.5])); K = np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]])
    th = np.linalg.norm(ang); Rc = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
    T_cl = np.eye(4); T_cl[:3, :3] = Rc; T_cl[:3, 3] = [0.03, -0.02, 0.05]
    F = a.frames
    lidars = [room_scan(k, rng, edges) for k in range(F)]
    frames, true = [], []
    for k in range(F):
        R_true, t_true = sy.true_pose(k)
        T_wc = lm_twin.pose4(R_true, t_true) @ np.linalg.inv(T_cl)
        true.append((T_wc[:3, :3].copy(), T_wc[:3, 3].copy()))
        T_cw = np.linalg.inv(T_wc)
        ends = np.array([np.concatenate([T_cw[:3, :3] @ p + T_cw[:3, 3], T_cw[:3, :3] @ q + T_cw[:3, 3]]) for p, q in edges]).reshape(-1, 3)
        px = cam_to_image(rows, cols, ends).reshape(-1, 4).astype(np.float32) + rng.normal(size=(len(edges), 4)).astype(np.float32)
        T_est = lm_twin.pose4(lidars[k]["R_wl"], lidars[k]["t_wl"]) @ np.linalg.inv(T_cl)
        frames.append(dict(id=k, rows=rows, cols=cols, valid=1, R_wc=T_est[:3, :3].copy(), t_wc=T_est[:3, 3].copy(), lines=px, keypoints=[]))
    # tracks: points on the room walls, each seen by a window of 4..8 consecutive frames
    M = a.points
    X = rng.uniform([-4, -1.5, -6], [4, 1.5, 6], size=(M, 3)); wall = rng.integers(0, 3, size=M)
    X[np.arange(M), wall] = np.sign(X[np.arange(M), wall]) * np.array([4.0, 1.5, 6.0])[wall]
    tracks = []
    for p in range(M):
        k = int(rng.integers(4, 9)); f0 = int(rng.integers(0, max(F - k, 1)))
        obs = []
        for fi in range(f0, min(f0 + k, F)):
            R, t = true[fi]
            pc = R.T @ (X[p] - t)
            frames[fi]["keypoints"].append((cam_to_image(rows, cols, pc[None])[0] + rng.normal(size=2) * 0.5).astype(np.float32))
            obs.append((fi, len(frames[fi]["keypoints"]) - 1))
        tracks.append(dict(point=X[p] + rng.normal(size=3) * 0.03, obs=obs))
    n_obs = sum(len(t["obs"]) for t in tracks)
    with tempfile.TemporaryDirectory() as d:
        if a.keep:
            d = a.keep; os.makedirs(d, exist_ok=True)
        lp, fp, sp = (os.path.join(d, n) for n in ("l.bin", "f.bin", "s.bin"))
        host_io.write_scans(lp, lidars, world=False); host_io.write_frames(fp, T_cl, frames); host_io.write_structure(sp, frames, tracks)
        if a.keep:
            print("driver command:", host_io.driver(), "joint", lp, fp, 3, a.iters, 0, 1, 0.05, 1.0, 0.3, 0.01, 25.0, 1.0, sp)
        t0 = time.perf_counter()
        out = host_io.run("joint", lp, fp, 3, a.iters, 0, 1, 0.05, 1.0, 0.3, 0.01, 25.0, 1.0, sp, timeout=3000)   # Room weights (config/Room.txt:81-83)
        wall = time.perf_counter() - t0
        results = lambda o: [l for l in o if l.startswith(("iter", "pose", "fpose", "point"))]
        same = all(results(host_io.run("joint", lp, fp, 3, a.iters, 0, 1, 0.05, 1.0, 0.3, 0.01, 25.0, 1.0, sp, timeout=3000)) == results(out) for _ in range(a.repeat - 1))
        if a.repeat > 1:
            print("reproducible over %d runs (costs, step counts, every pose and point): %s" % (a.repeat, same))
    print("GPU JointOptimize: %d frames + %d scans, %d tracks / %d observations, %.2f s wall" % (F, F, M, n_obs, wall))
    for l in out:
        if l.startswith("iter"):
            print("  ", l)
        if l.startswith("stage"):
            print(_stage_line(l))
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    e0 = np.mean([np.linalg.norm(lidars[k]["t_wl"] - sy.true_pose(k)[1]) for k in range(1, F)])
    e1 = np.mean([np.linalg.norm(poses[k][9:] - sy.true_pose(k)[1]) for k in range(1, F)])
    print("mean LiDAR translation error vs ground truth: %.4f m -> %.4f m" % (e0, e1))


if __name__ == "__main__":
    main()
