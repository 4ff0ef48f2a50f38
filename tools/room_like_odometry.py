#!/usr/bin/env python3
"""Room-like end-to-end run of the mirrored LidarOdometry::EstimatePose on the GPU (BASELINE.json configs[0]/[3]
shape: 16 x 1800 scans, <= 384 surfFlat queries per scan, 0.2 m voxel surfLessFlat targets, point-to-plane
Angle residual, <= 7 outer iterations).  Reports wall time per stage of the C++ driver and the pose error
before / after.  With --twin F' the first F' scans are also solved by the CPU oracle twin for a timing beside it.
usage: python tools/room_like_odometry.py [--scans 64] [--iters 7] [--twin 0]"""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panovlm_amd import synthetic as sy
from tests import host_io, lm_twin


def room_edges():
    """The 12 edges of the 8 x 3 x 12 m synthetic room (panovlm_amd/synthetic.py) + a few interior verticals:
    the straight structures the reference's edge extractor would turn into line segments."""
    x, y, z = 4.0, 1.5, 6.0
    c = [np.array([sx * x, sy_ * y, sz * z]) for sx in (-1, 1) for sy_ in (-1, 1) for sz in (-1, 1)]
    edges = [(a, b) for i, a in enumerate(c) for b in c[i + 1:] if np.count_nonzero(a != b) == 1]
    for px, pz in ((-2.0, 1.0), (1.5, -2.5), (2.5, 3.0), (-1.0, -4.0)):
        edges.append((np.array([px, -y, pz]), np.array([px, y, pz])))
    return edges


def room_scan(k, rng, lines=None):
    s = sy.make_scan(k, cols=1800, downsample_targets=0.2)
    R, t = s["R_wl"], s["t_wl"]
    Rl, tl = lm_twin.inv_pose(R, t)
    less_local = lm_twin.transform_f32(s["less_xyz"], Rl, tl)
    sel = np.sort(rng.choice(len(s["local_xyz"]), size=384, replace=False))   # 4 per sector x 6 sectors x 16 rings
    flat = s["local_xyz"][sel]
    out = dict(id=k, R_wl=R, t_wl=t, flat_local=flat, flat_tag=np.ones(len(flat), np.float32), less_local=less_local,
               less_tag=np.ones(len(less_local), np.float32))
    if lines:
        from tests import synth
        Rt, tt = sy.true_pose(k)
        ls = synth.make_line_scan(rng, k, Rt, tt, lines, pts_per_line=(30, 70), extra_pts=100, noise=0.01, shared_frac=0.0)   # sampled at the TRUE pose
        out.update(corner_local=ls["corner_local"], p2s=ls["p2s"], seg_coeffs=ls["seg_coeffs"], end_points=ls["end_points"],
                   seg_points=[[i for i, l in enumerate(ls["p2s"]) if sid in l] for sid in range(len(ls["seg_size"]))])
    return out


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
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--iters", type=int, default=7)
    ap.add_argument("--twin", type=int, default=0)
    ap.add_argument("--lines", type=int, default=0, help="1: add line features (room edges) and the line-to-line term (Room config)")
    ap.add_argument("--repeat", type=int, default=1, help="run the driver this many times on the same file and compare the results bit for bit")
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    edges = room_edges() if a.lines else None
    scans = [room_scan(k, rng, edges) for k in range(a.scans)]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "room.bin")
        host_io.write_scans(path, scans, world=False)
        os.environ.setdefault("PVLM_HOST_RESERVE_MB", "1536")       # the engine's pool, sized once at context creation
        os.environ.setdefault("PVLM_HOST_RESERVE_STAGING_MB", "64")  # ... and the pinned staging window of the scan uploads
        os.environ.setdefault("PVLM_HOST_PRELOAD", "1")             # ... and the code objects of the kernels (pvlm_preload)
        t0 = time.perf_counter()
        out = host_io.run("odometry", path, a.iters, 1, 1, 1 if a.lines else 0, 1, 0.05, 1.0, 0.3, timeout=3000)
        wall = time.perf_counter() - t0
        results = lambda o: [l for l in o if l.startswith(("iter", "pose"))]
        same = all(results(host_io.run("odometry", path, a.iters, 1, 1, 1 if a.lines else 0, 1, 0.05, 1.0, 0.3, timeout=3000)) == results(out) for _ in range(a.repeat - 1))
        if a.repeat > 1:
            print("reproducible over %d runs (costs, step counts, every pose): %s" % (a.repeat, same))
    iters = [l for l in out if l.startswith("iter")]
    poses = {int(l.split()[1]): np.array([float(v) for v in l.split()[2:]]) for l in out if l.startswith("pose")}
    e0 = np.mean([np.linalg.norm(scans[k]["t_wl"] - sy.true_pose(k)[1]) for k in range(1, a.scans)])
    e1 = np.mean([np.linalg.norm(poses[k][9:] - sy.true_pose(k)[1]) for k in range(1, a.scans)])
    print("GPU EstimatePose: %d scans, %d outer iterations, %.2f s wall (process start, upload, association, LM, write-back)" % (a.scans, len(iters), wall))
    for l in iters:
        print("  ", l)
    for l in out:
        if l.startswith("call"):
            print("   call  %8.3f s  %s" % (float(l.split()[1]), " ".join(l.split()[2:])))
    for l in out:
        if l.startswith("stage"):
            print(_stage_line(l))
    print("mean translation error vs ground truth: %.4f m -> %.4f m" % (e0, e1))
    if a.twin > 0:
        from oracle import oracle as orc
        tw = [dict(s) for s in scans[:a.twin]]
        t0 = time.perf_counter()
        log = lm_twin.estimate_pose(orc, tw, dict(angle=True, normalize=True, tol=0.05, thr=1.0), a.iters)
        print("CPU oracle twin: %d scans, %d outer iterations, %.2f s" % (a.twin, len(log), time.perf_counter() - t0))


if __name__ == "__main__":
    main()
