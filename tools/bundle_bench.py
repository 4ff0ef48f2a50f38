#!/usr/bin/env python3
"""Room-scale timing of the reprojection kernels (K9): F cameras on a trajectory, M points each seen by a
window of consecutive cameras (what SIFT tracks look like), N = sum of track lengths observations.
Prints wall times of pvlm_ba_reduce / step / cost (each includes its small host<->device copies)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=454)          # Room: 454 frames
    ap.add_argument("--points", type=int, default=200_000)
    ap.add_argument("--min-track", type=int, default=3)
    ap.add_argument("--max-track", type=int, default=9)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import panovlm_amd as pv
    rng = np.random.default_rng(1)
    F, M = args.cams, args.points
    aa = rng.normal(size=(F, 3)) * 0.05
    t = np.stack([np.zeros(F), np.zeros(F), -0.1 * np.arange(F)], axis=1) + rng.normal(size=(F, 3)) * 0.01
    k = rng.integers(args.min_track, args.max_track + 1, size=M)
    first = rng.integers(0, np.maximum(F - k, 1))
    off = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    cam = np.concatenate([np.arange(f, f + kk) for f, kk in zip(first, k)]).astype(np.int32)
    # points around the cameras that see them; bearings = true directions + noise
    centre = -t[np.minimum(first + k // 2, F - 1)]             # camera centres ~ -t for small rotations
    X = centre + rng.normal(size=(M, 3)) * np.array([2.0, 1.0, 2.0])
    pt = np.repeat(np.arange(M), k)
    pc = X[pt] + t[cam]                                        # R ~ I
    bearing = pc / np.linalg.norm(pc, axis=1, keepdims=True) + rng.normal(size=pc.shape) * 2e-3
    ctx = pv.Context(0)
    ctx.set_poses(aa, t)
    t0 = time.perf_counter()
    bs = pv.BundleSet(ctx, off, cam, bearing, X, weight=1.0)
    t_create = time.perf_counter() - t0
    a = 4.0 * np.pi / 180.0
    bs.reduce(pv.LOSS_HUBER, a, init_scale=True)
    res = {}
    for name, fn in (("reduce", lambda: bs.reduce(pv.LOSS_HUBER, a, radius=1e3)),
                     ("step", lambda: bs.step(np.full((bs.n_cams, 6), 1e-4), pv.LOSS_HUBER, a)),
                     ("cost", lambda: bs.cost(pv.LOSS_HUBER, a, candidate=True))):
        fn()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        res[name + "_ms"] = (time.perf_counter() - t0) / args.reps * 1e3
    print(json.dumps(dict(cams=F, points=M, observations=int(len(cam)), covisible_pairs=int(bs.n_upairs), packed_doubles=bs.size,
                          create_s=t_create, **res)))


if __name__ == "__main__":
    main()
