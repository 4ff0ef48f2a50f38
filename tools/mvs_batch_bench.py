#!/usr/bin/env python3
"""The sequential PatchMatch sweep (Room / Floor strategy) of MANY resident views per launch: ms per view and iteration against the
number of views in the batch.  The per-diagonal form is chosen by PVLM_MVS_LANE_BATCH_MIN (pixels on a diagonal over all jobs from
which the thread-per-pixel kernel is used; a huge value = always one wave per pixel): run once per setting, the variable is read once.
usage: python tools/mvs_batch_bench.py [--views 8,32,64,128] [--rows 720 --cols 1440]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=720)
    ap.add_argument("--cols", type=int, default=1440)
    ap.add_argument("--neighbors", type=int, default=4)
    ap.add_argument("--views", default="8,32,64,128")
    ap.add_argument("--half-window", type=int, default=3)
    a = ap.parse_args()
    from oracle import oracle as orc     # scene rendering only (a measurement tool, not the product)
    from tests import synth
    import panovlm_amd as pv
    from panovlm_amd.api import MvsViews
    n = a.neighbors + 1
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.2 * k - 0.3, 0.01])), np.array([0.3 * k - 0.5, 0.04 * k, 0.2 * k - 0.3])) for k in range(n)]
    views = [synth.render_panorama(orc, a.rows, a.cols, R, t) for R, t in poses]
    ref = n // 2
    nei = [k for k in range(n) if k != ref]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    gray, depth, normal = views[ref]
    neis = [views[k][0] for k in nei]
    nd = [views[k][1] for k in nei]
    ctx = pv.Context(0)
    prng = np.random.default_rng(7)
    d0 = (depth * prng.uniform(0.9, 1.1, size=depth.shape)).astype(np.float32)
    c0, d1, n1 = ctx.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), d0, normal, a.half_window, 1)
    out = {"rows": a.rows, "cols": a.cols, "neighbors": a.neighbors, "lane_batch_min": os.environ.get("PVLM_MVS_LANE_BATCH_MIN", "default"), "points": []}
    first = None
    for B in [int(x) for x in a.views.split(",")]:
        V = MvsViews(ctx, a.rows, a.cols, B + len(neis))
        for k, g in enumerate(neis):
            V.upload(B + k, gray=g, depth=nd[k], normal=np.zeros((a.rows, a.cols, 3), np.float32), conf=np.zeros((a.rows, a.cols), np.float32))
        for k in range(B):
            V.upload(k, gray=gray, depth=d1, normal=n1, conf=c0)
        jobs = [dict(ref=k, nei=list(range(B, B + len(neis))), R_nr=np.array(Rn), t_nr=np.array(tn), seed=5 + k) for k in range(B)]
        ctx.profile_enable(True)
        V.estimate_sequential_batch(jobs, half_window=a.half_window, step=1, max_iter=1)
        ms, cnt = ctx.profile_read(1)
        ctx.profile_enable(False)
        got = V.download(0, ("depth", "conf"))
        if first is None:
            first = got
        same = bool(np.array_equal(got["depth"], first["depth"]) and np.array_equal(got["conf"], first["conf"]))
        out["points"].append({"views": B, "ms_per_iteration": ms / max(cnt, 1), "ms_per_view_iteration": ms / max(cnt, 1) / B, "view0_equals_first_batch": same,
                              "view0_depth_checksum": float(np.float64(got["depth"]).sum())})
        V.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
