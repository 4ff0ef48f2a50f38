#!/usr/bin/env python3
"""Single-view SEQUENTIAL sweep only (PropagateSequential, mvs/MVS.cpp:1057-1097): ms per iteration of pvlm_mvs_propagate_sequential on a rendered
panorama, and a checksum of the maps so that launch forms (PVLM_MVS_FLOW=0: one launch per anti-diagonal; default: one persistent data-flow launch
per iteration) can be compared across processes.  python tools/mvs_seq_bench.py [--rows 720 --cols 1440] [--iters 2]"""
import argparse, hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=720); ap.add_argument("--cols", type=int, default=1440)
    ap.add_argument("--neighbors", type=int, default=4); ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--checkerboard", action="store_true", help="time the checkerboard sweep (PropagateCheckerBoard) instead")
    a = ap.parse_args()
    from oracle import oracle as orc     # scene rendering only
    from tests import synth
    import panovlm_amd as pv
    n = a.neighbors + 1
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.2 * k - 0.3, 0.01])), np.array([0.3 * k - 0.5, 0.04 * k, 0.2 * k - 0.3])) for k in range(n)]
    views = [synth.render_panorama(orc, a.rows, a.cols, R, t) for R, t in poses]
    ref = n // 2; nei = [k for k in range(n) if k != ref]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    gray, depth, normal = views[ref]; neis = [views[k][0] for k in nei]
    ctx = pv.Context(0)
    d0 = (depth * np.random.default_rng(7).uniform(0.9, 1.1, size=depth.shape)).astype(np.float32)
    c0, d1, n1 = ctx.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), d0, normal, 3, 1)
    sw = (gray, neis, np.array(Rn), np.array(tn), d1, n1, c0)
    ctx.mvs_propagate(*sw, half_window=3, step=1, max_iter=1, seed=5, sequential=not a.checkerboard)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    sq = ctx.mvs_propagate(*sw, half_window=3, step=1, max_iter=a.iters, seed=5, sequential=not a.checkerboard)
    wall = time.perf_counter() - t0
    ms, cnt = ctx.profile_read(1)          # checkerboard: one interval per colour pass
    ctx.profile_enable(False)
    h = hashlib.sha256(); [h.update(np.ascontiguousarray(x).tobytes()) for x in sq[:3]]
    print(json.dumps(dict(sweep="checkerboard" if a.checkerboard else "sequential", rows=a.rows, cols=a.cols, flow=os.environ.get("PVLM_MVS_FLOW", "1"), sleep=os.environ.get("PVLM_LIB", "base"), ms_per_iteration=ms / max(cnt, 1),
                          iterations=int(cnt), wall_ms=wall * 1e3, sha256=h.hexdigest()[:16], diags=os.environ.get("PVLM_MVS_FLOW_DIAGS", "2.5"))))


if __name__ == "__main__":
    main()
