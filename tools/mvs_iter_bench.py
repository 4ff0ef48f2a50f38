#!/usr/bin/env python3
"""Per-colour-pass time of the checkerboard PatchMatch sweep over several iterations from one start (the first iteration of a sweep sends
most pixels through the random phase, the later ones few): python tools/mvs_iter_bench.py [--iters 3]   (PVLM_LIB selects the library)"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=720)
    ap.add_argument("--cols", type=int, default=1440)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    from oracle import oracle as orc     # scene rendering only
    from tests import synth
    import panovlm_amd as pv
    n = 5
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.2 * k - 0.3, 0.01])), np.array([0.3 * k - 0.5, 0.04 * k, 0.2 * k - 0.3])) for k in range(n)]
    views = [synth.render_panorama(orc, a.rows, a.cols, R, t) for R, t in poses]
    ref = n // 2
    nei = [k for k in range(n) if k != ref]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    gray, depth, normal = views[ref]
    neis = [views[k][0] for k in nei]
    ctx = pv.Context(0)
    prng = np.random.default_rng(7)
    d0 = (depth * prng.uniform(0.9, 1.1, size=depth.shape)).astype(np.float32)
    c0, d1, n1 = ctx.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), d0, normal, 3, 1)
    state = (d1, n1, c0)
    ctx.mvs_propagate(gray, neis, np.array(Rn), np.array(tn), *state, half_window=3, step=1, max_iter=1, seed=5)
    out = {"lib": os.environ.get("PVLM_LIB", "default"), "low_conf_share_at_start": float((1 - c0[c0 > -1] >= 0.495).mean()), "iterations": []}
    for it in range(a.iters):
        ctx.profile_enable(True)
        state = ctx.mvs_propagate(gray, neis, np.array(Rn), np.array(tn), *state, half_window=3, step=1, max_iter=1, seed=5 + it)
        ms, cnt = ctx.profile_read(1)
        ctx.profile_enable(False)
        c = state[2]
        out["iterations"].append({"ms_per_colour_pass": ms / max(cnt, 1), "low_conf_share_after": float((1 - c[c > -1] >= 0.495).mean()), "checksum": float(np.float64(state[0]).sum())})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
