#!/usr/bin/env python3
"""Timing of the panoramic MVS scoring pass (pvlm_mvs_init_conf_map, kernel k_mvs_conf) at the Room MVS size (5.7K
panorama at scale -2 = 1440 x 720, 7 x 7 NCC window, 4 neighbours) beside the CPU oracle (OpenMP) on the same inputs."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=720)
    ap.add_argument("--cols", type=int, default=1440)
    ap.add_argument("--neighbors", type=int, default=4)
    ap.add_argument("--half-window", type=int, default=3)
    ap.add_argument("--step", type=int, default=1)
    a = ap.parse_args()
    from oracle import oracle as orc     # scene rendering + the CPU leg (this is a measurement tool, not the product)
    from tests import synth
    import panovlm_amd as pv
    n = a.neighbors + 1
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.2 * k - 0.3, 0.01])), np.array([0.3 * k - 0.5, 0.04 * k, 0.2 * k - 0.3])) for k in range(n)]
    views = [synth.render_panorama(orc, a.rows, a.cols, R, t) for R, t in poses]
    ref = n // 2
    nei = [k for k in range(n) if k != ref]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    gray, depth, normal = views[ref]
    neis = [views[k][0] for k in nei]
    ctx = pv.Context(0)
    ctx.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), depth, normal, a.half_window, a.step)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        cg, _, _ = ctx.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), depth, normal, a.half_window, a.step)
    wall = (time.perf_counter() - t0) / reps
    ms, cnt = ctx.profile_read(1)
    ctx.profile_enable(False)
    t0 = time.perf_counter()
    co, _, _ = orc.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), depth, normal, a.half_window, a.step)
    cpu = time.perf_counter() - t0
    # depth fusion filter (K12) on the same views: neighbours' true depth maps against the reference's
    nd = [views[k][1] for k in nei]
    ctx.mvs_filter_depth(nd, np.array(Rn), np.array(tn), depth, conf=cg, thr=0.01)
    t0 = time.perf_counter()
    for _ in range(reps):
        dfg, _ = ctx.mvs_filter_depth(nd, np.array(Rn), np.array(tn), depth, conf=cg, thr=0.01)
    filt_wall = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    dfo, _ = orc.mvs_filter_depth(nd, np.array(Rn), np.array(tn), depth, conf=co, thr=0.01)
    filt_cpu = time.perf_counter() - t0
    # the filter the pipeline runs (FilterDepthImageRefine): confidences of the scoring pass, negatives clamped (ConvertNCC2Conf)
    nc = [np.clip(orc.mvs_init_conf_map(views[k][0], [gray], *[np.array(x)[None] for x in synth.relative_pose(poses[k][0], poses[k][1], poses[ref][0], poses[ref][1])],
                                        views[k][1], views[k][2], a.half_window, a.step)[0], 0, None) for k in nei]
    cref = np.clip(cg, 0, None)
    ctx.mvs_filter_depth_refine(nd, nc, np.array(Rn), np.array(tn), depth, cref)
    t0 = time.perf_counter()
    for _ in range(reps):
        rdg, rcg, _ = ctx.mvs_filter_depth_refine(nd, nc, np.array(Rn), np.array(tn), depth, cref)
    ref_wall = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    rdo, rco, _ = orc.mvs_filter_depth_refine(nd, nc, np.array(Rn), np.array(tn), depth, cref)
    ref_cpu = time.perf_counter() - t0
    # PatchMatch sweep (K13): one checkerboard iteration from a perturbed state, GPU beside the oracle (OpenMP)
    prng = np.random.default_rng(7)
    d0 = (depth * prng.uniform(0.9, 1.1, size=depth.shape)).astype(np.float32)
    c0, d1, n1 = ctx.mvs_init_conf_map(gray, neis, np.array(Rn), np.array(tn), d0, normal, a.half_window, a.step)
    sw = (gray, neis, np.array(Rn), np.array(tn), d1, n1, c0)
    ctx.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    sg = ctx.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5)
    sweep_wall = time.perf_counter() - t0
    sweep_ms, sweep_cnt = ctx.profile_read(1)
    ctx.profile_enable(False)
    t0 = time.perf_counter()
    so = orc.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5)
    sweep_cpu = time.perf_counter() - t0
    # the same iteration with the geometric-consistency term (use_geometry: the neighbours' depth maps; the call at mvs/MVS.cpp:137)
    ctx.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5, nei_depths=nd)
    ctx.profile_enable(True)
    gg = ctx.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5, nei_depths=nd)
    geo_ms, geo_cnt = ctx.profile_read(1)
    ctx.profile_enable(False)
    go = orc.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5, nei_depths=nd)
    geo_same = bool(np.array_equal(gg[0], go[0]) and np.array_equal(gg[1], go[1]) and np.array_equal(gg[2], go[2]))
    # the SEQUENTIAL sweep of the Room / Floor configs: one launch per anti-diagonal (rows + cols - 1 per iteration); the oracle's
    # raster walk is single-threaded (minutes at this size), parity is pinned at 96 x 192 (tests/test_mvs_gpu.py)
    ctx.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=1, seed=5, sequential=True)
    ctx.profile_enable(True)
    sq = ctx.mvs_propagate(*sw, half_window=a.half_window, step=a.step, max_iter=2, seed=5, sequential=True)
    seq_ms, seq_cnt = ctx.profile_read(1)
    ctx.profile_enable(False)
    # ... and eight views per launch through the resident view set (upstream: one image per thread for this strategy)
    from panovlm_amd.api import MvsViews
    B = 8
    V = MvsViews(ctx, a.rows, a.cols, B + len(neis))
    for k, g in enumerate(neis):
        V.upload(B + k, gray=g, depth=nd[k], normal=np.zeros((a.rows, a.cols, 3), np.float32), conf=np.zeros((a.rows, a.cols), np.float32))
    for k in range(B):
        V.upload(k, gray=gray, depth=d1, normal=n1, conf=c0)
    jobs = [dict(ref=k, nei=list(range(B, B + len(neis))), R_nr=np.array(Rn), t_nr=np.array(tn), seed=5 + k) for k in range(B)]
    V.estimate_sequential_batch(jobs, half_window=a.half_window, step=a.step, max_iter=1)
    for k in range(B):
        V.upload(k, depth=d1, normal=n1, conf=c0)
    ctx.profile_enable(True)
    V.estimate_sequential_batch(jobs, half_window=a.half_window, step=a.step, max_iter=2)
    bat_ms, bat_cnt = ctx.profile_read(1)
    ctx.profile_enable(False)
    bat0 = V.download(0, ("depth", "normal", "conf"))
    batch_equals_single = bool(np.array_equal(bat0["depth"], sq[0]) and np.array_equal(bat0["conf"], sq[2]))
    V.close()
    # depth map -> coloured world points (MVS::DepthImageToCloud, the body of MergeDepthImages): count + scan + emit
    crng = np.random.default_rng(11)
    bgr = np.repeat(gray[..., None], 3, axis=2); bgr[..., 0] = np.minimum(255, bgr[..., 0].astype(np.int32) + crng.integers(0, 90, size=gray.shape)).astype(np.uint8)
    Twc = np.eye(4); Twc[:3, 3] = [0.5, 0.1, -2.0]
    ctx.mvs_depth_to_cloud(depth, bgr, Twc, 20.0)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    cl = ctx.mvs_depth_to_cloud(depth, bgr, Twc, 20.0)
    cloud_wall = time.perf_counter() - t0
    cloud_ms, cloud_cnt = ctx.profile_read(1)
    ctx.profile_enable(False)
    t0 = time.perf_counter()
    clo = orc.mvs_depth_to_cloud(depth, bgr, Twc, 20.0)
    cloud_cpu = time.perf_counter() - t0
    vs = c0 > -1
    same = (np.abs(sg[0] - so[0]) <= 1e-4 * np.maximum(np.abs(so[0]), 1e-3)) & (np.abs(sg[1] - so[1]).max(axis=2) <= 1e-4) & (np.abs(sg[2] - so[2]) <= 1e-4)
    rel = lambda d: float(np.median(np.abs(d[vs] / depth[vs] - 1)))
    w = 2 * a.half_window + 1; q = w // a.step + (1 if a.step > 1 else 0)
    texels = a.rows * a.cols * q * q * a.neighbors
    k_ms = ms / max(cnt, 1)
    print(json.dumps(dict(rows=a.rows, cols=a.cols, neighbors=a.neighbors, window=[w, a.step], valid=float((cg > -1).mean()),
                          max_abs_diff_vs_oracle=float(np.abs(cg - co)[(cg > -1) & (co > -1)].max()), kernel_ms=k_ms,
                          M_pixels_per_s=a.rows * a.cols / k_ms / 1e3, G_texel_projections_per_s=texels / k_ms / 1e6,
                          wall_ms_incl_copies=wall * 1e3, cpu_oracle_s=cpu,
                          filter=dict(wall_ms_incl_copies=filt_wall * 1e3, kept=float((dfg > 0).mean()), identical_to_oracle=bool(np.array_equal(dfg, dfo)),
                                      cpu_oracle_s_single_thread=filt_cpu),
                          filter_refine=dict(wall_ms_incl_copies=ref_wall * 1e3, kept=float((rdg > 0).mean()),
                                             identical_to_oracle=bool(np.array_equal(rdg, rdo) and np.array_equal(rcg, rco)), cpu_oracle_s_single_thread=ref_cpu),
                          sweep=dict(kernel_ms_per_colour_pass=sweep_ms / max(sweep_cnt, 1), colour_passes=int(sweep_cnt), wall_ms_one_iteration_incl_copies=sweep_wall * 1e3,
                                     cpu_oracle_s=sweep_cpu, agree_with_oracle=float(same[vs].mean()), mean_conf_gpu=float(sg[2][vs].mean()), mean_conf_oracle=float(so[2][vs].mean()),
                                     depth_err_before=rel(d1), depth_err_gpu=rel(sg[0]), depth_err_oracle=rel(so[0])),
                          sweep_sequential=dict(ms_per_iteration=seq_ms / max(seq_cnt, 1), launches_per_iteration=a.rows + a.cols - 1,
                                                depth_err=rel(sq[0]), mean_conf=float(sq[2][vs].mean()), batch_views=B,
                                                batch_ms_per_iteration=bat_ms / max(bat_cnt, 1), batch_ms_per_view_iteration=bat_ms / max(bat_cnt, 1) / B,
                                                batch_view0_equals_single_call=batch_equals_single),
                          depth_to_cloud=dict(kernels_ms=cloud_ms / max(cloud_cnt, 1), wall_ms_incl_copies=cloud_wall * 1e3, points=int(len(cl[0])),
                                              identical_to_oracle=bool(np.array_equal(cl[0], clo[0]) and np.array_equal(cl[1], clo[1])), cpu_oracle_s_single_thread=cloud_cpu),
                          sweep_geometric=dict(kernel_ms_per_colour_pass=geo_ms / max(geo_cnt, 1), identical_to_oracle=geo_same),
                          cpu_threads=orc.num_threads(), speedup_kernel_vs_cpu=cpu / (k_ms * 1e-3))))


if __name__ == "__main__":
    main()
