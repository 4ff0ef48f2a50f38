#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory (run from the repo root in the development
container: `python tests/golden/make_golden.py`).

  fast_atan2.npz   inputs + outputs of the REAL reference base/Math.h::FastAtan2 (float and double),
                   produced by oracle/_ref/libref_math.so, i.e. compiled from /root/reference itself.
  everything else  inputs + outputs of the CPU oracle (the restated reference algorithm; the reference
                   has no tests / vectors of its own and cannot be built here — "parity unpinned").
The fixtures are data only (numpy arrays): seeded inputs and expected outputs.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from panovlm_amd import synthetic as sy  # noqa: E402
from tests import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def fast_atan2():
    ref = orc.ref_math()
    assert ref is not None, "oracle/_ref/libref_math.so missing: run `make -C oracle` with /root/reference present"
    rng = np.random.default_rng(1)
    y = np.concatenate([rng.normal(size=3000) * 3, [0, 0, 1, -1, 0, 1e-30, -1e-30, 5, -5, 1, 1, -1, -1]])
    x = np.concatenate([rng.normal(size=3000) * 3, [0, 1, 0, 0, -1, 1e-30, 1e-30, 5, 5, 1e-9, -1e-9, 1e-9, -1e-9]])
    yf, xf = y.astype(np.float32), x.astype(np.float32)
    of = np.empty_like(yf); od = np.empty_like(y)
    ref.ref_fast_atan2_f_vec(C.c_long(len(yf)), yf.ctypes.data_as(C.POINTER(C.c_float)), xf.ctypes.data_as(C.POINTER(C.c_float)), of.ctypes.data_as(C.POINTER(C.c_float)))
    ref.ref_fast_atan2_d_vec(C.c_long(len(y)), y.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)), od.ctypes.data_as(C.POINTER(C.c_double)))
    np.savez_compressed(os.path.join(OUT, "fast_atan2.npz"), y=y, x=x, out_f32=of, out_f64=od)


def functors():
    d = {}
    for kind, normalize in [(0, False), (1, False), (1, True), (2, False), (3, False), (3, True), (4, False), (5, False)]:
        rng = np.random.default_rng(1000 + kind * 2 + int(normalize))
        F, P = 6, 16
        aa, t = synth.random_poses(rng, F)
        ref, nei = synth.random_pairs(rng, F, P)
        counts = rng.integers(0, 24, size=P)
        rows, off = synth.random_resset(rng, kind, aa, t, ref, nei, counts)
        rid, nid = synth.expand_ids(off, ref, nei)
        w = 1.3
        r, J = orc.evaluate(kind, synth.oracle_rows(kind, rows, w), rid, nid, aa, t, normalize=normalize)
        # the same statements in x87 extended precision: the exact value of the reference's formula (arbiter of the GPU test)
        rx, Jx = orc.evaluate(kind, synth.oracle_rows(kind, rows, w), rid, nid, aa, t, normalize=normalize, extended=True)
        k = "k%d_n%d_" % (kind, int(normalize))
        d.update({k + "aa": aa, k + "t": t, k + "ref": ref, k + "nei": nei, k + "off": off, k + "rows": rows, k + "r": r, k + "J": J,
                  k + "r_ext": rx, k + "J_ext": Jx})
    d["weight"] = np.array(1.3)
    np.savez_compressed(os.path.join(OUT, "functors.npz"), **d)


def _scan_arrays(s, keys=("R_wl", "t_wl", "flat_xyz", "flat_tag", "less_xyz", "less_tag")):
    return {k: np.asarray(s[k]) for k in keys}


def assoc():
    d = {}
    scans = {k: sy.make_scan(k, cols=128) for k in (0, 1)}
    scans[2] = sy.make_scan(2, cols=256, downsample_targets=0.2)
    for k, s in scans.items():
        for kk, v in _scan_arrays(s).items():
            d["s%d_%s" % (k, kk)] = v
    cases = [(0, 1, 0.05, 1.0), (1, 0, 0.01, 1.0), (2, 1, 0.05, 1.0), (1, 2, 0.05, 0.5)]
    d["cases"] = np.array(cases)
    for i, (r, n, tol, thr) in enumerate(cases):
        o = orc.assoc_point2plane(scans[r], scans[n], tol, thr)
        for kk in ("point", "plane", "qidx", "nn"):
            d["c%d_%s" % (i, kk)] = o[kk]
    np.savez_compressed(os.path.join(OUT, "assoc_point2plane.npz"), **d)


def equirect():
    rng = np.random.default_rng(5)
    cam = rng.normal(size=(4000, 3)) * np.array([3.0, 1.0, 3.0])
    cam[:20, 0] = 0; cam[20:40, 2] = 0; cam[40:60, 1] = 0
    d = dict(cam=cam)
    for rows, cols in [(2880, 5760), (720, 1440)]:
        d["px_f32_%d" % rows] = orc.cam_to_image(rows, cols, cam.astype(np.float32))
        d["px_f64_%d" % rows] = orc.cam_to_image(rows, cols, cam)
        px = rng.uniform([0, 0], [cols, rows], size=(500, 2))
        d["pix_%d" % rows] = px
        d["cam_f64_%d" % rows] = orc.image_to_cam(rows, cols, px, 1.0)
        d["seg_%d" % rows] = orc.break_to_segments(rows, cols, [100.0, 200.0], [cols - 150.0, rows - 300.0], 100.0)
    np.savez_compressed(os.path.join(OUT, "equirect.npz"), **d)


def _line_scan_arrays(s):
    off = np.zeros(len(s["p2s"]) + 1, np.int32)
    for i, l in enumerate(s["p2s"]):
        off[i + 1] = off[i] + len(l)
    ids = np.array([v for l in s["p2s"] for v in l], np.int32)
    return dict(R_wl=s["R_wl"], t_wl=s["t_wl"], corner_xyz=s["corner_xyz"], corner_local=s["corner_local"], p2s_off=off, p2s_ids=ids,
                seg_size=s["seg_size"], seg_coeffs=s["seg_coeffs"], end_points=s["end_points"])


def lines():
    rng = np.random.default_rng(21)
    lw = synth.random_world_lines(rng, 10)
    Ra, ta = sy.estimated_pose(3); Rb, tb = sy.estimated_pose(4)
    a = synth.make_line_scan(rng, 3, Ra, ta, lw[:8], pts_per_line=(6, 20), extra_pts=10)
    b = synth.make_line_scan(rng, 4, Rb, tb, lw[2:], pts_per_line=(6, 20), extra_pts=10)
    d = {}
    for name, s in (("a", a), ("b", b)):
        for k, v in _line_scan_arrays(s).items():
            d[name + "_" + k] = v
    for thr in (0.3, 0.4):
        o = orc.assoc_line2line(a, b, thr)
        t = "t%02d_" % int(thr * 10)
        d.update({t + "votes": o["votes"], t + "nei_idx": o["nei_idx"], t + "ref_idx": o["ref_idx"], t + "p1": o["p1"], t + "p2": o["p2"]})
    # camera <-> LiDAR by angle
    rows, cols = 2880, 5760
    rng = np.random.default_rng(8)
    lw = synth.random_world_lines(rng, 8, extent=3.0)
    scan = synth.make_line_scan(rng, 0, np.eye(3), np.zeros(3), lw, pts_per_line=(20, 40), extra_pts=30)
    local = dict(scan); local["corner_xyz"] = scan["corner_local"]
    ang = np.deg2rad(rng.uniform(-2, 2, size=3))
    T = np.eye(4); T[:3, :3] = synth.rodrigues(ang); T[:3, 3] = rng.uniform(-0.05, 0.05, size=3)
    ends_cam = scan["end_points"].reshape(-1, 3) @ T[:3, :3].T + T[:3, 3]
    px = orc.cam_to_image(rows, cols, ends_cam).reshape(-1, 4).astype(np.float32)
    px += rng.normal(size=px.shape).astype(np.float32) * 2.0
    img_lines = np.concatenate([px, rng.uniform([0, 0, 0, 0], [cols, rows, cols, rows], size=(4, 4)).astype(np.float32)])
    for k, v in _line_scan_arrays(scan).items():
        d["c_" + k] = v
    d["c_T_cl"] = T; d["c_lines"] = img_lines
    for mult in (1, 0):
        o = orc.assoc_by_angle(rows, cols, img_lines, local, T, multiple=bool(mult))
        m = "m%d_" % mult
        d.update({m + "image_line_id": o["image_line_id"], m + "lidar_line_id": o["lidar_line_id"], m + "score": o["score"],
                  m + "start": o["start"], m + "end": o["end"]})
    d["c_votes"] = o["votes"]
    np.savez_compressed(os.path.join(OUT, "lines.npz"), **d)


def neighbors():
    rng = np.random.default_rng(9)
    F = 40
    poses = np.zeros((F, 12)); valid = np.ones(F, np.int32)
    for i in range(F):
        R, t = sy.true_pose(i * 7)
        poses[i, :9] = R.reshape(-1); poses[i, 9:] = t + rng.normal(size=3) * 0.01
    poses[13, :9] = 0; poses[13, 9:] = np.inf; valid[13] = 0   # invalid pose sentinel (Velodyne.cpp:50-51)
    nb = orc.find_neighbors(poses, valid, 6)
    off = np.zeros(F + 1, np.int32)
    for i, l in enumerate(nb):
        off[i + 1] = off[i] + len(l)
    np.savez_compressed(os.path.join(OUT, "neighbors.npz"), poses=poses, valid=valid, off=off, ids=np.array([v for l in nb for v in l], np.int32))


def reproj():
    """PanoramaReprojResidual_1Angle blocks (base/CostFunction.h:218-247): r and the 1x9 Jacobian rows."""
    rng = np.random.default_rng(31)
    b = synth.random_bundle(rng, n_cams=5, n_points=40)
    pt = np.repeat(np.arange(len(b["off"]) - 1), np.diff(b["off"])).astype(np.int32)
    r, J = orc.evaluate_reproj(b["bearing"], 1.5, b["cam"], pt, b["aa"], b["t"], b["X"])
    np.savez_compressed(os.path.join(OUT, "reproj.npz"), aa=b["aa"], t=b["t"], X=b["X"], off=b["off"], cam=b["cam"], pt=pt, bearing=b["bearing"],
                        weight=np.array(1.5), r=r, J=J)


def depth():
    """ProjectLidar2PanoramaDepth (util/Visualization.h:407-441) of a small scan, window sizes 3 and 2."""
    rng = np.random.default_rng(41)
    s = sy.make_scan(5, cols=128)
    xyz = np.concatenate([s["local_xyz"], s["local_xyz"][::5] * np.float32(1.02), rng.normal(size=(100, 3)).astype(np.float32) * np.float32([0.01, 3.0, 0.01])])
    a = np.deg2rad([1.0, -2.0, 0.5]); T = np.eye(4); T[:3, :3] = synth.rodrigues(a); T[:3, 3] = [0.03, -0.02, 0.05]
    d = dict(xyz=xyz, T_cl=T, rows=np.array(360), cols=np.array(720))
    for size in (3, 2):
        d["depth_size%d" % size] = orc.project_lidar_depth(360, 720, xyz, T, size)
    np.savez_compressed(os.path.join(OUT, "depth.npz"), **d)


def mvs():
    """MVS scoring pass (InitPatchMap + InitConfMap, mvs/MVS.cpp:586-680, :774-923) on a small rendered scene: photometric
    and geometric-consistency confidences."""
    from tests.test_mvs_cpu import mvs_scene
    (gray, depth, normal), neis, Rn, tn, nd = mvs_scene(orc, 64, 128, with_depths=True)
    rng = np.random.default_rng(51)
    depth = depth * rng.uniform(0.97, 1.03, size=depth.shape).astype(np.float32)
    d = dict(gray=gray, depth=depth, normal=normal, R_nr=Rn, t_nr=tn)
    for k, (g, x) in enumerate(zip(neis, nd)):
        d["nei%d_gray" % k] = g; d["nei%d_depth" % k] = x
    d["conf_pho"], d["depth_pho"], _ = orc.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1)
    d["conf_geo"], _, _ = orc.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1, nei_depths=nd)
    # one checkerboard PatchMatch iteration (EstimateDepthMapSingle, mvs/MVS.cpp:682-772, :1098-1129, :1254-1431) from the scored state
    d["sweep_seed"] = np.uint64(5)
    d["depth_sweep"], d["normal_sweep"], d["conf_sweep"] = orc.mvs_propagate(gray, neis, Rn, tn, d["depth_pho"], normal, d["conf_pho"], max_iter=1, seed=5)
    # FilterDepthImageRefine (mvs/MVS.cpp:1794-1890) of the swept map against the neighbours' depth maps; confidences on a 1/16 grid
    for k in range(len(nd)):
        d["nei%d_conf" % k] = (np.round(rng.uniform(0, 1, size=depth.shape) * 16) / 16).astype(np.float32)
    ref_conf = np.clip(d["conf_sweep"], 0, None)
    d["depth_refine"], d["conf_refine"], d["conf_after_refine"] = orc.mvs_filter_depth_refine(nd, [d["nei%d_conf" % k] for k in range(len(nd))], Rn, tn,
                                                                                               d["depth_sweep"], ref_conf, thr=0.02, min_depth=0.1, max_depth=20.0)
    np.savez_compressed(os.path.join(OUT, "mvs.npz"), **d)


def mvs_cloud():
    """MVS::DepthImageToCloud / DepthNormalToCloud (mvs/MVS.cpp:2073-2142) of a 40 x 80 depth map with every branch present (invalid and far
    depths, sky-coloured, grey and black pixels); and one SEQUENTIAL PatchMatch iteration (PropagateSequential, :1057-1097) from the scored
    state of the mvs.npz scene (its inputs are read from that fixture)."""
    depth, bgr, normal, T = synth.cloud_scene(np.random.default_rng(61), 40, 80, 20.0)
    g = np.load(os.path.join(OUT, "mvs.npz"))
    neis = [g["nei%d_gray" % k] for k in range(3)]
    ds, ns, cs = orc.mvs_propagate(g["gray"], neis, g["R_nr"], g["t_nr"], g["depth_pho"], g["normal"], g["conf_pho"], max_iter=1, seed=int(g["sweep_seed"]), sequential=True)
    xyz, rgb = orc.mvs_depth_to_cloud(depth, bgr, T, 20.0)
    xyz_n, rgb_n, nrm_n = orc.mvs_depth_to_cloud(depth, bgr, T, 20.0, filter_sky=False, normal=normal)
    np.savez_compressed(os.path.join(OUT, "mvs_cloud.npz"), depth=depth, bgr=bgr, normal=normal, T_wc=T, max_depth=np.float32(20.0), xyz=xyz, rgb=rgb,
                        xyz_all=xyz_n, rgb_all=rgb_n, normal_all=nrm_n, depth_seq=ds, normal_seq=ns, conf_seq=cs)


def features():
    """LiDAR feature extraction, planar branch (sensors/Velodyne.cpp:371-526, :531-760, :883-1000, :1098-1189, :1438-1586) on a
    small raw scan (16 rings x 240 columns, clutter, dropouts)."""
    from panovlm_amd import synthetic as sy
    raw = sy.raw_vlp16_scan(6, cols=240, clutter=25, dropout=0.05)
    f = orc.ScanFeatures(raw, horizon=240)
    d = dict(raw=raw, horizon=np.int32(240))
    for name in ("cloud_scan", "cornerSharp", "cornerLessSharp", "surfFlat", "surfLessFlat", "rc", "scan_start", "scan_end", "range_image",
                 "image_to_point_idx", "curvature", "state", "sort_ind", "left", "right"):
        d[name] = getattr(f, name)
    np.savez_compressed(os.path.join(OUT, "features.npz"), **d)


def line_extraction():
    """LiDAR feature extraction, LINE branch (Velodyne::EdgeToLine sensors/Velodyne.cpp:1269-1324, ExtractLineFeatures
    sensors/LidarLineExtraction.cpp) on a raw 16 x 900 scan with clutter; the RANSAC of FuseLines is the exhaustive
    2-point maximum consensus (oracle/lines.hpp)."""
    from panovlm_amd import synthetic as sy
    raw = sy.raw_vlp16_scan(4, cols=900, clutter=20, dropout=0.02)
    f = orc.ScanFeatures(raw, horizon=900, edge_to_line=True)
    so = np.cumsum([0] + [len(s) for s in f.edge_segmented]).astype(np.int32)
    po = np.cumsum([0] + [len(l) for l in f.point_to_segment]).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "line_extraction.npz"), raw=raw, horizon=np.int32(900), cornerBeforeFilter=f.cornerBeforeFilter,
                        cornerLessSharp=f.cornerLessSharp, cornerSharp=f.cornerSharp, surfFlat=f.surfFlat, surfLessFlat=f.surfLessFlat,
                        seg_offsets=so, seg_points=np.concatenate(f.edge_segmented) if f.edge_segmented else np.zeros((0, 4), np.float32),
                        segment_coeffs=f.segment_coeffs, end_points=f.end_points, p2s_offsets=po,
                        p2s_ids=np.array([v for l in f.point_to_segment for v in l], np.int32))


def undistort():
    """Velodyne::UndistortCloud (sensors/Velodyne.cpp:1642-1674) of three small sweeps — a small motion, a vanishing one (slerp's linear branch) and a large one —
    and SlerpPose (base/Geometry.hpp:572-583) at ratios inside and outside [0, 1]."""
    rng = np.random.default_rng(77)
    d = {}
    for k, (angle, n) in enumerate(((0.03, 700), (1e-12, 64), (2.5, 300))):
        cloud = np.concatenate([rng.normal(0, 7, (n, 3)), rng.integers(0, 16, (n, 1))], axis=1).astype(np.float32)
        R_wl = synth.rodrigues(rng.normal(0, 1.0, 3)); t_wl = rng.normal(0, 5, 3)
        R_we = R_wl @ synth.rodrigues(rng.normal(0, angle, 3)); t_we = t_wl + R_wl @ rng.normal(0, 0.2, 3)
        done, out = orc.undistort_cloud(cloud, R_wl, t_wl, R_we, t_we)
        assert done
        d.update({"cloud%d" % k: cloud, "R_wl%d" % k: R_wl, "t_wl%d" % k: t_wl, "R_we%d" % k: R_we, "t_we%d" % k: t_we, "out%d" % k: out})
    T1 = np.eye(4); T2 = np.eye(4)
    T1[:3, :3] = synth.rodrigues(rng.normal(0, 0.8, 3)); T1[:3, 3] = rng.normal(0, 3, 3)
    T2[:3, :3] = synth.rodrigues(rng.normal(0, 0.8, 3)); T2[:3, 3] = rng.normal(0, 3, 3)
    ratios = np.array([0.0, 0.25, 0.6666666666666666, 1.0, 2.0, -0.5])
    poses = np.zeros((len(ratios), 4, 4))
    for i, r in enumerate(ratios):
        R, t = orc.slerp_pose(T1[:3, :3], T1[:3, 3], T2[:3, :3], T2[:3, 3], float(r))
        poses[i] = np.eye(4); poses[i, :3, :3] = R; poses[i, :3, 3] = t
    d.update(cases=np.array(3), pose_w1=T1, pose_w2=T2, ratios=ratios, slerp=poses)
    np.savez_compressed(os.path.join(OUT, "undistort.npz"), **d)


if __name__ == "__main__":
    orc.build()
    if len(sys.argv) > 1:          # regenerate only the named fixtures: python tests/golden/make_golden.py line_extraction
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    fast_atan2(); functors(); assoc(); equirect(); lines(); neighbors(); reproj(); depth(); mvs(); mvs_cloud(); features(); line_extraction(); undistort()
    tot = 0
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            sz = os.path.getsize(os.path.join(OUT, f)); tot += sz
            print("%-26s %7d bytes" % (f, sz))
    print("total", tot)
