"""CPU twin of panovlm_amd/host (Solve + LidarOdometry::RefinePose/EstimatePose) driven by the ORACLE:
same trust-region policy (restated Ceres 2.0 defaults), residuals/Jacobians by the
oracle's Jet AutoDiff, associations by the oracle's brute-force search.  Test infrastructure only."""
import numpy as np

from panovlm_amd import synthetic as sy
from tests import synth


def transform_f32(xyz, R, t):
    """TransformCloud of the host mirror / pcl::transformPointCloud: float(((R0 x + R1 y) + R2 z) + t)."""
    return sy.to_world_f32(np.asarray(xyz, np.float32), np.asarray(R, np.float64), np.asarray(t, np.float64))


def inv_pose(R, t):
    Rl = R.T.copy()
    rt = np.array([(Rl[i, 0] * t[0] + Rl[i, 1] * t[1]) + Rl[i, 2] * t[2] for i in range(3)])
    return Rl, -rt


class Options:
    max_num_iterations = 20
    initial_radius = 1e4
    max_radius = 1e16
    min_radius = 1e-32
    min_relative_decrease = 1e-3
    function_tolerance = 1e-6
    gradient_tolerance = 1e-10
    parameter_tolerance = 1e-8
    min_lm_diagonal = 1e-6
    max_lm_diagonal = 1e32


def solve(oracle, groups, aa, t, const_poses, opt=Options(), bundle=None):
    """groups: list of dict(kind, normalize, rows (oracle layout incl. weight), rid, nid, loss (0/1), a).
    aa, t: F x 3 arrays updated in place.  const_poses: set of pose ids held constant (both blocks).
    bundle: optional dict(bearing n x 3, cam n (pose ids), pt n, X M x 3 (updated in place), w, loss, a,
    frozen (bool)) — PanoramaReprojResidual_1Angle blocks; the twin solves the FULL damped system (pose +
    point columns, dense), which is what eliminating the points and back-substituting computes."""
    F = aa.shape[0]
    ids = [np.concatenate([g["rid"], g["nid"]]) for g in groups]
    if bundle is not None:
        ids.append(np.asarray(bundle["cam"]))
    used = sorted(set(np.concatenate(ids).tolist()))
    free = [p for p in used if p not in const_poses]
    col = {p: 6 * i for i, p in enumerate(free)}
    n_pose = 6 * len(free)
    M = 0 if bundle is None else bundle["X"].shape[0]
    pts_free = bundle is not None and not bundle.get("frozen", False)
    n = n_pose + (3 * M if pts_free else 0)
    X0 = None if bundle is None else bundle["X"].copy()

    def evaluate(a_, t_, X_=None):
        H = np.zeros((n, n)); g = np.zeros(n); cost = 0.0
        if bundle is not None:
            rb, Jb = oracle.evaluate_reproj(bundle["bearing"], bundle["w"], bundle["cam"], bundle["pt"], a_, t_, X_)
            wb, hb = synth.huber_weights(rb, bundle["loss"], bundle["a"])
            cost += hb.sum()
            for i in range(len(rb)):
                cols = []
                c = col.get(int(bundle["cam"][i]), None)
                if c is not None:
                    cols.append((c, Jb[i, :6]))
                if pts_free:
                    cols.append((n_pose + 3 * int(bundle["pt"][i]), Jb[i, 6:]))
                for (ci, Ji) in cols:
                    g[ci:ci + len(Ji)] += wb[i] * Ji * rb[i]
                    for (cj, Jj) in cols:
                        H[ci:ci + len(Ji), cj:cj + len(Jj)] += wb[i] * np.outer(Ji, Jj)
        for gr in groups:
            r, J = oracle.evaluate(gr["kind"], gr["rows"], gr["rid"], gr["nid"], a_, t_, normalize=gr["normalize"])
            w, half = synth.huber_weights(r, gr["loss"], gr["a"])
            cost += half.sum()
            # accumulate per pose pair
            order = np.lexsort((gr["nid"], gr["rid"]))
            rid, nid = gr["rid"][order], gr["nid"][order]
            Jw = J[order]; rw = r[order]; ww = w[order]
            bounds = np.flatnonzero(np.r_[True, (rid[1:] != rid[:-1]) | (nid[1:] != nid[:-1]), True])
            for s, e in zip(bounds[:-1], bounds[1:]):
                Jp = Jw[s:e]; W = ww[s:e, None]
                Hp = (Jp * W).T @ Jp; gp = (Jp * W).T @ rw[s:e]
                idx = []
                for pose, o in ((rid[s], 0), (nid[s], 6)):
                    idx.append((col.get(pose, None), o))
                for (ci, oi) in idx:
                    if ci is None:
                        continue
                    g[ci:ci + 6] += gp[oi:oi + 6]
                    for (cj, oj) in idx:
                        if cj is None:
                            continue
                        H[ci:ci + 6, cj:cj + 6] += Hp[oi:oi + 6, oj:oj + 6]
        return cost, H, g

    x_aa, x_t, x_X = aa.copy(), t.copy(), X0
    cost, H, g = evaluate(x_aa, x_t, x_X)
    out = dict(initial_cost=cost, successful=1, unsuccessful=0, message="", history=[float(cost)])
    if n == 0:
        out["final_cost"] = cost
        return out
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0.0)))
    radius, dec = opt.initial_radius, 2.0
    it = 0
    if np.abs(g).max() <= opt.gradient_tolerance:
        out["message"] = "gradient tolerance reached"
    while not out["message"] and it < opt.max_num_iterations:
        it += 1
        Hs = H * scale[:, None] * scale[None, :]
        rhs = -g * scale
        D = np.clip(np.diag(Hs), opt.min_lm_diagonal, opt.max_lm_diagonal) / radius
        ok = True
        try:
            L = np.linalg.cholesky(Hs + np.diag(D))
            dy = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        except np.linalg.LinAlgError:
            ok = False
        accepted = False
        if ok:
            model = -((-rhs) @ dy + 0.5 * dy @ Hs @ dy)
            ok = model > 0 and np.isfinite(model)
        if ok:
            step = dy * scale
            c_aa, c_t = x_aa.copy(), x_t.copy()
            for p in free:
                c_aa[p] += step[col[p]:col[p] + 3]; c_t[p] += step[col[p] + 3:col[p] + 6]
            c_X = x_X
            if pts_free:
                c_X = x_X + step[n_pose:].reshape(M, 3)
            c_cost, cH, cg = evaluate(c_aa, c_t, c_X)
            rho = (cost - c_cost) / model
            if np.isfinite(c_cost) and rho > opt.min_relative_decrease:
                accepted = True
                xn = np.sqrt(sum((x_aa[p] ** 2).sum() + (x_t[p] ** 2).sum() for p in free) + ((x_X ** 2).sum() if pts_free else 0.0))
                change = cost - c_cost
                prev = cost
                x_aa, x_t, x_X, cost, H, g = c_aa, c_t, c_X, c_cost, cH, cg
                radius = min(opt.max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
                dec = 2.0
                out["successful"] += 1
                out["history"].append(float(cost))
                if abs(change) <= opt.function_tolerance * prev:
                    out["message"] = "function tolerance reached"
                elif np.abs(g).max() <= opt.gradient_tolerance:
                    out["message"] = "gradient tolerance reached"
                elif np.linalg.norm(step) <= opt.parameter_tolerance * (xn + opt.parameter_tolerance):
                    out["message"] = "parameter tolerance reached"
        if not accepted:
            out["unsuccessful"] += 1
            radius /= dec
            dec *= 2.0
            if radius < opt.min_radius:
                out["message"] = "trust region collapsed"
    aa[:] = x_aa; t[:] = x_t
    if bundle is not None:
        bundle["X"][:] = x_X
    out["final_cost"] = cost
    return out


def refine_pose(oracle, scans, cfg):
    """One RefinePose (lidar_mapping/LidarOdometry.cpp:15-114) on scan dicts holding LOCAL float clouds
    (flat_local, less_local) and the current pose; point-to-plane term only.  Updates poses in place."""
    F = len(scans)
    world = []
    for s in scans:
        world.append(dict(id=s["id"], R_wl=s["R_wl"], t_wl=s["t_wl"], flat_xyz=transform_f32(s["flat_cur"], s["R_wl"], s["t_wl"]),
                          flat_tag=s["flat_tag"], less_xyz=transform_f32(s["less_cur"], s["R_wl"], s["t_wl"]), less_tag=s["less_tag"]))
    aa = np.zeros((F, 3)); t = np.zeros((F, 3))
    for i, s in enumerate(scans):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])
        aa[i] = oracle.matrix_to_angle_axis(Rl); t[i] = tl
    poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in scans])
    nb = oracle.find_neighbors(poses, np.ones(F, np.int32), 6)
    rows, rid, nid = [], [], []
    for i in range(F):
        for n_idx in nb[i]:
            if n_idx < 0 or n_idx == i or n_idx >= F:
                continue
            o = oracle.assoc_point2plane(world[i], world[n_idx], cfg["tol"], cfg["thr"])
            m = len(o["qidx"])
            rows.append(np.concatenate([o["point"], o["plane"], np.ones((m, 1))], axis=1))
            rid += [i] * m; nid += [n_idx] * m
    rows = np.concatenate(rows)
    kind = 1 if cfg["angle"] else 0
    group = dict(kind=kind, normalize=cfg["normalize"], rows=rows, rid=np.array(rid, np.int32), nid=np.array(nid, np.int32), loss=1,
                 a=2 * np.pi / 180 if cfg["angle"] else 0.2)
    res = solve(oracle, [group], aa, t, {0})
    res["blocks"] = len(rows)
    for i, s in enumerate(scans):
        # Transform2Local with the OLD pose, then the new pose is set (clouds round-trip through float)
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])
        s["flat_cur"] = transform_f32(world[i]["flat_xyz"], Rl, tl)
        s["less_cur"] = transform_f32(world[i]["less_xyz"], Rl, tl)
        R_lw = oracle.angle_axis_to_matrix(aa[i])
        R_wl = R_lw.T.copy()
        rt = np.array([(R_wl[r, 0] * t[i][0] + R_wl[r, 1] * t[i][1]) + R_wl[r, 2] * t[i][2] for r in range(3)])
        s["R_wl"] = R_wl; s["t_wl"] = -rt
    return res


def estimate_pose(oracle, scans, cfg, max_iteration):
    for s in scans:
        s["flat_cur"] = np.asarray(s["flat_local"], np.float32); s["less_cur"] = np.asarray(s["less_local"], np.float32)
    log = []
    last_cost, last_step = 0.0, 32767
    for _ in range(max_iteration):
        res = refine_pose(oracle, scans, cfg)
        log.append(res)
        with np.errstate(divide="ignore", invalid="ignore"):
            if abs(res["final_cost"] - last_cost) / last_cost < 0.01:
                break
        if res["successful"] < 5 and last_step < 5:
            break
        last_cost, last_step = res["final_cost"], res["successful"]
    return log


# ------------------------------------------------------------------------------------------------
# line-to-line term: LidarLineMatch::GenerateTracks + AddLidarLineToLineResidual2 on the oracle
# ------------------------------------------------------------------------------------------------
def world_line_scan(s):
    return dict(id=s["id"], R_wl=s["R_wl"], t_wl=s["t_wl"], corner_xyz=transform_f32(s["corner_cur"], s["R_wl"], s["t_wl"]), p2s=s["p2s"],
                seg_size=np.array([len(x) for x in s["seg_points"]], np.int32), seg_coeffs=s["seg_coeffs"], end_points=s["end_points"])


def line_tracks(oracle, world, nb, min_len):
    parent = {}

    def find(a):
        while parent.setdefault(a, a) != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    for i in range(len(world)):
        for j in nb[i]:
            o = oracle.assoc_line2line(world[j], world[i], 0.3)
            for a, b in zip(o["nei_idx"], o["ref_idx"]):
                ra, rb = find((i, int(a))), find((j, int(b)))
                if ra != rb:
                    parent[ra] = rb
    comps = {}
    for node in list(parent):
        comps.setdefault(find(node), set()).add(node)
    return [c for c in comps.values() if len({n[0] for n in c}) >= min_len and len(c) > 1]


def world2local(R, t, p):
    out = np.empty(3)
    for i in range(3):
        a = (R[0, i] * p[0] + R[1, i] * p[1]) + R[2, i] * p[2]
        b = (R[0, i] * t[0] + R[1, i] * t[1]) + R[2, i] * t[2]
        out[i] = a - b
    return out


def refine_pose_lines(oracle, scans, thr=0.3, normalize=True):
    F = len(scans)
    world = [world_line_scan(s) for s in scans]
    aa = np.zeros((F, 3)); t = np.zeros((F, 3))
    for i, s in enumerate(scans):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])
        aa[i] = oracle.matrix_to_angle_axis(Rl); t[i] = tl
    poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in scans])
    nb6 = oracle.find_neighbors(poses, np.ones(F, np.int32), 6)
    nb4 = oracle.find_neighbors(poses, np.ones(F, np.int32), 4)
    tracks = line_tracks(oracle, world, nb4, 3)
    member = {}
    for k, tr in enumerate(tracks):
        for node in tr:
            member.setdefault(node, []).append(k)
    rows, rid, nid = [], [], []
    for i in range(F):
        for n in nb6[i]:
            if n < 0 or n == i or n >= F:
                continue
            o = oracle.assoc_line2line(world[i], world[n], thr)
            for nl, rl, p1, p2 in zip(o["nei_idx"], o["ref_idx"], o["p1"], o["p2"]):
                if (i, int(rl)) not in member:
                    continue
                if not any((n, int(nl)) in tracks[k] for k in member[(i, int(rl))]):
                    continue
                for ci in scans[n]["seg_points"][int(nl)]:
                    lp = world2local(scans[n]["R_wl"], scans[n]["t_wl"], world[n]["corner_xyz"][ci].astype(np.float64))
                    rows.append(np.concatenate([lp, p1, p2, [1.0]])); rid.append(i); nid.append(n)
    group = dict(kind=3, normalize=normalize, rows=np.array(rows), rid=np.array(rid, np.int32), nid=np.array(nid, np.int32), loss=0, a=0.0)
    res = solve(oracle, [group], aa, t, {0})
    res["blocks"] = len(rows)
    for i, s in enumerate(scans):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])
        s["corner_cur"] = transform_f32(world[i]["corner_xyz"], Rl, tl)
        R_lw = oracle.angle_axis_to_matrix(aa[i])
        R_wl = R_lw.T.copy()
        rt = np.array([(R_wl[r, 0] * t[i][0] + R_wl[r, 1] * t[i][1]) + R_wl[r, 2] * t[i][2] for r in range(3)])
        s["R_wl"] = R_wl; s["t_wl"] = -rt
    return res


def refine_pose_full(oracle, scans, cfg):
    """One RefinePose with BOTH LiDAR terms, in the order lidar_mapping/LidarOdometry.cpp:38-56 adds them: line-to-line
    (GenerateTracks + AddLidarLineToLineResidual2, no loss for the angle functor) then point-to-plane (Huber).  scans: dicts
    with flat_cur / less_cur / corner_cur (LOCAL float clouds), flat_tag, less_tag, p2s, seg_points, seg_coeffs, end_points."""
    F = len(scans)
    wl = [world_line_scan(s) for s in scans]
    wp = [dict(id=s["id"], R_wl=s["R_wl"], t_wl=s["t_wl"], flat_xyz=transform_f32(s["flat_cur"], s["R_wl"], s["t_wl"]), flat_tag=s["flat_tag"],
               less_xyz=transform_f32(s["less_cur"], s["R_wl"], s["t_wl"]), less_tag=s["less_tag"]) for s in scans]
    aa = np.zeros((F, 3)); t = np.zeros((F, 3))
    for i, s in enumerate(scans):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])
        aa[i] = oracle.matrix_to_angle_axis(Rl); t[i] = tl
    poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in scans])
    nb6 = oracle.find_neighbors(poses, np.ones(F, np.int32), 6)
    nb4 = oracle.find_neighbors(poses, np.ones(F, np.int32), 4)
    groups = []
    segmented = any(len(s["seg_points"]) > 0 for s in scans)
    if cfg.get("lines", True) and segmented:
        tracks = line_tracks(oracle, wl, nb4, 3)
        member = {}
        for k, tr in enumerate(tracks):
            for node in tr:
                member.setdefault(node, []).append(k)
        rows, rid, nid = [], [], []
        for i in range(F):
            for n in nb6[i]:
                if n < 0 or n == i or n >= F:
                    continue
                o = oracle.assoc_line2line(wl[i], wl[n], cfg.get("line_thr", 0.3))
                for nl, rl, p1, p2 in zip(o["nei_idx"], o["ref_idx"], o["p1"], o["p2"]):
                    if (i, int(rl)) not in member:
                        continue
                    if not any((n, int(nl)) in tracks[k] for k in member[(i, int(rl))]):
                        continue
                    for ci in scans[n]["seg_points"][int(nl)]:
                        lp = world2local(scans[n]["R_wl"], scans[n]["t_wl"], wl[n]["corner_xyz"][ci].astype(np.float64))
                        rows.append(np.concatenate([lp, p1, p2, [1.0]])); rid.append(i); nid.append(n)
        if rows:
            groups.append(dict(kind=3, normalize=cfg["normalize"], rows=np.array(rows), rid=np.array(rid, np.int32), nid=np.array(nid, np.int32), loss=0, a=0.0))
    rows, rid, nid = [], [], []
    for i in range(F):
        for n_idx in nb6[i]:
            if n_idx < 0 or n_idx == i or n_idx >= F:
                continue
            o = oracle.assoc_point2plane(wp[i], wp[n_idx], cfg["tol"], cfg["thr"])
            m = len(o["qidx"])
            rows.append(np.concatenate([o["point"], o["plane"], np.ones((m, 1))], axis=1))
            rid += [i] * m; nid += [n_idx] * m
    rows = np.concatenate(rows)
    groups.append(dict(kind=1, normalize=cfg["normalize"], rows=rows, rid=np.array(rid, np.int32), nid=np.array(nid, np.int32), loss=1, a=2 * np.pi / 180))
    res = solve(oracle, groups, aa, t, {0})
    res["blocks"] = sum(len(g["rows"]) for g in groups)
    res["line_blocks"] = len(groups[0]["rows"]) if len(groups) == 2 else 0
    for i, s in enumerate(scans):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])
        s["flat_cur"] = transform_f32(wp[i]["flat_xyz"], Rl, tl); s["less_cur"] = transform_f32(wp[i]["less_xyz"], Rl, tl)
        s["corner_cur"] = transform_f32(wl[i]["corner_xyz"], Rl, tl)
        R_lw = oracle.angle_axis_to_matrix(aa[i])
        R_wl = R_lw.T.copy()
        rt = np.array([(R_wl[r, 0] * t[i][0] + R_wl[r, 1] * t[i][1]) + R_wl[r, 2] * t[i][2] for r in range(3)])
        s["R_wl"] = R_wl; s["t_wl"] = -rt
    return res


def estimate_pose_full(oracle, scans, cfg, max_iteration):
    for s in scans:
        s["flat_cur"] = np.asarray(s["flat_local"], np.float32); s["less_cur"] = np.asarray(s["less_local"], np.float32)
        s["corner_cur"] = np.asarray(s["corner_local"], np.float32)
    log = []
    last_cost, last_step = 0.0, 32767
    for _ in range(max_iteration):
        res = refine_pose_full(oracle, scans, cfg)
        log.append(res)
        with np.errstate(divide="ignore", invalid="ignore"):
            if abs(res["final_cost"] - last_cost) / last_cost < 0.01:
                break
        if res["successful"] < 5 and last_step < 5:
            break
        last_cost, last_step = res["final_cost"], res["successful"]
    return log


def twin_scan_from_features(sid, R_wl, t_wl, f):
    """Scan dict of the twins above from an oracle.ScanFeatures(..., edge_to_line=True)."""
    ids = [int(v) for v in f.cornerLessSharp[:, 3]]
    where = {v: k for k, v in enumerate(ids)}
    return dict(id=sid, R_wl=R_wl, t_wl=t_wl, flat_local=f.surfFlat[:, :3], flat_tag=f.surfFlat[:, 3], less_local=f.surfLessFlat[:, :3],
                less_tag=f.surfLessFlat[:, 3], corner_local=f.cornerLessSharp[:, :3], p2s=f.point_to_segment,
                seg_points=[[where[int(v)] for v in seg[:, 3]] for seg in f.edge_segmented], seg_coeffs=f.segment_coeffs,
                end_points=f.end_points.reshape(-1, 6))


# ------------------------------------------------------------------------------------------------
# CameraLidarOptimizer (mapping mode, no SfM term): AssociateLineMulti + Optimize + JointOptimize
# ------------------------------------------------------------------------------------------------
def pose4(R, t):
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T


def associate_line_multi(oracle, lidars, frames, T_cl_init, neighbor_size):
    out = {}
    L = len(lidars)
    for f, fr in enumerate(frames):
        start = max(0, f - neighbor_size // 2); end = min(L, start + neighbor_size); start = max(0, end - neighbor_size)
        for lid in range(start, end):
            s = lidars[lid]
            T_cl = np.linalg.inv(pose4(fr["R_wc"], fr["t_wc"])) @ pose4(s["R_wl"], s["t_wl"])
            local = dict(corner_xyz=s["corner_cur"], p2s=s["p2s"], seg_size=np.array([len(x) for x in s["seg_points"]], np.int32),
                         seg_coeffs=s["seg_coeffs"], end_points=s["end_points"])
            out[(f, lid)] = (oracle.assoc_by_angle(fr["rows"], fr["cols"], fr["lines"], local, T_cl, multiple=True), fr["lines"])
    return out


# ------------------------------------------------------------------------------------------------
# SfM reprojection term: AddCameraResidual (util/Optimization.cpp:172-222) on the oracle
# ------------------------------------------------------------------------------------------------
def reproj_blocks(oracle, frames, tracks, cam_offset=0):
    """(cam ids, point ids, bearings) of the PanoramaReprojResidual_1Angle blocks: one per (track, observation)
    in a frame with a valid pose; bearing = float ImageToCam of the keypoint ROUNDED to the nearest pixel
    (the cv::Point2i overload the reference's call binds to)."""
    cam, pt, bearing = [], [], []
    for ti, tr in enumerate(tracks):
        for (fi, ki) in sorted(set((int(a), int(b)) for a, b in tr["obs"])):
            fr = frames[fi]
            if not fr.get("valid", 1):
                continue
            px = np.rint(np.asarray(fr["keypoints"][ki], np.float32)).astype(np.float32)[None]
            b = oracle.image_to_cam(fr["rows"], fr["cols"], px, 1.0)[0]
            cam.append(cam_offset + fi); pt.append(ti); bearing.append(b.astype(np.float64))
    return np.array(cam, np.int32), np.array(pt, np.int32), np.array(bearing, np.float64).reshape(-1, 3)


def frame_params(oracle, frames):
    aa = np.zeros((len(frames), 3)); t = np.zeros((len(frames), 3))
    for i, fr in enumerate(frames):
        if not fr.get("valid", 1):
            continue
        Rl, tl = inv_pose(np.asarray(fr["R_wc"], np.float64), np.asarray(fr["t_wc"], np.float64))
        aa[i] = oracle.matrix_to_angle_axis(Rl); t[i] = tl
    return aa, t


def bundle_adjust(oracle, frames, tracks, weight, refine_structure=True, max_iter=50):
    """Twin of the driver's `bundle` command: camera-only BA, camera 0 constant.  Returns (summary, aa, t, X)."""
    aa, t = frame_params(oracle, frames)
    cam, pt, bearing = reproj_blocks(oracle, frames, tracks)
    X = np.array([tr["point"] for tr in tracks], np.float64)
    opt = Options(); opt.max_num_iterations = max_iter
    b = dict(bearing=bearing, cam=cam, pt=pt, X=X, w=weight, loss=1, a=4.0 * np.pi / 180.0, frozen=not refine_structure)
    res = solve(oracle, [], aa, t, {0}, opt, bundle=b)
    res["blocks"] = len(cam)
    return res, aa, t, X


def joint_optimize_step(oracle, lidars, frames, pairs, cfg, structure=None):
    Fc, L = len(frames), len(lidars)
    aa = np.zeros((Fc + L, 3)); t = np.zeros((Fc + L, 3))
    for i, fr in enumerate(frames):
        Rl, tl = inv_pose(fr["R_wc"], fr["t_wc"]); aa[i] = oracle.matrix_to_angle_axis(Rl); t[i] = tl
    for i, s in enumerate(lidars):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"]); aa[Fc + i] = oracle.matrix_to_angle_axis(Rl); t[Fc + i] = tl
    groups = []
    r4, r5, rid, nid = [], [], [], []
    w = cfg["camera_lidar_weight"]
    for (f, lid), (o, lines) in sorted(pairs.items()):
        fr = frames[f]
        for k, li in enumerate(o["image_line_id"]):
            px = np.asarray(lines[li], np.float64)
            p1 = oracle.image_to_cam(fr["rows"], fr["cols"], px[None, 0:2], 1.0)[0]; p2 = oracle.image_to_cam(fr["rows"], fr["cols"], px[None, 2:4], 1.0)[0]
            pl = np.cross(p2 - p1, -p1); plane = np.concatenate([pl, [-(pl @ p1)]])
            st, en = o["start"][k], o["end"][k]
            r4.append(np.concatenate([plane[:3], en, st, [1.0 * w]]))
            c = float(np.clip(p1 @ p2, -1, 1))
            r5.append(np.concatenate([plane, (en + st) / 2, (p1 + p2) / 2, [np.arccos(c)], [2.0 * w]]))
            rid.append(f); nid.append(Fc + lid)
    a3 = 3 * np.pi / 180
    if r4:
        groups.append(dict(kind=4, normalize=False, rows=np.array(r4), rid=np.array(rid, np.int32), nid=np.array(nid, np.int32), loss=1, a=a3))
        groups.append(dict(kind=5, normalize=False, rows=np.array(r5), rid=np.array(rid, np.int32), nid=np.array(nid, np.int32), loss=1, a=a3))
    blocks = 2 * len(r4)
    # Optimize moves every valid scan to the world frame (Transform2LidarWorld) whatever terms are enabled
    world = [dict(id=s["id"], R_wl=s["R_wl"], t_wl=s["t_wl"], flat_xyz=transform_f32(s["flat_cur"], s["R_wl"], s["t_wl"]), flat_tag=s["flat_tag"],
                  less_xyz=transform_f32(s["less_cur"], s["R_wl"], s["t_wl"]), less_tag=s["less_tag"],
                  corner_xyz=transform_f32(s["corner_cur"], s["R_wl"], s["t_wl"])) for s in lidars]
    if cfg["p2plane"]:
        poses = np.array([np.concatenate([s["R_wl"].reshape(-1), s["t_wl"]]) for s in lidars])
        nb = oracle.find_neighbors(poses, np.ones(L, np.int32), 6)
        rows, rr, nn = [], [], []
        for i in range(L):
            for n_idx in nb[i]:
                if n_idx < 0 or n_idx == i or n_idx >= L:
                    continue
                o = oracle.assoc_point2plane(world[i], world[n_idx], cfg["tol"], cfg["thr"])
                m = len(o["qidx"])
                rows.append(np.concatenate([o["point"], o["plane"], np.full((m, 1), cfg["lidar_weight"])], axis=1))
                rr += [Fc + i] * m; nn += [Fc + n_idx] * m
        rows = np.concatenate(rows)
        groups.append(dict(kind=1, normalize=True, rows=rows, rid=np.array(rr, np.int32), nid=np.array(nn, np.int32), loss=1, a=2 * np.pi / 180))
        blocks += len(rows)
    opt = Options(); opt.max_num_iterations = 50
    bundle = None
    if structure is not None and len(structure["tracks"]):
        cam, pt, bearing = reproj_blocks(oracle, frames, structure["tracks"])
        bundle = dict(bearing=bearing, cam=cam, pt=pt, X=structure["X"], w=cfg.get("camera_weight", 1.0), loss=1, a=4.0 * np.pi / 180.0)
        blocks += len(cam)
    res = solve(oracle, groups, aa, t, {0}, opt, bundle=bundle)
    res["blocks"] = blocks
    for i, fr in enumerate(frames):
        R_cw = oracle.angle_axis_to_matrix(aa[i]); R_wc = R_cw.T.copy()
        rt = np.array([(R_wc[r, 0] * t[i][0] + R_wc[r, 1] * t[i][1]) + R_wc[r, 2] * t[i][2] for r in range(3)])
        fr["R_wc"] = R_wc; fr["t_wc"] = -rt
    for i, s in enumerate(lidars):
        Rl, tl = inv_pose(s["R_wl"], s["t_wl"])   # Transform2Local with the OLD pose: the float clouds round-trip
        s["flat_cur"] = transform_f32(world[i]["flat_xyz"], Rl, tl); s["less_cur"] = transform_f32(world[i]["less_xyz"], Rl, tl)
        s["corner_cur"] = transform_f32(world[i]["corner_xyz"], Rl, tl)
        R_lw = oracle.angle_axis_to_matrix(aa[Fc + i]); R_wl = R_lw.T.copy()
        rt = np.array([(R_wl[r, 0] * t[Fc + i][0] + R_wl[r, 1] * t[Fc + i][1]) + R_wl[r, 2] * t[Fc + i][2] for r in range(3)])
        s["R_wl"] = R_wl; s["t_wl"] = -rt
    return res


def joint_optimize(oracle, lidars, frames, T_cl_init, cfg, neighbor_size, iters, structure=None):
    """structure: optional dict(tracks=[dict(point, obs)], X=M x 3 array refined in place) — the SfM term."""
    for s in lidars:
        s["corner_cur"] = np.asarray(s["corner_local"], np.float32)
        s["flat_cur"] = np.asarray(s.get("flat_local", np.zeros((0, 3))), np.float32); s["less_cur"] = np.asarray(s.get("less_local", np.zeros((0, 3))), np.float32)
    log = []
    last_cost, last_step = 0.0, 2 ** 31 - 1
    pairs = associate_line_multi(oracle, lidars, frames, T_cl_init, neighbor_size)
    for _ in range(iters):
        npairs = sum(len(o["image_line_id"]) for o, _ in pairs.values())
        res = joint_optimize_step(oracle, lidars, frames, pairs, cfg, structure)
        res["pairs"] = npairs
        log.append(res)
        pairs = associate_line_multi(oracle, lidars, frames, T_cl_init, neighbor_size)
        with np.errstate(divide="ignore", invalid="ignore"):
            if abs(res["final_cost"] - last_cost) / last_cost < 0.01:
                break
        if res["successful"] < 5 and last_step < 5:
            break
        last_cost, last_step = res["final_cost"], res["successful"]
    return log
