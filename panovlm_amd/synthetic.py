"""Deterministic synthetic VLP-16 scans (BASELINE.md §2 / SURVEY.md §8d): workload generator
shared by bench.py and the tests.  numpy only; not part of the compute path.

Scene: a 8 x 3 x 12 m box room (x right, y down, z forward — the axis convention of
sensors/Velodyne.cpp:125-132) with 6 interior boxes.  Scan k is ray-cast from the TRUE pose
(0.1 m per step along z, bouncing between the end walls, 0.5 deg yaw per step) with 16 lasers
(-15..+15 deg, 2 deg apart, sensors/Velodyne.cpp:176) x `cols` azimuth steps and sigma = 1 cm range
noise.  The feature clouds handed to the association are the LOCAL float32 points moved to the world
frame with the ESTIMATED pose (true pose perturbed by <= 2 cm / 0.5 deg), rounded to float32 per
coordinate exactly like pcl::transformPointCloud(cloud, Matrix4d) does (sensors/Velodyne.cpp:1791).
Every point is tagged POINT_NORMAL (=1) and is both a surfFlat query and a surfLessFlat target.
"""
import numpy as np

SEED = 20240601
ROOM_MIN = np.array([-4.0, -1.5, -6.0])
ROOM_MAX = np.array([4.0, 1.5, 6.0])


def _yaw(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def scene_boxes(seed=SEED):
    rng = np.random.default_rng(seed)
    boxes = []
    for i in range(6):
        side = -1.0 if i % 2 == 0 else 1.0
        cx = side * rng.uniform(1.8, 3.2)
        cz = -5.0 + 10.0 * (i + 0.5) / 6.0 + rng.uniform(-0.3, 0.3)
        half = np.array([rng.uniform(0.3, 0.7), rng.uniform(0.4, 1.2), rng.uniform(0.3, 0.8)])
        c = np.array([cx, 1.5 - half[1], cz])  # standing on the floor (y down)
        boxes.append((c - half, c + half))
    return boxes


def true_pose(k):
    """T_wl of frame k: 0.1 m steps along z bouncing in [-3.5, 3.5], yaw 0.5 deg per step."""
    span = 70  # steps per leg
    m = k % (2 * span)
    z = -3.5 + 0.1 * (m if m <= span else 2 * span - m)
    return _yaw(np.deg2rad(0.5) * k), np.array([0.0, 0.0, z])


def estimated_pose(k, seed=SEED):
    rng = np.random.default_rng(seed + 7919 * (k + 1))
    R, t = true_pose(k)
    if k == 0:
        return R, t
    d = rng.uniform(-np.deg2rad(0.5), np.deg2rad(0.5), size=3)
    th = np.linalg.norm(d)
    K = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    dR = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)
    return dR @ R, t + rng.uniform(-0.02, 0.02, size=3)


def _ray_aabb(o, d, lo, hi):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t1 = (lo - o) * inv
        t2 = (hi - o) * inv
    tn = np.minimum(t1, t2).max(axis=1)
    tf = np.maximum(t1, t2).min(axis=1)
    return tn, tf


def raycast_local(k, cols=4096, rings=16, seed=SEED, noise=0.01):
    """Local-frame float32 point cloud of scan k, ring-major (rings x cols, 3)."""
    R, t = true_pose(k)
    el = np.deg2rad(-15.0 + 2.0 * np.arange(rings))
    az = 2 * np.pi * (np.arange(cols) + 0.5) / cols
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    dl = np.stack([ce * np.sin(az)[None, :], -se * np.ones_like(az)[None, :], ce * np.cos(az)[None, :]], axis=-1).reshape(-1, 3)
    dw = dl @ R.T
    o = t[None, :]
    _, tf = _ray_aabb(o, dw, ROOM_MIN, ROOM_MAX)
    rng_ = tf.copy()
    for lo, hi in scene_boxes(seed):
        tn, tf2 = _ray_aabb(o, dw, lo, hi)
        hit = (tn > 0) & (tn <= tf2)
        rng_ = np.where(hit & (tn < rng_), tn, rng_)
    rng_ = rng_ + np.random.default_rng(seed + 104729 * (k + 1)).normal(size=rng_.shape) * noise
    return (dl * rng_[:, None]).astype(np.float32)


VLP16_FIRING_ORDER = np.array([0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15])   # ring ids in firing order (sensors/Velodyne.cpp:407-414)


def raw_vlp16_scan(k, cols=1800, seed=SEED, noise=0.01, skew=0.3, jitter=0.05, dropout=0.01, start_deg=37.0, clutter=0, elevation_noise=0.0,
                   return_truth=False):
    """Raw VLP-16 scan k as LoadLidar hands it to ReOrderVLP (sensors/Velodyne.cpp:92-145, :371-526): n x 4 float32
    (x, y, z, intensity) in FIRING order — azimuth column by column, the 16 lasers of a column in the sensor's
    interleaved order — camera-style axes (x right, y down, z forward), azimuth atan2(x, z) increasing from `start_deg`
    through one full turn.  skew: azimuth progress inside one column (fraction of the column width), jitter: random
    azimuth error per return (same unit), dropout: fraction of returns missing (no echo), clutter: number of small
    (5-25 cm) boxes scattered through the room (the small segments Velodyne::Segmentation removes), elevation_noise:
    sigma in degrees added to the beam elevation.  return_truth: also return the (ring, column) each return was fired at."""
    R, t = true_pose(k)
    rng = np.random.default_rng(seed + 15485863 * (k + 1))
    small = []
    for _ in range(clutter):
        c = np.array([rng.uniform(-3.5, 3.5), rng.uniform(-1.2, 1.2), rng.uniform(-5.5, 5.5)])
        h = rng.uniform(0.025, 0.125, size=3)
        if np.linalg.norm(c - t) > 1.0:
            small.append((c - h, c + h))
    res = 2 * np.pi / cols
    ring = np.tile(VLP16_FIRING_ORDER, cols)
    col = np.repeat(np.arange(cols), 16)
    fire = np.tile(np.arange(16), cols)
    az = np.deg2rad(start_deg) + res * (col + skew * fire / 16.0 + jitter * rng.uniform(-1, 1, size=col.shape))
    el = np.deg2rad(-15.0 + 2.0 * ring + elevation_noise * rng.normal(size=ring.shape))
    dl = np.stack([np.cos(el) * np.sin(az), -np.sin(el), np.cos(el) * np.cos(az)], axis=-1)
    dw = dl @ R.T
    o = t[None, :]
    _, tf = _ray_aabb(o, dw, ROOM_MIN, ROOM_MAX)
    rng_ = tf.copy()
    for lo, hi in scene_boxes(seed) + small:
        tn, tf2 = _ray_aabb(o, dw, lo, hi)
        hit = (tn > 0) & (tn <= tf2)
        rng_ = np.where(hit & (tn < rng_), tn, rng_)
    rng_ = rng_ + rng.normal(size=rng_.shape) * noise
    keep = rng.uniform(size=rng_.shape) >= dropout
    xyz = (dl * rng_[:, None]).astype(np.float32)[keep]
    inten = rng.uniform(0, 100, size=len(xyz)).astype(np.float32)
    raw = np.concatenate([xyz, inten[:, None]], axis=1)
    return (raw, ring[keep], col[keep]) if return_truth else raw


def to_world_f32(local_f32, R_wl, t_wl):
    """pcl::transformPointCloud(float cloud, Matrix4d): per coordinate float(m0*x + m1*y + m2*z + m3)."""
    p = local_f32.astype(np.float64)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((R_wl[r, 0] * p[:, 0] + R_wl[r, 1] * p[:, 1]) + R_wl[r, 2] * p[:, 2]) + t_wl[r]
    return out.astype(np.float32)


def make_scan(k, cols=4096, rings=16, seed=SEED, downsample_targets=0.0):
    """Scan dict for panovlm_amd.Scan.  downsample_targets > 0 replaces the surfLessFlat target cloud
    by its voxel-grid centroids, as the reference's extractor does at 0.2 m."""
    local = raycast_local(k, cols, rings, seed)
    R, t = estimated_pose(k, seed)
    world = to_world_f32(local, R, t)
    less = world
    if downsample_targets > 0:
        # pcl::VoxelGrid semantics: one centroid per occupied voxel (leaf 0.2 m in the extractor)
        key = np.floor(local.astype(np.float64) / downsample_targets).astype(np.int64)
        key = (key[:, 0] + 4096) * (8192 * 8192) + (key[:, 1] + 4096) * 8192 + (key[:, 2] + 4096)
        uniq, inv = np.unique(key, return_inverse=True)
        sums = np.zeros((len(uniq), 3)); np.add.at(sums, inv, local.astype(np.float64))
        cnt = np.bincount(inv, minlength=len(uniq))[:, None]
        less = to_world_f32((sums / cnt).astype(np.float32), R, t)
    return dict(id=k, R_wl=R, t_wl=t, flat_xyz=world, flat_tag=np.ones(len(world), np.float32),
                less_xyz=less, less_tag=np.ones(len(less), np.float32), local_xyz=local)


def pose_params(R_wl, t_wl):
    """(angleAxis_lw, t_lw) parameter blocks of T_lw = T_wl^-1 (lidar_mapping/LidarOdometry.cpp:27-33)."""
    R = R_wl.T
    t = -R @ t_wl
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-12:
        aa = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    else:
        aa = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return aa, t


def pair_list(F, nb):
    """Ordered (ref, nei) pairs: every scan against its nb temporally nearest scans."""
    ref, nei = [], []
    half = nb // 2
    for i in range(F):
        cand = [i + d for d in range(-half, half + 1) if d != 0 and 0 <= i + d < F]
        j = 1
        while len(cand) < min(nb, F - 1):
            for c in (i - half - j, i + half + j):
                if 0 <= c < F and c not in cand and len(cand) < nb:
                    cand.append(c)
            j += 1
        for c in cand:
            ref.append(i); nei.append(c)
    return np.array(ref, np.int32), np.array(nei, np.int32)


# ---- line features (cornerLessSharp + segments) ------------------------------------------------------------
def make_line_scan(rng, scan_id, R_wl, t_wl, lines_world, pts_per_line=(8, 40), noise=0.01, shared_frac=0.1, extra_pts=30):
    """lines_world: list of (a, b) 3-D endpoints in the world frame.  Returns a scan dict whose
    corner cloud samples those lines (+ a few unsegmented points); coefficients/end points are in
    the LiDAR-local frame, the corner cloud in BOTH frames (corner_xyz = world float32 as during
    LiDAR-LiDAR association, corner_local for camera-LiDAR association)."""
    pts_local, p2s, seg_size, coeffs, ends = [], [], [], [], []
    Rlw = R_wl.T
    for s, (a, b) in enumerate(lines_world):
        n = int(rng.integers(*pts_per_line))
        u = np.sort(rng.uniform(0, 1, size=n))
        pw = a[None, :] + u[:, None] * (b - a)[None, :] + rng.normal(size=(n, 3)) * noise
        pl = (pw - t_wl) @ Rlw.T
        al, bl = Rlw @ (a - t_wl), Rlw @ (b - t_wl)
        d = (bl - al) / np.linalg.norm(bl - al)
        first = len(pts_local)
        for q in pl:
            pts_local.append(q); p2s.append([s])
        seg_size.append(n)
        coeffs.append(np.concatenate([pl.mean(0), d]))
        ends.append(np.concatenate([al, bl]))
        # a few points shared with the previous segment (point_to_segment is a set of ids)
        if s > 0 and rng.uniform() < shared_frac * 5:
            k = first + int(rng.integers(0, n))
            p2s[k] = sorted(set(p2s[k] + [s - 1]))
            seg_size[s - 1] += 1
    for _ in range(extra_pts):
        pts_local.append(rng.normal(size=3) * 3); p2s.append([])
    pts_local = np.array(pts_local, np.float32)
    world = to_world_f32(pts_local, R_wl, t_wl)
    return dict(id=scan_id, R_wl=R_wl, t_wl=t_wl, corner_xyz=world, corner_local=pts_local, p2s=p2s,
                seg_size=np.array(seg_size, np.int32), seg_coeffs=np.array(coeffs), end_points=np.array(ends))


def random_world_lines(rng, n, extent=4.0):
    out = []
    for _ in range(n):
        a = rng.uniform(-extent, extent, size=3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        out.append((a, a + d * rng.uniform(0.5, 3.0)))
    return out
