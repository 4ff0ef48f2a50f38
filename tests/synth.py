"""Seeded synthetic inputs shared by the tests, smoke() and bench.py (numpy only)."""
import numpy as np


def rodrigues(aa):
    aa = np.asarray(aa, np.float64)
    th = np.linalg.norm(aa)
    K = np.array([[0, -aa[2], aa[1]], [aa[2], 0, -aa[0]], [-aa[1], aa[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def random_poses(rng, F, rot=0.3, trans=2.0):
    return rng.normal(size=(F, 3)) * rot, rng.normal(size=(F, 3)) * trans


def random_pairs(rng, F, P):
    ref = rng.integers(0, F, size=P)
    nei = (ref + rng.integers(1, F, size=P)) % F
    return ref.astype(np.int32), nei.astype(np.int32)


def p_ref(aa, t, r, n, P):
    return rodrigues(aa[r]) @ rodrigues(-aa[n]) @ (P - t[n]) + t[r]


def random_resset(rng, kind, aa, t, ref, nei, counts, offsets_scale=(1e-4, 5e-3, 0.05, 0.5)):
    """ICP-like residual rows for each (ref[p], nei[p]) segment; returns rows (n x stride), pair_offsets."""
    rows = []
    off = [0]
    for p, c in enumerate(counts):
        for _ in range(int(c)):
            P = rng.normal(size=3) * 4.0
            Pr = p_ref(aa, t, ref[p], nei[p], P)
            o = rng.choice(offsets_scale) * rng.choice([-1.0, 1.0])
            if kind in (0, 1):
                n = rng.normal(size=3); n /= np.linalg.norm(n)
                rows.append(np.concatenate([P, n, [-(n @ Pr) + o]]))
            elif kind in (2, 3):
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                u = rng.normal(size=3); u -= (u @ d) * d; u /= np.linalg.norm(u)
                A0 = Pr + u * abs(o) + d * rng.normal()
                rows.append(np.concatenate([P, A0 + 0.1 * d, A0 - 0.1 * d]))
            elif kind == 4:
                a = rng.normal(size=3) * 3; b = a + rng.normal(size=3)
                ac = p_ref(aa, t, ref[p], nei[p], a); bc = p_ref(aa, t, ref[p], nei[p], b)
                nrm = np.cross(ac, bc); nrm /= np.linalg.norm(nrm)
                pert = nrm + rng.normal(size=3) * abs(o)
                rows.append(np.concatenate([pert * rng.uniform(0.5, 2.0), a, b, [rng.uniform(0.5, 2.0)]]))
            else:
                m = rng.normal(size=3) * 3
                mc = p_ref(aa, t, ref[p], nei[p], m)
                n = np.cross(mc, rng.normal(size=3)); n /= np.linalg.norm(n)
                n = n + rng.normal(size=3) * 0.05
                mid = mc / np.linalg.norm(mc) + rng.normal(size=3) * rng.choice([0.01, 0.3])
                rows.append(np.concatenate([n * rng.uniform(0.5, 2), [0.0], m, mid, [rng.uniform(0.005, 0.4)], [rng.uniform(0.5, 2.0)]]))
        off.append(len(rows))
    stride = {0: 7, 1: 7, 2: 9, 3: 9, 4: 10, 5: 12}[kind]
    rows = np.array(rows, np.float64).reshape(-1, stride)
    return rows, np.array(off, np.int64)


def oracle_rows(kind, rows, weight=1.0):
    """Converts ABI rows to the oracle's row layout (per-row weight column for kinds 0..3)."""
    if kind in (0, 1, 2, 3):
        return np.concatenate([rows, np.full((rows.shape[0], 1), weight)], axis=1)
    return rows


def expand_ids(pair_offsets, ref, nei):
    counts = np.diff(pair_offsets)
    return np.repeat(ref, counts).astype(np.int32), np.repeat(nei, counts).astype(np.int32)


def huber_weights(r, loss, a):
    s = r * r
    if loss == 0:
        return np.ones_like(r), 0.5 * s
    rho1 = np.ones_like(r); rho = s.copy()
    m = s > a * a
    rr = np.sqrt(s[m])
    rho[m] = 2 * a * rr - a * a
    rho1[m] = a / rr
    return rho1, 0.5 * rho


def pair_blocks_from_jacobian(r, J, pair_offsets, loss, a):
    """Reference assembly of the 121-double pair blocks from materialised r, J (numpy)."""
    P = len(pair_offsets) - 1
    out = np.zeros((P, 121))
    w, half_rho = huber_weights(r, loss, a)
    for p in range(P):
        s, e = pair_offsets[p], pair_offsets[p + 1]
        Jp = J[s:e]; rp = r[s:e]; wp = w[s:e]
        H = (Jp * wp[:, None]).T @ Jp
        g = (Jp * wp[:, None]).T @ rp
        out[p, 0:36] = H[0:6, 0:6].reshape(-1)
        out[p, 36:72] = H[0:6, 6:12].reshape(-1)
        out[p, 72:108] = H[6:12, 6:12].reshape(-1)
        out[p, 108:120] = g
        out[p, 120] = half_rho[s:e].sum()
    return out


# ------------------------------------------------------------------------------------------------
# segment-bearing scans (cornerLessSharp + edge_segmented + point_to_segment + segment_coeffs)
# ------------------------------------------------------------------------------------------------
from panovlm_amd.synthetic import make_line_scan, random_world_lines  # noqa: E402,F401  (shared with bench.py)


# ---- reprojection ("bundle") problems: PanoramaReprojResidual_1Angle blocks ---------------------------------
def pose_table(aa, t):
    """numpy restatement of the device pose table rows [R row-major | J_l(aa) row-major | t] (21 doubles)."""
    aa = np.asarray(aa, np.float64); t = np.asarray(t, np.float64)
    out = np.zeros((aa.shape[0], 21))
    for i, w in enumerate(aa):
        th2 = float(w @ w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        if th2 > 2.220446049250313e-16:
            th = np.sqrt(th2); k = K / th
            R = np.eye(3) + np.sin(th) * k + (1 - np.cos(th)) * (k @ k)
        else:
            R = np.eye(3) + K
        if th2 > 1e-6:
            th = np.sqrt(th2); A = (1 - np.cos(th)) / th2; B = (th - np.sin(th)) / (th2 * th)
        else:
            A = 0.5 - th2 / 24 + th2 * th2 / 720; B = 1 / 6 - th2 / 120 + th2 * th2 / 5040
        Jl = np.eye(3) + A * K + B * (K @ K)
        out[i, :9] = R.reshape(-1); out[i, 9:18] = Jl.reshape(-1); out[i, 18:] = t[i]
    return out


def random_bundle(rng, n_cams=5, n_points=30, min_track=2, max_track=5, noise=2e-3, outliers=0.1, empty_points=1):
    """Cameras looking at a cloud of points 2..6 m away; every point is seen by min..max_track cameras.
    Returns dict(aa, t [camera poses cw], X [perturbed points], off, cam, bearing [noisy, un-normalised])."""
    aa = rng.normal(size=(n_cams, 3)) * 0.3
    t = rng.normal(size=(n_cams, 3)) * 0.5
    Xt = rng.normal(size=(n_points, 3)) * 2.0 + np.array([0.0, 0.0, 1.0])
    tab = pose_table(aa, t)
    off = [0]; cam = []; bearing = []
    for p in range(n_points):
        k = 0 if p < empty_points else int(rng.integers(min_track, max_track + 1))
        cs = rng.choice(n_cams, size=min(k, n_cams), replace=False)
        if k > 2 and rng.uniform() < 0.2:
            cs = np.concatenate([cs, cs[:1]])          # the same camera twice in one track
        for c in cs:
            pc = tab[c, :9].reshape(3, 3) @ Xt[p] + tab[c, 18:]
            b = pc / np.linalg.norm(pc) + rng.normal(size=3) * (0.2 if rng.uniform() < outliers else noise)
            cam.append(int(c)); bearing.append(b * rng.uniform(0.5, 2.0))
        off.append(len(cam))
    X = Xt + rng.normal(size=Xt.shape) * 0.05
    return dict(aa=aa, t=t, X=X, off=np.array(off, np.int64), cam=np.array(cam, np.int32), bearing=np.array(bearing).reshape(-1, 3))


def bundle_reference(r, J, off, cam, n_cams, loss, a, scale, radius, min_diag, max_diag):
    """numpy Schur complement of the point blocks from materialised r, J (n x 9): returns dict with the dense
    reduced camera system S (6F x 6F), g (6F), cost, Udiag (F x 6), the per-point Vinv, gp, scale, gmax."""
    rho1, half = huber_weights(r, loss, a)
    F = n_cams; M = len(off) - 1
    U = np.zeros((6 * F, 6 * F)); gc = np.zeros(6 * F); Ud = np.zeros((F, 6))
    S = np.zeros_like(U); g = np.zeros_like(gc)
    Vinv = np.zeros((M, 3, 3)); gp = np.zeros((M, 3)); sc = np.zeros((M, 3)) if scale is None else np.array(scale, np.float64)
    for i in range(len(r)):
        c = cam[i]; Jc = J[i, :6]
        U[6 * c:6 * c + 6, 6 * c:6 * c + 6] += rho1[i] * np.outer(Jc, Jc)
        gc[6 * c:6 * c + 6] += rho1[i] * Jc * r[i]
        Ud[c] += rho1[i] * Jc * Jc
    S += U; g += gc
    for p in range(M):
        idx = np.arange(off[p], off[p + 1])
        V = np.zeros((3, 3)); W = np.zeros((6 * F, 3))
        for i in idx:
            Jp = J[i, 6:]; c = cam[i]
            V += rho1[i] * np.outer(Jp, Jp); gp[p] += rho1[i] * Jp * r[i]
            W[6 * c:6 * c + 6] += rho1[i] * np.outer(J[i, :6], Jp)
        if scale is None:
            sc[p] = 1.0 / (1.0 + np.sqrt(np.diag(V)))
        lam = np.clip(np.diag(V) * sc[p] ** 2, min_diag, max_diag) / (radius * sc[p] ** 2)
        Vd = V + np.diag(lam)
        Vinv[p] = np.linalg.inv(Vd)
        S -= W @ Vinv[p] @ W.T
        g -= W @ Vinv[p] @ gp[p]
    return dict(S=S, g=g, cost=float(half.sum()), Udiag=Ud, Vinv=Vinv, gp=gp, scale=sc, gmax=float(np.abs(gp).max()) if M else 0.0)


def bundle_unpack(packed, n_cams, ui, uj):
    """packed buffer of pvlm_ba_reduce -> dense (S 6F x 6F symmetric, g 6F, cost, Udiag F x 6, gmax); the
    un-eliminated camera gradient is packed[-1 - 6F:-1]."""
    F = n_cams; U = len(ui)
    Hd = packed[:F * 36].reshape(F, 6, 6); Ho = packed[F * 36:F * 36 + U * 36].reshape(U, 6, 6)
    o = F * 36 + U * 36
    g = packed[o:o + 6 * F].copy(); cost = float(packed[o + 6 * F]); Ud = packed[o + 6 * F + 1:o + 12 * F + 1].reshape(F, 6).copy()
    S = np.zeros((6 * F, 6 * F))
    for c in range(F):
        S[6 * c:6 * c + 6, 6 * c:6 * c + 6] = Hd[c]
    for u in range(U):
        a, b = ui[u], uj[u]
        S[6 * a:6 * a + 6, 6 * b:6 * b + 6] = Ho[u]; S[6 * b:6 * b + 6, 6 * a:6 * a + 6] = Ho[u].T
    return S, g, cost, Ud, float(packed[-1])


def covisible_pairs(off, cam):
    up = set()
    for p in range(len(off) - 1):
        cs = cam[off[p]:off[p + 1]]
        for i in range(len(cs)):
            for j in range(i + 1, len(cs)):
                if cs[i] != cs[j]:
                    up.add((min(cs[i], cs[j]), max(cs[i], cs[j])))
    up = sorted(up)
    return np.array([u[0] for u in up], np.int32), np.array([u[1] for u in up], np.int32)


# ---- panoramic MVS scenes: a textured box room rendered as equirectangular grey images -------------------------------
def _room_texture(P, frequency=1.0):
    x, y, z = P[..., 0] * frequency, P[..., 1] * frequency, P[..., 2] * frequency
    g = 128 + 45 * np.sin(3.1 * x + 1.3 * y) * np.cos(2.3 * z) + 35 * np.sin(6.7 * y + 2.1 * z + 0.5 * x) + 25 * np.cos(9.3 * x - 4.1 * z) * np.sin(5.9 * y)
    return np.clip(g, 0, 255)


def render_panorama(oracle, rows, cols, R_wc, t_wc, half=(4.0, 1.5, 6.0), texture_frequency=1.0):
    """Grey equirectangular image of the inside of the box [-half, half] seen from the camera pose (R_wc, t_wc), with the
    true depth (distance along the ray) and the surface normal in the camera frame, facing the camera."""
    jj, ii = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
    px = np.stack([jj.reshape(-1), ii.reshape(-1)], axis=1)
    dirs = oracle.image_to_cam(rows, cols, px, 1.0).astype(np.float64)                   # unit rays, camera frame
    dw = dirs @ np.asarray(R_wc, np.float64).T
    o = np.asarray(t_wc, np.float64)
    h = np.asarray(half)
    with np.errstate(divide="ignore", invalid="ignore"):
        tpos = np.where(dw > 0, (h - o) / dw, np.where(dw < 0, (-h - o) / dw, np.inf))
    axis = np.argmin(tpos, axis=1); t = tpos[np.arange(len(tpos)), axis]
    P = o + dw * t[:, None]
    n_w = np.zeros_like(P); n_w[np.arange(len(P)), axis] = -np.sign(dw[np.arange(len(P)), axis])
    gray = np.rint(_room_texture(P, texture_frequency)).astype(np.uint8).reshape(rows, cols)   # frequency > 1: finer texture for large images
    normal = (n_w @ np.asarray(R_wc, np.float64)).astype(np.float32).reshape(rows, cols, 3)
    return gray, t.astype(np.float32).reshape(rows, cols), normal


def relative_pose(R_wr, t_wr, R_wn, t_wn):
    """(R_nr, t_nr): X_n = R_nr X_r + t_nr — reference camera frame to neighbour camera frame (mvs NeighborInfo)."""
    R = np.asarray(R_wn).T @ np.asarray(R_wr)
    t = np.asarray(R_wn).T @ (np.asarray(t_wr) - np.asarray(t_wn))
    return R.astype(np.float32), t.astype(np.float32)


def cloud_scene(rng, rows, cols, max_depth=20.0):
    """Inputs of MVS::DepthImageToCloud with every branch present: invalid / far depths, sky-coloured, grey (delta = 0: hue NaN) and black pixels,
    colours on the edges of the sky box, a tilted pose."""
    depth = rng.uniform(0.3, 0.9 * max_depth, size=(rows, cols)).astype(np.float32)
    depth[rng.random((rows, cols)) < 0.15] = 0.0
    depth[rng.random((rows, cols)) < 0.03] = -1.0
    depth[0, :4] = np.float32(max_depth * 0.8) * np.array([1.0, 0.9999999, 1.0000001, 0.5], np.float32)
    bgr = rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)
    sky = rng.random((rows, cols)) < 0.3                                  # b > g > r, bright: hue 200..248 deg -> inside the box
    bgr[sky] = np.stack([rng.integers(200, 256, sky.sum()), rng.integers(120, 200, sky.sum()), rng.integers(60, 120, sky.sum())], 1).astype(np.uint8)
    grey = rng.random((rows, cols)) < 0.05
    bgr[grey] = bgr[grey][:, :1]
    bgr[rng.random((rows, cols)) < 0.02] = 0
    # the inclusive edges of the box: V = 150 / 255, S = 43 / 255 ..., reached by exact byte triples
    bgr[1, :6] = np.array([[150, 100, 75], [149, 100, 75], [255, 213, 212], [255, 212, 212], [255, 128, 55], [255, 128, 54]], np.uint8)
    normal = rng.normal(size=(rows, cols, 3)); normal = (normal / np.linalg.norm(normal, axis=2, keepdims=True)).astype(np.float32)
    T = np.eye(4); T[:3, :3] = rodrigues(np.array([0.3, -0.7, 0.2])); T[:3, 3] = [1.5, -0.25, 7.0]
    return depth, bgr, normal, T
