"""Pins the CPU oracle's residual functors (Jet AutoDiff restatement of base/CostFunction.h) against an
INDEPENDENT torch.float64 autograd transcription of the same formulas, and its fits / k-NN against
numpy / scipy.  (The reference ships no tests or golden vectors for these — SURVEY.md §4.)"""
import math

import numpy as np
import pytest
import torch



@pytest.fixture(autouse=True)
def _float64_default():
    """The transcription below builds its tensors without dtypes: float64 is the default for the duration of each test of THIS
    module only (a module-level torch.set_default_dtype leaked into every module collected after it)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def rodrigues(aa):
    th = torch.sqrt((aa * aa).sum())
    K = torch.zeros(3, 3)
    K[0, 1], K[0, 2], K[1, 0], K[1, 2], K[2, 0], K[2, 1] = -aa[2], aa[1], aa[2], -aa[0], -aa[1], aa[0]
    if th.item() < 1e-12:
        return torch.eye(3) + K
    return torch.eye(3) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * (K @ K)


def p_ref(aa_r, t_r, aa_n, t_n, P):
    return rodrigues(aa_r) @ rodrigues(-aa_n) @ (P - t_n) + t_r


def vangle(a, b):
    c = (a * b).sum() / (a.norm() * b.norm())
    if c.item() >= 1.0:
        return c * 0
    if c.item() <= -1.0:
        return c * 0 + math.pi
    return torch.acos(c)


def normalized_angle(P, Pp, normalize):
    if normalize:
        n = Pp.norm()
        c = Pp * (n - 1.0) / n
        return vangle(Pp - c, P - c)
    return vangle(P, Pp)


def torch_residual(kind, rec, normalize, aa_r, t_r, aa_n, t_n):
    rec = torch.tensor(rec)
    if kind in (0, 1):
        P = p_ref(aa_r, t_r, aa_n, t_n, rec[0:3]); n = rec[3:6]; d = rec[6]; w = rec[7]
        sd = (n * P).sum() + d
        if kind == 0:
            return w * sd.abs()
        dis = sd.abs()
        if dis.item() < 1e-3:
            return dis * 0
        Pp = P - dis * n
        if abs(((n * Pp).sum() + d).item()) > 1e-4:
            Pp = P + dis * n
        return normalized_angle(P, Pp, normalize)
    if kind in (2, 3):
        P = p_ref(aa_r, t_r, aa_n, t_n, rec[0:3]); A = rec[3:6]; B = rec[6:9]; w = rec[9]
        dirv = (A - B) / (A - B).norm()
        k = (dirv * (P - A)).sum()
        Pp = A + k * dirv
        dis = (P - Pp).norm()
        if kind == 2:
            return w * dis
        if dis.item() < 1e-3:
            return dis * 0
        return normalized_angle(P, Pp, normalize)
    if kind == 4:
        n_img = rec[0:3] / rec[0:3].norm(); w = rec[9]
        a = p_ref(aa_r, t_r, aa_n, t_n, rec[3:6]); b = p_ref(aa_r, t_r, aa_n, t_n, rec[6:9])
        nrm = torch.linalg.cross(a, b)
        c = (n_img * nrm).sum().abs() / nrm.norm()
        if c.item() >= 1.0:
            return c * 0
        return w * torch.acos(c)
    if kind == 5:
        pl = rec[0:4] / rec[0:3].norm(); w = rec[11]; ang = rec[10]
        m = p_ref(aa_r, t_r, aa_n, t_n, rec[4:7]); mr = rec[7:10]
        n = pl[0:3]; d = pl[3]
        dis = ((n * m).sum() + d).abs()
        mp = m - dis * n
        if abs(((n * mp).sum() + d).item()) > 1e-4:
            mp = m + dis * n
        cur = vangle(mp, mr)
        if cur.item() < ang.item():
            return cur * 0
        return w * (cur - ang)
    raise ValueError(kind)


def random_case(rng, kind):
    aa_r = rng.normal(size=3) * 0.4; aa_n = rng.normal(size=3) * 0.4
    t_r = rng.normal(size=3) * 2.0; t_n = rng.normal(size=3) * 2.0
    P = rng.normal(size=3) * 4.0
    w = rng.uniform(0.5, 2.0)
    if kind in (0, 1):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        # place the plane near the transformed point so residuals are ICP-like (cm..dm)
        Pr = p_ref(torch.tensor(aa_r), torch.tensor(t_r), torch.tensor(aa_n), torch.tensor(t_n), torch.tensor(P)).numpy()
        off = rng.choice([1e-4, 5e-3, 0.05, 0.5]) * rng.choice([-1, 1])
        d = -(n @ Pr) + off
        rec = np.concatenate([P, n, [d], [w]])
    elif kind in (2, 3):
        Pr = p_ref(torch.tensor(aa_r), torch.tensor(t_r), torch.tensor(aa_n), torch.tensor(t_n), torch.tensor(P)).numpy()
        dirv = rng.normal(size=3); dirv /= np.linalg.norm(dirv)
        off = rng.normal(size=3); off -= (off @ dirv) * dirv; off /= np.linalg.norm(off)
        A0 = Pr + off * rng.choice([1e-4, 5e-3, 0.05, 0.5]) + dirv * rng.normal()
        rec = np.concatenate([P, A0 + 0.1 * dirv, A0 - 0.1 * dirv, [w]])
    elif kind == 4:
        rec = np.concatenate([rng.normal(size=3), rng.normal(size=3) * 3, rng.normal(size=3) * 3, [w]])
    else:
        n = rng.normal(size=3)
        rec = np.concatenate([n, [0.0], rng.normal(size=3) * 3, rng.normal(size=3), [rng.uniform(0.01, 1.5)], [w]])
    return rec, aa_r, t_r, aa_n, t_n


@pytest.mark.parametrize("kind,normalize", [(0, False), (1, False), (1, True), (2, False), (3, False), (3, True), (4, False), (5, False)])
def test_functor_matches_torch_autograd(oracle, kind, normalize):
    rng = np.random.default_rng(20240601 + kind * 7 + int(normalize))
    n_zero = 0
    for _ in range(40):
        rec, aa_r, t_r, aa_n, t_n = random_case(rng, kind)
        aa = np.stack([aa_r, aa_n]); t = np.stack([t_r, t_n])
        r, J = oracle.evaluate(kind, rec[None, :], [0], [1], aa, t, normalize=normalize)
        params = [torch.tensor(x, requires_grad=True) for x in (aa_r, t_r, aa_n, t_n)]
        rt = torch_residual(kind, rec, normalize, *params)
        rt.backward()
        Jt = np.concatenate([p.grad.numpy() if p.grad is not None else np.zeros(3) for p in params])
        assert abs(r[0] - rt.item()) <= 1e-10 * max(1.0, abs(rt.item()))
        if rt.item() == 0.0:
            n_zero += 1
            assert np.all(J[0] == 0) or kind == 0
            continue
        scale = max(1.0, np.abs(Jt).max())
        assert np.abs(J[0] - Jt).max() <= 2e-7 * scale, (kind, normalize, J[0], Jt)
    if kind in (1, 3):
        assert n_zero > 0  # the dis < 1e-3 early-out must be exercised


def test_cost_only_path_matches(oracle):
    rng = np.random.default_rng(5)
    rec, aa_r, t_r, aa_n, t_n = random_case(rng, 1)
    aa = np.stack([aa_r, aa_n]); t = np.stack([t_r, t_n])
    r1, _ = oracle.evaluate(1, rec[None, :], [0], [1], aa, t, normalize=True, jac=True)
    r2, J2 = oracle.evaluate(1, rec[None, :], [0], [1], aa, t, normalize=True, jac=False)
    assert J2 is None and r1[0] == r2[0]


def test_rotation_roundtrip(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        aa = rng.normal(size=3) * rng.choice([1e-9, 1e-3, 0.5, 2.5])
        R = oracle.angle_axis_to_matrix(aa)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert np.allclose(R, rodrigues(torch.tensor(aa)).numpy(), atol=1e-12)
        aa2 = oracle.matrix_to_angle_axis(R)
        assert np.allclose(oracle.angle_axis_to_matrix(aa2), R, atol=1e-9)
        if np.linalg.norm(aa) < 3.0:
            assert np.allclose(aa2, aa, atol=1e-9)


def test_huber(oracle):
    a = 2 * math.pi / 180
    s = np.array([0.0, 1e-6, a * a, a * a * 1.0001, 0.1, 4.0])
    rho = oracle.huber(a, s)
    for si, r in zip(s, rho):
        if si <= a * a:
            assert r[0] == si and r[1] == 1.0 and r[2] == 0.0
        else:
            assert np.isclose(r[0], 2 * a * math.sqrt(si) - a * a) and np.isclose(r[1], a / math.sqrt(si))


def test_plane_fit_matches_lstsq(oracle):
    rng = np.random.default_rng(2)
    for _ in range(100):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        c = rng.normal(size=3) * 5 + n * 3
        u = np.cross(n, rng.normal(size=3)); u /= np.linalg.norm(u); v = np.cross(n, u)
        pts = c + rng.normal(size=(10, 1)) * 0.3 * u + rng.normal(size=(10, 1)) * 0.3 * v + rng.normal(size=(10, 1)) * 0.004 * n
        ok, plane = oracle.form_plane_lsq(pts, 0.05)
        x = np.linalg.lstsq(pts, -np.ones(10), rcond=None)[0]
        d = 1 / np.linalg.norm(x); x = x / np.linalg.norm(x)
        assert ok
        assert np.allclose(plane[:3], x, atol=1e-9) and abs(plane[3] - d) < 1e-8
        ok2, plane2 = oracle.form_plane_lsq(pts, 1e-4)
        assert not ok2 and np.all(plane2 == 0)


def test_line_test_matches_eigh(oracle):
    rng = np.random.default_rng(3)
    flips = 0
    for _ in range(200):
        pts = rng.normal(size=(10, 3)) * np.array([1.0, rng.uniform(0.2, 1.0), rng.uniform(0.01, 0.5)])
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        pts = pts @ Q.T + rng.normal(size=3) * 4
        ok, line = oracle.form_line_pca(pts, 3.0)
        c = pts.mean(0); S = (pts - c).T @ (pts - c)
        w, V = np.linalg.eigh(S)
        ww, VV = oracle.eig_sym3(S)
        assert np.allclose(w, ww, rtol=1e-10, atol=1e-12)
        expect = w[2] > 3.0 * w[1]
        if abs(w[2] - 3 * w[1]) > 1e-9:
            assert ok == expect
        if ok:
            assert np.allclose(line[:3], c) and abs(abs(line[3:] @ V[:, 2]) - 1) < 1e-9
            flips += 1
    assert 0 < flips < 200


def test_knn_matches_ckdtree(oracle):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(4)
    tgt = (rng.normal(size=(3000, 3)) * 3).astype(np.float32)
    q = (rng.normal(size=(500, 3)) * 3).astype(np.float32)
    idx, sqd = oracle.knn(tgt, q, 10)
    _, ref = cKDTree(tgt.astype(np.float64)).query(q.astype(np.float64), k=10)
    # float32 vs float64 distance rounding can swap near-equal neighbours: compare as sets, allow rare diffs
    same = sum(set(a) == set(b) for a, b in zip(idx, ref))
    assert same >= 498
    assert np.all(np.diff(sqd, axis=1) >= 0)
    d = ((q[:, None, :] - tgt[idx]) ** 2)
    manual = (d[..., 0] + d[..., 1]) + d[..., 2]
    assert np.array_equal(manual.astype(np.float32), sqd)


def test_calibration_functors_are_the_global_ones_at_an_identity_pose(oracle):
    """Plane2Plane_Relative / PlaneRelativeIOUResidual (base/CostFunction.h:294-348, :509-565: ONE pose, the calibration mode of
    CameraLidarOptimizer::Optimize) against Plane2Plane_Global / PlaneIOUResidual with the second pose at the identity — the form
    in which the product evaluates them (kinds 4 and 5, the first six columns of the Jacobian row): the IOU term is the same to
    the last bit, the plane term differs only by where the constant 180 / pi is multiplied in."""
    rng = np.random.default_rng(17)
    n = 400
    aa = rng.normal(size=3) * 0.4; t = rng.normal(size=3)
    A = np.stack([aa, np.zeros(3)]); T = np.stack([t, np.zeros(3)])
    r4 = np.concatenate([rng.normal(size=(n, 3)), rng.normal(size=(n, 3)) * 3, rng.normal(size=(n, 3)) * 3, np.full((n, 1), 1.3)], axis=1)
    rr, Jr = oracle.evaluate_relative(4, r4, aa, t)
    g4 = r4.copy(); g4[:, 9] = 1.3 * 180.0 / np.pi
    rg, Jg = oracle.evaluate(4, g4, [0] * n, [1] * n, A, T)
    assert np.abs(rr - rg).max() <= 1e-14 * np.abs(rr).max() and np.abs(Jr - Jg[:, :6]).max() <= 1e-14 * np.abs(Jr).max()
    assert rr.min() >= 0 and rr.max() > 10                                    # degrees
    r5 = np.concatenate([rng.normal(size=(n, 3)), np.zeros((n, 1)), rng.normal(size=(n, 3)) * 3, rng.normal(size=(n, 3)), rng.uniform(0, 1.5, size=(n, 1)),
                         np.full((n, 1), 2.0)], axis=1)
    rr, Jr = oracle.evaluate_relative(5, r5, aa, t)
    rg, Jg = oracle.evaluate(5, r5, [0] * n, [1] * n, A, T)
    assert np.array_equal(rr, rg) and np.array_equal(Jr, Jg[:, :6])
    assert 0.1 < (rr > 0).mean() < 0.9                                        # both sides of the half-arc threshold
    # cost-only path
    assert np.abs(oracle.evaluate_relative(5, r5, aa, t, jac=False)[0] - rr).max() <= 1e-13


def test_reproj_functor_matches_torch_autograd(oracle):
    """PanoramaReprojResidual_1Angle (base/CostFunction.h:218-247): the oracle's Jet<9> AutoDiff against torch.float64
    autograd through an independent Rodrigues implementation."""
    rng = np.random.default_rng(12)
    F, M, N = 4, 10, 40
    aa = rng.normal(size=(F, 3)) * 0.5; t = rng.normal(size=(F, 3)); X = rng.normal(size=(M, 3)) * 3 + np.array([0, 0, 5.0])
    cam = rng.integers(0, F, N); pt = rng.integers(0, M, N); b = rng.normal(size=(N, 3))
    r, J = oracle.evaluate_reproj(b, 1.7, cam, pt, aa, t, X)

    def rod(w):
        th = w.norm(); k = w / th
        K = torch.zeros(3, 3, dtype=torch.float64)
        K[0, 1] = -k[2]; K[0, 2] = k[1]; K[1, 0] = k[2]; K[1, 2] = -k[0]; K[2, 0] = -k[1]; K[2, 1] = k[0]
        return torch.eye(3, dtype=torch.float64) + torch.sin(th) * K + (1 - torch.cos(th)) * K @ K
    for i in range(N):
        a_ = torch.tensor(aa[cam[i]], requires_grad=True); t_ = torch.tensor(t[cam[i]], requires_grad=True); x_ = torch.tensor(X[pt[i]], requires_grad=True)
        s = torch.tensor(b[i] / np.linalg.norm(b[i]))
        p = rod(a_) @ x_ + t_
        rr = 1.7 * torch.acos((p @ s) / p.norm())
        rr.backward()
        Jt = np.concatenate([a_.grad.numpy(), t_.grad.numpy(), x_.grad.numpy()])
        assert abs(rr.item() - r[i]) <= 1e-12 and np.abs(Jt - J[i]).max() <= 1e-10 * max(1.0, np.abs(Jt).max())


def test_point2line_variants_match_independent_search(oracle):
    """AssociatePoint2Line / ...SegmentKNN / ...Segment / AssociateLine2LineKNN of the oracle against scipy's kd-tree,
    numpy eigh and brute-force numpy distances."""
    from scipy.spatial import cKDTree
    from panovlm_amd import synthetic as sy
    rng = np.random.default_rng(33)
    lines = sy.random_world_lines(rng, 10)
    Ra, ta = sy.estimated_pose(2); Rb, tb = sy.estimated_pose(3)
    ref = sy.make_line_scan(rng, 0, Ra, ta, lines, pts_per_line=(10, 24), extra_pts=12, noise=0.005)
    nei = sy.make_line_scan(rng, 1, Rb, tb, lines, pts_per_line=(10, 24), extra_pts=12, noise=0.005)
    thr = np.float32(0.4)
    tree = cKDTree(ref["corner_xyz"].astype(np.float64))
    q = nei["corner_xyz"].astype(np.float64)
    d, idx = tree.query(q, k=5)
    near = (d[:, 4].astype(np.float32) ** 2 <= thr * thr + 1e-6)      # float32 distances in the oracle: keep away from the threshold
    far = (d[:, 4].astype(np.float32) ** 2 >= thr * thr - 1e-6)
    assert not np.any(near & far & (np.abs(d[:, 4] ** 2 - float(thr) ** 2) > 1e-5))
    # --- SegmentKNN: all five neighbours on one ref segment
    o = oracle.assoc_point2line(ref, nei, float(thr), mode="segment_knn")
    exp = []
    for i in range(len(q)):
        if d[i, 4] ** 2 > float(thr) ** 2:
            continue
        cnt = {}
        for j in idx[i]:
            for sgm in ref["p2s"][j]:
                cnt[sgm] = cnt.get(sgm, 0) + 1
        for sgm in sorted(cnt):
            if cnt[sgm] >= 5:
                exp.append((i, sgm))
    assert [int(v) for v in o["qidx"]] == [e[0] for e in exp] and len(exp) > 20
    for k, (i, sgm) in enumerate(exp):
        c = ref["seg_coeffs"][sgm]
        assert np.allclose(o["a"][k], c[:3] + 0.1 * c[3:], atol=1e-12) and np.allclose(o["b"][k], c[:3] - 0.1 * c[3:], atol=1e-12)
        assert np.allclose(o["point"][k], Rb.T @ (q[i] - tb), atol=1e-9)
    # --- plain k-NN + PCA line test (FormLine(points, 10.0, 0.05))
    o = oracle.assoc_point2line(ref, nei, float(thr), mode="knn")
    exp = []
    for i in range(len(q)):
        if d[i, 4] ** 2 > float(thr) ** 2:
            continue
        P = ref["corner_xyz"][idx[i]].astype(np.float64)
        c = P.mean(0); w, V = np.linalg.eigh((P - c).T @ (P - c))
        if not w[2] > 10.0 * w[1]:
            continue
        dirv = V[:, 2] / np.linalg.norm(V[:, 2])
        if np.any(np.linalg.norm(np.cross(P - c, dirv), axis=1) > 0.05):
            continue
        exp.append((i, c, dirv))
    assert [int(v) for v in o["qidx"]] == [e[0] for e in exp] and len(exp) > 20
    for k, (i, c, dirv) in enumerate(exp):
        pa, pb = Ra.T @ (c + 0.1 * dirv - ta), Ra.T @ (c - 0.1 * dirv - ta)
        assert (np.allclose(o["a"][k], pa, atol=1e-9) and np.allclose(o["b"][k], pb, atol=1e-9)) or \
               (np.allclose(o["a"][k], pb, atol=1e-9) and np.allclose(o["b"][k], pa, atol=1e-9))      # eigenvector sign
    # --- Segment: nearest ref line (world) by point-to-line distance
    o = oracle.assoc_point2line(ref, nei, float(thr), mode="segment")
    lw_p = ref["seg_coeffs"][:, :3] @ Ra.T + ta; lw_d = ref["seg_coeffs"][:, 3:] @ Ra.T
    dist = np.array([[np.linalg.norm(np.cross(p - lp, ld)) / np.linalg.norm(ld) for lp, ld in zip(lw_p, lw_d)] for p in q])
    best = dist.argmin(1); keep = dist.min(1) <= float(thr)
    assert [int(v) for v in o["qidx"]] == np.flatnonzero(keep).tolist()
    for k, i in enumerate(np.flatnonzero(keep)):
        c = ref["seg_coeffs"][best[i]]
        assert np.allclose(o["a"][k], c[:3] + 0.1 * c[3:], atol=1e-12)
    # --- Line2LineKNN vote matrix: >= 3 of the 5 neighbours on one ref segment
    ol = oracle.assoc_line2line(ref, nei, float(thr), knn=True)
    votes = np.zeros((len(nei["seg_size"]), len(ref["seg_size"])), np.int32)
    for i in range(len(q)):
        if d[i, 4] ** 2 > float(thr) ** 2:
            continue
        cnt = {}
        for j in idx[i]:
            for sgm in ref["p2s"][j]:
                cnt[sgm] = cnt.get(sgm, 0) + 1
        for sgm, c_ in cnt.items():
            if c_ >= 3:
                for ns in nei["p2s"][i]:
                    votes[ns, sgm] += 1
    assert np.array_equal(ol["votes"], votes) and votes.sum() > 50


def test_project_lidar_depth_matches_numpy(oracle):
    """ProjectLidar2PanoramaDepth of the oracle against a vectorised numpy implementation of the same painting rule
    (true atan2 replaced by the oracle's own CamToImage for the pixel; window, border rejection, last-point-wins)."""
    from panovlm_amd import synthetic as sy
    rows, cols, size = 360, 720, 3
    s = sy.make_scan(4, cols=128)
    xyz = np.concatenate([s["local_xyz"], s["local_xyz"][::3] * np.float32(1.03)])
    T = np.eye(4); T[:3, 3] = [0.02, -0.01, 0.03]
    img = oracle.project_lidar_depth(rows, cols, xyz, T, size)
    p = np.stack([(T[r, 0] * xyz[:, 0].astype(np.float64) + T[r, 1] * xyz[:, 1] + T[r, 2] * xyz[:, 2] + T[r, 3]).astype(np.float32) for r in range(3)], axis=1)
    px = oracle.cam_to_image(rows, cols, p)
    exp = np.zeros((rows, cols), np.uint16)
    half = size // 2
    for i in range(len(p)):
        rbx, rby = int(np.ceil(px[i, 0]) + half), int(np.ceil(px[i, 1]) + half)
        ltx, lty = int(np.floor(px[i, 0]) - half), int(np.floor(px[i, 1]) - half)
        if not (0 <= rbx < cols and 0 <= rby < rows and 0 <= ltx < cols and 0 <= lty < rows):
            continue
        depth = np.sqrt(p[i, 0] * p[i, 0] + p[i, 1] * p[i, 1] + p[i, 2] * p[i, 2], dtype=np.float32)
        exp[lty:rby + 1, ltx:rbx + 1] = np.uint16(np.float64(depth) * 256.0)
    assert np.array_equal(img, exp) and (img > 0).mean() > 0.05


def test_mvs_score_pixel_matches_numpy(oracle):
    """ScorePixel of the oracle (bilateral patch, plane-induced homography, bilinear samples, weighted NCC, smoothness
    factors, best-two average; mvs/MVS.cpp:637-680, :774-923) against an independent float64 numpy implementation.  Only the
    equirectangular projection is shared (oracle.cam_to_image, itself pinned against the real base/Math.h)."""
    import ctypes as C
    from tests.test_mvs_cpu import mvs_scene
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle, 96, 192)
    rows, cols = gray.shape
    hw, L = 3, oracle.lib()
    L.orc_mvs_score_pixel.restype = C.c_float
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ptrs = (C.POINTER(C.c_ubyte) * len(neis))(*[np.ascontiguousarray(g).ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
    R32 = np.ascontiguousarray(Rn, np.float32); t32 = np.ascontiguousarray(tn, np.float32)
    unit = lambda xs, ys: oracle.image_to_cam(rows, cols, np.stack([xs, ys], 1).astype(np.float32), 1.0).astype(np.float64)
    rng = np.random.default_rng(12)
    dy, dx = np.mgrid[-hw:hw + 1, -hw:hw + 1]
    checked, diffs = 0, []
    for _ in range(60):
        px, py = int(rng.integers(hw + 2, cols - hw - 2)), int(rng.integers(hw + 12, rows - hw - 12))
        e_depth = float(depth[py, px])
        if e_depth <= 0:
            continue
        dep = e_depth * float(rng.uniform(0.97, 1.03))
        nrm = normal[py, px].astype(np.float64) + 0.1 * rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        close = []
        for (cx, cy) in ((px - 1, py), (px, py - 1), (px, py + 1), (px + 1, py))[:int(rng.integers(0, 5))]:
            if depth[cy, cx] > 0:
                close.append(np.concatenate([unit(np.array([cx]), np.array([cy]))[0] * depth[cy, cx], normal[cy, cx], [depth[cy, cx]]]))
        cl = np.ascontiguousarray(np.array(close, np.float32).reshape(-1, 7))
        got = L.orc_mvs_score_pixel(C.c_int(rows), C.c_int(cols), C.c_int(hw), C.c_int(1), gray.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(neis)), ptrs,
                                    fp(R32), fp(t32), C.c_int(px), C.c_int(py), fp(nrm.astype(np.float32)), C.c_float(dep), None, C.c_int(len(cl)), fp(cl) if len(cl) else None)
        # ---- numpy
        nrm = nrm.astype(np.float32).astype(np.float64); dep = float(np.float32(dep))
        tex = gray[py + dy, px + dx].astype(np.float64).ravel()
        w = np.exp(((tex - float(gray[py, px])) / 255.0) ** 2 * (-1 / (2 * 0.2 * 0.2)) + (dx.ravel() ** 2 + dy.ravel() ** 2) * (-1 / (2.0 * hw * hw)))
        w /= w.sum()
        t0 = tex - (w * tex).sum()
        sq0 = (t0 * t0 * w).sum()
        X0 = unit(np.array([px]), np.array([py]))[0] * dep
        d = X0 @ nrm
        if d > 0 or sq0 <= 1e-6:
            assert got < 0
            continue
        u = unit((px + dx).ravel(), (py + dy).ravel())
        plane = np.concatenate([nrm, [-(nrm @ X0)]])
        factor = 1.0
        fs = []
        for c in cl.astype(np.float64):
            dd = abs(plane[:3] @ c[:3] + plane[3]) / dep
            ang = np.arccos(np.clip(nrm @ c[3:6], -1, 1))
            fs.append((1 - 0.05 * np.exp(dd * dd * (-1 / (2 * 0.02 ** 2)))) * (1 - 0.05 * 0.96 * np.exp(ang * ang * (-1 / (2 * 0.22 ** 2)))))
        scores = []
        for b in range(len(neis)):
            H = Rn[b].reshape(3, 3).astype(np.float64) + np.outer(tn[b].astype(np.float64), nrm) / d
            x1 = oracle.cam_to_image(rows, cols, (u @ H.T).astype(np.float32)).astype(np.float64)
            if not np.all((x1[:, 0] >= 1) & (x1[:, 1] >= 1) & (x1[:, 0] < cols - 1) & (x1[:, 1] < rows - 1)):
                continue
            lx, ly = np.floor(x1[:, 0]).astype(int), np.floor(x1[:, 1]).astype(int)
            fx, fy = x1[:, 0] - lx, x1[:, 1] - ly
            g = neis[b].astype(np.float64)
            t1 = (g[ly, lx] * (1 - fx) + g[ly, lx + 1] * fx) * (1 - fy) + (g[ly + 1, lx] * (1 - fx) + g[ly + 1, lx + 1] * fx) * fy
            t1 = t1 - (t1 * w).sum()
            nrm2 = sq0 * (t1 * t1 * w).sum()
            if nrm2 <= 0:
                continue
            s = float(np.clip((t0 * w * t1).sum() / np.sqrt(nrm2), -1, 1))
            if fs:
                s = 1 - s
                for f in fs:
                    s *= f
                s = float(np.clip(1 - s, -1, 1))
            scores.append(s)
        want = -1.0 if not scores else (scores[0] if len(scores) == 1 else float(np.mean(sorted(scores, reverse=True)[:2])))
        diffs.append(abs(got - want)); checked += 1
    diffs = np.array(diffs)
    assert checked > 30 and np.median(diffs) < 1e-6 and diffs.max() < 1e-4, (checked, np.median(diffs), diffs.max())     # observed: 6e-8 / 2.4e-7


def test_depth_fusion_filters_match_numpy(oracle):
    """FilterDepthImage and FilterDepthImageRefine of the oracle (mvs/MVS.cpp:1735-1890) against a numpy restatement of
    the per-pixel rules, on top of the oracle's own forward projection (checked separately in tests/test_mvs_cpu.py)."""
    from tests.test_mvs_cpu import _refine_scene
    nd, nc, Rn, tn, depth, conf, const = _refine_scene(oracle, 48, 96)
    rows, cols = depth.shape
    proj = [oracle.mvs_project_depth_conf(nd[b], nc[b], Rn[b], tn[b]) for b in range(len(nd))]
    f32 = np.float32
    thr = f32(0.01)
    loose, strict = thr * f32(1.2), thr * f32(0.8)
    # ---- FilterDepthImage: >= 2 neighbours agree at the pixel (strict), >= 5 (neighbour, 4-neighbourhood) samples agree (loose) or depth_constant
    want = np.zeros_like(depth)
    for r in range(rows):
        for c in range(cols):
            d = depth[r, c]
            if d <= 0:
                continue
            if sum(1 for pd, _ in proj if pd[r, c] > 0 and abs((d - pd[r, c]) / d) < strict) < 2:
                continue
            votes = 0
            for pd, _ in proj:
                for dc, dr in ((-1, 0), (1, 0), (0, 1), (0, -1)):
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < rows and 0 <= cc < cols and pd[rr, cc] > 0 and abs((d - pd[rr, cc]) / d) < loose:
                        votes += 1
            if votes >= 5 or const[r, c]:
                want[r, c] = d
    got, gotc = oracle.mvs_filter_depth(nd, Rn, tn, depth, conf=conf, depth_constant=const, thr=0.01)
    assert np.array_equal(got, want) and np.array_equal(gotc, np.where(want > 0, conf, 0)) and 0.1 < (want > 0).mean() < 0.95
    # ---- FilterDepthImageRefine
    px = np.stack(np.meshgrid(np.arange(cols), np.arange(rows)), -1).reshape(-1, 2).astype(np.float32)
    unit = oracle.image_to_cam(rows, cols, px, 1.0).reshape(rows, cols, 3)
    min_d, max_d = f32(0.1), f32(3.0)
    wd = np.zeros_like(depth); wc = np.zeros_like(depth); conf_after = conf.copy()
    for r in range(rows):
        for c in range(cols):
            d = depth[r, c]
            if d <= 0:
                conf_after[r, c] = 0
                continue
            pos, neg, avg, npos, nneg, bad = conf[r, c], f32(0), d * conf[r, c], 0, 0, False
            for n in range(len(proj) - 1, -1, -1):
                dn, cn = proj[n][0][r, c], proj[n][1][r, c]
                if dn <= 0 and npos + nneg + n < 2:
                    bad = True
                    break
                if abs((d - dn) / d) < loose:
                    avg = f32(avg + dn * cn); pos = f32(pos + cn); npos += 1
                else:
                    if dn < d:
                        neg = f32(neg + cn)                                   # occlusion
                    else:                                                     # free-space violation: the neighbour's own confidence where we land in it
                        X1 = (Rn[n].reshape(3, 3).astype(f32) @ (unit[r, c] * d).astype(f32) + tn[n].astype(f32)).astype(f32)
                        x1 = oracle.cam_to_image(rows, cols, X1[None, :])[0]
                        xr, yr = int(np.floor(x1[0] + 0.5)), int(np.floor(x1[1] + 0.5))           # round half away from zero (positive values)
                        cc = nc[n][yr, xr] if (0 <= xr < cols and 0 <= yr < rows) else f32(0)
                        neg = f32(neg + (cc if cc > 0 else cn))
                    nneg += 1
            if not bad:
                with np.errstate(divide="ignore", invalid="ignore"):
                    avg = f32(avg / pos)
                if npos >= 2 and pos > neg and min_d <= avg <= max_d:
                    wd[r, c] = avg; wc[r, c] = f32(pos - neg)
                    continue
            if const[r, c]:
                wd[r, c] = d; wc[r, c] = 1
    gd, gc, ga = oracle.mvs_filter_depth_refine(nd, nc, Rn, tn, depth, conf, depth_constant=const, thr=0.01, min_depth=0.1, max_depth=3.0)
    assert np.array_equal(ga, conf_after)
    same = (np.abs(gd - wd) <= 1e-6 * np.maximum(wd, 1)) & (np.abs(gc - wc) <= 1e-5)
    assert same.all() and np.array_equal(gd > 0, wd > 0)
    assert 0.05 < (wd > 0).mean() < 0.95
