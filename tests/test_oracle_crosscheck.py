"""Pins the CPU oracle's residual functors (Jet AutoDiff restatement of base/CostFunction.h) against an
INDEPENDENT torch.float64 autograd transcription of the same formulas, and its fits / k-NN against
numpy / scipy.  (The reference ships no tests or golden vectors for these — SURVEY.md §4.)"""
import math

import numpy as np
import pytest
import torch

torch.set_default_dtype(torch.float64)


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
