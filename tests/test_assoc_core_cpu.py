"""The per-query bodies of the association kernels (csrc/pvlm_assoc_core.h), compiled for the host by
tests/cpp/assoc_core_check.cpp and driven serially: the pruned voxel-grid search must return the exact, tie-broken
k-NN of the oracle's brute force (indices AND float32 distances, bit for bit), and the plane fit / certified
collinearity test must take the oracle's decisions.  No GPU: this is where a change of the search or of the fits is
validated before it is measured."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from panovlm_amd import synthetic as sy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("assoc_core") / "assoc_core_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", out,
                           os.path.join(ROOT, "tests", "cpp", "assoc_core_check.cpp")])
    lib = ctypes.CDLL(out)
    lib.chk_knn.restype = ctypes.c_int
    lib.chk_line_sweeps.restype = ctypes.c_longlong
    lib.chk_fit_fast.restype = ctypes.c_longlong
    lib.chk_is_line_fast.restype = ctypes.c_longlong
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def knn(lib, tgt, q, k, max_dist, cell=0.0, force_hash=0, xf=4):
    tgt = np.ascontiguousarray(tgt, np.float32); q = np.ascontiguousarray(q, np.float32)
    idx = np.empty((len(q), k), np.int32); sqd = np.empty((len(q), k), np.float32); stats = np.zeros(3, np.int64)
    rc = lib.chk_knn(_p(tgt, ctypes.c_float), len(tgt), _p(q, ctypes.c_float), len(q), k, ctypes.c_float(max_dist), ctypes.c_float(cell), force_hash, xf,
                     _p(idx, ctypes.c_int), _p(sqd, ctypes.c_float), _p(stats, ctypes.c_longlong))
    assert rc == 0
    return idx, sqd, stats


def check(lib, oracle, tgt, q, k, max_dist, **kw):
    idx, sqd, stats = knn(lib, tgt, q, k, max_dist, **kw)
    oi, od = oracle.knn(tgt, q, k)
    thr2 = np.float32(max_dist) * np.float32(max_dist)
    valid = od <= thr2
    exp_i = np.where(valid, oi, -1)
    exp_d = np.where(valid, od, np.float32(np.inf))
    bad = np.argwhere(idx != exp_i)
    assert len(bad) == 0, (bad[:5], idx[idx != exp_i][:5], exp_i[idx != exp_i][:5], kw)
    assert np.array_equal(sqd, exp_d)
    return stats


def test_search_equals_brute_force_random_cloud(chk, oracle):
    rng = np.random.default_rng(11)
    tgt = (rng.normal(size=(5000, 3)) * 2).astype(np.float32)
    q = (rng.normal(size=(1500, 3)) * 2.2).astype(np.float32)
    for kw in ({}, {"force_hash": 1}, {"cell": 0.11}, {"cell": 0.9}, {"cell": 3.0}, {"cell": 0.35, "force_hash": 1}, {"xf": 1}, {"xf": 3, "cell": 0.5}, {"xf": 16, "cell": 1.1}):
        check(chk, oracle, tgt, q, 10, 1.0, **kw)
        check(chk, oracle, tgt, q, 5, 0.3, **kw)
    # a threshold larger than the cloud, queries far outside the grid
    check(chk, oracle, tgt, q * 4, 10, 30.0, cell=1.5)
    check(chk, oracle, tgt[:300], q * 3, 5, 2.0)


def test_search_vlp_geometry_ties_and_zero_distances(chk, oracle):
    a = sy.make_scan(2, cols=512)["flat_xyz"]
    b = sy.make_scan(3, cols=512)["flat_xyz"]
    s = check(chk, oracle, a, b[::3], 10, 1.0)
    check(chk, oracle, a, b[::3], 10, 1.0, xf=1)
    check(chk, oracle, a, b[::3], 10, 1.0, xf=7)
    assert s[2] == 1                                  # the dense table is what the bench clouds use
    check(chk, oracle, a, b[::3], 10, 1.0, force_hash=1)
    dup = np.concatenate([a[:2000], a[:2000]])         # exact distance ties, resolved by ascending index
    check(chk, oracle, dup, b[:800], 10, 1.0)
    check(chk, oracle, dup, b[:800], 10, 1.0, force_hash=1)
    check(chk, oracle, a[:3000], a[:3000:2], 5, 0.5)   # queries identical to targets (distance 0)
    check(chk, oracle, a[:3000], a[:3000:2], 10, 0.05)  # most lists stay incomplete


def test_search_voxel_targets_prunes_and_stays_exact(chk, oracle):
    """The bench's shape: every point of a scan queries the 0.2 m voxel centroids of another scan.  The pruned search scans
    fewer candidates than the full shells would (the figure DESIGN.md quotes comes from here)."""
    t = sy.make_scan(5, cols=1024, downsample_targets=0.2)
    s = sy.make_scan(6, cols=1024, downsample_targets=0.2)
    q = s["flat_xyz"][::5]
    coarse = check(chk, oracle, t["less_xyz"], q, 10, 1.0, xf=1)
    stats = check(chk, oracle, t["less_xyz"], q, 10, 1.0)
    print("x-refinement 1: candidates per query %.1f" % (coarse[0] / len(q)))
    per_query = stats[0] / len(q)
    print("candidates per query %.1f, rows per query %.1f, targets %d" % (per_query, stats[1] / len(q), len(t["less_xyz"])))
    assert per_query < 45          # full shells (round 2): ~90; row pruning: ~50; x-refined cells: ~35


def test_search_degenerate_clouds(chk, oracle):
    rng = np.random.default_rng(3)
    line = np.zeros((400, 3), np.float32); line[:, 0] = np.linspace(-5, 5, 400)
    plane = np.zeros((900, 3), np.float32); plane[:, :2] = rng.uniform(-3, 3, size=(900, 2))
    point = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (40, 1))
    q = rng.uniform(-4, 4, size=(300, 3)).astype(np.float32)
    for tgt in (line, plane, point, plane[:9], plane[:1], line[:, [1, 0, 2]], line[:, [2, 1, 0]], plane[:, [2, 0, 1]]):
        for kw in ({}, {"force_hash": 1, "cell": 0.4}, {"xf": 1}):     # hashed tables have no extent to clip the shells to: a usable cell edge
            if len(tgt) >= 10:
                check(chk, oracle, tgt, q, 10, 1.5, **kw)
            if len(tgt) >= 5:
                check(chk, oracle, tgt, q, 5, 6.0, **kw)


def _fits(lib, pts, plane_tol, line_tol):
    pts = np.ascontiguousarray(pts, np.float64)
    m = len(pts)
    ok = np.empty(m, np.int32); plane = np.empty((m, 4)); line = np.empty(m, np.int32)
    lib.chk_fit(_p(pts, ctypes.c_double), m, ctypes.c_double(plane_tol), ctypes.c_double(line_tol), _p(ok, ctypes.c_int), _p(plane, ctypes.c_double), _p(line, ctypes.c_int))
    return ok.astype(bool), plane, line.astype(bool)


def _point_sets(rng, m):
    """10-point neighbourhoods from clean planes to clean lines, with everything in between (the decision w2 > 3 w1 has to be
    taken on a continuum of eigenvalue ratios) and degenerate sets."""
    sets = []
    for _ in range(m):
        kind = rng.integers(0, 6)
        c = rng.uniform(-20, 20, size=3)
        B = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if kind == 0:      # planar patch, isotropic
            s = np.array([rng.uniform(0.05, 0.5), rng.uniform(0.05, 0.5), rng.uniform(0, 0.01)])
        elif kind == 1:    # elongated patch: ratio of the two large eigenvalues around the threshold 3 (std ratio sqrt 3)
            a = rng.uniform(0.05, 0.5); s = np.array([a, a / np.sqrt(3.0) * rng.uniform(0.8, 1.25), rng.uniform(0, 0.01)])
        elif kind == 2:    # line
            s = np.array([rng.uniform(0.1, 1.0), rng.uniform(0, 0.005), rng.uniform(0, 0.005)])
        elif kind == 3:    # blob
            s = rng.uniform(0.05, 0.5, size=3)
        elif kind == 4:    # two equal eigenvalues (the slow case of Jacobi)
            a = rng.uniform(0.05, 0.5); s = np.array([a, a, a * rng.uniform(0, 1)])
        else:              # exactly coplanar / collinear lattices
            g = rng.integers(-3, 4, size=(10, 3)).astype(np.float64)
            g[:, 2] = 0
            if rng.integers(0, 2):
                g[:, 1] = 0
            sets.append(g * 0.125 + np.round(c))
            continue
        sets.append(c + (rng.normal(size=(10, 3)) * s) @ B.T)
    return np.array(sets)


def test_fits_take_the_oracles_decisions(chk, oracle):
    rng = np.random.default_rng(21)
    pts = _point_sets(rng, 6000)
    for plane_tol in (0.05, 0.01):
        ok, plane, line = _fits(chk, pts, plane_tol, 3.0)
        for s in range(len(pts)):
            o_ok, o_plane = oracle.form_plane_lsq(pts[s], plane_tol)
            o_line = oracle.form_line_pca(pts[s], 3.0)[0]
            assert bool(o_ok) == bool(ok[s]), s
            if o_ok:
                assert np.array_equal(np.asarray(o_plane), plane[s]), s          # same arithmetic: bit for bit
            assert bool(o_line) == bool(line[s]), (s, pts[s])
    assert 0.1 < line.mean() < 0.9 and 0.1 < ok.mean() < 0.9                   # both branches of both tests were exercised


def test_certified_line_test_near_the_threshold(chk, oracle):
    """Sets whose eigenvalue ratio sits within 1e-9 ... 1e-15 of the threshold: the certified early exit must hand these to the
    full sweep and still agree with the oracle."""
    rng = np.random.default_rng(8)
    sets = []
    for k in range(1500):
        # ten points symmetric about the origin: the scatter matrix is diagonal in the chosen basis with exactly known entries
        a = rng.uniform(0.1, 1.0)
        eps = 10.0 ** rng.uniform(-15, -6) * rng.choice([-1, 1])
        b = a / np.sqrt(3.0) * (1 + eps)
        base = np.array([[a, 0, 0], [-a, 0, 0], [0, b, 0], [0, -b, 0], [0, 0, 0.01], [0, 0, -0.01], [a, 0, 0], [-a, 0, 0], [0, b, 0], [0, -b, 0]])
        B = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        sets.append(base @ B.T + rng.uniform(-5, 5, size=3))
    pts = np.array(sets)
    _, _, line = _fits(chk, pts, 0.05, 3.0)
    exp = np.array([bool(oracle.form_line_pca(p, 3.0)[0]) for p in pts])
    assert np.array_equal(line, exp)
    assert 0.2 < exp.mean() < 0.8


def test_certified_line_test_saves_sweeps(chk):
    rng = np.random.default_rng(5)
    pts = np.ascontiguousarray(_point_sets(rng, 4000))
    hist = np.zeros(13, np.int32)
    total = chk.chk_line_sweeps(_p(pts, ctypes.c_double), len(pts), ctypes.c_double(3.0), 0, _p(hist, ctypes.c_int))
    print("exact loop with the certified exit, Jacobi sweeps per set: %.2f, histogram %s" % (total / len(pts), hist.tolist()))
    assert total / len(pts) < 4.0          # the uncertified loop runs 6-10 sweeps to an exactly zero off-diagonal part
    total = chk.chk_line_sweeps(_p(pts, ctypes.c_double), len(pts), ctypes.c_double(3.0), 1, _p(hist, ctypes.c_int))
    print("with the closed-form screen in front: %.4f sweeps per set, histogram %s" % (total / len(pts), hist.tolist()))
    assert hist[0] >= 0.99 * len(pts)


def test_closed_form_screen_accuracy_and_fallback_band(chk):
    """The screen's eigenvalues against LAPACK on scatter-like matrices of every conditioning, and its decisions: wrong never, undecided
    only inside the guard band."""
    rng = np.random.default_rng(17)
    m = 300000
    lam = np.sort(np.abs(rng.normal(size=(m, 3))) * 10.0 ** rng.uniform(-6, 2, size=(m, 1)), axis=1)
    lam[: m // 4, 0] *= 10.0 ** rng.uniform(-12, 0, size=m // 4)            # planar patches: one tiny eigenvalue
    lam[m // 4: m // 2, 1] = lam[m // 4: m // 2, 0] * (1 + 10.0 ** rng.uniform(-12, -1, size=m // 4))   # nearly equal pairs
    lam[m // 2: 5 * m // 8, 1] = lam[m // 2: 5 * m // 8, 2] / 3.0 * (1 + rng.choice([-1, 1], size=m // 8) * 10.0 ** rng.uniform(-9, -2, size=m // 8))   # around the threshold
    lam = np.sort(lam, axis=1)
    Q = np.linalg.qr(rng.normal(size=(m, 3, 3)))[0]
    A = np.einsum("mij,mj,mkj->mik", Q, lam, Q)
    A = 0.5 * (A + A.transpose(0, 2, 1))
    packed = np.ascontiguousarray(np.stack([A[:, 0, 0], A[:, 0, 1], A[:, 0, 2], A[:, 1, 1], A[:, 1, 2], A[:, 2, 2]], axis=1))
    dec = np.empty(m, np.int32); eig = np.zeros((m, 3))
    chk.chk_line_screen(_p(packed, ctypes.c_double), m, ctypes.c_double(3.0), _p(dec, ctypes.c_int), _p(eig, ctypes.c_double))
    w = np.linalg.eigvalsh(A)
    err = np.abs(eig - w).max(axis=1) / w[:, 2]
    print("screen: max eigenvalue error %.2e of the largest eigenvalue; undecided %.4f %%" % (err.max(), 100.0 * (dec < 0).mean()))
    assert err.max() < 1e-7                                                 # the bound in the header: 6e-8
    margin = (w[:, 2] - 3.0 * w[:, 1]) / (w[:, 2] + 3.0 * w[:, 1])
    decided = dec >= 0
    assert np.array_equal(dec[decided] == 1, margin[decided] > 0)          # never wrong
    assert np.all(np.abs(margin[~decided]) < 1.1e-5)                        # undecided only inside the guard band ...
    assert np.all(decided[np.abs(margin) > 1.1e-5])                         # ... and always decided outside it


def _fit_fast(lib, pts, tol, quad=False):
    pts = np.ascontiguousarray(pts, np.float64)
    out = np.zeros((len(pts), 8)); plane = np.zeros((len(pts), 4))
    wrong = lib.chk_fit_fast(_p(pts, ctypes.c_double), len(pts), ctypes.c_double(tol), 1 if quad else 0, _p(out, ctypes.c_double), _p(plane, ctypes.c_double))
    return wrong, out, plane


def _planar_sets(rng, m, tol, near_threshold=False, far=False):
    """Neighbourhoods like the association sees them: a patch of spread 0.1-0.6 m at 1-40 m (far: up to 300 m) from the sensor, out-of-plane noise
    drawn so that the largest distance lands anywhere from well inside to well outside the tolerance (near_threshold: within 1e-6 relative of it)."""
    c = rng.normal(size=(m, 3)); c /= np.linalg.norm(c, axis=1, keepdims=True)
    c *= rng.uniform(1.0, 300.0 if far else 40.0, size=(m, 1))
    B = np.linalg.qr(rng.normal(size=(m, 3, 3)))[0]
    spread = rng.uniform(0.1, 0.6, size=(m, 1, 1)) * np.array([1.0, 1.0, 0.0]) * rng.uniform(0.6, 1.0, size=(m, 1, 3))
    local = rng.normal(size=(m, 10, 3)) * spread
    noise = rng.normal(size=(m, 10)) * tol * rng.uniform(0.0, 1.2, size=(m, 1))
    if near_threshold:       # scale the noise so that the exact fit's largest distance sits within ~1e-7 .. 1e-12 of the tolerance
        noise = rng.normal(size=(m, 10)) * tol * 0.4
    local[:, :, 2] = noise
    return c[:, None, :] + np.einsum("mij,mkj->mik", local, B)


def test_certified_fast_fit_takes_the_exact_fits_decisions(chk):
    """Fit10::form_plane_fast (normal equations + an a-posteriori bound; the default of K3) against Fit10::form_plane (the reference's pivoted
    Householder QR restated; base/Geometry.hpp:345-373, lidar_mapping/LidarFeatureAssociate.cpp:592-602): wherever the fast fit answers, the answer is
    the exact path's; it answers almost always; the distance between the two solutions stays under the bound E it computed; the accepted planes agree
    to 5e-7 relative by construction, 1e-11 typically (that is the record the residual set stores; the bar of the path is 1e-6)."""
    rng = np.random.default_rng(31)
    total = decided = 0
    worst_ratio = 0.0
    for tol in (0.05, 0.01):
        for far in (False, True):
            pts = _planar_sets(rng, 400_000, tol, far=far)
            wrong, out, plane = _fit_fast(chk, pts, tol)
            assert wrong == 0
            have = out[:, 2] > 0
            assert np.all(out[have, 3] <= out[have, 2]), "||x - x_qr|| above the bound E"
            assert np.all(out[have, 7] <= out[have, 6] + 1e-300), "distance bound B violated"
            worst_ratio = max(worst_ratio, float((out[have, 3] / out[have, 2]).max()))
            dec = out[:, 0] >= 0
            if not far:                                                                # beyond 40 m at grazing incidence the systems are refused (condition cap): correct, just not fast
                total += len(pts); decided += int(dec.sum())
            acc = out[:, 0] == 1
            assert 0.15 < acc.mean() < 0.95 and (out[:, 0] == 0).mean() > 0.03        # both answers exercised
            # accepted records: unit normal + d against the exact path's, relative to the record's size
            _, exact_plane, _ = _fits(chk, pts[acc][:50_000], tol, 3.0)
            rel = np.abs(plane[acc][:50_000] - exact_plane) / np.maximum(1.0, np.abs(exact_plane).max(axis=1, keepdims=True))
            assert rel.max() <= 5e-7 and np.median(rel.max(axis=1)) <= 1e-11, (rel.max(), np.median(rel.max(axis=1)))   # guaranteed (e <= 2.5e-7) / typical
    assert decided >= 0.98 * total, (decided, total)                               # up to 40 m at random incidence; indoor ranges: > 0.9999 (bench.py reports the rate)
    assert worst_ratio < 0.2, worst_ratio                                          # the bound has room (it is built from worst-case constants)
    # the sets of the collinearity / degeneracy tests: lines, blobs, exactly coplanar lattices (rank-deficient systems are refused, never decided wrongly)
    pts = _point_sets(rng, 60_000)
    for tol in (0.05, 0.01):
        wrong, out, _ = _fit_fast(chk, pts, tol)
        assert wrong == 0
        have = out[:, 2] > 0
        assert np.all(out[have, 3] <= out[have, 2])


def test_certified_fast_fit_near_the_tolerance_and_against_quad_precision(chk):
    """Largest distance within a hair of the tolerance: the fast fit must refuse (or be right).  And both solutions against a __float128 solve of the
    normal equations: each lies within the share of the bound the derivation gives it."""
    rng = np.random.default_rng(32)
    tol = 0.05
    base = _planar_sets(rng, 20_000, tol, near_threshold=True)
    # bisect a scale of the out-of-plane component (about the centroid plane of the exact fit) so that the exact path flips between accept and reject
    sets = []
    for P in base[:3000]:
        c = P.mean(0)
        _, _, Vt = np.linalg.svd(P - c)
        n = Vt[2]
        h = (P - c) @ n
        lo, hi = 0.0, 8.0
        flat = P - np.outer(h, n)
        f = lambda a: _fits(chk, (flat + np.outer(h * a, n))[None], tol, 3.0)[0][0]
        if not f(lo) or f(hi):
            continue
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if f(mid):
                lo = mid
            else:
                hi = mid
        for a in (lo, hi, lo * (1 - 1e-12), hi * (1 + 1e-12), lo * (1 - 1e-9), hi * (1 + 1e-9), lo * (1 - 1e-6), hi * (1 + 1e-6), lo * (1 - 1e-3), hi * (1 + 1e-3)):
            sets.append(flat + np.outer(h * a, n))
    sets = np.array(sets)
    assert len(sets) > 10000
    wrong, out, _ = _fit_fast(chk, sets, tol)
    assert wrong == 0
    undecided = (out[:, 0] < 0).reshape(-1, 10).mean(0)
    assert undecided[0] > 0.99 and undecided[1] > 0.99 and undecided[6] < 0.7 and undecided[8] < 0.03, undecided   # at the flip: refused; 1e-6 away: half answered again; 1e-3 away: answered
    # quad-precision reference
    pts = _planar_sets(rng, 30_000, tol, far=True)
    wrong, out, _ = _fit_fast(chk, pts, tol, quad=True)
    assert wrong == 0
    have = out[:, 2] > 0
    assert np.all(out[have, 4] <= out[have, 2]) and np.all(out[have, 5] <= out[have, 6]), "a distance to the exact minimiser exceeds its share of the bound"
    assert (out[have, 4] / out[have, 2]).max() < 0.2 and (out[have, 5] / out[have, 6]).max() < 0.05       # with room: the shares are built from worst-case constants


def test_collinearity_from_the_moments_takes_the_exact_decision(chk):
    """Fit10::is_line_fast — the closed-form screen on the scatter matrix derived from the raw second moments (M - s s^T / 10), what the streaming form of the fast
    kernel evaluates without keeping the ten points — against Fit10::is_line (the reference's FormLine decision, base/Geometry.hpp:220-260): every decided case
    equal, undecided ones rare, near and far from the sensor (the derivation cancels against the distance)."""
    rng = np.random.default_rng(41)
    for scale, shift in ((1.0, 0.0), (1.0, 300.0), (0.02, 50.0)):
        pts = _point_sets(rng, 60_000) * scale + shift
        for tol in (3.0,):
            dec = np.zeros(len(pts), np.int32); exact = np.zeros(len(pts), np.int32)
            wrong = chk.chk_is_line_fast(_p(np.ascontiguousarray(pts), ctypes.c_double), len(pts), ctypes.c_double(tol), _p(dec, ctypes.c_int), _p(exact, ctypes.c_int))
            assert wrong == 0
            assert (dec < 0).mean() < (0.25 if scale < 1 else 0.2), (scale, shift, (dec < 0).mean())     # the lattice sets (a sixth) are exactly degenerate: refused
            assert 0.1 < exact.mean() < 0.9
    # the thin band around the threshold: refused or right
    sets = []
    for _ in range(4000):
        a = rng.uniform(0.05, 0.5)
        B = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        s_ = np.array([a, a / np.sqrt(3.0) * (1 + rng.normal() * 1e-6), rng.uniform(0, 0.01)])
        sets.append(rng.uniform(-20, 20, size=3) + (rng.normal(size=(10, 3)) * s_) @ B.T)
    sets = np.array(sets)
    dec = np.zeros(len(sets), np.int32); exact = np.zeros(len(sets), np.int32)
    assert chk.chk_is_line_fast(_p(np.ascontiguousarray(sets), ctypes.c_double), len(sets), ctypes.c_double(3.0), _p(dec, ctypes.c_int), _p(exact, ctypes.c_int)) == 0
