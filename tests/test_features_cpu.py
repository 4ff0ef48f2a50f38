"""LiDAR feature extraction in front of the hot path (planar branch) — no GPU needed, like upstream:
the host mirror (panovlm_amd/host/pvlm_features.cpp: counting sort + union-find + flat arrays) against the oracle
(oracle/features.hpp: the reference's statements in order), the oracle against independent numpy / scipy restatements,
and both against a committed golden scan."""
import os

import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import host_io

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("cloud_scan", "cornerSharp", "cornerLessSharp", "surfFlat", "surfLessFlat", "rc", "scan_start", "scan_end", "range_image",
          "image_to_point_idx", "curvature", "state", "sort_ind", "left", "right")

CASES = [
    dict(k=3),
    dict(k=1, clutter=60),
    dict(k=2, clutter=150, dropout=0.3),                      # many small segments, ragged rings
    dict(k=4, jitter=0.6, skew=0.9),                          # columns collide: the col_offset correction and overwritten cells
    dict(k=5, start_deg=359.0, elevation_noise=0.6),          # the +z crossing right at the start; returns jumping between rings
    dict(k=6, cols=360, clutter=30),
    dict(k=7, start_deg=180.0, dropout=0.9),                  # segmentation removes > 90 %: the scan is declared invalid (:551-556)
    dict(k=8, segment=False, max_curvature=5.0, angle_threshold=10.0),
]


def _both(oracle, raw, cols=1800, n_scans=16, **kw):
    o = oracle.ScanFeatures(raw, n_scans=n_scans, horizon=cols, max_curvature=kw.get("max_curvature", 1000.0),
                            intersect_angle_threshold=kw.get("angle_threshold", 5.0), segment=kw.get("segment", True), extract=kw.get("extract", True))
    g = host_io.extract_features(raw, n_scans=n_scans, horizon=cols, **kw)
    return o, g


def _assert_same(o, g):
    assert o.valid == g["valid"]
    for name in FIELDS:
        a, b = getattr(o, name), g[name]
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), name


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_host_mirror_matches_oracle(oracle, case):
    c = dict(case); k = c.pop("k")
    ext = {n: c.pop(n) for n in ("segment", "max_curvature", "angle_threshold") if n in c}
    raw = sy.raw_vlp16_scan(k, **c)
    o, g = _both(oracle, raw, cols=c.get("cols", 1800), **ext)
    _assert_same(o, g)
    if o.valid:
        assert len(o.surfFlat) <= 4 * 6 * 16 and len(o.surfLessFlat) > 100      # <= 384 queries per scan
    else:
        assert len(o.surfFlat) == 0 and len(o.cloud_scan) < 0.1 * len(raw)


def test_edge_cases(oracle):
    raw = sy.raw_vlp16_scan(9, cols=180)
    for kw in (dict(n_scans=32), dict(n_scans=64), dict(n_scans=8)):           # other ring tables (:190-207); 8 rings: unsupported, nothing happens
        o, g = _both(oracle, raw, cols=180, **kw)
        _assert_same(o, g)
    assert len(oracle.ScanFeatures(raw, n_scans=8, horizon=180).cloud_scan) == 0
    o, g = _both(oracle, np.zeros((0, 4), np.float32), cols=180)              # empty cloud
    _assert_same(o, g)
    one_ring = raw[np.abs(np.degrees(np.arctan(-raw[:, 1] / np.hypot(raw[:, 0], raw[:, 2]))) - 1.0) < 0.5]   # only the +1 degree laser
    o, g = _both(oracle, one_ring, cols=180, segment=False)
    _assert_same(o, g)
    assert set(np.unique(o.cloud_scan[:, 3])) == {8.0}
    few = raw[:40]                                                             # rings with fewer than 6 usable points: no sector at all
    o, g = _both(oracle, few, cols=180, segment=False)
    _assert_same(o, g)
    assert len(o.surfFlat) == 0 and len(o.cornerLessSharp) == 0


def test_reorder_recovers_ring_and_column(oracle):
    """ReOrderVLP against the generator's truth: every return lands on its laser's ring; its column is the firing column
    (up to the rounding of the within-column azimuth skew), also when the sweep starts just before the +z axis."""
    for start in (37.0, 359.9, 180.0):
        raw, ring, col = sy.raw_vlp16_scan(12, start_deg=start, jitter=0.02, skew=0.3, return_truth=True)
        o = oracle.ScanFeatures(raw, extract=False)
        assert len(o.cloud_scan) == len(raw)
        order = np.argsort(ring, kind="stable")                                # ring by ring, firing order inside a ring
        assert np.array_equal(o.cloud_scan[:, :3], raw[order, :3]) and np.array_equal(o.cloud_scan[:, 3], ring[order].astype(np.float32))
        assert np.array_equal(o.rc[:, 0], ring[order])
        d = (o.rc[:, 1] - col[order]) % 1800
        assert np.isin(d, (0, 1, 1799)).all() and (d == 0).mean() > 0.6
        counts = np.bincount(ring, minlength=16); ends = np.cumsum(counts)
        assert np.array_equal(o.scan_start, ends - counts + 5) and np.array_equal(o.scan_end, ends - 6)
        rng_ = np.sqrt((o.cloud_scan[:, 0] ** 2 + o.cloud_scan[:, 1] ** 2 + o.cloud_scan[:, 2] ** 2).astype(np.float32))
        last = o.image_to_point_idx[o.rc[:, 0], o.rc[:, 1]]                    # the last return written to each cell
        assert np.array_equal(o.range_image[o.rc[:, 0], o.rc[:, 1]], rng_[last])


def test_segmentation_matches_graph_components(oracle):
    """Velodyne::Segmentation against scipy connected components of the same neighbour relation on the range image."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    raw = sy.raw_vlp16_scan(2, clutter=150, dropout=0.2)
    before = oracle.ScanFeatures(raw, extract=False)
    after = oracle.ScanFeatures(raw, segment=True)
    R = before.range_image
    rows, cols = R.shape
    theta = np.float32(20.0 / 180.0 * np.pi)

    def joined(a, b, alpha):
        far, near = np.maximum(a, b), np.minimum(a, b)
        alpha = np.float32(alpha)
        return np.arctan2(near * np.sin(alpha), far - near * np.cos(alpha)) > theta
    idx = np.arange(rows * cols).reshape(rows, cols)
    h = joined(R, np.roll(R, -1, axis=1), 0.2 / 180.0 * np.pi)                 # (r, c) - (r, c + 1), columns wrap
    v = joined(R[:-1], R[1:], 2.0 / 180.0 * np.pi)
    ei = np.concatenate([idx[h], idx[:-1][v]]); ej = np.concatenate([np.roll(idx, -1, axis=1)[h], idx[1:][v]])
    n, lab = connected_components(coo_matrix((np.ones(len(ei)), (ei, ej)), shape=(rows * cols,) * 2), directed=False)
    size = np.bincount(lab, minlength=n)
    seed = np.full(n, rows * cols); np.minimum.at(seed, lab, np.arange(rows * cols))
    keep = size >= 30
    for c in np.nonzero((size >= 5) & (size < 30))[0]:
        cells = np.nonzero(lab == c)[0]
        keep[c] = len(set(cells[cells != seed[c]] // cols)) >= 3              # rows of the pushed cells (the seed is never pushed)
    cell = before.rc[:, 0] * cols + before.rc[:, 1]
    survive = keep[lab[cell]]
    assert 0.02 < (~survive).mean() < 0.5
    assert np.array_equal(after.cloud_scan, before.cloud_scan[survive]) and np.array_equal(after.rc, before.rc[survive])


def test_voxel_grid_against_numpy(oracle):
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-3, 3, size=(4000, 3)), rng.uniform(0, 50, size=(4000, 1))], axis=1).astype(np.float32)
    pts[100:140] = pts[100]                                                    # duplicates (upstream feeds picked points twice)
    out = oracle.voxel_grid(pts, 0.2)
    inv = np.float32(1.0) / np.float32(0.2)
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64)
    ijk -= np.floor(pts[:, :3].min(0) * inv).astype(np.int64)
    dim = ijk.max(0) + 1
    key = ijk[:, 0] + ijk[:, 1] * dim[0] + ijk[:, 2] * dim[0] * dim[1]
    uniq, invk = np.unique(key, return_inverse=True)
    assert len(out) == len(uniq)                                               # one centroid per occupied voxel, ascending voxel index
    sums = np.zeros((len(uniq), 4)); np.add.at(sums, invk, pts.astype(np.float64))
    want = sums / np.bincount(invk)[:, None]
    assert np.abs(out - want).max() < 1e-4
    assert len(oracle.voxel_grid(np.zeros((0, 4), np.float32), 0.2)) == 0
    far = np.array([[0, 0, 0, 1], [1e6, 1e6, 1e6, 1]], np.float32)             # extent / leaf overflows the index: input passed through
    assert np.array_equal(oracle.voxel_grid(far, 0.01), far)


def test_curvature_and_picks_against_numpy(oracle):
    """The ADAPTIVE curvature of sampled points re-derived with numpy, and the invariants of the greedy picks."""
    raw = sy.raw_vlp16_scan(3, clutter=40)
    f = oracle.ScanFeatures(raw)
    P = f.cloud_scan[:, :3]
    dist = f.range_image[f.rc[:, 0], f.rc[:, 1]]
    rng = np.random.default_rng(1)
    has = np.nonzero(f.curvature >= 0)[0]
    assert len(has) > 0.95 * len(P)
    for i in rng.choice(has, 300, replace=False):
        a, b = f.left[i], f.right[i]
        assert i - a == b - i and b - i >= 5
        w = dist[a:b + 1].astype(np.float64)
        assert abs(abs((w.sum() - len(w) * float(dist[i])) / (b - a)) - f.curvature[i]) < 1e-4
        h = b - i
        if h > 5:                                                              # the window grew because one side was closer than 8 cm
            near = lambda j: np.sum((P[j] - P[i]).astype(np.float32) ** 2) < 0.0064
            assert near(i - h + 1) or near(i + h - 1)
    # picks: flat points are NORMAL|FLAT with curvature <= 0.02, at most 4 per sector; edge points have curvature in [0.1, max]
    flat_idx = np.nonzero(f.state & 8)[0]
    assert len(flat_idx) == len(f.surfFlat) and np.all(f.curvature[flat_idx] <= 0.02)
    assert np.array_equal(np.sort(f.cornerLessSharp[:, 3].astype(int)), np.nonzero((f.state == 2) | (f.state == 4))[0])
    edge_idx = f.cornerLessSharp[:, 3].astype(int)
    assert np.all(f.curvature[edge_idx] >= 0.1) and np.array_equal(f.cornerLessSharp[:, :3], P[edge_idx])
    assert np.array_equal(np.sort(f.cornerSharp[:, 3].astype(int)), np.nonzero(f.state == 4)[0])
    for r in range(16):
        lo, hi = f.scan_start[r], f.scan_end[r]
        for j in range(6):
            sp, ep = lo + (hi - lo) * j // 6, lo + (hi - lo) * (j + 1) // 6 - 1
            sec = f.sort_ind[sp:ep + 1]
            assert np.array_equal(np.sort(sec), np.arange(sp, ep + 1)) and np.all(np.diff(f.curvature[sec]) >= 0)
            assert np.sum((f.state[sec] & 8) > 0) <= 4 and np.sum(f.state[sec] == 4) <= 3
    # surfLessFlat: voxel centroids, tagged POINT_NORMAL, every one within a leaf diagonal of a cloud point
    assert np.all(f.surfLessFlat[:, 3] == 1.0)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(P).query(f.surfLessFlat[:, :3])
    assert d.max() < 0.2 * np.sqrt(3)


def test_golden_scan(oracle):
    """tests/golden/features.npz (made by tests/golden/make_golden.py): the oracle and the host mirror both reproduce it."""
    g = np.load(os.path.join(G, "features.npz"))
    o, h = _both(oracle, g["raw"], cols=int(g["horizon"]))
    for name in FIELDS:
        assert np.array_equal(getattr(o, name), g[name], equal_nan=True), name
        assert np.array_equal(h[name], g[name], equal_nan=True), name


def test_fuzz_unstructured_clouds(oracle):
    """Clouds that do not look like a VLP-16 sweep at all (random order, several turns, backwards sweeps, duplicated
    directions, out-of-band elevations): whatever path the re-ordering state machine takes, mirror and oracle agree."""
    rng = np.random.default_rng(123)
    for it in range(40):
        mode = it % 5
        n = int(rng.integers(50, 3000))
        if mode == 0:
            az = rng.uniform(0, 2 * np.pi, n); el = np.deg2rad(rng.uniform(-17, 17, n)); r = rng.uniform(0.5, 10, n)
        elif mode == 1:
            az = np.sort(rng.uniform(0, 2 * np.pi, n)); el = np.deg2rad(-15 + 2 * rng.integers(0, 16, n)); r = rng.uniform(0.5, 1.5, n)
        elif mode == 2:
            az = np.linspace(0, 5 * np.pi, n) + rng.normal(0, 0.01, n); el = np.deg2rad(-15 + 2 * (np.arange(n) % 16)); r = 3 + np.sin(az * 3)
        elif mode == 3:
            az = np.linspace(2 * np.pi, 0, n); el = np.deg2rad(-15 + 2 * (np.arange(n) % 16) + rng.normal(0, 0.7, n)); r = rng.uniform(1, 4, n)
        else:
            az = rng.choice(np.linspace(0, 2 * np.pi, 40), n); el = np.deg2rad(rng.choice([-15, -13, 1, 15, 40], n)); r = rng.choice([0.5, 1.0, 2.0], n)
        xyz = np.stack([r * np.cos(el) * np.sin(az), -r * np.sin(el), r * np.cos(el) * np.cos(az)], 1).astype(np.float32)
        raw = np.concatenate([xyz, np.zeros((n, 1), np.float32)], 1)
        with np.errstate(all="ignore"):
            o, g = _both(oracle, raw, cols=int(rng.choice([90, 360, 1800])), segment=bool(it % 2))
        _assert_same(o, g)


def test_pcd_file_to_features(oracle, tmp_path):
    """A scan as it lies on disk (binary .pcd, sensor axes x right / y forward / z up, with NaN returns and returns
    closer than 0.5 m) -> Velodyne::LoadLidar -> ReOrderVLP -> ExtractFeatures, against the oracle fed with what LoadLidar
    must produce (sensors/Velodyne.cpp:92-168: NaN and near points dropped, axes swapped to x, -z, y)."""
    raw = sy.raw_vlp16_scan(13, clutter=20)
    lidar = np.stack([raw[:, 0], raw[:, 2], -raw[:, 1], raw[:, 3]], axis=1).astype(np.float32)        # camera-style (x, y, z) = (x_l, -z_l, y_l)
    rec = np.concatenate([lidar, np.float32([[np.nan, 1, 1, 0], [0.1, 0.2, 0.1, 5]])])              # one invalid, one too close
    rec = np.concatenate([rec[:100], rec[-2:], rec[100:-2]]).astype(np.float32)
    path = os.path.join(str(tmp_path), "scan.pcd")
    hdr = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n" \
          "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (len(rec), len(rec))
    with open(path, "wb") as f:
        f.write(hdr.encode()); f.write(rec.tobytes())
    g = host_io.extract_features(path)
    o = oracle.ScanFeatures(raw)
    _assert_same(o, g)
    assert len(g["surfFlat"]) == 384 and len(g["cloud_scan"]) > 20000

