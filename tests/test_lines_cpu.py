"""Line branch of the LiDAR feature extraction: Velodyne::EdgeToLine (sensors/Velodyne.cpp:1269-1324) with
ExtractLineFeatures / ExpandLine / FuseLineSegments / FuseLines / the two filters (sensors/LidarLineExtraction.cpp).

Parity contract.  Upstream fits every fused group of segments with pcl::SACSegmentation RANSAC, whose samples come from a
generator inside PCL: no restatement can reproduce one particular run.  Product (panovlm_amd/host/pvlm_lines.cpp) and
oracle (oracle/lines.hpp) therefore both use the limit RANSAC approximates — the exhaustive 2-point maximum-consensus
line — and the contract is
  (1) product == oracle on every output array (edge_segmented, segment_coeffs, end_points, point_to_segment, the filtered
      cornerLessSharp / cornerSharp, and the planar clouds extracted after it), bit for bit;
  (2) properties that hold for ANY RANSAC seed: a consensus set found from a random pair is never larger than the
      exhaustive one, all of its points lie within 0.02 m of a line through two points of the group, groups with <= 4
      inliers are dropped;
  (3) the structural invariants of upstream's filters hold on the output (>= 5 points, >= 3 rings, longer than 0.3 m ...).
All CPU: the extraction is host code (like upstream)."""
import os

import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import host_io

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCANS = [(0, {}), (3, dict(clutter=25)), (7, dict(clutter=40, dropout=0.03, jitter=0.1)), (11, dict(clutter=60, elevation_noise=0.15)),
         (19, dict(clutter=10, dropout=0.2)), (42, dict(clutter=80, skew=0.6))]


def _same(o, h):
    assert len(o.edge_segmented) == len(h["edge_segmented"])
    for a, b in zip(o.edge_segmented, h["edge_segmented"]):
        assert np.array_equal(a, b)
    assert np.array_equal(o.segment_coeffs, h["segment_coeffs"], equal_nan=True)
    assert np.array_equal(o.end_points, h["end_points"], equal_nan=True)
    assert o.point_to_segment == h["point_to_segment"]
    for name in ("cornerBeforeFilter", "cornerLessSharp", "cornerSharp", "surfFlat", "surfLessFlat", "cloud_scan"):
        assert np.array_equal(getattr(o, name), h[name]), name


@pytest.mark.parametrize("k,kw", SCANS)
def test_host_mirror_equals_oracle(oracle, k, kw):
    raw = sy.raw_vlp16_scan(k, cols=1800, **kw)
    o = oracle.ScanFeatures(raw, edge_to_line=True)
    h = host_io.extract_features(raw, edge_to_line=True)
    assert len(o.edge_segmented) >= 5                      # box edges of the synthetic room are found
    _same(o, h)
    # the planar clouds do not depend on the line branch (EdgeToLine does not touch the per-point state)
    p = oracle.ScanFeatures(raw, edge_to_line=False)
    assert np.array_equal(p.surfFlat, o.surfFlat) and np.array_equal(p.surfLessFlat, o.surfLessFlat)
    assert np.array_equal(p.cornerLessSharp, o.cornerBeforeFilter)


@pytest.mark.parametrize("k,kw", SCANS)
def test_task_formulation_of_the_growth_equals_oracle(oracle, monkeypatch, k, kw):
    """K27's formulation on the CPU: every (start point, neighbour pair) grown as an independent task by the kernel's own code (csrc/pvlm_linegrow_core.h, compiled
    for the host: PVLM_EDGE_GROW=tasks), then upstream's walk over the start points replayed on the finished segments — the same segments, coefficients and filtered
    clouds as the oracle's segment-after-segment growth, bit for bit (incl. the 1-degree turn test decided on the cosine against thresholds taken from this
    machine's acos)."""
    monkeypatch.setenv("PVLM_EDGE_GROW", "tasks")
    raw = sy.raw_vlp16_scan(k, cols=1800, **kw)
    o = oracle.ScanFeatures(raw, edge_to_line=True)
    h = host_io.extract_features(raw, edge_to_line=True)
    assert len(o.edge_segmented) >= 5
    _same(o, h)


def test_degenerate_inputs(oracle):
    """No edge points, fewer edge points than neighbours, a scan without any structure: empty segment lists on both sides."""
    rng = np.random.default_rng(5)
    flat = sy.raw_vlp16_scan(1, cols=1800)
    flat[:, :3] *= (2.0 / np.linalg.norm(flat[:, :3], axis=1, keepdims=True)).astype(np.float32)       # a sphere: no edges at all
    noisy = sy.raw_vlp16_scan(2, cols=1800)
    noisy[:, :3] += rng.normal(0, 0.2, size=(len(noisy), 3)).astype(np.float32)                        # edges everywhere, no lines
    for raw in (flat, noisy, sy.raw_vlp16_scan(5, cols=1800)[:3000]):
        o = oracle.ScanFeatures(raw, edge_to_line=True)
        h = host_io.extract_features(raw, edge_to_line=True)
        if not o.valid:
            assert not h["valid"]
            continue
        _same(o, h)


def _numpy_consensus(c, thr):
    """Independent statement of the consensus step: all pairs, vectorised, first best pair in (i, j) order."""
    p = c[:, :3].astype(np.float64)
    n = len(p)
    best, arg = 0, None
    for i in range(n - 1):
        d = p[i + 1:] - p[i]                                        # (m, 3)
        len2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        v = p - p[i]                                                # (n, 3)
        cx = v[None, :, 1] * d[:, None, 2] - v[None, :, 2] * d[:, None, 1]
        cy = v[None, :, 2] * d[:, None, 0] - v[None, :, 0] * d[:, None, 2]
        cz = v[None, :, 0] * d[:, None, 1] - v[None, :, 1] * d[:, None, 0]
        inl = ((cx * cx + cy * cy) + cz * cz) < (thr * thr) * len2[:, None]
        inl &= (len2 > 0)[:, None]
        cnt = inl.sum(axis=1)
        j = int(np.argmax(cnt))
        if cnt[j] > best:
            best, arg = int(cnt[j]), np.flatnonzero(inl[j])
    return arg if arg is not None else np.zeros(0, np.int64)


def test_consensus_against_numpy_and_random_sampling(oracle):
    rng = np.random.default_rng(17)
    for trial in range(12):
        n_line, n_out = int(rng.integers(4, 40)), int(rng.integers(0, 25))
        a, d = rng.normal(size=3) * 3, rng.normal(size=3)
        d /= np.linalg.norm(d)
        on = a + np.outer(rng.uniform(-1.5, 1.5, n_line), d) + rng.normal(0, 0.008, size=(n_line, 3))
        off = a + rng.normal(0, 0.6, size=(n_out, 3))
        c = np.concatenate([on, off]).astype(np.float32)
        c = np.concatenate([c, np.arange(len(c), dtype=np.float32)[:, None]], axis=1)
        if trial % 4 == 0:
            c = np.concatenate([c, c[:3]])                      # duplicated points: zero-length pairs are skipped
        c = c[rng.permutation(len(c))]
        got = oracle.line_consensus(c, 0.02)
        want = _numpy_consensus(c, 0.02)
        assert np.array_equal(got, want), trial
        # what any RANSAC run can return: the consensus set of SOME pair — never larger than the exhaustive one,
        # and every exhaustive inlier is within the threshold of a line through two points of the cloud
        p = c[:, :3].astype(np.float64)
        for _ in range(50):
            i, j = rng.choice(len(c), 2, replace=False)
            dd = p[j] - p[i]
            if not dd.any():
                continue
            dist = np.linalg.norm(np.cross(p - p[i], dd), axis=1) / np.linalg.norm(dd)
            assert (dist < 0.02 - 1e-12).sum() <= len(got)
        if len(got) >= 2:
            ok = False
            for i in got:
                for j in got:
                    if j <= i or not (p[j] - p[i]).any():
                        continue
                    dist = np.linalg.norm(np.cross(p[got] - p[i], p[j] - p[i]), axis=1) / np.linalg.norm(p[j] - p[i])
                    ok = ok or bool(np.all(dist < 0.02 + 1e-9))
                if ok:
                    break
            assert ok


@pytest.mark.parametrize("k,kw", SCANS[:4])
def test_invariants_of_the_extracted_segments(oracle, k, kw):
    raw = sy.raw_vlp16_scan(k, cols=1800, **kw)
    f = oracle.ScanFeatures(raw, edge_to_line=True)
    before_ids = set(int(v) for v in f.cornerBeforeFilter[:, 3])
    ring_of = f.rc[:, 0]
    seen = []
    for s, (seg, co, ends) in enumerate(zip(f.edge_segmented, f.segment_coeffs, f.end_points)):
        ids = [int(v) for v in seg[:, 3]]
        assert len(ids) >= 5 and len(set(ids)) == len(ids) and set(ids) <= before_ids          # ExtractLineFeatures :374, FuseLines :162
        assert np.array_equal(seg[:, :3], f.cloud_scan[ids, :3])                               # intensity = index into cloud_scan
        rings = set(int(ring_of[i]) for i in ids)
        assert len(rings) >= 3 and len(rings) >= len(ids) // 2                                 # FilterLineByScan
        p = seg[:, :3].astype(np.float64)
        dmax = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)).max()
        assert dmax > 0.3                                                                      # FilterLineByLength
        if np.any(co != 0):
            assert abs(np.linalg.norm(co[3:]) - 1) < 1e-12
            for e in ends:                                                                     # end points lie on the fitted line
                assert np.linalg.norm(np.cross(e - co[:3], co[3:])) < 1e-9
            assert np.linalg.norm(ends[0] - ends[1]) <= dmax + 1e-9
        seen += [i for i in ids if i not in seen]
    # cornerLessSharp = the members of the segments, once each, in order of first appearance; point_to_segment is its inverse
    assert [int(v) for v in f.cornerLessSharp[:, 3]] == seen
    for i, segs in enumerate(f.point_to_segment):
        pid = int(f.cornerLessSharp[i, 3])
        assert segs == [s for s, seg in enumerate(f.edge_segmented) if pid in set(int(v) for v in seg[:, 3])]
    assert set(int(v) for v in f.cornerSharp[:, 3]) <= set(seen)


def test_golden_scan(oracle):
    """tests/golden/line_extraction.npz (tests/golden/make_golden.py line_extraction): oracle and host mirror reproduce it."""
    g = np.load(os.path.join(G, "line_extraction.npz"))
    o = oracle.ScanFeatures(g["raw"], horizon=int(g["horizon"]), edge_to_line=True)
    h = host_io.extract_features(g["raw"], horizon=int(g["horizon"]), edge_to_line=True)
    so, po = g["seg_offsets"], g["p2s_offsets"]
    segs = [g["seg_points"][so[k]:so[k + 1]] for k in range(len(so) - 1)]
    p2s = [g["p2s_ids"][po[k]:po[k + 1]].tolist() for k in range(len(po) - 1)]
    for got_segs, got_co, got_ep, got_p2s, cl, cs in ((o.edge_segmented, o.segment_coeffs, o.end_points, o.point_to_segment, o.cornerLessSharp, o.cornerSharp),
                                                      (h["edge_segmented"], h["segment_coeffs"], h["end_points"], h["point_to_segment"], h["cornerLessSharp"],
                                                       h["cornerSharp"])):
        assert len(got_segs) == len(segs) and all(np.array_equal(a, b) for a, b in zip(got_segs, segs))
        assert np.array_equal(got_co, g["segment_coeffs"]) and np.array_equal(got_ep, g["end_points"])
        assert got_p2s == p2s and np.array_equal(cl, g["cornerLessSharp"]) and np.array_equal(cs, g["cornerSharp"])
