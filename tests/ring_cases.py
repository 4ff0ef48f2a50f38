"""Raw-scan cases shared by tests/test_ring_core_cpu.py (host-compiled device bodies) and tests/test_ring_gpu.py (the kernels):
every array the range-image stages of the LiDAR feature extractor produce is compared with oracle/features.hpp, bit for bit."""
import numpy as np

from panovlm_amd import synthetic as sy

CASES = [
    dict(k=3),
    dict(k=1, clutter=60),
    dict(k=2, clutter=150, dropout=0.3),                      # many small segments, ragged rings
    dict(k=4, jitter=0.6, skew=0.9),                          # columns collide: the col_offset correction and overwritten cells
    dict(k=5, start_deg=359.0, elevation_noise=0.6),          # the +z crossing right at the start; returns jumping between rings
    dict(k=6, cols=360, clutter=30),
    dict(k=7, start_deg=180.0, dropout=0.9),                  # segmentation removes > 90 %
    dict(k=8, segment=False),
    dict(k=9, cols=4096, clutter=40),                         # the 16 x 4096 range image of the bench scans
    dict(k=10, cols=4096, jitter=0.0, skew=0.0, noise=0.0),   # noise-free: every return of a firing column has the same azimuth up to rounding
    dict(k=11, scale=0.02, segment=False),                    # everything within 8 cm: the curvature windows run out of their rings (:637, :647)
    dict(k=12, scale=0.05, cols=600),
    dict(k=13, start_deg=0.0, jitter=0.0, skew=0.0),          # azimuths at +-0: the + 2 pi of :445-446 sits on the interval
    dict(k=14, n_scans=32), dict(k=15, n_scans=64, cols=360),
]


def raw_of(case):
    c = dict(case)
    k = c.pop("k"); scale = c.pop("scale", None)
    seg = c.pop("segment", True); n_scans = c.pop("n_scans", 16)
    raw = sy.raw_vlp16_scan(k, **c)
    if scale:
        raw = raw.copy(); raw[:, :3] *= np.float32(scale)
    return raw, n_scans, c.get("cols", 1800), seg


def case_id(c):
    return "-".join(f"{k}{v}" for k, v in c.items())


def starts_ends(ring_count, n_scans):
    ends = np.cumsum(ring_count[:n_scans])
    return (ends - ring_count[:n_scans] + 5).astype(np.int32), (ends - 6).astype(np.int32)


def assert_matches_oracle(oracle, raw, n_scans, horizon, segment, got):
    """got: dict with the arrays of both states as produced by the device path (or its host-compiled twin)."""
    before = oracle.ScanFeatures(raw, n_scans=n_scans, horizon=horizon, extract=False)
    n = len(before.cloud_scan)
    assert got["n_reordered"] == n
    assert np.array_equal(got["cloud_reordered"][:n].view(np.uint32), before.cloud_scan.view(np.uint32))
    assert np.array_equal(got["rc_reordered"][:n], before.rc)
    assert np.array_equal(got["range_image"].view(np.uint32), before.range_image.view(np.uint32))
    assert np.array_equal(got["image_to_point_reordered"], before.image_to_point_idx)
    if len(raw) == 0:                          # upstream reads cloud.points[0]; the oracle and the host mirror leave an empty scan untouched
        return before, before
    s, e = starts_ends(got["ring_count_reordered"], n_scans)
    assert np.array_equal(s, before.scan_start) and np.array_equal(e, before.scan_end)
    after = oracle.ScanFeatures(raw, n_scans=n_scans, horizon=horizon, segment=segment, extract=True, max_curvature=1000.0)
    if not after.valid:                        # < 10 % survive (:551-556): the oracle stops before the curvature
        assert got["n_kept"] < 0.1 * n
        return before, after
    m = len(after.cloud_scan)
    assert got["n_kept"] == m
    assert np.array_equal(got["cloud_kept"][:m].view(np.uint32), after.cloud_scan.view(np.uint32))
    assert np.array_equal(got["rc_kept"][:m], after.rc)
    assert np.array_equal(got["image_to_point_kept"], after.image_to_point_idx)
    s, e = starts_ends(got["ring_count"], n_scans)
    assert np.array_equal(s, after.scan_start) and np.array_equal(e, after.scan_end)
    if m:
        assert np.array_equal(got["curvature"][:m].view(np.uint32), after.curvature.view(np.uint32))
        half = got["half_window"][:m]
        idx = np.arange(m)
        assert np.array_equal(np.where(half >= 0, idx - half, -1), after.left) and np.array_equal(np.where(half >= 0, idx + half, -1), after.right)
        cell_range = after.range_image[after.rc[:, 0], after.rc[:, 1]]
        assert np.array_equal(got["range"][:m].view(np.uint32), cell_range.view(np.uint32))
    if m and "sorted" in got:
        sector_order_check(got["sorted"][:m], got["sector_host"], after, n_scans)
    return before, after


def sector_order_check(order, sector_host, after, n_scans):
    """K23: every sector is in the oracle's order — the order ITS std::sort left, equal curvatures included (the device runs libstdc++'s introsort on a sector
    with ties, csrc/pvlm_stdsort.h) — except a sector with a NaN or more than 2048 points, which is flagged and in index order like everything outside the
    sectors.  Returns (sectors with equal curvatures, sectors left to the host)."""
    expect = np.arange(len(order), dtype=np.int32)
    tied = left = 0
    for ring in range(n_scans):
        lo, hi = int(after.scan_start[ring]), int(after.scan_end[ring])
        span = hi - lo
        for j in range(6):
            flag = int(sector_host[ring * 6 + j])
            if span < 6:
                assert flag == 0
                continue
            sp, ep = lo + span * j // 6, lo + span * (j + 1) // 6 - 1
            c = after.curvature[sp:ep + 1]
            assert flag == (1 if (np.isnan(c).any() and len(c) > 1) or len(c) > 2048 else 0), (ring, j, flag)
            tied += len(np.unique(np.where(c == 0, np.float32(0), c))) < len(c)
            if flag:
                left += 1
            else:
                expect[sp:ep + 1] = after.sort_ind[sp:ep + 1]
    assert np.array_equal(order, expect)
    return tied, left
