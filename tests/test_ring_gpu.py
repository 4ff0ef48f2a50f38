"""Range-image stages of the LiDAR feature extractor on the GPU (csrc/pvlm_ring.hip through pvlm_ring_extract_batch): ring / column
order, range image, segmentation labels (as the surviving points), curvature (float bits) and window ends against
oracle/features.hpp, bit for bit, at 16 x 1800 and 16 x 4096 — one batch, so that scans of different sizes, ring tables and
outcomes share the launches.  sensors/Velodyne.cpp:371-526, :623-657, :1438-1586."""
import numpy as np
import pytest

import panovlm_amd as pv
from tests import ring_cases as rc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pv.Context(0)
    yield c
    c.close()


def _groups():
    """Cases that can share a batch: same ring table, horizon and segmentation switch."""
    groups = {}
    for case in rc.CASES:
        raw, n_scans, horizon, segment = rc.raw_of(case)
        groups.setdefault((n_scans, horizon, segment), []).append((case, raw))
    return groups


@pytest.mark.parametrize("key", sorted(_groups().keys()), ids=lambda k: "rings%d-cols%d-seg%d" % k)
def test_batch_matches_oracle(ctx, oracle, key):
    n_scans, horizon, segment = key
    members = _groups()[key]
    extra = [members[0][1][:0], members[0][1][:1], members[0][1][:40]]                     # an empty, a one-point and a 40-point scan ride along
    batch = pv.RingBatch(ctx, [raw for _, raw in members] + extra, n_rings=n_scans, horizon=horizon, segment=segment)
    listed = 0
    for k, (case, raw) in enumerate(members):
        g = batch.arrays(k)
        rc.assert_matches_oracle(oracle, raw, n_scans, horizon, segment, g)
        assert np.array_equal(g["cloud_kept"][:, :3], raw[g["source"], :3])               # `source` indexes the raw scan
        assert np.array_equal(g["ring_col"] >> 16, g["rc_kept"][:, 0]) and np.array_equal(g["ring_col"] & 0xFFFF, g["rc_kept"][:, 1])
        listed += g["resolved_points"]
        print("%s: %d points, %d decided by the host libm, %d edges, %d replays, %d of %d sectors left to the host's std::sort" %
              (rc.case_id(case), len(raw), g["resolved_points"], g["resolved_edges"], g["replayed"], int(g["sector_host"].sum()), 6 * n_scans))
        assert int(g["sector_host"].sum()) == 0                                            # no NaN curvature, no sector beyond 2048 points in these cases
    for k, raw in enumerate(extra):
        rc.assert_matches_oracle(oracle, raw, n_scans, horizon, segment, batch.arrays(len(members) + k))
    print("stage ms:", {k: round(v, 3) for k, v in batch.timing().items()})
    batch.close()


def test_batch_equals_scan_by_scan(ctx):
    """One launch per stage for the batch == the same scans one by one (no cross-talk through the shared arrays)."""
    raws = [rc.raw_of(c)[0] for c in rc.CASES[:4]]
    together = pv.RingBatch(ctx, raws)
    for k, raw in enumerate(raws):
        alone = pv.RingBatch(ctx, [raw])
        a, b = together.arrays(k), alone.arrays(0)
        for name in ("cloud_kept", "rc_kept", "curvature", "half_window", "range", "range_image", "image_to_point_kept", "source", "sorted", "sector_host"):
            assert np.array_equal(a[name], b[name], equal_nan=True), name
        alone.close()
    together.close()


def test_errors(ctx):
    raw = rc.raw_of(rc.CASES[0])[0]
    with pytest.raises(pv.PvlmError):
        pv.RingBatch(ctx, [raw], n_rings=8)                                                  # no ring table (:386-390)
    bad = raw.copy(); bad[7, 1] = np.nan
    with pytest.raises(pv.PvlmError):
        pv.RingBatch(ctx, [raw, bad])
    pv.RingBatch(ctx, []).close()
