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


@pytest.mark.parametrize("key", sorted(_groups().keys()), ids=lambda k: "rings%d-cols%d-seg%d" % k)
def test_picks_match_oracle(ctx, oracle, key):
    """K24: the edge / plane picks (with their suppression chains), the point states and the voxel-grid centroids of every scan the device decides equal the oracle's
    ExtractFeatures; a scan with a ring left to the host is counted, not compared (the host mirror's PickFeatures takes it: test_host_gpu.py)."""
    n_scans, horizon, segment = key
    members = _groups()[key]
    batch = pv.RingBatch(ctx, [raw for _, raw in members], n_rings=n_scans, horizon=horizon, segment=segment, picks=(1000.0, 5.0))
    decided = 0
    for k, (case, raw) in enumerate(members):
        after = oracle.ScanFeatures(raw, n_scans=n_scans, horizon=horizon, segment=segment, extract=True, max_curvature=1000.0, intersect_angle_threshold=5.0)
        if not after.valid or len(after.cloud_scan) == 0:
            continue
        pk = batch.picks(k)
        assert pk is not None
        if pk["ring_host"].any():
            print("%s: %d rings left to the host" % (rc.case_id(case), int(pk["ring_host"].sum())))
            continue
        decided += 1
        cloud = after.cloud_scan
        assert np.array_equal(pk["state"], after.state), rc.case_id(case)
        assert np.array_equal(pk["corner"], after.cornerLessSharp[:, 3].astype(np.int32))
        assert np.array_equal(cloud[pk["corner"], :3].view(np.uint32), after.cornerLessSharp[:, :3].view(np.uint32))
        assert np.array_equal(pk["corner"][pk["sharp"]], after.cornerSharp[:, 3].astype(np.int32))
        assert np.array_equal(cloud[pk["flat"], :3].view(np.uint32), after.surfFlat[:, :3].view(np.uint32)) and np.all(after.surfFlat[:, 3] == 1)
        assert np.array_equal(pk["less_flat"].view(np.uint32), after.surfLessFlat.view(np.uint32)), rc.case_id(case)
        print("%s: %d edge picks (%d sharp), %d flat, %d less-flat centroids" % (rc.case_id(case), len(pk["corner"]), int(pk["sharp"].sum()), len(pk["flat"]), len(pk["less_flat"])))
    assert decided > 0 or all(not oracle.ScanFeatures(raw, n_scans=n_scans, horizon=horizon, segment=segment).valid for _, raw in members)
    print("stage ms:", {k: round(v, 3) for k, v in batch.timing().items()})
    batch.close()


def test_device_sort_equals_std_sort(ctx, tmp_path):
    """pvlm_stdsort::sort_wave (the form K23 / K24 run: long ranges partitioned by the whole wave, short ones side by side on the lanes, insertion leaf by leaf)
    leaves the permutation of the real std::sort — taken from tests/cpp/stdsort_check.cpp, which sorts with the toolchain's std::sort — on keys that tie."""
    import ctypes
    import os
    import subprocess
    from panovlm_amd import api
    so = str(tmp_path / "stdsort_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "stdsort_check.cpp")])
    chk = ctypes.CDLL(so)
    rng = np.random.default_rng(11)

    def reference(keys):
        f = np.ascontiguousarray(keys, np.float32)
        out = np.zeros(len(f), np.int32)
        assert chk.chk_sort_by_float(f.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(f), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 0
        return out

    cases = []
    for trial in range(160):
        n = int(rng.integers(1, 4097)) if trial % 4 else int(rng.integers(1, 40))
        kind = trial % 5
        if kind == 0:
            keys = rng.permutation(n)                                        # distinct
        elif kind == 1:
            keys = rng.integers(0, max(2, n // int(rng.integers(2, 40)) + 1), n)   # voxel-like: many equal keys
        elif kind == 2:
            keys = np.sort(rng.integers(0, max(2, n // 8), n))               # nearly sorted runs
        elif kind == 3:
            keys = np.zeros(n, np.int64)                                     # all equal
        else:
            keys = np.minimum(np.arange(n), n - 1 - np.arange(n)) // 3       # organ pipe with ties
        cases.append(keys)
    killer = np.zeros(3000, np.int32)
    chk.chk_killer_keys(3000, killer.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))   # drives the quicksort to its depth limit (heap-sorted ranges)
    cases += [killer, killer // 3]
    for keys in cases:
        got = api.device_sort(ctx, keys)
        assert np.array_equal(got, reference(keys)), (len(keys), keys[:8])
