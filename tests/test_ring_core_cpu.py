"""The per-element bodies of the range-image kernels (csrc/pvlm_ring_core.h), compiled for the host by tests/cpp/ring_core_check.cpp
and driven serially through the stages of pvlm_ring_extract_batch: ring / column order, range image, segmentation, curvature must equal
oracle/features.hpp bit for bit — with the decisions certified in interval form, with every decision taken from the host libm, and
through the replay path.  No GPU: this is where a change of the certificates or of the state machine is validated before it is run."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import ring_cases as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ring_core") / "ring_core_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "ring_core_check.cpp")])
    lib = ctypes.CDLL(out)
    lib.chk_ring.restype = ctypes.c_int
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def run(lib, raw, n_scans, horizon, segment, force=0, threads=0):
    raw = np.ascontiguousarray(raw, np.float32)
    n = len(raw); cells = n_scans * horizon
    f32 = lambda *s: np.zeros(s, np.float32)
    i32 = lambda *s: np.zeros(s, np.int32)
    g = dict(cloud_reordered=f32(max(n, 1), 4), rc_reordered=i32(max(n, 1), 2), range_image=f32(n_scans, horizon), image_to_point_reordered=i32(n_scans, horizon),
             ring_count_reordered=i32(64), cloud_kept=f32(max(n, 1), 4), rc_kept=i32(max(n, 1), 2), image_to_point_kept=i32(n_scans, horizon), ring_count=i32(64),
             curvature=f32(max(n, 1)), half_window=i32(max(n, 1)), range=f32(max(n, 1)))
    stats = np.zeros(5, np.int64)
    c_f, c_i = ctypes.c_float, ctypes.c_int
    rcode = lib.chk_ring(_p(raw, c_f), n, n_scans, horizon, int(segment), force, threads, _p(g["cloud_reordered"], c_f), _p(g["rc_reordered"], c_i), _p(g["range_image"], c_f),
                         _p(g["image_to_point_reordered"], c_i), _p(g["ring_count_reordered"], c_i), _p(g["cloud_kept"], c_f), _p(g["rc_kept"], c_i),
                         _p(g["image_to_point_kept"], c_i), _p(g["ring_count"], c_i), _p(g["curvature"], c_f), _p(g["half_window"], c_i), _p(g["range"], c_f),
                         _p(stats, ctypes.c_longlong))
    assert rcode == 0 and cells > 0
    g.update(n_reordered=int(stats[3]), n_kept=int(stats[4]), listed=int(stats[0]), undecided_edges=int(stats[1]), replayed=int(stats[2]))
    return g


@pytest.mark.parametrize("case", rc.CASES, ids=rc.case_id)
def test_device_bodies_match_oracle(chk, oracle, case):
    raw, n_scans, horizon, segment = rc.raw_of(case)
    g = run(chk, raw, n_scans, horizon, segment)
    before, after = rc.assert_matches_oracle(oracle, raw, n_scans, horizon, segment, g)
    # the certificates decide all but a sliver of the points on their own
    print("%s: %d points, %d listed (%.2e), %d undecided edges, replayed %d" % (rc.case_id(case), len(raw), g["listed"], g["listed"] / max(len(raw), 1),
                                                                               g["undecided_edges"], g["replayed"]))
    if case.get("jitter", 0.05) < 0.5 and "start_deg" not in case:
        assert g["listed"] <= max(20, 2e-3 * len(raw))
    assert g["undecided_edges"] <= 4


@pytest.mark.parametrize("threads", [1, 7, 64, 1024])
def test_workgroup_form_of_the_column_machine(chk, oracle, threads):
    """columns_block (chunks + block scans, what the kernel runs) == the sequential loop == the oracle, for every chunking."""
    for case in rc.CASES:
        raw, n_scans, horizon, segment = rc.raw_of(case)
        g = run(chk, raw, n_scans, horizon, segment, threads=threads)
        rc.assert_matches_oracle(oracle, raw, n_scans, horizon, segment, g)
        s = run(chk, raw, n_scans, horizon, segment)
        assert g["replayed"] == s["replayed"] and g["listed"] == s["listed"]
    raw = rc.raw_of(dict(k=9, cols=180))[0]
    for sub in (raw[:1], raw[:5], raw[:40], raw[::7]):
        rc.assert_matches_oracle(oracle, sub, 16, 180, True, run(chk, sub, 16, 180, True, threads=threads))


@pytest.mark.parametrize("force", [1, 2, 3])
def test_host_libm_paths_give_the_same_arrays(chk, oracle, force):
    """force 1: every point is decided by the host libm (no certificate used); force 2: the scan goes through the replay path."""
    for case in (rc.CASES[0], rc.CASES[3], rc.CASES[4]):
        raw, n_scans, horizon, segment = rc.raw_of(case)
        rc.assert_matches_oracle(oracle, raw, n_scans, horizon, segment, run(chk, raw, n_scans, horizon, segment, force=force))


def test_empty_and_tiny_scans(chk, oracle):
    raw, n_scans, horizon, segment = rc.raw_of(dict(k=9, cols=180))
    for sub in (raw[:0], raw[:1], raw[:40], raw[::7]):
        rc.assert_matches_oracle(oracle, sub, 16, 180, True, run(chk, sub, 16, 180, True))
        rc.assert_matches_oracle(oracle, sub, 16, 180, False, run(chk, sub, 16, 180, False))
    origin = raw[:200].copy(); origin[5, :3] = 0                               # a return at the sensor origin: no elevation
    rc.assert_matches_oracle(oracle, origin, 16, 180, False, run(chk, origin, 16, 180, False))
