"""K27 — pvlm_line_grow_batch (csrc/pvlm_linegrow.hip): the growth phase of ExtractLineFeatures / ExpandLine (sensors/LidarLineExtraction.cpp:296-389, :10-70) for a
batch of edge clouds, in rounds of speculative start points with upstream's walk replayed on the device, against the same walk run on the CPU with the kernel's own
task code (tests/cpp/linegrow_check.cpp; that code is held against the oracle's segment-after-segment growth in tests/test_lines_cpu.py).  Bar: bit-exact — segment
order, member ids, the six line coefficients."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import host_io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("linegrow") / "liblinegrow_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "linegrow_check.cpp")])
    return ctypes.CDLL(out)


def cpu_walk(chk, cloud):
    a = np.ascontiguousarray(cloud, np.float32).reshape(len(cloud), -1) if len(cloud) else np.zeros((0, 3), np.float32)
    sizes = (ctypes.c_longlong * 2)()
    fp = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if len(a) else None
    st = chk.chk_line_walk(fp, len(a), a.shape[1] if len(a) else 3, sizes, None, None, None, None)
    if st:
        return dict(status=st)
    ns, nm = int(sizes[0]), int(sizes[1])
    task = np.zeros(ns, np.int32); off = np.zeros(ns + 1, np.int32); mem = np.zeros(max(nm, 1), np.int32); co = np.zeros((max(ns, 1), 6))
    ip = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    chk.chk_line_walk(fp, len(a), a.shape[1] if len(a) else 3, sizes, ip(task), ip(off), ip(mem), co.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return dict(status=0, seg_task=task, seg_offset=off, members=mem[:nm], coeffs=co[:ns])


def edge_cloud(k, **kw):
    h = host_io.extract_features(sy.raw_vlp16_scan(k, cols=1800, **kw), edge_to_line=False)
    return h["cornerLessSharp"][:, :4].astype(np.float32)


def same(g, c):
    assert g["status"] == c["status"]
    if c["status"]:
        return
    assert np.array_equal(g["seg_task"], c["seg_task"]) and np.array_equal(g["seg_offset"], c["seg_offset"]) and np.array_equal(g["members"], c["members"])
    assert g["coeffs"].tobytes() == c["coeffs"].tobytes()


def test_batch_of_edge_clouds_equals_the_cpu_walk(chk):
    import panovlm_amd as pv
    ctx = pv.Context(0)
    rng = np.random.default_rng(4)
    clouds = [edge_cloud(0), edge_cloud(3, clutter=25), edge_cloud(7, clutter=40, dropout=0.03, jitter=0.1), edge_cloud(42, clutter=80, skew=0.6)]
    clouds.append(np.zeros((0, 4), np.float32))                                   # no edge points
    clouds.append(clouds[0][:3].copy())                                            # fewer points than neighbours
    clouds.append(clouds[1][:, :3].copy())                                         # stride 3
    # straight rods, 8 mm apart along the rod: long segments (a rod of 90 points outgrows the 64-member lists: status 2, the scan goes back to the caller)
    t = np.arange(40, dtype=np.float32)[:, None] * 0.008
    rods = np.concatenate([np.array([[1, 0, 0]], np.float32) + t * np.array([[0, 1, 0]], np.float32), np.array([[0, 2, 1]], np.float32) + t * np.array([[0.6, 0, 0.8]], np.float32),
                           rng.uniform(-3, 3, size=(60, 3)).astype(np.float32)])
    clouds.append(rods)
    t = np.arange(90, dtype=np.float32)[:, None] * 0.008
    clouds.append(np.array([[1, 0, 0]], np.float32) + t * np.array([[0, 1, 0]], np.float32))
    dup = clouds[0].copy(); dup[5] = dup[4]; dup[20] = dup[4]                       # coincident points: distance ties in the neighbour table, index order decides
    clouds.append(dup)
    got = ctx.line_grow_batch(clouds)
    assert len(got) == len(clouds) and got[0]["tasks_run"] > 0 and got[0]["kernel_ms"] > 0
    n_seg = 0
    for g, c in zip(got, clouds):
        want = cpu_walk(chk, c)
        same(g, want)
        n_seg += 0 if want["status"] else len(want["seg_task"])
    assert n_seg > 60 and got[8]["status"] == 2 and got[7]["status"] == 0 and len(got[7]["seg_task"]) >= 2
    # the two halves (begin / finish) and a second batch on the same context: the same bits
    again = ctx.line_grow_batch(clouds[:4], two_halves=True)
    for g, h in zip(again, got[:4]):
        same(g, h)


def test_large_edge_cloud_takes_the_points_from_global_memory(chk):
    """More edge points than the kernel keeps in LDS (2048): the variant that reads them from global memory, same bits."""
    import panovlm_amd as pv
    ctx = pv.Context(0)
    parts = [edge_cloud(k, clutter=60) for k in (1, 2, 5, 9, 11, 12, 13, 14, 15, 16)]
    big = np.concatenate([p[:, :3] + np.float32(7.0 * i) for i, p in enumerate(parts)])
    assert len(big) > 2048
    got = ctx.line_grow_batch([big, parts[0]])
    same(got[0], cpu_walk(chk, big)); same(got[1], cpu_walk(chk, parts[0]))
    assert len(got[0]["seg_task"]) > 50
