"""csrc/pvlm_stdsort.h (the permutation libstdc++'s std::sort leaves, restated for device code) against the toolchain's own std::sort: the
device may order a sector of equal curvatures — and, later, a voxel's points — only if this restatement is exact.  Host-compiled, no GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stdsort") / "stdsort_check.so")
    # $CXX: `make repin` runs this file with the reference machine's compiler — the order of equal keys is that toolchain's libstdc++'s (tools/refvec.py table)
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "stdsort_check.cpp")])
    lib = ctypes.CDLL(out)
    lib.chk_heap_sorted_ranges.restype = ctypes.c_longlong
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def by_float(lib, key, levels=True):
    """Both forms of the restatement — the serial one and the level-by-level one the device runs — against std::sort; 0 = both agree with it."""
    key = np.ascontiguousarray(key, np.float32)
    serial = lib.chk_sort_by_float(_p(key, ctypes.c_float), len(key), None)
    return serial if serial != 0 or not levels else lib.chk_sort_by_levels(_p(key, ctypes.c_float), len(key))


def test_sector_like_keys_with_ties(chk):
    rng = np.random.default_rng(5)
    for trial in range(3000):
        n = int(rng.integers(0, 700))
        kind = trial % 6
        if kind == 0:
            key = rng.random(n)                                            # distinct
        elif kind == 1:
            key = rng.integers(0, max(2, n // 8 + 1), n)                   # many ties
        elif kind == 2:
            key = np.where(rng.random(n) < 0.3, -1.0, rng.random(n))       # unset curvatures (-1) among real ones
        elif kind == 3:
            key = np.round(rng.random(n), 2)                               # quantised
        elif kind == 4:
            key = np.sort(rng.integers(0, 5, n))[::(1 if trial % 12 < 6 else -1)]   # sorted / reversed runs of equal keys
        else:
            key = np.zeros(n)                                              # all equal
        assert by_float(chk, key) == 0, (trial, n, kind)


def test_structured_inputs(chk):
    for n in (1, 2, 15, 16, 17, 18, 31, 32, 33, 100, 257, 1024, 2048, 5000):
        i = np.arange(n)
        for key in (i, i[::-1], np.minimum(i, n - 1 - i), np.maximum(i, n - 1 - i), i % 2, i % 3, (i * 7919) % 17, i // 16, np.zeros(n)):
            assert by_float(chk, key) == 0, n


def test_voxel_like_pairs(chk):
    rng = np.random.default_rng(9)
    for trial in range(1500):
        n = int(rng.integers(0, 2500))
        cell = rng.integers(0, max(1, n // int(rng.integers(1, 12)) + 1), n).astype(np.uint32)
        if trial % 3 == 0:
            cell = np.sort(cell)                                            # a ring walks through its voxels almost in order
            swap = rng.integers(0, max(n - 1, 1), n // 10)
            for k in swap:
                if k + 1 < n:
                    cell[k], cell[k + 1] = cell[k + 1], cell[k]
        assert chk.chk_sort_pairs(_p(cell, ctypes.c_uint), n) == 0, (trial, n)


def test_depth_limit_path_with_ties(chk):
    """Keys built by McIlroy's adversary against this very std::sort, then coarsened so that they tie: the ranges that reach the depth limit are heap-sorted,
    and the heap's order of equal keys must match too."""
    before = chk.chk_heap_sorted_ranges()
    hit = 0
    for n in (200, 1000, 2048, 6000):
        keys = np.zeros(n, np.int32)
        chk.chk_killer_keys(n, _p(keys, ctypes.c_int))
        assert sorted(keys.tolist()) == list(range(n))
        for coarse in (1, 2, 3, 5):
            h0 = chk.chk_heap_sorted_ranges()
            assert by_float(chk, (keys // coarse).astype(np.float32)) == 0, (n, coarse)
            hit += (chk.chk_heap_sorted_ranges() > h0) and coarse > 1
    assert chk.chk_heap_sorted_ranges() > before and hit > 0       # the heap path ran, also on tied keys


def test_nan_keys_are_refused_or_equal(chk):
    """With NaN keys `<` is no strict weak order; the serial restatement either reports that a loop bound stopped it (-1: the host's std::sort takes over) or
    agrees.  (The level-by-level form is only ever given integer keys: the device never sorts a sector with a NaN.)"""
    rng = np.random.default_rng(2)
    for trial in range(300):
        n = int(rng.integers(2, 400))
        key = rng.random(n).astype(np.float32)
        key[rng.integers(0, n, max(1, n // 10))] = np.nan
        assert by_float(chk, key, levels=False) in (0, -1)
