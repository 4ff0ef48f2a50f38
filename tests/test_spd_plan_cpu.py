"""Ordering + symbolic factorisation of the tile-sparse pose solve (csrc/pvlm_spd_plan.h, what pvlm_spd_solve_blocks uploads as per-block-column
tile lists), compiled for the host: the permutation is one, and the lists COVER a numeric factorisation — every nonzero of the Cholesky factor
of the permuted matrix lies in a row tile its block column lists, every tile a rank-32 update changes is listed as a pair.  A missing tile would
be a wrong answer on the GPU; this is where the planner is validated before it is run (the numeric solve itself: tests/test_linalg_gpu.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB, TILE = 32, 64


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("spd_plan") / "spd_plan_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "spd_plan_check.cpp")])
    lib = ctypes.CDLL(out)
    lib.chk_spd_plan.restype = ctypes.c_int
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def plan(lib, n, rows, cols):
    rows = np.ascontiguousarray(rows, np.int32); cols = np.ascontiguousarray(cols, np.int32)
    sizes = np.zeros(3, np.int64); frac = ctypes.c_double()
    lib.chk_spd_plan(n, len(rows) // 6, _p(rows, ctypes.c_int), _p(cols, ctypes.c_int), NB, None, None, None, None, None, _p(sizes, ctypes.c_longlong), ctypes.byref(frac))
    perm = np.zeros(n, np.int32); row_off = np.zeros(sizes[2] + 1, np.int32); pair_off = np.zeros(sizes[2] + 1, np.int32)
    row_tiles = np.zeros(max(sizes[0], 1), np.int32); pairs = np.zeros((max(sizes[1], 1), 2), np.int32)
    ordered = lib.chk_spd_plan(n, len(rows) // 6, _p(rows, ctypes.c_int), _p(cols, ctypes.c_int), NB, _p(perm, ctypes.c_int), _p(row_off, ctypes.c_int), _p(row_tiles, ctypes.c_int),
                               _p(pair_off, ctypes.c_int), _p(pairs, ctypes.c_int), _p(sizes, ctypes.c_longlong), ctypes.byref(frac))
    return dict(ordered=bool(ordered), perm=perm, row_off=row_off, row_tiles=row_tiles[:sizes[0]], pair_off=pair_off, pairs=pairs[:sizes[1]], fraction=frac.value)


def system(rng, P, pairs, constant=()):
    off = np.full((P, 6), -1, np.int64); n = 0
    for p in range(P):
        for half in range(2):
            if (p, half) in constant:
                continue
            off[p, 3 * half:3 * half + 3] = np.arange(n, n + 3); n += 3
    H = np.zeros((n, n))
    rows, cols = [], []
    for a, b in pairs:
        blk = rng.normal(size=(6, 6)) * (0.2 if a != b else 1.0)
        rows.append(off[a]); cols.append(off[b])
        va, vb = off[a] >= 0, off[b] >= 0
        H[np.ix_(off[a][va], off[b][vb])] += blk[np.ix_(va, vb)]
        if a != b:
            H[np.ix_(off[b][vb], off[a][va])] += blk[np.ix_(va, vb)].T
    H = H @ H.T * 0 + H + H.T                               # symmetric pattern
    M = H + np.eye(n) * (np.abs(H).sum(axis=1).max() + 1.0)  # diagonally dominant: positive definite, no cancellation to exact zeros
    return n, np.concatenate(rows), np.concatenate(cols), M


@pytest.mark.parametrize("shape", ["chain", "loops", "star", "grid"])
def test_lists_cover_the_numeric_factor(chk, shape):
    rng = np.random.default_rng(7)
    P = 160
    pairs = [(p, p) for p in range(P)]
    if shape == "chain":
        pairs += [(p, q) for p in range(P) for q in range(p + 1, min(p + 3, P))]
    elif shape == "loops":
        pairs += [(p, p + 1) for p in range(P - 1)] + [(int(a), int(b)) for a, b in zip(rng.integers(0, P, 80), rng.integers(0, P, 80)) if a < b]
    elif shape == "star":
        pairs += [(0, p) for p in range(1, P)] + [(p, p + 1) for p in range(1, P - 1, 7)]
    else:
        side = 12
        pairs = [(p, p) for p in range(side * side)] + [(r * side + c, r * side + c + 1) for r in range(side) for c in range(side - 1)] + \
                [(r * side + c, (r + 1) * side + c) for r in range(side - 1) for c in range(side)]
        P = side * side
    n, rows, cols, M = system(rng, P, pairs, constant={(0, 0), (3, 1)})
    pl = plan(chk, n, rows, cols)
    assert pl["ordered"] and sorted(pl["perm"].tolist()) == list(range(n))
    Mp = np.zeros_like(M); Mp[np.ix_(pl["perm"], pl["perm"])] = M          # new index of unknown i = perm[i]
    L = np.linalg.cholesky(Mp)
    C = (n + NB - 1) // NB
    assert len(pl["row_off"]) == C + 1
    listed_updates = set()
    for k in range(C):
        base = min(n, (k + 1) * NB)
        tiles = set(pl["row_tiles"][pl["row_off"][k]:pl["row_off"][k + 1]].tolist())
        nzrows = np.flatnonzero(np.abs(L[base:, k * NB:base]).max(axis=1) > 1e-13) + base
        assert set((nzrows // TILE).tolist()) <= tiles, (shape, k)
        for a, b in pl["pairs"][pl["pair_off"][k]:pl["pair_off"][k + 1]]:
            assert a >= b and a in tiles and b in tiles
            listed_updates.add((k, int(a), int(b)))
        # every pair of nonzero panel rows is an update the kernel must perform
        t = sorted(set((nzrows // TILE).tolist()))
        for i, a in enumerate(t):
            for b in t[:i + 1]:
                assert (k, a, b) in listed_updates, (shape, k, a, b)
    assert 0 < pl["fraction"] <= 1.2
    if shape in ("chain", "grid"):
        assert pl["fraction"] < 0.5


def test_degenerate_structures(chk):
    rng = np.random.default_rng(1)
    n, rows, cols, _ = system(rng, 1, [(0, 0)])
    pl = plan(chk, n, rows, cols)
    assert sorted(pl["perm"].tolist()) == list(range(n)) and len(pl["pairs"]) == 0
    # unknowns no block mentions stay where a permutation puts them; blocks with every index constant are ignored
    rows = np.array([0, 1, 2, 3, 4, 5, -1, -1, -1, -1, -1, -1], np.int32); cols = np.array([0, 1, 2, 3, 4, 5, -1, -1, -1, -1, -1, -1], np.int32)
    pl = plan(chk, 9, rows, cols)
    assert sorted(pl["perm"].tolist()) == list(range(9))
