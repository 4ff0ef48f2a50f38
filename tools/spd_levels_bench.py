#!/usr/bin/env python3
"""pvlm_spd_solve_blocks on the Floor-shaped pose system of tools/spd_floor_bench.py through the Python binding: wall time per solve (median of --reps) and the
plan's schedule.  For rocprofv3 --kernel-trace --stats passes of the level-scheduled factorisation.  python tools/spd_levels_bench.py [--scans 1593] [--reps 7]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.spd_floor_bench import neighbours


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--scans", type=int, default=1593); ap.add_argument("--reps", type=int, default=7); a = ap.parse_args()
    F = a.scans
    rng = np.random.default_rng(3)
    pairs = set()
    for i, nb in enumerate(neighbours(F)):
        for j in nb:
            if i != j: pairs.add((min(i, j), max(i, j)))
    pairs = [(p, p) for p in range(F)] + sorted(pairs)
    n = 6 * (F - 1)
    off = np.arange(-6, n).reshape(F, 6); off[0] = -1
    rows = np.array([off[pa] for pa, pb in pairs], np.int32); cols = np.array([off[pb] for pa, pb in pairs], np.int32)
    mirror = np.array([int(pa != pb) for pa, pb in pairs], np.int32)
    blocks = np.empty((len(pairs), 36))
    for k, (pa, pb) in enumerate(pairs):
        if pa == pb:
            J = rng.normal(size=(9, 6)); blocks[k] = (J.T @ J + 30 * np.eye(6)).reshape(-1)
        else:
            blocks[k] = (rng.normal(size=(6, 6)) * 0.2).reshape(-1)
    scale = np.full(n, 0.2); diag = np.full(n, 1.0); rhs = rng.normal(size=n)
    import panovlm_amd as pv
    ctx = pv.Context(0)
    t0 = time.perf_counter()
    x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    first = time.perf_counter() - t0
    walls = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
        walls.append(time.perf_counter() - t0)
    print("poses %d unknowns %d blocks %d | plan %s | first call (plan + solve) %.2f ms | solve %.3f ms (median of %d: %s) | info %d" %
          (F, n, len(pairs), ctx.spd_plan(), first * 1e3, np.median(walls) * 1e3, a.reps, " ".join("%.2f" % (w * 1e3) for w in walls), info))


if __name__ == "__main__":
    main()
