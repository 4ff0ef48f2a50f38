#!/usr/bin/env python3
"""Wall time of pvlm_spd_solve_blocks (K10: device assembly + blocked Cholesky + triangular solves) on a Room-sized
reduced pose system: P poses (6 unknowns each), pose 0 constant, every pose coupled to its next `band` poses.
PVLM_CHOL_VALU=1 selects the VALU trailing update instead of the MFMA-f64 one — only in a library built with
-DPVLM_MEASURED_VARIANTS=1 (PVLM_LIB=build/var/libpvlm_measured.so); the default library has the MFMA update only.  Checks the residual of the solution."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=908)
    ap.add_argument("--band", type=int, default=12)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import panovlm_amd as pv
    rng = np.random.default_rng(3)
    P = a.poses
    off = np.full((P, 6), -1, np.int64); off[1:] = np.arange(6 * (P - 1)).reshape(P - 1, 6)
    n = 6 * (P - 1)
    rows, cols, mirror, blocks = [], [], [], []
    for p in range(P):
        J = rng.normal(size=(12, 6)); rows.append(off[p]); cols.append(off[p]); mirror.append(0); blocks.append((J.T @ J + 30 * np.eye(6)).reshape(-1))
        for q in range(p + 1, min(p + 1 + a.band, P)):
            rows.append(off[p]); cols.append(off[q]); mirror.append(1); blocks.append((rng.normal(size=(6, 6)) * 0.3).reshape(-1))
    rows, cols, mirror, blocks = np.array(rows), np.array(cols), np.array(mirror), np.array(blocks)
    scale = np.ones(n); diag = np.full(n, 1.0); rhs = rng.normal(size=n)
    ctx = pv.Context(0)
    x, info = ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0
    # residual with a sparse mat-vec of the same blocks
    y = diag * x
    for r, c, m, b in zip(rows, cols, mirror, blocks):
        B = b.reshape(6, 6); ri, ci = r >= 0, c >= 0
        if ri.any() and ci.any():
            y[r[ri]] += B[np.ix_(ri, ci)] @ x[c[ci]]
            if m:
                y[c[ci]] += B[np.ix_(ri, ci)].T @ x[r[ri]]
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter(); ctx.spd_solve_blocks(n, rows, cols, mirror, blocks, scale, diag, rhs); ts.append(time.perf_counter() - t0)
    print(json.dumps(dict(unknowns=n, blocks=len(blocks), variant="valu" if os.environ.get("PVLM_CHOL_VALU") else "mfma_f64",
                          residual_max=float(np.abs(y - rhs).max()), wall_ms_min=min(ts) * 1e3, wall_ms_median=float(np.median(ts)) * 1e3,
                          gflops_factor_only=n ** 3 / 3 / min(ts) / 1e9)))


if __name__ == "__main__":
    main()
