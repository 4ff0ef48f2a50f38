"""pvlm_spd_solve_blocks on a Floor-shaped pose system: the block structure FindNeighbors (lidar_mapping/LidarFeatureAssociate.cpp:19-111) gives for the
synthetic 1593-scan trajectory of tools/floor_like_odometry.py (6-NN + previous / next + loop candidates), random SPD blocks — the dense factorisation
against the tile-sparse plan.  python tools/spd_floor_bench.py [--scans 1593]"""
import argparse, os, struct, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panovlm_amd import synthetic as sy
from tests import host_io


def neighbours(F):
    C = np.array([sy.estimated_pose(k)[1] for k in range(F)], np.float32)
    out = []
    for i in range(F):
        d = (C - C[i]) ** 2; s = d[:, 0] + d[:, 1] + d[:, 2]
        order = np.argsort(s, kind="stable")
        nb = [int(j) for j in order[:6]][1:]; ns = set(nb)
        for ni in (i - 1, i + 1):
            if 0 <= ni < F and ni not in ns: nb.append(ni)
        for j in order:
            if s[j] >= 400.0: break
            if sum(1 for v in ns if abs(int(j) - v) <= 200) < 2 and int(j) not in ns: nb.append(int(j)); ns.add(int(j))
        out.append(nb)
    return out


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--scans", type=int, default=1593); a = ap.parse_args()
    F = a.scans
    rng = np.random.default_rng(3)
    pairs = set()
    for i, nb in enumerate(neighbours(F)):
        for j in nb:
            if i != j: pairs.add((min(i, j), max(i, j)))
    pairs = [(p, p) for p in range(F)] + sorted(pairs)
    n = 6 * (F - 1)                                         # pose 0 constant (the gauge)
    off = np.arange(-6, n).reshape(F, 6); off[0] = -1
    rows = np.array([off[a] for a, b in pairs], np.int32); cols = np.array([off[b] for a, b in pairs], np.int32)
    mirror = np.array([int(a != b) for a, b in pairs], np.int32)
    blocks = np.empty((len(pairs), 36))
    for k, (a, b) in enumerate(pairs):
        if a == b:
            J = rng.normal(size=(9, 6)); blocks[k] = (J.T @ J + 30 * np.eye(6)).reshape(-1)
        else:
            blocks[k] = (rng.normal(size=(6, 6)) * 0.2).reshape(-1)
    scale = np.full(n, 0.2); diag = np.full(n, 1.0); rhs = rng.normal(size=n)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "blk.bin")
        with open(path, "wb") as f:
            f.write(struct.pack("<ii", n, len(pairs))); f.write(rows.tobytes()); f.write(cols.tobytes()); f.write(mirror.tobytes())
            f.write(blocks.tobytes()); f.write(scale.tobytes()); f.write(diag.tobytes()); f.write(rhs.tobytes())
        xs = {}
        for label, env in (("dense", "1000000000"), ("tile-sparse", "0")):
            os.environ["PVLM_SPD_SPARSE_MIN"] = env
            out = host_io.run("spdblocks", path, 4, timeout=900)
            plan = [l for l in out if l.startswith("plan")][0]
            xs[label] = np.array([float.fromhex(l.split()[1]) for l in out if l.startswith("x ")])
            print("%d poses, %d unknowns, %d blocks | forced %s -> %s" % (F, n, len(pairs), label, plan))
        print("largest relative difference of the two solutions: %.3e" % (np.abs(xs["dense"] - xs["tile-sparse"]).max() / np.abs(xs["dense"]).max()))


if __name__ == "__main__":
    main()
