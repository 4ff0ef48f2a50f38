"""GPU dense / block-sparse SPD solve of the LM driver (pvlm_spd_solve*: hand-written blocked Cholesky) against numpy,
and the host mirror's Solve with the GPU factorisation forced on against the CPU twin."""
import os

import numpy as np
import pytest

from tests import host_io, lm_twin, synth

pytestmark = pytest.mark.gpu


# driven through the C++ test driver (the real caller of these entry points)
def _run_spd(tmp_path, A, B):
    import struct
    n = A.shape[0]; B2 = np.asarray(B, np.float64).reshape(n, -1)
    path = os.path.join(str(tmp_path), "spd.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, B2.shape[1])); f.write(np.ascontiguousarray(A, np.float64).tobytes()); f.write(np.asfortranarray(B2).tobytes(order="F"))
    out = host_io.run("spd", path)
    info = int(out[0].split()[1])
    x = np.array([float.fromhex(l.split()[1]) for l in out[1:]])
    return (x.reshape(B2.shape[1], n).T.reshape(np.shape(B)) if info == 0 else None), info


@pytest.mark.parametrize("n", [7, 32, 33, 300, 1800])
def test_dense_spd_solve(tmp_path, n):
    rng = np.random.default_rng(n)
    Q = rng.normal(size=(n, n))
    A = Q @ Q.T + n * np.eye(n)
    B = rng.normal(size=(n, 3))
    X, info = _run_spd(tmp_path, A, B)
    assert info == 0
    assert np.allclose(A @ X, B, rtol=0, atol=1e-9 * np.abs(B).max() * n)
    assert np.allclose(X, np.linalg.solve(A, B), rtol=1e-9, atol=1e-12)
    # not positive definite: reported, not an error
    A2 = A.copy(); A2[n // 2, n // 2] = -1.0
    _, info = _run_spd(tmp_path, A2, B)
    assert info == n // 2 + 1


def _block_system(rng, P, pairs, constant=()):
    """Poses with 6 unknowns each (those in `constant`: (pose, half) pairs are fixed), one 6x6 block per (a, b) of `pairs`."""
    off = np.full((P, 6), -1, np.int64); n = 0
    for p in range(P):
        for half in range(2):
            if (p, half) in constant:
                continue
            off[p, 3 * half:3 * half + 3] = np.arange(n, n + 3); n += 3
    rows, cols, mirror, blocks = [], [], [], []
    H = np.zeros((n, n))
    for (a, b) in pairs:
        if a == b:
            J = rng.normal(size=(9, 6)); blk = J.T @ J
        else:
            blk = rng.normal(size=(6, 6)) * 0.2
        rows.append(off[a]); cols.append(off[b]); mirror.append(int(a != b)); blocks.append(blk.reshape(-1))
        ia, ib = off[a], off[b]
        va, vb = ia >= 0, ib >= 0
        H[np.ix_(ia[va], ib[vb])] += blk[np.ix_(va, vb)]
        if a != b:
            H[np.ix_(ib[vb], ia[va])] += blk[np.ix_(va, vb)].T
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0)))
    diag = rng.uniform(0.5, 1.0, size=n) + 10.0
    rhs = rng.normal(size=n)
    M = H * scale[:, None] * scale[None, :] + np.diag(diag)
    return n, rows, cols, mirror, blocks, scale, diag, rhs, M


def _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag, rhs, repeat=1):
    import struct
    path = os.path.join(str(tmp_path), "blk.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, len(blocks)))
        f.write(np.array(rows, np.int32).tobytes()); f.write(np.array(cols, np.int32).tobytes()); f.write(np.array(mirror, np.int32).tobytes())
        f.write(np.array(blocks, np.float64).tobytes()); f.write(scale.tobytes()); f.write(diag.tobytes()); f.write(rhs.tobytes())
    out = host_io.run("spdblocks", path, repeat)
    info = int(out[0].split()[1])
    x = np.array([float.fromhex(l.split()[1]) for l in out if l.startswith("x ")])
    return info, x, out


def test_block_sparse_assembly_and_solve(tmp_path):
    """M = D (sum of 6x6 blocks) D + diag: pose-pair blocks mirrored, constant blocks (-1) dropped, repeated blocks summed."""
    rng = np.random.default_rng(5)
    P = 40                                   # poses; pose 0 fully constant, pose 1 has a constant rotation block
    pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(p + 4, P))] + [(2, 3), (5, 5)]   # two repeated
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1), (1, 0)})
    info, x, _ = _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag, rhs)
    assert info == 0
    assert np.allclose(x, np.linalg.solve(M, rhs), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("P,links", [(300, "chain"), (400, "loops"), (260, "dense")])
def test_tile_sparse_factorisation(tmp_path, monkeypatch, P, links):
    """The tile-sparse plan of pvlm_spd_solve_blocks (minimum-degree + elimination-tree postorder of the pose graph, symbolic
    factorisation on 64-row x 32-column cells, per block column only the tiles that hold a nonzero): same solution as a dense
    solve — for a temporal chain, for a chain with long loop closures (fill-in), and for a graph too dense to pay (dense kernels)."""
    monkeypatch.setenv("PVLM_SPD_SPARSE_MIN", "0")
    rng = np.random.default_rng(P)
    pairs = [(p, p) for p in range(P)] + [(p, q) for p in range(P) for q in range(p + 1, min(p + 3, P))]
    if links == "loops":
        pairs += [(int(a), int(b)) for a, b in zip(rng.integers(0, P, 150), rng.integers(0, P, 150)) if a < b]
    if links == "dense":
        pairs += [(p, q) for p in range(P) for q in range(p + 1, P) if rng.uniform() < 0.3]
    n, rows, cols, mirror, blocks, scale, diag, rhs, M = _block_system(rng, P, pairs, constant={(0, 0), (0, 1), (7, 1)})
    info, x, out = _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag, rhs, repeat=2)      # the second solve reuses the cached plan
    want = np.linalg.solve(M, rhs)
    assert info == 0
    assert np.allclose(x, want, rtol=1e-9, atol=1e-12)
    plan = [l for l in out if l.startswith("plan")][0].split()
    print(" ".join(plan))
    assert (plan[1] == "tile-sparse") == (links != "dense")
    if links == "chain":
        assert float(plan[3]) < 0.15        # fraction of the dense factorisation's tile updates
    # not positive definite -> reported through info, as the dense path does
    diag2 = diag.copy(); diag2[n // 2] = -1e6
    info, _, _ = _solve_blocks(tmp_path, n, rows, cols, mirror, blocks, scale, diag2, rhs)
    assert info != 0


def test_solve_with_gpu_cholesky_matches_twin(oracle, tmp_path, monkeypatch):
    """The bundle-adjustment solve (reprojection sets + Schur complement) with the reduced camera system factorised on
    the GPU instead of the host skyline: same poses / points / step counts as the twin."""
    from tests.test_host_gpu import _bundle_scene
    monkeypatch.setenv("PVLM_GPU_CHOLESKY_MIN", "1")
    rng = np.random.default_rng(91)
    frames, tracks, _ = _bundle_scene(rng)
    fpath, spath = os.path.join(str(tmp_path), "bf.bin"), os.path.join(str(tmp_path), "bs.bin")
    host_io.write_frames(fpath, np.eye(4), frames)
    host_io.write_structure(spath, frames, tracks)
    out = host_io.run("bundle", fpath, spath, 1.5, 1, 12)
    summ = [l.split() for l in out if l.startswith("summary")][0]
    cams = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("cam ")])
    pts = np.array([[float(v) for v in l.split()[2:]] for l in out if l.startswith("point ")])
    res, aa, t, X = lm_twin.bundle_adjust(oracle, [dict(f) for f in frames], tracks, 1.5, True, 12)
    assert abs(float(summ[6]) - res["final_cost"]) <= 1e-6 * res["final_cost"]
    assert int(summ[8]) == res["successful"] and int(summ[10]) == res["unsuccessful"]
    assert np.abs(cams[:, :3] - aa).max() <= 1e-6 and np.abs(cams[:, 3:] - t).max() <= 1e-6
    assert np.abs(pts - X).max() <= 1e-6 * max(1.0, np.abs(X).max())
