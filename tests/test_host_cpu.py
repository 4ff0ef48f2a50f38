"""Host-side logic of the C++ mirror that needs no GPU (runs in the CPU suite through the test driver):
FindNeighbors against the committed golden vector, pose-file I/O (util/FileIO.cpp format)."""
import os
import tempfile

import numpy as np

from tests import host_io
from tests.test_golden_cpu import load


def test_find_neighbors_mirror_matches_golden():
    g = load("neighbors.npz")
    poses, valid = g["poses"], g["valid"]
    scans = []
    for i in range(len(poses)):
        R = poses[i, :9].reshape(3, 3); t = poses[i, 9:]
        scans.append(dict(id=i, valid=int(valid[i]), R_wl=R, t_wl=t))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "n.bin")
        host_io.write_scans(path, scans, world=False)
        nb = [[int(v) for v in l.split()[2:]] for l in host_io.run("neighbors", path, 6)]
    off, ids = g["off"], g["ids"]
    assert nb == [ids[off[i]:off[i + 1]].tolist() for i in range(len(poses))]


def test_find_neighbors_loop_closures_match_oracle(oracle):
    """FindNeighbors' second half (lidar_mapping/LidarFeatureAssociate.cpp:60-111): a 20 m radius search over ALL scan centres,
    walked in ascending distance, adds the scans of an earlier pass through the same place (index gap > 200).  Trajectories that
    come back to where they were — a figure of eight and a doubled circle, with invalid scans in between — against the oracle."""
    rng = np.random.default_rng(5)
    for F, shape in ((520, "circle"), (640, "eight")):
        a = np.linspace(0, 2.3 * 2 * np.pi, F)                      # 2.3 turns: every place is visited two or three times
        if shape == "circle":
            c = np.stack([30 * np.cos(a), 30 * np.sin(a), 0.2 * np.sin(3 * a)], axis=1)
        else:
            c = np.stack([25 * np.sin(a), 18 * np.sin(2 * a), 0.1 * np.cos(a)], axis=1)
        c += rng.normal(size=c.shape) * 0.05
        valid = np.ones(F, np.int32); valid[rng.choice(F, 12, replace=False)] = 0
        poses = np.zeros((F, 12)); poses[:, [0, 4, 8]] = 1.0; poses[:, 9:] = c
        scans = [dict(id=i, valid=int(valid[i]), R_wl=np.eye(3), t_wl=c[i]) for i in range(F)]
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "n.bin")
            host_io.write_scans(path, scans, world=False)
            nb = [[int(v) for v in l.split()[2:]] for l in host_io.run("neighbors", path, 6)]
        want = oracle.find_neighbors(poses, valid, 6)
        assert nb == want
        far = sum(1 for i, l in enumerate(want) for v in l if abs(v - i) > 200)
        assert far > 100                                            # the loop-closure branch really fired


def test_pose_file_roundtrip():
    rng = np.random.default_rng(4)
    with tempfile.TemporaryDirectory() as d:
        src, dst, dst6 = os.path.join(d, "in.txt"), os.path.join(d, "out.txt"), os.path.join(d, "out6.txt")
        rows = []
        with open(src, "w") as f:
            for i in range(6):
                v = rng.normal(size=12)
                if i == 2:
                    f.write("scan_%d.pcd " % i + " ".join(["inf"] * 12) + "\n"); rows.append(None); continue
                if i == 4:
                    f.write(" ".join(repr(float(x)) for x in v) + "\n"); rows.append(("", v)); continue   # no name
                f.write("scan_%d.pcd " % i + " ".join(repr(float(x)) for x in v) + "\n"); rows.append(("scan_%d.pcd" % i, v))
        out = host_io.run("poseio", src, dst, 1, 17)
        assert out[0] == "poses 6" and out[3].endswith("valid=0") and out[1].endswith("valid=1")
        got = [l.split() for l in open(dst).read().splitlines()]
        assert len(got) == 6
        for r, g_ in zip(rows, got):
            if r is None:
                assert "inf" in " ".join(g_)
                continue
            name, v = r
            nums = g_[1:] if name else g_
            assert (g_[0] == name) if name else len(g_) == 12
            assert np.array_equal(np.array([float(x) for x in nums]), v)
        out2 = host_io.run("poseio", src, dst6, 0, 6)       # invalid rows dropped, reference's 6-digit precision
        assert out2[0] == "poses 5"
        first = open(dst6).readline().split()
        assert np.allclose([float(x) for x in first[1:]], rows[0][1], rtol=1e-5)


def _lzf_compress(data):
    """Small LZF encoder (the codec of PCD binary_compressed): greedy 3-byte hash matches, literal runs <= 32,
    back references of length 3..264 at offsets <= 8192 — enough to exercise both token kinds of the decoder."""
    out = bytearray(); lit = bytearray(); table = {}
    i, n = 0, len(data)

    def flush():
        for k in range(0, len(lit), 32):
            chunk = lit[k:k + 32]
            out.append(len(chunk) - 1); out.extend(chunk)
        lit.clear()
    while i < n:
        key = bytes(data[i:i + 3]); j = table.get(key, -1)
        if len(key) == 3:
            table[key] = i
        if j >= 0 and 0 < i - j <= 8192:
            length = 3
            while i + length < n and length < 264 and data[j + length] == data[i + length]:
                length += 1
            flush()
            off = i - j - 1; l = length - 2
            if l < 7:
                out.append((l << 5) | (off >> 8))
            else:
                out.append((7 << 5) | (off >> 8)); out.append(l - 7)
            out.append(off & 0xff)
            i += length
        else:
            lit.append(data[i]); i += 1
    flush()
    return bytes(out)


def test_load_lidar_pcd_formats():
    """Velodyne::LoadLidar (sensors/Velodyne.cpp:92-172): PCD ascii / binary / binary_compressed -> NaN removal,
    0.5 m near-point removal (float), axis swap (x, y, z) -> (x, -z, y), < 4000 points = invalid scan."""
    import struct
    rng = np.random.default_rng(17)
    n = 5000
    pts = (rng.normal(size=(n, 3)) * 4).astype(np.float32)
    inten = rng.uniform(0, 255, size=n).astype(np.float32)
    inten[:2000] = 100.0                                                           # compressible run: back references in the LZF stream
    pts[::97] = np.float32([np.nan, 1.0, 2.0]); pts[5::131, 2] = np.inf          # invalid returns
    pts[3::53] = (rng.normal(size=(len(pts[3::53]), 3)) * 0.2).astype(np.float32)   # closer than 0.5 m
    pts[7] = np.float32([0.3, 0.4, 0.0])                                           # exactly on the 0.5 m sphere: kept (dis < thr^2 is false)
    finite = np.isfinite(pts).all(axis=1)
    d2 = (pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1] + pts[:, 2] * pts[:, 2]).astype(np.float32)
    keep = finite & ~(d2 < np.float32(0.25))
    exp = np.stack([pts[keep, 0], -pts[keep, 2], pts[keep, 1], inten[keep]], axis=1)
    assert 4000 < len(exp) < n and keep[7]
    hdr = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n" \
          "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
    rec = np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)

    def parse(out):
        head = out[0].split()
        got = np.array([[float.fromhex(v) for v in l.split()[1:]] for l in out[1:] if l.startswith("p ")], np.float32).reshape(-1, 4)
        return int(head[1]), int(head[3]), int(head[5]), got
    with tempfile.TemporaryDirectory() as d:
        files = {}
        files["ascii"] = os.path.join(d, "a.pcd")
        with open(files["ascii"], "w") as f:
            f.write(hdr % (n, n, "ascii"))
            for p, i in zip(pts, inten):
                f.write(" ".join("nan" if np.isnan(v) else repr(float(v)) for v in p) + " " + repr(float(i)) + "\n")
        files["binary"] = os.path.join(d, "b.pcd")
        with open(files["binary"], "wb") as f:
            f.write((hdr % (n, n, "binary")).encode()); f.write(rec.tobytes())
        files["binary_compressed"] = os.path.join(d, "c.pcd")
        soa = rec.T.copy().tobytes()                       # field-major block
        comp = _lzf_compress(soa)
        assert len(comp) < 0.95 * len(soa)                  # back references were emitted
        with open(files["binary_compressed"], "wb") as f:
            f.write((hdr % (n, n, "binary_compressed")).encode()); f.write(struct.pack("<II", len(comp), len(soa))); f.write(comp)
        for mode, path in files.items():
            ok, valid, count, got = parse(host_io.run("loadpcd", path))
            assert ok == 1 and valid == 1 and count == len(exp), mode
            assert np.array_equal(got, exp), mode     # ascii: repr() round-trips float32 values exactly through strtod
        # fewer than 4000 points left -> invalid scan; a missing file -> not loaded
        small = os.path.join(d, "s.pcd")
        with open(small, "wb") as f:
            f.write((hdr % (3000, 3000, "binary")).encode()); f.write(rec[:3000].tobytes())
        ok, valid, count, _ = parse(host_io.run("loadpcd", small))
        assert ok == 1 and valid == 0 and count == int(keep[:3000].sum())
        ok, valid, count, _ = parse(host_io.run("loadpcd", os.path.join(d, "missing.pcd")))
        assert ok == 0 and count == 0


def test_point2line_segment_mirror_matches_oracle():
    """AssociatePoint2LineSegment (LidarFeatureAssociate.cpp:319-383) is host-only (brute-force point-to-line distances):
    the mirror against the oracle without a GPU."""
    from oracle import oracle as orc
    from panovlm_amd import synthetic as sy
    rng = np.random.default_rng(23)
    lines = sy.random_world_lines(rng, 10)
    scans = []
    for k in range(2):
        R, t = sy.estimated_pose(k + 2)
        s = sy.make_line_scan(rng, k, R, t, lines, pts_per_line=(10, 24), extra_pts=12, noise=0.005)
        seg_points = [[i for i, l in enumerate(s["p2s"]) if sid in l] for sid in range(len(s["seg_size"]))]
        scans.append(dict(id=k, R_wl=R, t_wl=t, corner_local=s["corner_local"], p2s=s["p2s"], seg_points=seg_points,
                          seg_coeffs=s["seg_coeffs"], end_points=s["end_points"], _oracle=s))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "l.bin")
        host_io.write_scans(path, scans)
        got = np.array([[float(v) for v in l.split()[1:]] for l in host_io.run("p2line", path, 0, 1, 2, 0.3)]).reshape(-1, 9)
    o = orc.assoc_point2line(scans[0]["_oracle"], scans[1]["_oracle"], 0.3, mode="segment")
    assert len(got) == len(o["point"]) > 50
    assert np.allclose(got[:, :3], o["point"], rtol=0, atol=1e-12) and np.allclose(got[:, 3:6], o["a"], rtol=0, atol=1e-12)
    assert np.allclose(got[:, 6:9], o["b"], rtol=0, atol=1e-12)


def test_pcd_reader_refuses_corrupt_files():
    """Velodyne::LoadLidar on truncated / inconsistent / hostile .pcd files: refused quickly (the reference logs and returns
    false, sensors/Velodyne.cpp:100-104), never a crash, a hang or an allocation the file cannot back."""
    import struct
    import subprocess
    rng = np.random.default_rng(0)
    n = 5000
    pts = (rng.normal(size=(n, 4)) * 3).astype(np.float32)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
           "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n")
    good = (hdr % (n, n, "binary")).encode() + pts.tobytes()
    bad = {
        "truncated_body": good[:len(good) // 2], "truncated_header": good[:60],
        "huge_points": (hdr % (n, 2 ** 31 - 1, "binary")).encode() + pts.tobytes(),
        "negative_points": (hdr % (n, -5, "binary")).encode() + pts.tobytes(),
        "bad_data_kw": (hdr % (n, n, "weird")).encode() + pts.tobytes(),
        "compressed_garbage": (hdr % (n, n, "binary_compressed")).encode() + struct.pack("<II", 100, n * 16) + bytes(rng.integers(0, 256, 100, dtype=np.uint8)),
        "compressed_sizes_lie": (hdr % (n, n, "binary_compressed")).encode() + struct.pack("<II", 2 ** 31, 2 ** 31) + b"abc",
        "compressed_huge_points": (hdr % (n, 2 ** 28 - 1, "binary_compressed")).encode() + struct.pack("<II", 3, (2 ** 28 - 1) * 16) + b"abc",
        "ascii_short_rows": (hdr % (n, n, "ascii")).encode() + b"1 2\n3 4 5 6\nfoo bar baz qux\n",
        "ascii_huge_points": (hdr % (n, 2 ** 40, "ascii")).encode() + b"1 2 3 4\n",
        "size8_coordinates": good.replace(b"SIZE 4 4 4 4", b"SIZE 8 8 8 8"), "size0_field": good.replace(b"SIZE 4 4 4 4", b"SIZE 4 4 4 0"),
        "count_huge": good.replace(b"COUNT 1 1 1 1", b"COUNT 1 1 1 99999999"), "no_xyz": good.replace(b"FIELDS x y z intensity", b"FIELDS a b c d"),
        "random_bytes": bytes(rng.integers(0, 256, 4096, dtype=np.uint8)), "empty": b"",
    }
    drv = host_io.driver()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "good.pcd")
        open(p, "wb").write(good)
        out = subprocess.run([drv, "loadpcd", p], capture_output=True, timeout=30).stdout.decode().split("\n")[0].split()
        assert out[1] == "1" and int(out[5]) > 4900
        for name, data in bad.items():
            p = os.path.join(d, name + ".pcd")
            open(p, "wb").write(data)
            r = subprocess.run([drv, "loadpcd", p], capture_output=True, timeout=30)
            head = r.stdout.decode(errors="ignore").split("\n")[0].split()
            assert r.returncode == 0 and head[:2] == ["loaded", "0"] and head[5] == "0", (name, r.returncode, head)


def test_mvs_select_neighbor_knn_mirror_matches_oracle():
    """MVS::SelectNeighborKNN (mvs/MVS.cpp:334-382): host mirror (rigid inverse) against the oracle (general 4x4 inverse, as
    upstream) — neighbour ids equal, R_nr / t_nr equal after the rounding to float except for last-bit flips — and against
    numpy for the geometry (X_n = R_nr X_r + t_nr)."""
    import struct
    from oracle import oracle as orc
    from tests import synth
    rng = np.random.default_rng(17)
    n = 40
    R = np.array([synth.rodrigues(rng.normal(size=3) * 0.4) for _ in range(n)])
    t = np.cumsum(rng.normal(size=(n, 3)) * 0.3, axis=0)
    t[7] = t[6] + 1e-4                                      # closer than the threshold: skipped as a neighbour of each other
    valid = np.ones(n, np.int32); valid[[3, 20]] = 0        # frames without a pose neither get nor become neighbours
    for k, thr in ((4, 0.01), (8, 0.25), (30, 0.0)):
        ids, oR, ot = orc.mvs_select_neighbors(valid, R, t, k, thr)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "poses.bin")
            with open(path, "wb") as f:
                f.write(struct.pack("<i", n))
                for i in range(n):
                    f.write(struct.pack("<i", int(valid[i]))); f.write(R[i].astype(np.float64).tobytes()); f.write(t[i].astype(np.float64).tobytes())
            out = host_io.run("mvsneighbors", path, k, repr(thr))
        got = {}
        for l in out:
            if l.startswith("nb "):
                w = l.split()
                got.setdefault(int(w[1]), []).append((int(w[2]), np.array([float.fromhex(x) for x in w[3:12]], np.float32), np.array([float.fromhex(x) for x in w[12:15]], np.float32)))
        total = 0
        for i in range(n):
            want = [j for j in ids[i] if j >= 0]
            assert [g[0] for g in got.get(i, [])] == want, (i, want)
            if not valid[i]:
                assert not want
            assert i not in want and all(valid[j] for j in want) and len(want) <= k
            for q, (j, Rg, tg) in enumerate(got.get(i, [])):
                assert np.abs(Rg - oR[i, q]).max() <= 2e-7 and np.abs(tg - ot[i, q]).max() <= 1e-6 * max(1.0, np.abs(ot[i, q]).max())
                Rn = R[j].T @ R[i]; tn = R[j].T @ (t[i] - t[j])
                assert np.abs(Rg.reshape(3, 3) - Rn).max() < 1e-6 and np.abs(tg - tn).max() < 1e-5
                d2 = np.float32(((t[i].astype(np.float32) - t[j].astype(np.float32)) ** 2).sum())
                assert d2 >= np.float32(thr) * (1 - 1e-6)
                total += 1
        assert total > n
        if thr == 0.01:
            assert 7 not in [j for j in ids[6] if j >= 0] and 6 not in [j for j in ids[7] if j >= 0]



def _fusion_scene(n=5, rows=40, cols=80, seed=3, noise=0.002, baseline=1.0):
    """n panoramas of the synthetic room with their true depth maps (+ a little noise), confidences, colour images and poses."""
    from oracle import oracle as orc
    from tests import synth
    rng = np.random.default_rng(seed)
    poses = [(synth.rodrigues(np.array([0.03 * k, 0.3 * k - 0.5, 0.02])), baseline * np.array([0.5 * k - 1.0, 0.05 * k, 0.4 * k - 0.8])) for k in range(n)]
    depth, bgr, conf, T = [], [], [], []
    for R, t in poses:
        g, d, _ = synth.render_panorama(orc, rows, cols, R, t)
        depth.append((d * rng.uniform(1 - noise, 1 + noise, size=d.shape)).astype(np.float32))
        c = np.stack([g, (g.astype(np.int32) * 3 // 4).astype(np.uint8), (g.astype(np.int32) // 2).astype(np.uint8)], -1)       # b > g > r: bluish, mostly dark
        bgr.append(np.ascontiguousarray(c)); conf.append(rng.uniform(0.2, 1.0, size=d.shape).astype(np.float32))
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = t; T.append(M)
    nb = [[(k,) + synth.relative_pose(poses[v][0], poses[v][1], poses[k][0], poses[k][1]) for k in range(n) if k != v] for v in range(n)]
    return depth, bgr, conf, T, nb


def test_fuse_depth_images_oracle_properties():
    """MVS::FuseDepthImages (mvs/MVS.cpp:2168-2334), the oracle's restatement on views of the synthetic room: fused points lie on the room's
    surfaces, a pixel is used at most once, nothing is fused without agreement, and the accepted points zero the depths they stand in front of."""
    from oracle import oracle as orc
    # upstream compares the reference's range with the NEIGHBOUR's range at the projected pixel (:2262), i.e. it asks both cameras to be
    # equally far from the point: short baselines, so that most pixels pass
    depth, bgr, conf, T, nb = _fusion_scene(baseline=0.1)
    n, (rows, cols) = len(depth), depth[0].shape
    xyz, rgb, after = orc.mvs_fuse_depth_images(depth, [None] * n, conf, bgr, T, nb, max_depth=20.0, thr=0.02)
    valid = sum(int(((d > 0) & (d < 16.0)).sum()) for d in depth)
    assert 0.1 * valid < len(xyz) < 0.34 * valid                     # every point consumed >= 3 pixels (its own + >= 2 neighbours')
    half = np.array([4.0, 1.5, 6.0])
    off = np.abs(np.abs(xyz) - half).min(axis=1)                     # distance to the nearest wall plane
    assert np.median(off) < 0.02 and np.quantile(off, 0.99) < 0.25 and np.all(np.abs(xyz) < half + 0.3)
    assert rgb.max() > 60 and np.all(rgb[:, 2] >= rgb[:, 0])         # colours are weighted means of the bluish images (r <= b)
    none, _, _ = orc.mvs_fuse_depth_images(depth, [None] * n, conf, bgr, T, nb, max_depth=20.0, thr=0.0)
    assert len(none) == 0                                            # |d - d'| / d < 0 never holds: no agreement, every claim withdrawn
    lonely, _, _ = orc.mvs_fuse_depth_images(depth, [None] * n, conf, bgr, T, [[] for _ in range(n)], thr=0.02)
    assert len(lonely) == 0
    # a frame with one neighbour only can never collect two agreeing views
    one = [[nb[v][0]] for v in range(n)]
    assert len(orc.mvs_fuse_depth_images(depth, [None] * n, conf, bgr, T, one, thr=0.02)[0]) == 0
    # an occluder in front of the scene in view 1 (too-small depths): the points of the other views stand behind it -> those depths are zeroed
    d2 = [d.copy() for d in depth]; d2[1][10:20, 30:50] *= 0.5
    _, _, after2 = orc.mvs_fuse_depth_images(d2, [None] * n, conf, bgr, T, nb, thr=0.02)
    assert all(a is None for a in after2)                            # symmetric lists: neighbours + 1 visits each, every map released at the end
    # ... visible on a frame that is nobody's neighbour: it keeps its last reference until it has been the reference itself, but is
    # visited last (fewest neighbours) — give it none and it is never a neighbour nor has neighbours: its map is released untouched.
    lists = [[x for x in nb[v] if x[0] != 1] for v in range(n)]; lists[1] = []
    ref_only = orc.mvs_fuse_depth_images(d2, [None] * n, conf, bgr, T, lists, thr=0.02)
    assert len(ref_only[0]) > 0 and all(a is None for a in ref_only[2])
    # frame 1 (the occluder) keeps references: 4 neighbours of its own, but a neighbour of frames 0 and 2 only -> its map survives, with the
    # occluding depths zeroed by the points of 0 and 2 that stand behind them (cv::norm(X1) < n_depth is false there; the occluder's depths are
    # SMALLER than |X1|... so it is the scene BEHIND a too-far depth that is cleared: make the patch too far instead)
    d3 = [d.copy() for d in depth]; d3[1][10:20, 30:50] *= 1.5
    lists = [[x for x in nb[v] if x[0] != 1 or v in (0, 2)] for v in range(n)]
    kept = orc.mvs_fuse_depth_images(d3, [None] * n, conf, bgr, T, lists, thr=0.02)[2]
    assert kept[1] is not None and all(kept[k] is None for k in (0, 2, 3, 4))
    patch = np.zeros((rows, cols), bool); patch[10:20, 30:50] = True
    cleared = (kept[1] == 0) & (d3[1] > 0)
    # (no margin in the test: on consistent surfaces noise and pixel rounding put about half of the projected points in front, too)
    assert cleared[patch].mean() > cleared[~patch].mean() + 0.15 and 0 < cleared[~patch].mean() < 0.4 and np.array_equal(kept[1][~cleared], d3[1][~cleared])
    # max_depth: pixels at >= 0.8 max_depth do not start points
    far, _, _ = orc.mvs_fuse_depth_images(depth, [None] * n, conf, bgr, T, nb, max_depth=5.0, thr=0.02)
    assert 0 < len(far) < len(xyz)


def test_fuse_depth_images_host_mirror_equals_oracle():
    """pvlm::FuseDepthImages (host code, as upstream) against the oracle, bit for bit, on inputs that drive upstream's state machine: sky-coloured
    fused points (the `continue` that skips the list clears), asymmetric neighbour lists (maps released early, read again from the depth file,
    references skipped), invalid depths, an occluder."""
    from oracle import oracle as orc
    from tests import host_io
    depth, bgr, conf, T, nb = _fusion_scene(n=6, seed=9, baseline=0.15)
    n, (rows, cols) = len(depth), depth[0].shape
    rng = np.random.default_rng(1)
    for k in range(n):
        depth[k][rng.random((rows, cols)) < 0.05] = 0.0
        sky = rng.random((rows, cols)) < 0.25
        bgr[k][sky] = np.stack([rng.integers(200, 256, sky.sum()), rng.integers(120, 200, sky.sum()), rng.integers(60, 120, sky.sum())], 1).astype(np.uint8)
    depth[2][5:15, 20:40] *= 0.6
    files = [(d * rng.uniform(0.99, 1.01, size=d.shape)).astype(np.float32) for d in depth]
    files[4] = None                                                  # no depth file for frame 4: once released it stays empty
    # asymmetric lists: 0 and 1 are everybody's neighbours (released and re-read), 5 is nobody's; frame 3 arrives without a filtered map
    lists = [[x for x in nb[v] if x[0] in (0, 1, (v + 1) % n, (v + 2) % n)] for v in range(n)]
    lists[5] = [x for x in nb[5] if x[0] in (0, 1, 2, 3)]
    filt = list(depth); filt[3] = None
    for kw in (dict(thr=0.02), dict(thr=0.05, max_depth=9.0)):
        want = orc.mvs_fuse_depth_images(filt, files, conf, bgr, T, lists, frame_id=np.arange(n) + 7, **kw)
        got = host_io.fuse_depth_images(filt, files, conf, bgr, T, lists, frame_id=np.arange(n) + 7, **kw)
        assert len(want[0]) > 200
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        for a, b in zip(got[2], want[2]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    # the scene does reach those paths: without the depth files the released maps stay empty (fewer points), and the result is not the
    # sky-free result minus the sky points (the lists that survive the `continue` admit points the clean run rejects)
    assert len(orc.mvs_fuse_depth_images(filt, [None] * n, conf, bgr, T, lists, thr=0.02)[0]) < 0.7 * len(orc.mvs_fuse_depth_images(filt, files, conf, bgr, T, lists, thr=0.02)[0])
    plain = [np.stack([b[..., 0], b[..., 0] * 0 + 90, b[..., 0] * 0 + 90], -1).astype(np.uint8) for b in bgr]      # no sky-coloured pixel anywhere
    with_sky = {tuple(p) for p in orc.mvs_fuse_depth_images(filt, files, conf, bgr, T, lists, thr=0.02)[0].tolist()}
    without = {tuple(p) for p in orc.mvs_fuse_depth_images(filt, files, conf, plain, T, lists, thr=0.02)[0].tolist()}
    assert len(with_sky - without) > 20
    # the symmetric, well-behaved case as well
    depth, bgr, conf, T, nb = _fusion_scene(baseline=0.1)
    want = orc.mvs_fuse_depth_images(depth, [None] * 5, conf, bgr, T, nb, thr=0.02)
    got = host_io.fuse_depth_images(depth, [None] * 5, conf, bgr, T, nb, thr=0.02)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_balanced_ranges_of_a_sharded_run():
    """Exchange::BalancedRange (SURVEY.md §8 row E's rebalancing by sum Nq): contiguous, disjoint, covering ranges whose summed weights
    differ by at most one scan's weight; every rank computes the same boundaries; zero weights fall back to the block partition."""
    rng = np.random.default_rng(4)
    for world, w in ((2, [5, 1, 1, 1, 1, 1]), (3, rng.integers(1, 50, size=40).tolist()), (8, [7] * 100 + [70] * 5 + [7] * 60), (4, [0] * 10), (8, [3, 3]),
                     (5, [0, 0, 9, 0, 0, 0, 1])):
        out = host_io.run("balance", world, *w)
        rg = [[int(v) for v in l.split()[1:]] for l in out if l.startswith("range")]
        assert len(rg) == world
        assert all(r[1:3] == r[3:5] for r in rg)                     # own range == the range any other rank computes for it
        assert rg[0][1] == 0 and rg[-1][2] == len(w) and all(rg[k][2] == rg[k + 1][1] for k in range(world - 1))
        assert all(r[1] <= r[2] for r in rg)
        tot = float(sum(w))
        if tot > 0:
            sums = [sum(w[r[1]:r[2]]) for r in rg]
            assert max(sums) <= tot / world + max(w)                 # no rank carries more than its share plus one scan
        else:
            assert [r[1:3] for r in rg] == [[len(w) * k // world, len(w) * (k + 1) // world] for k in range(world)]


def test_frame_neighbors_spatial_and_temporal(tmp_path):
    """CameraLidarOptimizer::NeighborEachFrame (joint_optimization/CameraLidarOptimizer.cpp:551-610): the temporal window JointOptimize uses, and the spatial
    branch — the k scans whose centres are nearest to the camera centre (float32, first among equals), plus the scans before / after the frame's own index —
    against a numpy restatement.  Scans marked invalid do not take part; a frame without a pose gets no list."""
    rng = np.random.default_rng(4)
    n = 23
    t_l = np.cumsum(rng.normal(0, 0.6, (n, 3)), axis=0)
    t_l[7] = t_l[3]                                                   # equal distances: position decides
    scans = [dict(id=k, valid=0 if k in (5, 11) else 1, R_wl=np.eye(3), t_wl=t_l[k]) for k in range(n)]
    frames = [dict(id=k, rows=64, cols=128, valid=0 if k == 9 else 1, R_wc=np.eye(3), t_wc=t_l[min(k, n - 1)] + rng.normal(0, 0.3, 3), lines=np.zeros((0, 4)))
              for k in range(n + 2)]
    lp, fp = str(tmp_path / "l.bin"), str(tmp_path / "f.bin")
    host_io.write_scans(lp, scans)
    host_io.write_frames(fp, np.eye(4), frames)

    def lists(k, temporal):
        out = [[int(v) for v in l.split()[1:]] for l in host_io.run("frame_neighbors", lp, fp, k, 1 if temporal else 0) if l.startswith("nb")]
        assert len(out) == len(frames)
        return out

    for k in (3, 6, 40):
        got = lists(k, True)
        for f in range(len(frames)):
            start = max(0, f - k // 2); end = min(n, start + k); start = max(0, end - k)
            assert got[f] == list(range(start, end))
        got = lists(k, False)
        owner = [i for i in range(n) if scans[i]["valid"]]
        c = np.asarray([t_l[i] for i in owner], np.float32)
        for f, fr in enumerate(frames):
            if not fr["valid"]:
                assert got[f] == []
                continue
            d = c - np.asarray(fr["t_wc"], np.float32)
            s = np.zeros(len(c), np.float32)
            for a in range(3):
                s = (s + d[:, a] * d[:, a]).astype(np.float32)
            nearest = [owner[j] for j in np.argsort(s, kind="stable")[:k]]
            want = list(nearest)
            if f - 1 >= 0 and f - 1 not in nearest:
                want.append(f - 1)
            if f + 1 < n and f + 1 not in nearest:
                want.append(f + 1)
            assert got[f] == want, (k, f)
