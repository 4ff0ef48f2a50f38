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
