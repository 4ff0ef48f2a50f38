"""CPU suite: the oracle against the committed golden vectors, and the oracle's FastAtan2 against
the REAL reference header (oracle/_ref, compiled from /root/reference/base/Math.h)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from tests import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_fast_atan2_golden_is_the_real_reference(oracle):
    g = load("fast_atan2.npz")
    y, x = g["y"], g["x"]
    of = np.array([oracle.fast_atan2_f(a, b) for a, b in zip(y.astype(np.float32), x.astype(np.float32))], np.float32)
    od = np.array([oracle.fast_atan2_d(a, b) for a, b in zip(y, x)])
    assert np.array_equal(of, g["out_f32"], equal_nan=True)
    assert np.array_equal(od, g["out_f64"], equal_nan=True)
    # accuracy claim of base/Math.h:9-10 (~0.3 degree) on non-degenerate inputs
    m = (np.abs(x) + np.abs(y)) > 1e-6
    assert np.abs(od[m] - np.arctan2(y[m], x[m])).max() < np.deg2rad(0.35)


def test_ref_build_matches_oracle_bitwise(oracle):
    ref = oracle.ref_math()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(77)
    y = rng.normal(size=20000) * rng.choice([1e-3, 1, 100], size=20000)
    x = rng.normal(size=20000) * rng.choice([1e-3, 1, 100], size=20000)
    for a, b in zip(y[:4000], x[:4000]):
        assert ref.ref_fast_atan2_d(a, b) == oracle.fast_atan2_d(a, b)
        af, bf = np.float32(a), np.float32(b)
        assert ref.ref_fast_atan2_f(C.c_float(af), C.c_float(bf)) == oracle.fast_atan2_f(af, bf)


@pytest.mark.parametrize("kind,normalize", [(0, 0), (1, 0), (1, 1), (2, 0), (3, 0), (3, 1), (4, 0), (5, 0)])
def test_functor_golden(oracle, kind, normalize):
    g = load("functors.npz")
    k = "k%d_n%d_" % (kind, normalize)
    rid, nid = synth.expand_ids(g[k + "off"], g[k + "ref"], g[k + "nei"])
    r, J = oracle.evaluate(kind, synth.oracle_rows(kind, g[k + "rows"], float(g["weight"])), rid, nid, g[k + "aa"], g[k + "t"], normalize=bool(normalize))
    assert np.allclose(r, g[k + "r"], rtol=1e-12, atol=1e-15) and np.allclose(J, g[k + "J"], rtol=1e-11, atol=1e-13)


def _scan(g, k):
    return {kk: g["s%d_%s" % (k, kk)] for kk in ("R_wl", "t_wl", "flat_xyz", "flat_tag", "less_xyz", "less_tag")} | {"id": k}


def test_assoc_point2plane_golden(oracle):
    g = load("assoc_point2plane.npz")
    for i, (r, n, tol, thr) in enumerate(g["cases"]):
        o = oracle.assoc_point2plane(_scan(g, int(r)), _scan(g, int(n)), float(tol), float(thr))
        assert np.array_equal(o["qidx"], g["c%d_qidx" % i]) and np.array_equal(o["nn"], g["c%d_nn" % i])
        assert np.array_equal(o["point"], g["c%d_point" % i]) and np.array_equal(o["plane"], g["c%d_plane" % i])
        assert len(o["qidx"]) > 50


def test_equirect_golden(oracle):
    g = load("equirect.npz")
    for rows, cols in [(2880, 5760), (720, 1440)]:
        assert np.array_equal(oracle.cam_to_image(rows, cols, g["cam"].astype(np.float32)), g["px_f32_%d" % rows], equal_nan=True)
        assert np.array_equal(oracle.cam_to_image(rows, cols, g["cam"]), g["px_f64_%d" % rows], equal_nan=True)
        assert np.allclose(oracle.image_to_cam(rows, cols, g["pix_%d" % rows], 1.0), g["cam_f64_%d" % rows], atol=1e-15)
        seg = oracle.break_to_segments(rows, cols, [100.0, 200.0], [cols - 150.0, rows - 300.0], 100.0)
        assert np.allclose(seg, g["seg_%d" % rows], atol=1e-3)
        # the polyline wraps once around the image seam: left (x=0) and right (x=cols-1) appear
        assert np.any(seg[:, 0] == 0) and np.any(seg[:, 0] == cols - 1)


def line_scan(g, p, local=False):
    off = g[p + "_p2s_off"]; ids = g[p + "_p2s_ids"]
    return dict(R_wl=g[p + "_R_wl"], t_wl=g[p + "_t_wl"], corner_xyz=g[p + ("_corner_local" if local else "_corner_xyz")],
                p2s=[ids[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)], seg_size=g[p + "_seg_size"],
                seg_coeffs=g[p + "_seg_coeffs"], end_points=g[p + "_end_points"])


def test_line2line_golden(oracle):
    g = load("lines.npz")
    a, b = line_scan(g, "a"), line_scan(g, "b")
    for thr in (0.3, 0.4):
        t = "t%02d_" % int(thr * 10)
        o = oracle.assoc_line2line(a, b, thr)
        assert np.array_equal(o["votes"], g[t + "votes"]) and np.array_equal(o["nei_idx"], g[t + "nei_idx"]) and np.array_equal(o["ref_idx"], g[t + "ref_idx"])
        assert np.allclose(o["p1"], g[t + "p1"], atol=1e-14) and np.allclose(o["p2"], g[t + "p2"], atol=1e-14)
        assert len(o["nei_idx"]) >= 4


def test_by_angle_golden(oracle):
    g = load("lines.npz")
    scan = line_scan(g, "c", local=True)
    for mult in (1, 0):
        o = oracle.assoc_by_angle(2880, 5760, g["c_lines"], scan, g["c_T_cl"], multiple=bool(mult))
        m = "m%d_" % mult
        assert np.array_equal(o["image_line_id"], g[m + "image_line_id"]) and np.array_equal(o["lidar_line_id"], g[m + "lidar_line_id"])
        assert np.allclose(o["score"], g[m + "score"], atol=1e-9) and np.allclose(o["start"], g[m + "start"], atol=1e-12)
    assert np.array_equal(o["votes"], g["c_votes"])


def test_find_neighbors_golden(oracle):
    g = load("neighbors.npz")
    nb = oracle.find_neighbors(g["poses"], g["valid"], 6)
    off, ids = g["off"], g["ids"]
    for i, l in enumerate(nb):
        assert l == ids[off[i]:off[i + 1]].tolist()
    assert all(i not in l for i, l in enumerate(nb) if g["valid"][i])   # self removed
    assert nb[13] == [16, 15, 14, 13, 12, 11, 10]                       # invalid pose -> temporal window (LidarFeatureAssociate.cpp:103-107)


def test_reproj_golden(oracle):
    g = load("reproj.npz")
    r, J = oracle.evaluate_reproj(g["bearing"], float(g["weight"]), g["cam"], g["pt"], g["aa"], g["t"], g["X"])
    assert np.array_equal(r, g["r"]) and np.array_equal(J, g["J"])


def test_depth_golden(oracle):
    g = load("depth.npz")
    for size in (3, 2):
        img = oracle.project_lidar_depth(int(g["rows"]), int(g["cols"]), g["xyz"], g["T_cl"], size)
        assert np.array_equal(img, g["depth_size%d" % size]) and (img > 0).mean() > 0.02


def test_undistort_golden(oracle):
    """Velodyne::UndistortCloud and SlerpPose (oracle/undistort.hpp) against tests/golden/undistort.npz."""
    g = load("undistort.npz")
    for k in range(int(g["cases"])):
        done, out = oracle.undistort_cloud(g["cloud%d" % k], g["R_wl%d" % k], g["t_wl%d" % k], g["R_we%d" % k], g["t_we%d" % k])
        assert done and np.array_equal(out.view(np.uint32), g["out%d" % k].view(np.uint32)), k
    for r, want in zip(g["ratios"], g["slerp"]):
        R, t = oracle.slerp_pose(g["pose_w1"][:3, :3], g["pose_w1"][:3, 3], g["pose_w2"][:3, :3], g["pose_w2"][:3, 3], float(r))
        assert np.array_equal(R, want[:3, :3]) and np.array_equal(t, want[:3, 3])


def test_mvs_golden(oracle):
    g = load("mvs.npz")
    neis = [g["nei%d_gray" % k] for k in range(3)]; nd = [g["nei%d_depth" % k] for k in range(3)]
    c, d, _ = oracle.mvs_init_conf_map(g["gray"], neis, g["R_nr"], g["t_nr"], g["depth"], g["normal"], 3, 1)
    assert np.array_equal(c, g["conf_pho"]) and np.array_equal(d, g["depth_pho"])
    cg, _, _ = oracle.mvs_init_conf_map(g["gray"], neis, g["R_nr"], g["t_nr"], g["depth"], g["normal"], 3, 1, nei_depths=nd)
    assert np.array_equal(cg, g["conf_geo"]) and (c > -1).mean() > 0.6
    ds, ns, cs = oracle.mvs_propagate(g["gray"], neis, g["R_nr"], g["t_nr"], g["depth_pho"], g["normal"], g["conf_pho"], max_iter=1, seed=int(g["sweep_seed"]))
    assert np.array_equal(ds, g["depth_sweep"]) and np.array_equal(ns, g["normal_sweep"]) and np.array_equal(cs, g["conf_sweep"])
    nc = [g["nei%d_conf" % k] for k in range(3)]
    dr, cr, ca = oracle.mvs_filter_depth_refine(nd, nc, g["R_nr"], g["t_nr"], g["depth_sweep"], np.clip(g["conf_sweep"], 0, None), thr=0.02, min_depth=0.1, max_depth=20.0)
    assert np.array_equal(dr, g["depth_refine"]) and np.array_equal(cr, g["conf_refine"]) and np.array_equal(ca, g["conf_after_refine"])
    assert 0.05 < (dr > 0).mean() < 0.95


def test_mvs_cloud_golden(oracle):
    """tests/golden/mvs_cloud.npz: DepthImageToCloud / DepthNormalToCloud and one sequential PatchMatch iteration (make_golden.py mvs_cloud)."""
    g = np.load(os.path.join(G, "mvs_cloud.npz"))
    xyz, rgb = oracle.mvs_depth_to_cloud(g["depth"], g["bgr"], g["T_wc"], float(g["max_depth"]))
    assert np.array_equal(xyz, g["xyz"]) and np.array_equal(rgb, g["rgb"]) and 0.3 * g["depth"].size < len(xyz) < 0.8 * g["depth"].size
    a, b, c = oracle.mvs_depth_to_cloud(g["depth"], g["bgr"], g["T_wc"], float(g["max_depth"]), filter_sky=False, normal=g["normal"])
    assert np.array_equal(a, g["xyz_all"]) and np.array_equal(b, g["rgb_all"]) and np.array_equal(c, g["normal_all"]) and len(a) > len(xyz)
    m = np.load(os.path.join(G, "mvs.npz"))
    neis = [m["nei%d_gray" % k] for k in range(3)]
    ds, ns, cs = oracle.mvs_propagate(m["gray"], neis, m["R_nr"], m["t_nr"], m["depth_pho"], m["normal"], m["conf_pho"], max_iter=1, seed=int(m["sweep_seed"]), sequential=True)
    assert np.array_equal(ds, g["depth_seq"]) and np.array_equal(ns, g["normal_seq"]) and np.array_equal(cs, g["conf_seq"])
    assert not np.array_equal(ds, m["depth_sweep"])          # a different sweep from the checkerboard one of mvs.npz


def test_refvec_container_round_trip(tmp_path):
    """tools/refvec.py (re-pinning recipe against a real PanoVLM build): export -> read back == fixture inputs,
    and `compare` accepts the fixtures' own expectations."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("refvec", os.path.join(root, "tools", "refvec.py"))
    rv = importlib.util.module_from_spec(spec); spec.loader.exec_module(rv)
    rv.export(str(tmp_path))
    for fx in rv.FIXTURES:
        z = np.load(os.path.join(rv.GOLDEN, fx + ".npz"))
        b = rv.read_pvv(os.path.join(str(tmp_path), fx + ".in.pvv"))
        for k, v in b.items():
            assert np.array_equal(np.asarray(z[k], v.dtype).reshape(v.shape), v), (fx, k)
        rv.write_pvv(os.path.join(str(tmp_path), fx + ".ref.pvv"),
                     {k: z[k] for k in z.files if rv._is_output(fx, k) and not rv.is_internal(fx, k)})
    assert rv.compare(str(tmp_path)) == 0


def test_repin_kit_one_command(tmp_path):
    """`make repin REFERENCE=...` (the one command of tools/repin/README.md) as a dry run: it must name the reference's own build, the dump tool,
    the regeneration of the fixtures and the CPU suite; `refvec.py table` must cover every fixture; `regenerate` must rewrite exactly the
    expected-output arrays (fed here with the fixtures' own expectations standing in for a reference run) and tag the files."""
    import importlib.util, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "PanoVLM" / "base"
    fake.mkdir(parents=True)
    (fake / "CostFunction.h").write_text("// stand-in for the dry run\n")
    out = subprocess.run(["make", "-n", "repin", "REFERENCE=" + str(tmp_path / "PanoVLM")], cwd=root, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    for needle in ("cmake -S \"%s\"" % (tmp_path / "PanoVLM"), "cmake -S tools/repin", "--target repin_regenerate", "refvec.py table", "pytest tests -q -m \"not gpu\""):
        assert needle in out.stdout, (needle, out.stdout)
    bad = subprocess.run(["make", "repin"], cwd=root, capture_output=True, text=True)
    assert bad.returncode != 0 and "usage: make repin REFERENCE=" in bad.stdout
    spec = importlib.util.spec_from_file_location("refvec", os.path.join(root, "tools", "refvec.py"))
    rv = importlib.util.module_from_spec(spec); spec.loader.exec_module(rv)
    assert set(rv.PINS) == set(rv.FIXTURES)
    tab = subprocess.run([sys.executable, os.path.join(root, "tools", "refvec.py"), "table"], capture_output=True, text=True).stdout
    for fx in rv.FIXTURES:
        assert "`%s.npz`" % fx in tab
    pvv = tmp_path / "pvv"; gold = tmp_path / "golden"
    rv.export(str(pvv))
    for fx in rv.FIXTURES:                      # a "reference run" that returns what the fixtures expect
        z = np.load(os.path.join(rv.GOLDEN, fx + ".npz"))
        rv.write_pvv(str(pvv / (fx + ".ref.pvv")), {k: z[k] for k in z.files if rv._is_output(fx, k) and not rv.is_internal(fx, k)})
    assert rv.compare(str(pvv)) == 0
    assert rv.regenerate(str(pvv), str(gold), "PanoVLM deadbeef") == 0
    for fx in rv.FIXTURES:
        a, b = np.load(os.path.join(rv.GOLDEN, fx + ".npz")), np.load(str(gold / (fx + ".npz")))
        assert str(b["__pinned__"][0]) == "PanoVLM deadbeef"
        assert set(a.files) | {"__pinned__"} == set(b.files)
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (fx, k)
