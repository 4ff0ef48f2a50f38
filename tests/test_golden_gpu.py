"""HIP path against the committed golden vectors (through the C ABI)."""
import os

import numpy as np
import pytest

from tests import synth
from tests.test_golden_cpu import _scan, line_scan, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kind,normalize", [(0, 0), (1, 0), (1, 1), (2, 0), (3, 0), (3, 1), (4, 0), (5, 0)])
def test_functor_golden(ctx, kind, normalize):
    import panovlm_amd as pv
    g = load("functors.npz")
    k = "k%d_n%d_" % (kind, normalize)
    rs = pv.ResidualSet.upload(ctx, kind, g[k + "rows"], g[k + "off"], g[k + "ref"], g[k + "nei"], flags=normalize, weight=float(g["weight"]))
    ctx.set_poses(g[k + "aa"], g[k + "t"])
    r, J = rs.eval()
    # arbiter: the reference's statements evaluated in x87 extended precision (oracle/costfunction.hpp AutoDiffEvaluateExt) —
    # plain 1e-6, no conditioning allowance; the double evaluation of the same formula is compared with the same gate
    ro, Jo = g[k + "r_ext"], g[k + "J_ext"]
    assert np.array_equal(r == 0, g[k + "r"] == 0)
    assert np.all(np.abs(r - ro) <= 1e-6 * np.abs(ro) + 1e-13)
    assert np.all(np.abs(J - Jo) <= 1e-6 * np.maximum(np.abs(Jo).max(axis=1, keepdims=True), 1e-9))


def test_assoc_point2plane_golden(ctx):
    import panovlm_amd as pv
    g = load("assoc_point2plane.npz")
    dev = {k: pv.Scan(ctx, _scan(g, k)) for k in (0, 1, 2)}
    from tests.test_assoc_gpu import same_planes
    for i, (r, n, tol, thr) in enumerate(g["cases"]):
        for exact in (True, False):        # PVLM_FLAG_ASSOC_EXACT_FIT: bit for bit; default (certified fast fit): same correspondences, planes to 1e-6
            rs = ctx.assoc_point2plane([dev[int(r)]], [dev[int(n)]], float(tol), float(thr), flags=0x101 | (0x200 if exact else 0))
            off, ref, nei, rows = rs.download()
            qidx, nn = rs.assoc_debug()
            assert np.array_equal(qidx, g["c%d_qidx" % i]) and np.array_equal(nn, g["c%d_nn" % i])
            assert np.array_equal(rows[:, :3], g["c%d_point" % i]) and same_planes(rows[:, 3:], g["c%d_plane" % i], exact)
            rs.close()


def test_equirect_golden(ctx):
    g = load("equirect.npz")
    for rows, cols in [(2880, 5760), (720, 1440)]:
        assert np.array_equal(ctx.cam_to_image(rows, cols, g["cam"].astype(np.float32)), g["px_f32_%d" % rows], equal_nan=True)
        assert np.array_equal(ctx.cam_to_image(rows, cols, g["cam"]), g["px_f64_%d" % rows], equal_nan=True)
        assert np.allclose(ctx.image_to_cam(rows, cols, g["pix_%d" % rows], 1.0), g["cam_f64_%d" % rows], atol=1e-15)


def test_fast_atan2_golden_through_projection(ctx):
    """lon = FastAtan2(x, z): recover it from the u pixel coordinate and compare with the vectors of the
    real reference header (u = cols * (0.5 + lon / 2pi) is monotone in lon)."""
    g = load("fast_atan2.npz")
    y, x = g["y"], g["x"]
    cam = np.stack([y, np.zeros_like(y), x], axis=1)   # CamToSphere: lon = FastAtan2(point.x, point.z)
    px = ctx.cam_to_image(2880, 5760, cam)
    lon = (px[:, 0] / 5760 - 0.5) * 2 * np.pi
    assert np.abs(lon - g["out_f64"]).max() < 1e-12


def test_lines_golden(ctx):
    import panovlm_amd as pv
    g = load("lines.npz")
    a, b = pv.Scan(ctx, line_scan(g, "a") | {"id": 3}), pv.Scan(ctx, line_scan(g, "b") | {"id": 4})
    for thr in (0.3, 0.4):
        assert np.array_equal(ctx.line2line_votes(a, b, thr), g["t%02d_votes" % int(thr * 10)])
    c = pv.Scan(ctx, line_scan(g, "c", local=True))
    assert np.array_equal(ctx.cam_lidar_votes(2880, 5760, g["c_lines"], c, g["c_T_cl"]), g["c_votes"])


def test_reproj_golden(ctx):
    import panovlm_amd as pv
    g = load("reproj.npz")
    ctx.set_poses(g["aa"], g["t"])
    bs = pv.BundleSet(ctx, g["off"], g["cam"], g["bearing"], g["X"], weight=float(g["weight"]))
    r, J = bs.evaluate()
    assert np.all(np.abs(r - g["r"]) <= 1e-6 * np.abs(g["r"]) + 8 * 2.2e-16 / np.maximum(g["r"], 1e-7))
    assert np.all(np.abs(J - g["J"]) <= 1e-6 * np.abs(g["J"]).max(axis=1, keepdims=True) + 1e-15 / np.maximum(g["r"], 1e-7)[:, None] ** 2)
    bs.close()


def test_depth_golden(ctx):
    g = load("depth.npz")
    for size in (3, 2):
        assert np.array_equal(ctx.project_lidar_depth(int(g["rows"]), int(g["cols"]), g["xyz"], g["T_cl"], size), g["depth_size%d" % size])


def test_undistort_golden(ctx):
    """K25 (pvlm_undistort_batch) against tests/golden/undistort.npz: floating point through double sines, tolerance 1e-6 relative."""
    from panovlm_amd import api
    g = load("undistort.npz")
    n = int(g["cases"])
    got = api.undistort_batch(ctx, [g["cloud%d" % k] for k in range(n)], [(g["R_wl%d" % k], g["t_wl%d" % k]) for k in range(n)],
                              [(g["R_we%d" % k], g["t_we%d" % k]) for k in range(n)])
    for k in range(n):
        want = g["out%d" % k]
        scale = np.maximum(np.abs(want[:, :3]).max(axis=1, keepdims=True), 1.0)
        assert np.all(np.abs(got[k][:, :3] - want[:, :3]) <= 1e-6 * scale) and np.array_equal(got[k][:, 3], want[:, 3]), k


def test_mvs_golden(ctx):
    g = load("mvs.npz")
    neis = [g["nei%d_gray" % k] for k in range(3)]; nd = [g["nei%d_depth" % k] for k in range(3)]
    for key, kw in (("conf_pho", {}), ("conf_geo", dict(nei_depths=nd))):
        c, d, _ = ctx.mvs_init_conf_map(g["gray"], neis, g["R_nr"], g["t_nr"], g["depth"], g["normal"], 3, 1, **kw)
        assert np.array_equal(c == -1, g[key] == -1)
        assert np.abs(c - g[key]).max() <= 1e-4
    # depth fusion of the golden swept map: bit-exact
    nc = [g["nei%d_conf" % k] for k in range(3)]
    dr, cr, ca = ctx.mvs_filter_depth_refine(nd, nc, g["R_nr"], g["t_nr"], g["depth_sweep"], np.clip(g["conf_sweep"], 0, None), thr=0.02, min_depth=0.1, max_depth=20.0)
    assert np.array_equal(dr, g["depth_refine"]) and np.array_equal(cr, g["conf_refine"]) and np.array_equal(ca, g["conf_after_refine"])
    # the PatchMatch iteration: the great majority of pixels land on the golden hypothesis (see test_mvs_gpu.py for why not all)
    ds, ns, cs = ctx.mvs_propagate(g["gray"], neis, g["R_nr"], g["t_nr"], g["depth_pho"], g["normal"], g["conf_pho"], max_iter=1, seed=int(g["sweep_seed"]))
    valid = g["conf_pho"] > -1
    same = (np.abs(ds - g["depth_sweep"]) <= 1e-4 * np.maximum(g["depth_sweep"], 1e-3)) & (np.abs(cs - g["conf_sweep"]) <= 1e-4)
    assert same[valid].mean() > 0.9 and abs(float(cs[valid].mean()) - float(g["conf_sweep"][valid].mean())) < 2e-3
