"""GPU parity of the equirectangular model (K7) and the camera<->LiDAR voting loop (K8)."""
import numpy as np
import pytest

from panovlm_amd import synthetic as sy
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import panovlm_amd as pv
    c = pv.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("rows,cols", [(2880, 5760), (720, 1440)])
def test_cam_to_image_bit_exact(ctx, oracle, rows, cols):
    rng = np.random.default_rng(5)
    cam = rng.normal(size=(20000, 3)) * np.array([3.0, 1.0, 3.0])
    cam[:50, 0] = 0.0; cam[50:100, 2] = 0.0; cam[100:150, 1] = 0.0; cam[150:160] = 0.0   # axis / degenerate cases
    cam[160:200] *= 1e-6
    f32 = cam.astype(np.float32)
    assert np.array_equal(ctx.cam_to_image(rows, cols, f32), oracle.cam_to_image(rows, cols, f32), equal_nan=True)
    assert np.array_equal(ctx.cam_to_image(rows, cols, cam), oracle.cam_to_image(rows, cols, cam), equal_nan=True)


@pytest.mark.parametrize("rows,cols", [(2880, 5760), (720, 1440)])
def test_image_to_cam(ctx, oracle, rows, cols):
    rng = np.random.default_rng(6)
    px = rng.uniform([0, 0], [cols, rows], size=(20000, 2))
    g = ctx.image_to_cam(rows, cols, px, 1.0); o = oracle.image_to_cam(rows, cols, px, 1.0)
    assert np.abs(g - o).max() <= 4e-16          # device vs glibc sin/cos differ by <= 1 ulp
    g32 = ctx.image_to_cam(rows, cols, px.astype(np.float32), 5.0); o32 = oracle.image_to_cam(rows, cols, px.astype(np.float32), 5.0)
    assert np.abs(g32 - o32).max() <= 1e-6 and np.mean(g32 == o32) > 0.999
    # round trip pixel -> ray -> pixel through the fast-atan2 model: ~0.3 deg accuracy (Math.h:9-10)
    back = ctx.cam_to_image(rows, cols, g)
    err = np.abs(back - px); err[:, 0] = np.minimum(err[:, 0], cols - err[:, 0])
    assert err.max() < cols * 0.35 / 360 * 2


def test_image_to_cam_f32_fast_trig_equals_library_trig_on_every_pixel_of_a_5p7k_panorama(ctx, oracle):
    """Equirectangular::ImageToCam<float> (sensors/Equirectangular.h:149-170) on all 5760 x 2880 integer pixels and on 4 M
    sub-pixel positions: the kernel's own sin/cos (one Cody-Waite step + fdlibm kernel polynomials in double, rounded to
    float) gives the same floats as the device library's double sin / cos rounded to float (PVLM_EXACT_TRIG=1, the round-1
    path), bit for bit, and both equal the oracle's (float)sin((double)x) on a sample."""
    import os, subprocess, sys, tempfile
    rows, cols = 2880, 5760
    yy, xx = np.mgrid[0:rows, 0:cols]
    grid = np.stack([xx.ravel(), yy.ravel()], axis=1).astype(np.float32)
    rng = np.random.default_rng(8)
    sub = rng.uniform([-50, -50], [cols + 50, rows + 50], size=(4_000_000, 2)).astype(np.float32)     # incl. positions outside the image
    px = np.concatenate([grid, sub])
    fast = ctx.image_to_cam(rows, cols, px, 1.0)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "px.npy"), px)
        code = ("import numpy as np, sys; sys.path.insert(0, %r)\nimport panovlm_amd as pv\nctx = pv.Context(0)\n"
                "px = np.load(sys.argv[1] + '/px.npy'); np.save(sys.argv[1] + '/exact.npy', ctx.image_to_cam(%d, %d, px, 1.0))\n"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), rows, cols))
        env = dict(os.environ); env["PVLM_EXACT_TRIG"] = "1"
        subprocess.run([sys.executable, "-c", code, d], check=True, env=env, timeout=900)
        exact = np.load(os.path.join(d, "exact.npy"))
    assert fast.dtype == np.float32 and fast.shape == exact.shape == (len(px), 3)
    mism = np.flatnonzero((fast != exact).any(axis=1))
    assert len(mism) == 0, (len(mism), px[mism[:5]], fast[mism[:5]], exact[mism[:5]])
    idx = rng.choice(len(px), 200_000, replace=False)
    o = oracle.image_to_cam(rows, cols, px[idx], 1.0)
    assert np.mean((fast[idx] == o).all(axis=1)) > 0.9999 and np.abs(fast[idx] - o).max() <= 1.2e-7


def test_cam_lidar_votes(ctx, oracle):
    import panovlm_amd as pv
    rng = np.random.default_rng(8)
    rows, cols = 2880, 5760
    lines_w = synth.random_world_lines(rng, 10, extent=3.0)
    R, t = np.eye(3), np.zeros(3)
    scan = synth.make_line_scan(rng, 0, R, t, lines_w, pts_per_line=(20, 60), extra_pts=50)
    local = dict(scan); local["corner_xyz"] = scan["corner_local"]
    # camera ~ LiDAR with a small calibration offset
    a = np.deg2rad(rng.uniform(-2, 2, size=3))
    T = np.eye(4); T[:3, :3] = synth.rodrigues(a); T[:3, 3] = rng.uniform(-0.05, 0.05, size=3)
    ends_cam = (scan["end_points"].reshape(-1, 3) @ T[:3, :3].T + T[:3, 3])
    px = oracle.cam_to_image(rows, cols, ends_cam).reshape(-1, 4).astype(np.float32)
    px += rng.normal(size=px.shape).astype(np.float32) * 2.0
    extra = rng.uniform([0, 0, 0, 0], [cols, rows, cols, rows], size=(6, 4)).astype(np.float32)
    lines = np.concatenate([px, extra])
    dscan = pv.Scan(ctx, local)
    v = ctx.cam_lidar_votes(rows, cols, lines, dscan, T)
    o = oracle.assoc_by_angle(rows, cols, lines, local, T, multiple=True)
    assert v.shape == o["votes"].shape
    assert np.array_equal(v, o["votes"])
    assert v.sum() > 50 and len(o["image_line_id"]) > 0
    # batched form: different line sets / calibrations per pair in one launch == pair by pair
    T2 = T.copy(); T2[:3, 3] += 0.02
    jobs = [(lines, T), (lines[:5], T2), (np.zeros((0, 4), np.float32), T), (lines[3:], T2)]
    got = ctx.cam_lidar_votes_batch(rows, cols, [j[0] for j in jobs], [dscan] * len(jobs), [j[1] for j in jobs])
    for (l, Tj), g in zip(jobs, got):
        assert g.shape == (len(l), dscan.n_segments)
        if len(l):
            assert np.array_equal(g, ctx.cam_lidar_votes(rows, cols, l, dscan, Tj))
    # the thread-per-point kernel of the batch (round 6: angles compared on squared cosines, the reference's acos chain only inside a 10^-11 band) against the
    # thread-per-test kernel (PVLM_CAM_VOTES=tests: roots, quotient and acos for every test) over many calibrations: the same blocks
    import os
    many_jobs = []
    for k in range(48):
        Tk = np.eye(4); Tk[:3, :3] = synth.rodrigues(np.deg2rad(rng.uniform(-4, 4, size=3))); Tk[:3, 3] = rng.uniform(-0.2, 0.2, size=3)
        many_jobs.append((lines[rng.permutation(len(lines))[:rng.integers(1, len(lines) + 1)]], Tk))
    new_kernel = ctx.cam_lidar_votes_batch(rows, cols, [j[0] for j in many_jobs], [dscan] * len(many_jobs), [j[1] for j in many_jobs])
    os.environ["PVLM_CAM_VOTES"] = "tests"
    try:
        old_kernel = ctx.cam_lidar_votes_batch(rows, cols, [j[0] for j in many_jobs], [dscan] * len(many_jobs), [j[1] for j in many_jobs])
    finally:
        del os.environ["PVLM_CAM_VOTES"]
    assert sum(int(g.sum()) for g in new_kernel) > 500
    for a_, b_ in zip(new_kernel, old_kernel):
        assert np.array_equal(a_, b_)
    # sparse read-back of the same launch: the non-zero counters, in dense order
    voff, nzi, nzc = ctx.cam_lidar_votes_batch_sparse(rows, cols, [j[0] for j in jobs], [dscan] * len(jobs), [j[1] for j in jobs])
    dense = np.concatenate([g.reshape(-1) for g in got])
    assert voff[-1] == len(dense) and np.array_equal(np.diff(voff), [g.size for g in got])
    assert np.array_equal(nzi, np.flatnonzero(dense)) and np.array_equal(nzc, dense[dense != 0])
    many = ctx.cam_lidar_votes_batch_sparse(rows, cols, [lines] * 40, [dscan] * 40, [T] * 40)       # more than one tile of counters
    one = got[0].reshape(-1)
    assert np.array_equal(many[1], np.concatenate([np.flatnonzero(one) + k * one.size for k in range(40)])) and np.array_equal(many[2], np.tile(one[one != 0], 40))
    dscan.close()


@pytest.mark.parametrize("size", [3, 2, 0])
def test_project_lidar_depth_matches_oracle(ctx, oracle, size):
    """ProjectLidar2PanoramaDepth (util/Visualization.h:407-441): bit-exact uint16 image, including the
    last-point-wins rule where windows overlap and the skipped windows at the image border."""
    from panovlm_amd import synthetic as sy
    rng = np.random.default_rng(12 + size)
    rows, cols = 720, 1440
    s = sy.make_scan(3, cols=512)
    xyz = np.concatenate([s["local_xyz"], s["local_xyz"][::7] * np.float32(1.01),     # overlapping windows, different depths
                          rng.normal(size=(500, 3)).astype(np.float32) * np.float32([0.01, 3.0, 0.01]),   # poles: windows leave the image
                          np.float32([[0.0, 0.0, -2.0], [1e-4, 0.3, -2.0], [-1e-4, -0.3, -2.0]])])        # the +-pi seam
    a = np.deg2rad([1.0, -2.0, 0.5]); T = np.eye(4); T[:3, :3] = synth.rodrigues(a); T[:3, 3] = [0.03, -0.02, 0.05]
    got = ctx.project_lidar_depth(rows, cols, xyz, T, size)
    ref = oracle.project_lidar_depth(rows, cols, xyz, T, size)
    assert got.dtype == np.uint16 and got.shape == (rows, cols)
    assert np.array_equal(got, ref)
    assert 0.02 < (ref > 0).mean() < 0.9
    assert np.array_equal(ctx.project_lidar_depth(rows, cols, np.zeros((0, 3), np.float32), T, size), np.zeros((rows, cols), np.uint16))
