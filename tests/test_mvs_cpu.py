"""CPU checks of the panoramic MVS scoring pass (mvs/MVS.cpp:586-680, :774-923): the oracle's behaviour on a rendered
scene, and the per-texel device bodies (panovlm_amd/csrc/pvlm_mvs_core.h, compiled for the host by
tests/cpp/mvs_math_check.cpp) against the oracle.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mvs_scene(oracle, rows=96, cols=192, n_views=4, with_depths=False):
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.25 * k - 0.3, 0.01])), np.array([0.35 * k - 0.5, 0.04 * k, 0.25 * k - 0.3])) for k in range(n_views)]
    views = [synth.render_panorama(oracle, rows, cols, R, t) for R, t in poses]
    ref = 1
    nei = [k for k in range(n_views) if k != ref]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    out = (views[ref], [views[k][0] for k in nei], np.array(Rn), np.array(tn))
    if with_depths:      # the neighbours' own depth maps (Frame::depth_filter), with holes and a band of wrong depths
        nd = []
        for k in nei:
            d = views[k][1].copy(); d[20:30, 40:90] = 0; d[50:60, :] *= 1.2
            nd.append(d)
        out = out + (nd,)
    return out


def test_oracle_scores_prefer_the_true_geometry(oracle):
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle)
    conf, d_out, n_out = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1)
    valid = conf > -1
    assert valid.mean() > 0.85 and conf[valid].mean() > 0.98
    # invalidated pixels lose their hypothesis (InitConfMap :610-614); valid ones keep it
    assert np.all(d_out[~valid] == 0) and np.all(n_out[~valid] == 0) and np.array_equal(d_out[valid], depth[valid])
    worse, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth * 1.2, normal, 3, 1)
    both = valid & (worse > -1)
    assert worse[both].mean() < conf[both].mean() - 0.01
    # a plane facing away from the camera (d > 0) and a pixel without depth
    flipped, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, -normal, 3, 1)
    assert np.all(flipped == -1)
    dz = depth.copy(); dz[10:20, 30:50] = 0
    keep = np.full(depth.shape, 7.0, np.float32)
    c2, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, dz, normal, 3, 1, conf=keep)
    assert np.all(c2[10:20, 30:50] == 7.0) and np.array_equal(c2[40:50], conf[40:50])
    # patch statistics: weights sum to one, weighted texels to zero
    w, t0w, sq0 = oracle.mvs_fill_patch(gray, 60, 40, 3, 1)
    assert abs(w.sum() - 1) < 1e-5 and abs(t0w.sum()) < 1e-3 and sq0 > 0
    assert oracle.mvs_fill_patch(gray, 1, 40, 3, 1)[2] == -1.0          # window leaves the image


def test_geometric_consistency_term(oracle):
    """InitConfMap(use_geometry=true): a consistent neighbour depth leaves the score almost untouched, an inconsistent or
    missing one costs 0.2 x min(angle, 2 deg) (ScorePixel :857-893)."""
    (gray, depth, normal), neis, Rn, tn, nd = mvs_scene(oracle, with_depths=True)
    pho, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1)
    geo, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1, nei_depths=nd)
    valid = (pho > -1) & (geo > -1)
    drop = pho[valid] - geo[valid]
    assert np.all(drop >= -1e-6) and drop.max() <= 0.4 + 1e-5          # penalty in [0, 0.2 * 2]
    assert np.median(drop) < 0.02 and (drop > 0.3).mean() > 0.01      # mostly consistent, the tampered band / holes are punished
    zero = [np.zeros_like(x) for x in nd]                              # no neighbour depth at all: the full penalty everywhere
    worst, _, _ = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, 3, 1, nei_depths=zero)
    both = valid & (worst > -1)
    assert np.allclose(pho[both] - worst[both], 0.4, atol=1e-5) or np.all(worst[both] >= -1)


@pytest.mark.parametrize("hw,step,geometric", [(3, 1, False), (5, 2, False), (3, 1, True)])
def test_device_bodies_match_oracle(oracle, hw, step, geometric):
    out = os.path.join(ROOT, "build", "libmvs_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "mvs_math_check.cpp")])
    lib = C.CDLL(out)
    (gray, depth, normal), neis, Rn, tn, nd = mvs_scene(oracle, with_depths=True)
    rng = np.random.default_rng(3)
    depth = depth * rng.uniform(0.97, 1.03, size=depth.shape).astype(np.float32)   # hypotheses, not the truth
    depth[5:9, 7:30] = 0
    co, do, no = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, hw, step, nei_depths=nd if geometric else None)
    d = depth.copy(); nrm = normal.copy(); c = np.zeros_like(depth)
    ptrs = (C.POINTER(C.c_ubyte) * len(neis))(*[np.ascontiguousarray(g).ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
    R = np.ascontiguousarray(Rn, np.float32); t = np.ascontiguousarray(tn, np.float32)
    nd = [np.ascontiguousarray(x, np.float32) for x in nd]
    dptrs = (C.POINTER(C.c_float) * len(nd))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd]) if geometric else None
    lib.chk_mvs_conf(C.c_int(gray.shape[0]), C.c_int(gray.shape[1]), C.c_int(hw), C.c_int(step), gray.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(neis)),
                     ptrs, R.ctypes.data_as(C.POINTER(C.c_float)), t.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.POINTER(C.c_float)),
                     nrm.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float)), dptrs)
    assert np.array_equal(co == -1, c == -1)                      # every validity decision
    if geometric:      # acosf / the double-rounded angle may differ in the last bit between the two restatements
        assert np.abs(c - co).max() <= 1e-6
    else:
        assert np.array_equal(c, co)                              # same float arithmetic, same order: bit for bit
    assert np.array_equal(d, do) and np.array_equal(nrm, no)
    assert (co > -1).mean() > 0.7


def _filter_scene(oracle, rows=96, cols=192):
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.25 * k - 0.3, 0.01])), np.array([0.35 * k - 0.5, 0.04 * k, 0.25 * k - 0.3])) for k in range(4)]
    views = [synth.render_panorama(oracle, rows, cols, R, t) for R, t in poses]
    ref, nei = 1, [0, 2, 3]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    depth = views[ref][1].copy()
    depth[30:40, 50:80] *= 1.1          # a wrong patch: must be filtered out
    depth[60:64, 100:140] = 0           # holes stay holes
    nd = [views[k][1].copy() for k in nei]
    nd[0][10:20, :] = 0                 # zero-depth neighbour pixels are splat too (at the epipole), as upstream
    conf = np.random.default_rng(5).uniform(0.2, 1.0, size=depth.shape).astype(np.float32)
    const = np.zeros(depth.shape, np.uint8); const[32:35, 55:60] = 1     # depth_constant pixels survive the second vote
    return nd, np.array(Rn), np.array(tn), depth, conf, const


def test_depth_filter_oracle_behaviour_and_device_bodies(oracle):
    nd, Rn, tn, depth, conf, const = _filter_scene(oracle)
    df, cf = oracle.mvs_filter_depth(nd, Rn, tn, depth, conf=conf, depth_constant=const, thr=0.01)
    kept = df > 0
    assert 0.15 < kept.mean() < 0.9
    assert not kept[30:40, 50:80][const[30:40, 50:80] == 0].any() and not kept[60:64, 100:140].any()
    assert np.array_equal(df[kept], depth[kept]) and np.array_equal(cf[kept], conf[kept]) and np.all(cf[~kept] == 0)
    looser, _ = oracle.mvs_filter_depth(nd, Rn, tn, depth, thr=0.03)
    assert (looser > 0).sum() > kept.sum()
    # the device bodies, driven serially on the host: identical (same float arithmetic; the splat's minimum is order-free)
    out = os.path.join(ROOT, "build", "libmvs_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "mvs_math_check.cpp")])
    lib = C.CDLL(out)
    nd = [np.ascontiguousarray(x, np.float32) for x in nd]
    dptrs = (C.POINTER(C.c_float) * len(nd))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
    R = np.ascontiguousarray(Rn, np.float32); t = np.ascontiguousarray(tn, np.float32)
    od = np.zeros_like(depth); oc = np.zeros_like(depth)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    lib.chk_mvs_filter(C.c_int(depth.shape[0]), C.c_int(depth.shape[1]), C.c_int(len(nd)), dptrs, fp(R), fp(t), fp(depth), fp(conf),
                       const.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_float(0.01), fp(od), fp(oc))
    assert np.array_equal(od, df) and np.array_equal(oc, cf)


def _refine_scene(oracle, rows=96, cols=192):
    nd, Rn, tn, depth, conf, const = _filter_scene(oracle, rows, cols)
    rng = np.random.default_rng(11)
    nc = [rng.uniform(-0.3, 1.0, size=depth.shape).astype(np.float32).clip(0, None) for _ in nd]   # ConvertNCC2Conf: negatives are already 0
    depth = depth * rng.uniform(0.995, 1.005, size=depth.shape).astype(np.float32)                  # estimates scatter around the surface
    depth[20:26, 100:130] *= 0.7        # in front of what the neighbours see: free-space violation
    depth[70:76, 20:60] *= 1.4          # behind it: occlusion
    return nd, nc, Rn, tn, depth, conf, const


def test_depth_conf_projection_keeps_last_closest_writer(oracle):
    """ProjectDepthConfToRef with both outputs: range = the minimum, confidence = that of the last raster-order source
    among those at the minimum range (checked with a numpy restatement of the sequential rule)."""
    nd, nc, Rn, tn, *_ = _refine_scene(oracle, 48, 96)
    pd, pc = oracle.mvs_project_depth_conf(nd[1], nc[1], Rn[1], tn[1])
    only_d, _ = oracle.mvs_filter_depth([], np.zeros((0, 9)), np.zeros((0, 3)), nd[1])   # shapes only
    assert pd.shape == only_d.shape
    rows, cols = pd.shape
    jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
    unit = oracle.image_to_cam(rows, cols, np.stack([jj.ravel(), ii.ravel()], 1).astype(np.float32), 1.0)
    R_rn = Rn[1].reshape(3, 3).T.astype(np.float32); t_rn = (-R_rn) @ tn[1].astype(np.float32)
    pr = (unit * nd[1].reshape(-1, 1)) @ R_rn.T + t_rn
    rng_ = np.linalg.norm(pr.astype(np.float64), axis=1).astype(np.float32)
    px = oracle.cam_to_image(rows, cols, pr.astype(np.float32))
    want_d = np.zeros(rows * cols, np.float32); want_c = np.zeros(rows * cols, np.float32)
    for e in range(rows * cols):
        for y in (int(np.ceil(px[e, 1])), int(np.floor(px[e, 1]))):
            for x in (int(np.ceil(px[e, 0])), int(np.floor(px[e, 0]))):
                if 0 <= x < cols and 0 <= y < rows:
                    k = y * cols + x
                    if want_d[k] != 0 and want_d[k] < rng_[e]:
                        continue
                    want_d[k] = rng_[e]; want_c[k] = nc[1].ravel()[e]
    hit = want_d > 0
    assert hit.mean() > 0.5
    # the independent numpy transform may round a range differently in the last bit; the structure must agree
    assert np.allclose(pd.ravel(), want_d, rtol=2e-6, atol=0) and (pc.ravel() == want_c).mean() > 0.995


def test_depth_filter_refine_oracle_behaviour_and_device_bodies(oracle):
    nd, nc, Rn, tn, depth, conf, const = _refine_scene(oracle)
    df, cf, conf_after = oracle.mvs_filter_depth_refine(nd, nc, Rn, tn, depth, conf, depth_constant=const, thr=0.01, min_depth=0.1, max_depth=20.0)
    kept = df > 0
    assert 0.15 < kept.mean() < 0.95
    assert np.all(conf_after[depth <= 0] == 0) and np.array_equal(conf_after[depth > 0], conf[depth > 0])
    assert not kept[60:64, 100:140].any()                                   # holes stay holes
    fused = kept & (const == 0)
    assert np.abs(df[fused] / depth[fused] - 1).max() < 0.012               # an average of depths that agree within 1.2 %
    assert (cf[fused] > 0).all()                                            # positive - negative, accepted only when positive wins
    assert np.all(df[(const == 1) & ~fused & (depth > 0)] == depth[(const == 1) & ~fused & (depth > 0)])
    for band in (np.s_[20:26, 100:130], np.s_[70:76, 20:60]):               # the displaced bands disagree with every neighbour
        assert kept[band].mean() < 0.05
    # a tight depth range rejects the averaged depth (IsInside(avg, min_depth, max_depth))
    near, _, _ = oracle.mvs_filter_depth_refine(nd, nc, Rn, tn, depth, conf, thr=0.01, min_depth=0.1, max_depth=float(np.median(depth[depth > 0])))
    assert 0 < (near > 0).sum() < (df[const == 0] > 0).sum()
    # device bodies on the host (reverse-order splat): bit for bit
    out = os.path.join(ROOT, "build", "libmvs_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "mvs_math_check.cpp")])
    lib = C.CDLL(out)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    nd = [np.ascontiguousarray(x, np.float32) for x in nd]; nc = [np.ascontiguousarray(x, np.float32) for x in nc]
    dptrs = (C.POINTER(C.c_float) * len(nd))(*[fp(x) for x in nd]); cptrs = (C.POINTER(C.c_float) * len(nc))(*[fp(x) for x in nc])
    R = np.ascontiguousarray(Rn, np.float32); t = np.ascontiguousarray(tn, np.float32)
    od = np.zeros_like(depth); oc = np.zeros_like(depth); cc = conf.copy()
    lib.chk_mvs_filter_refine(C.c_int(depth.shape[0]), C.c_int(depth.shape[1]), C.c_int(len(nd)), dptrs, cptrs, fp(R), fp(t), fp(depth), fp(cc),
                              const.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_float(0.01), C.c_float(0.1), C.c_float(20.0), fp(od), fp(oc))
    assert np.array_equal(od, df) and np.array_equal(oc, cf) and np.array_equal(cc, conf_after)


def sweep_scene(oracle, rows=96, cols=192):
    """A perturbed initial state for the PatchMatch sweep: depths off by up to 10 %, normals tilted, scored by InitConfMap."""
    (gray, depth, normal), neis, Rn, tn, nd = mvs_scene(oracle, rows, cols, with_depths=True)
    rng = np.random.default_rng(7)
    d0 = (depth * rng.uniform(0.9, 1.1, size=depth.shape)).astype(np.float32)
    n0 = normal + 0.15 * rng.normal(size=normal.shape).astype(np.float32)
    n0 = (n0 / np.linalg.norm(n0, axis=2, keepdims=True)).astype(np.float32)
    c0, d1, n1 = oracle.mvs_init_conf_map(gray, neis, Rn, tn, d0, n0, 3, 1)
    const = np.zeros(depth.shape, np.uint8); const[rows // 2 - 5:rows // 2 + 5, cols // 3:cols // 2] = 1
    return dict(gray=gray, neis=neis, Rn=Rn, tn=tn, nd=nd, depth=d1, normal=n1, conf=c0, truth=depth, const=const)


def test_patchmatch_sweep_oracle_behaviour_and_device_bodies(oracle):
    """EstimateDepthMapSingle (checkerboard): the oracle converges towards the rendered geometry, never lowers a pixel's
    score, honours depth_constant and the confidence threshold, and repeats for a seed; process_pixel and everything under
    it (panovlm_amd/csrc/pvlm_mvs_core.h) compiled for the host gives the same maps bit for bit, pixels visited backwards."""
    S = sweep_scene(oracle)
    args = (S["gray"], S["neis"], S["Rn"], S["tn"], S["depth"], S["normal"], S["conf"])
    valid = S["conf"] > -1
    err = lambda d: np.median(np.abs(d[valid] / S["truth"][valid] - 1))
    d1, n1, c1 = oracle.mvs_propagate(*args, max_iter=1, seed=5)
    d3, n3, c3 = oracle.mvs_propagate(*args, max_iter=3, seed=5)
    assert np.all(c1[valid] >= S["conf"][valid]) and np.all(c3[valid] >= c1[valid] - 0)          # a hypothesis is only ever replaced by a better one
    assert err(S["depth"]) > 0.04 and err(d1) < 0.5 * err(S["depth"]) and err(d3) < 0.6 * err(d1)
    assert np.array_equal(d1[~valid], S["depth"][~valid])                                          # pixels without a hypothesis are not touched
    again = oracle.mvs_propagate(*args, max_iter=1, seed=5)
    assert all(np.array_equal(a, b) for a, b in zip((d1, n1, c1), again))
    other = oracle.mvs_propagate(*args, max_iter=1, seed=6)[0]
    assert (other != d1)[valid].mean() > 0.5
    dk, nk, ck = oracle.mvs_propagate(*args, max_iter=1, seed=5, depth_constant=S["const"], conf_threshold=0.97)
    m = (S["const"] == 1) & valid
    assert np.array_equal(dk[m], S["depth"][m]) and (nk[m] != S["normal"][m]).any()               # depth pinned, normal still refined
    dropped = (ck == -1) & valid
    assert dropped.any() and np.all(dk[dropped] == 0) and not dropped[m].any() and np.all(ck[valid & ~dropped & ~m] >= 0.97)
    # random stream: uniform and independent enough for the purpose
    oracle.lib().orc_mvs_random_u32.restype = C.c_uint
    u = np.array([oracle.lib().orc_mvs_random_u32(C.c_ulonglong(3), C.c_ulonglong(p), C.c_uint(k)) for p in range(200) for k in range(20)], np.float64) / 2 ** 32
    assert abs(u.mean() - 0.5) < 0.02 and abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.05
    out = os.path.join(ROOT, "build", "libmvs_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "mvs_math_check.cpp")])
    lib = C.CDLL(out)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ptrs = (C.POINTER(C.c_ubyte) * len(S["neis"]))(*[np.ascontiguousarray(g).ctypes.data_as(C.POINTER(C.c_ubyte)) for g in S["neis"]])
    R = np.ascontiguousarray(S["Rn"], np.float32); t = np.ascontiguousarray(S["tn"], np.float32)
    nd = [np.ascontiguousarray(x, np.float32) for x in S["nd"]]
    lib.chk_mvs_random_u32.restype = C.c_uint
    assert lib.chk_mvs_random_u32(C.c_ulonglong(3), C.c_ulonglong(17), C.c_uint(4)) == oracle.lib().orc_mvs_random_u32(C.c_ulonglong(3), C.c_ulonglong(17), C.c_uint(4))
    for geo, kw in ((False, dict(max_iter=1, seed=5)), (True, dict(max_iter=2, seed=9, conf_threshold=0.9, depth_constant=S["const"]))):
        want = oracle.mvs_propagate(*args, nei_depths=S["nd"] if geo else None, **kw)
        d = S["depth"].copy(); n = S["normal"].copy(); c = S["conf"].copy()
        dptrs = (C.POINTER(C.c_float) * len(nd))(*[fp(x) for x in nd]) if geo else None
        dc = kw.get("depth_constant")
        lib.chk_mvs_propagate(C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(3), C.c_int(1), S["gray"].ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(nd)), ptrs,
                              fp(R), fp(t), fp(d), fp(n), fp(c), dptrs, dc.ctypes.data_as(C.POINTER(C.c_ubyte)) if dc is not None else None, C.c_float(0.1),
                              C.c_float(20.0), C.c_ulonglong(kw["seed"]), C.c_int(kw["max_iter"]), C.c_float(kw.get("conf_threshold", -1.0)))
        assert np.array_equal(d, want[0]) and np.array_equal(n, want[1]) and np.array_equal(c, want[2])
        # the thread-per-pixel bodies (fill_patch_column + ColumnScorer: the program of k_mvs_propagate_lane) — the same maps
        d = S["depth"].copy(); n = S["normal"].copy(); c = S["conf"].copy()
        lib.chk_mvs_propagate_column(C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(3), C.c_int(1), S["gray"].ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(nd)), ptrs,
                                     fp(R), fp(t), fp(d), fp(n), fp(c), dptrs, dc.ctypes.data_as(C.POINTER(C.c_ubyte)) if dc is not None else None, C.c_float(0.1),
                                     C.c_float(20.0), C.c_ulonglong(kw["seed"]), C.c_int(kw["max_iter"]), C.c_float(kw.get("conf_threshold", -1.0)))
        assert np.array_equal(d, want[0]) and np.array_equal(n, want[1]) and np.array_equal(c, want[2])
        # the SEQUENTIAL sweep (the Room / Floor strategy): the oracle walks in raster order, the device bodies go anti-diagonal
        # by anti-diagonal, bottom row first inside a diagonal — the kernel's launch structure.  Three iterations: forward, back, forward.
        kw3 = dict(kw); kw3["max_iter"] = 3
        want = oracle.mvs_propagate(*args, nei_depths=S["nd"] if geo else None, sequential=True, **kw3)
        d = S["depth"].copy(); n = S["normal"].copy(); c = S["conf"].copy()
        lib.chk_mvs_propagate_sequential(C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(3), C.c_int(1), S["gray"].ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(nd)),
                                         ptrs, fp(R), fp(t), fp(d), fp(n), fp(c), dptrs, dc.ctypes.data_as(C.POINTER(C.c_ubyte)) if dc is not None else None,
                                         C.c_float(0.1), C.c_float(20.0), C.c_ulonglong(kw["seed"]), C.c_int(3), C.c_float(kw.get("conf_threshold", -1.0)))
        assert np.array_equal(d, want[0]) and np.array_equal(n, want[1]) and np.array_equal(c, want[2])
        # ... and through process_pixel_spec, the form k_mvs_propagate_diag_spec runs: the independent hypotheses of a pixel (the two
        # propagated ones; batches of W consecutive refinements built as if none were accepted) scored side by side and resolved in
        # order — the same maps for every batch width
        for width in (1, 2, 3, 4):
            d = S["depth"].copy(); n = S["normal"].copy(); c = S["conf"].copy()
            lib.chk_mvs_propagate_sequential_spec(C.c_int(d.shape[0]), C.c_int(d.shape[1]), C.c_int(3), C.c_int(1), S["gray"].ctypes.data_as(C.POINTER(C.c_ubyte)),
                                                  C.c_int(len(nd)), ptrs, fp(R), fp(t), fp(d), fp(n), fp(c), dptrs,
                                                  dc.ctypes.data_as(C.POINTER(C.c_ubyte)) if dc is not None else None, C.c_float(0.1), C.c_float(20.0),
                                                  C.c_ulonglong(kw["seed"]), C.c_int(3), C.c_float(kw.get("conf_threshold", -1.0)), C.c_int(width))
            assert np.array_equal(d, want[0]) and np.array_equal(n, want[1]) and np.array_equal(c, want[2]), width
        chk = oracle.mvs_propagate(*args, nei_depths=S["nd"] if geo else None, **kw3)
        assert not np.array_equal(chk[0], want[0])                  # and it is a different sweep from the checkerboard


def _hsv_numpy(bgr):
    """BGR2HSV (util/Visualization.cpp:57-77) x (180, 255, 255), in float32 numpy."""
    f = np.float32
    c = bgr.astype(np.float32) / f(255)
    b, g, r = c[..., 0], c[..., 1], c[..., 2]
    cmax = np.maximum(r, np.maximum(g, b)); cmin = np.minimum(r, np.minimum(g, b)); delta = cmax - cmin
    with np.errstate(invalid="ignore", divide="ignore"):
        hr = f(60) * ((g - b) / delta + (f(6) * (g < b)).astype(np.float32))
        hg = f(60) * ((b - r) / delta + f(2))
        hb = f(60) * ((r - g) / delta + f(4))
        h = np.where(cmax == r, hr, np.where(cmax == g, hg, hb)) / f(360)
        s = delta / cmax
    h = np.where(cmax == 0, f(0), h); s = np.where(cmax == 0, f(0), s)
    return h * f(180), s * f(255), cmax * f(255)


@pytest.mark.parametrize("with_normal", [False, True])
def test_depth_to_cloud_oracle_against_numpy_and_device_bodies(oracle, with_normal):
    """MVS::DepthImageToCloud / DepthNormalToCloud: the oracle against a vectorised numpy restatement, and the per-pixel bodies the HIP
    kernels run (pvlm_mvs_core.h, compiled for the host) against the oracle, bit for bit."""
    rows, cols, max_depth = 60, 120, 20.0
    depth, bgr, normal, T = synth.cloud_scene(np.random.default_rng(12), rows, cols, max_depth)
    res = oracle.mvs_depth_to_cloud(depth, bgr, T, max_depth, filter_sky=not with_normal, normal=normal if with_normal else None)
    # numpy restatement
    h, s, v = _hsv_numpy(bgr)
    sky = (h >= 100) & (h <= 124) & (s >= 43) & (s <= 200) & (v >= 150) & (v <= 255)
    keep = (depth > 0) & (depth.astype(np.float64) < np.float64(np.float32(max_depth)) * 0.8)
    if not with_normal:
        keep &= ~sky
        assert 0.15 * sky.size < sky.sum() < 0.5 * sky.size
    px = np.stack(np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32)), -1).reshape(-1, 2)
    unit = oracle.image_to_cam(rows, cols, px, 1.0).reshape(rows, cols, 3)
    pc = (unit * depth[..., None]).astype(np.float32)[keep].astype(np.float64)
    want = (pc[:, 0:1] * T[:3, 0] + pc[:, 1:2] * T[:3, 1] + pc[:, 2:3] * T[:3, 2] + T[:3, 3]).astype(np.float32)
    assert len(res[0]) == keep.sum() and 0.2 * keep.size < keep.sum() < keep.size
    assert np.array_equal(res[0], want) and np.array_equal(res[1], bgr[keep][:, ::-1])
    if with_normal:
        nc = normal[keep].astype(np.float64)
        assert np.array_equal(res[2], (T[:3, 0] * nc[:, 0:1] + T[:3, 1] * nc[:, 1:2] + T[:3, 2] * nc[:, 2:3]).astype(np.float32))
    # device bodies on the host
    out = os.path.join(ROOT, "build", "libmvs_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "mvs_math_check.cpp")])
    lib = C.CDLL(out)
    lib.chk_mvs_depth_to_cloud.restype = C.c_longlong
    xyz = np.zeros((rows * cols, 3), np.float32); rgb = np.zeros((rows * cols, 3), np.uint8); nout = np.zeros((rows * cols, 3), np.float32)
    T12 = np.ascontiguousarray(T.reshape(-1)[:12])
    n = lib.chk_mvs_depth_to_cloud(C.c_int(rows), C.c_int(cols), depth.ctypes.data_as(C.POINTER(C.c_float)), bgr.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                   normal.ctypes.data_as(C.POINTER(C.c_float)), T12.ctypes.data_as(C.POINTER(C.c_double)), C.c_float(max_depth),
                                   C.c_int(0 if with_normal else 1), xyz.ctypes.data_as(C.POINTER(C.c_float)), rgb.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                   nout.ctypes.data_as(C.POINTER(C.c_float)) if with_normal else None)
    assert n == len(res[0]) and np.array_equal(xyz[:n], res[0]) and np.array_equal(rgb[:n], res[1])
    if with_normal:
        assert np.array_equal(nout[:n], res[2])


def test_patchmatch_helpers_against_numpy(oracle):
    """The small pieces of the sweep, one by one, against what they are meant to compute."""
    L = oracle.lib()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rng = np.random.default_rng(2)
    # InterpolatePixel = intersection of the pixel's view ray with the neighbour's tangent plane
    L.orc_mvs_interpolate_pixel.restype = C.c_float
    rows, cols = 90, 180
    unit = lambda x, y: oracle.image_to_cam(rows, cols, np.array([[x, y]], np.float32), 1.0)[0].astype(np.float64)
    for _ in range(50):
        px, py = int(rng.integers(1, cols - 1)), int(rng.integers(10, rows - 10))
        nx, ny = px + int(rng.integers(-1, 2)), py + int(rng.integers(-1, 2))
        n = -unit(px, py) + 0.3 * rng.normal(size=3); n /= np.linalg.norm(n)
        d = float(rng.uniform(1, 8))
        got = L.orc_mvs_interpolate_pixel(C.c_int(rows), C.c_int(cols), C.c_int(px), C.c_int(py), C.c_int(nx), C.c_int(ny), C.c_float(d),
                                          fp(n.astype(np.float32)), C.c_float(0.1), C.c_float(20.0))
        want = (unit(nx, ny) * d) @ n / (unit(px, py) @ n)
        assert abs(got - (want if 0.1 <= want <= 20 else d)) < 1e-4 * d
        assert abs((unit(px, py) * got - unit(nx, ny) * d) @ n) < 1e-4 * d or not (0.1 <= want <= 20)     # the new point lies on the plane
    # CorrectNormal: a normal facing away from the camera comes back to just past 90 degrees; Eigen's AngleAxis formula in float64
    for _ in range(50):
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        n = v * rng.uniform(0.05, 1.0) + rng.normal(size=3) * 0.5; n /= np.linalg.norm(n)
        if n @ v < 0:
            n = -n
        out = n.astype(np.float32).copy()
        L.orc_mvs_correct_normal(fp(v.astype(np.float32)), fp(out))
        a = np.cross(n, v); rad = min((np.arccos(n @ v) - np.pi / 2) * 1.01, -0.001)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.cos(rad) * np.eye(3) + (1 - np.cos(rad)) * np.outer(a, a) + np.sin(rad) * K                 # AngleAxis::toRotationMatrix, axis as given
        assert np.abs(out - R @ n).max() < 1e-5
        assert out @ v < n @ v                                                                             # turned towards the camera
    facing = -np.array([0.0, 0.6, 0.8], np.float32); keep = facing.copy()
    L.orc_mvs_correct_normal(fp(np.array([0.0, 0.6, 0.8], np.float32)), fp(keep))
    assert np.array_equal(keep, facing)                                                                    # already facing the camera: untouched
    # PerturbNormal: a rotation (norm kept) by at most ~ perturbation; three draws.  PerturbDepth: inside +- perturbation
    L.orc_mvs_perturb_depth.restype = C.c_float
    for k in range(50):
        n = rng.normal(size=3); n = (n / np.linalg.norm(n)).astype(np.float32)
        out = np.zeros(3, np.float32)
        used = L.orc_mvs_perturb_normal(C.c_ulonglong(9), C.c_ulonglong(k), fp(n), C.c_float(0.3), fp(out))
        assert used == 3 and abs(np.linalg.norm(out) - 1) < 1e-5
        assert np.arccos(np.clip(out @ n, -1, 1)) <= 0.3 * np.sqrt(3) / 2 + 1e-3
        d = L.orc_mvs_perturb_depth(C.c_ulonglong(9), C.c_ulonglong(k), C.c_float(4.0), C.c_float(0.02))
        assert 4.0 * 0.98 - 1e-5 <= d <= 4.0 * 1.02 + 1e-5
    # GenerateRandomNormal: unit, facing the camera, spread over the hemisphere; an even number of draws (rejection sampling)
    v = np.array([0.0, 0.0, 1.0], np.float32)
    ns = []
    for k in range(400):
        out = np.zeros(3, np.float32)
        used = L.orc_mvs_random_normal(C.c_ulonglong(4), C.c_ulonglong(k), fp(v), fp(out))
        assert used >= 2 and used % 2 == 0 and abs(np.linalg.norm(out) - 1) < 1e-5 and out @ v <= 0
        ns.append(out)
    ns = np.array(ns)
    assert abs(ns[:, 2].mean() + 0.5) < 0.06 and np.abs(ns[:, :2].mean(0)).max() < 0.1                      # uniform on the hemisphere: E[n.v] = -1/2



def test_cheap_sequences_equal_the_reference_statements_for_every_float(tmp_path):
    """panovlm_amd/csrc/pvlm_exact_math.h replaces x / pi, x / (2 pi) (double division of a promoted float) and
    (float)sqrt((double)v) by cheaper sequences in the kernels; tests/cpp/exact_math_check.cpp walks all 2^32 float bit patterns
    and counts the arguments for which the result differs from the reference's statement: none."""
    exe = str(tmp_path / "exact_math_check")
    subprocess.check_call(["g++", "-O2", "-march=native", "-fopenmp", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "cpp", "exact_math_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "division mismatches 0  sqrt mismatches 0" in out.stdout and "floats 4278190080" in out.stdout
    assert ", 0 visible" in out.stdout
