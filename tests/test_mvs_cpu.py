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


def mvs_scene(oracle, rows=96, cols=192, n_views=4):
    poses = [(synth.rodrigues(np.array([0.02 * k, 0.25 * k - 0.3, 0.01])), np.array([0.35 * k - 0.5, 0.04 * k, 0.25 * k - 0.3])) for k in range(n_views)]
    views = [synth.render_panorama(oracle, rows, cols, R, t) for R, t in poses]
    ref = 1
    nei = [k for k in range(n_views) if k != ref]
    Rn, tn = zip(*[synth.relative_pose(poses[ref][0], poses[ref][1], poses[k][0], poses[k][1]) for k in nei])
    return views[ref], [views[k][0] for k in nei], np.array(Rn), np.array(tn)


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


@pytest.mark.parametrize("hw,step", [(3, 1), (5, 2)])
def test_device_bodies_match_oracle(oracle, hw, step):
    out = os.path.join(ROOT, "build", "libmvs_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "mvs_math_check.cpp")])
    lib = C.CDLL(out)
    (gray, depth, normal), neis, Rn, tn = mvs_scene(oracle)
    rng = np.random.default_rng(3)
    depth = depth * rng.uniform(0.9, 1.1, size=depth.shape).astype(np.float32)     # hypotheses, not the truth
    depth[5:9, 7:30] = 0
    co, do, no = oracle.mvs_init_conf_map(gray, neis, Rn, tn, depth, normal, hw, step)
    d = depth.copy(); nrm = normal.copy(); c = np.zeros_like(depth)
    ptrs = (C.POINTER(C.c_ubyte) * len(neis))(*[np.ascontiguousarray(g).ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
    R = np.ascontiguousarray(Rn, np.float32); t = np.ascontiguousarray(tn, np.float32)
    lib.chk_mvs_conf(C.c_int(gray.shape[0]), C.c_int(gray.shape[1]), C.c_int(hw), C.c_int(step), gray.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(neis)),
                     ptrs, R.ctypes.data_as(C.POINTER(C.c_float)), t.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.POINTER(C.c_float)),
                     nrm.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(co == -1, c == -1)                      # every validity decision
    assert np.array_equal(c, co)                                  # same float arithmetic, same order: bit for bit
    assert np.array_equal(d, do) and np.array_equal(nrm, no)
    assert (co > -1).mean() > 0.7
