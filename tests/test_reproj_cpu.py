"""CPU checks of the reprojection ("bundle") path: the oracle's PanoramaReprojResidual_1Angle (Jet AutoDiff,
base/CostFunction.h:218-247) against the closed form and the Schur algebra of the device bodies
(panovlm_amd/csrc/pvlm_ba_core.h), which tests/cpp/reproj_math_check.cpp compiles for the host.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class View(C.Structure):
    _fields_ = [("n_points", C.c_int), ("n_cams", C.c_int), ("n_upairs", C.c_int), ("n_obs", C.c_longlong),
                ("pt_off", C.c_void_p), ("cam", C.c_void_p), ("obs_pt", C.c_void_p), ("s", C.c_void_p), ("X", C.c_void_p), ("Xc", C.c_void_p),
                ("scale", C.c_void_p), ("Vinv", C.c_void_p), ("gp", C.c_void_p), ("adj_off", C.c_void_p), ("adj_cam", C.c_void_p),
                ("adj_slot", C.c_void_p), ("frozen", C.c_void_p), ("w", C.c_double), ("loss", C.c_int), ("a", C.c_double)]


@pytest.fixture(scope="module")
def chk():
    out = os.path.join(ROOT, "build", "libreproj_check.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "cpp", "reproj_math_check.cpp")])
    lib = C.CDLL(out)
    lib.chk_cost.restype = C.c_double
    lib.chk_packed_size.restype = C.c_longlong
    return lib


class Problem:
    def __init__(self, b, w, loss, a):
        self.b = b
        self.F = int(b["cam"].max()) + 1
        self.M = len(b["off"]) - 1
        self.ui, self.uj = synth.covisible_pairs(b["off"], b["cam"])
        self.keep = dict(
            off=np.ascontiguousarray(b["off"], np.int64), cam=np.ascontiguousarray(b["cam"], np.int32),
            obs_pt=np.repeat(np.arange(self.M), np.diff(b["off"])).astype(np.int32),
            s=np.ascontiguousarray(b["bearing"] / np.linalg.norm(b["bearing"], axis=1, keepdims=True)),
            X=np.ascontiguousarray(b["X"], np.float64), Xc=np.zeros((self.M, 3)), scale=np.zeros((self.M, 3)), Vinv=np.zeros((self.M, 6)),
            gp=np.zeros((self.M, 3)))
        adj_off = np.zeros(self.F + 1, np.int32)
        for u in self.ui:
            adj_off[u + 1] += 1
        self.keep["adj_off"] = np.cumsum(adj_off).astype(np.int32)
        self.keep["adj_cam"] = np.ascontiguousarray(self.uj, np.int32)
        self.keep["adj_slot"] = np.arange(len(self.ui), dtype=np.int32)
        k = self.keep
        self.view = View(self.M, self.F, len(self.ui), len(b["cam"]), *[k[n].ctypes.data for n in
                         ("off", "cam", "obs_pt", "s", "X", "Xc", "scale", "Vinv", "gp", "adj_off", "adj_cam", "adj_slot")], None, w, loss, a)
        self.tab = np.ascontiguousarray(synth.pose_table(b["aa"], b["t"]))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("loss", [0, 1])
def test_device_bodies_match_oracle_schur(chk, oracle, loss):
    rng = np.random.default_rng(40 + loss)
    b = synth.random_bundle(rng, n_cams=6, n_points=40)
    w, a = 1.7, 4.0 * np.pi / 180.0 * 0.2
    P = Problem(b, w, loss, a)
    n = len(b["cam"])
    # closed-form r, J vs Jet AutoDiff
    r = np.zeros(n); J = np.zeros((n, 9))
    chk.chk_eval(C.byref(P.view), _dp(P.tab), _dp(r), _dp(J))
    ro, Jo = oracle.evaluate_reproj(b["bearing"], w, b["cam"], P.keep["obs_pt"], b["aa"], b["t"], b["X"])
    assert np.allclose(r, ro, rtol=1e-9, atol=1e-12)
    assert np.allclose(J, Jo, rtol=1e-8, atol=1e-9 * np.abs(Jo).max())
    assert loss == 0 or (ro > a).any()          # the Huber branch is exercised
    # Schur complement
    radius, mn, mx = 1e4, 1e-6, 1e32
    packed = np.zeros(chk.chk_packed_size(P.F, len(P.ui)))
    chk.chk_reduce(C.byref(P.view), _dp(P.tab), 1, C.c_double(radius), C.c_double(mn), C.c_double(mx), _dp(packed))
    ref = synth.bundle_reference(ro, Jo, b["off"], b["cam"], P.F, loss, a, None, radius, mn, mx)
    S, g, cost, Ud, gmax = synth.bundle_unpack(packed, P.F, P.ui, P.uj)
    sc = np.abs(ref["S"]).max()
    assert np.allclose(S, ref["S"], rtol=0, atol=1e-9 * sc)
    assert np.allclose(S, S.T, rtol=0, atol=1e-12 * sc)
    assert np.allclose(g, ref["g"], rtol=0, atol=1e-9 * np.abs(ref["g"]).max())
    assert np.isclose(cost, ref["cost"], rtol=1e-12)
    assert np.allclose(Ud, ref["Udiag"], rtol=1e-9)
    assert np.isclose(gmax, ref["gmax"], rtol=1e-9)
    assert np.allclose(P.keep["scale"], ref["scale"], rtol=1e-12)
    # second radius re-uses the stored scaling
    chk.chk_reduce(C.byref(P.view), _dp(P.tab), 0, C.c_double(10.0), C.c_double(mn), C.c_double(mx), _dp(packed))
    ref2 = synth.bundle_reference(ro, Jo, b["off"], b["cam"], P.F, loss, a, ref["scale"], 10.0, mn, mx)
    S2, g2, _, _, _ = synth.bundle_unpack(packed, P.F, P.ui, P.uj)
    assert np.allclose(S2, ref2["S"], rtol=0, atol=1e-9 * sc) and np.allclose(g2, ref2["g"], rtol=0, atol=1e-9 * np.abs(ref2["g"]).max())
    assert not np.allclose(S2, S, rtol=0, atol=1e-6 * sc)
    # back-substitution and model decrease for a camera step
    dcam = rng.normal(size=(P.F, 6)) * 1e-2
    out3 = np.zeros(3)
    chk.chk_step(C.byref(P.view), _dp(P.tab), _dp(dcam), _dp(out3))
    rho1, _ = synth.huber_weights(ro, loss, a)
    dX = np.zeros((P.M, 3)); model = 0.0
    for p in range(P.M):
        idx = np.arange(b["off"][p], b["off"][p + 1])
        rhs = ref2["gp"][p].copy()
        for i in idx:
            rhs += rho1[i] * Jo[i, 6:] * (Jo[i, :6] @ dcam[b["cam"][i]])
        dX[p] = -ref2["Vinv"][p] @ rhs
        for i in idx:
            d = Jo[i, :6] @ dcam[b["cam"][i]] + Jo[i, 6:] @ dX[p]
            model -= rho1[i] * (ro[i] * d + 0.5 * d * d)
    assert np.allclose(P.keep["Xc"], b["X"] + dX, rtol=0, atol=1e-10)
    assert np.isclose(out3[0], model, rtol=1e-8, atol=1e-14)
    assert np.isclose(out3[1], (dX ** 2).sum(), rtol=1e-8) and np.isclose(out3[2], (b["X"] ** 2).sum(), rtol=1e-12)
    assert np.allclose(P.keep["Xc"][0], b["X"][0])      # a point without observations does not move
    # cost at the candidate points
    c1 = chk.chk_cost(C.byref(P.view), _dp(P.tab), 1)
    r1, _ = oracle.evaluate_reproj(b["bearing"], w, b["cam"], P.keep["obs_pt"], b["aa"], b["t"], P.keep["Xc"], jac=False)
    assert np.isclose(c1, synth.huber_weights(r1, loss, a)[1].sum(), rtol=1e-12)


def test_schur_step_equals_full_damped_system(chk, oracle):
    """Solving the reduced camera system + back-substitution == solving the full (cameras + points) damped
    Gauss-Newton system: the elimination is exact, whatever Ceres' linear solver is called."""
    rng = np.random.default_rng(7)
    b = synth.random_bundle(rng, n_cams=4, n_points=25, empty_points=0)
    w, loss, a = 1.0, 1, 0.01
    P = Problem(b, w, loss, a)
    n = len(b["cam"]); F, M = P.F, P.M
    ro, Jo = oracle.evaluate_reproj(b["bearing"], w, b["cam"], P.keep["obs_pt"], b["aa"], b["t"], b["X"])
    rho1, _ = synth.huber_weights(ro, loss, a)
    radius, mn, mx = 50.0, 1e-6, 1e32
    packed = np.zeros(chk.chk_packed_size(F, len(P.ui)))
    chk.chk_reduce(C.byref(P.view), _dp(P.tab), 1, C.c_double(radius), C.c_double(mn), C.c_double(mx), _dp(packed))
    S, g, _, Ud, _ = synth.bundle_unpack(packed, F, P.ui, P.uj)
    # camera columns: same scaling/damping rule on the un-eliminated diagonal
    sc_c = 1.0 / (1.0 + np.sqrt(Ud.reshape(-1)))
    lam_c = np.clip(Ud.reshape(-1) * sc_c ** 2, mn, mx) / (radius * sc_c ** 2)
    dc = np.linalg.solve(S + np.diag(lam_c), -g)
    out3 = np.zeros(3)
    chk.chk_step(C.byref(P.view), _dp(P.tab), _dp(np.ascontiguousarray(dc.reshape(F, 6))), _dp(out3))
    # full system
    N = 6 * F + 3 * M
    Jf = np.zeros((n, N))
    for i in range(n):
        Jf[i, 6 * b["cam"][i]:6 * b["cam"][i] + 6] = Jo[i, :6]
        Jf[i, 6 * F + 3 * P.keep["obs_pt"][i]:6 * F + 3 * P.keep["obs_pt"][i] + 3] = Jo[i, 6:]
    H = Jf.T @ (rho1[:, None] * Jf); gf = Jf.T @ (rho1 * ro)
    sc = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    lam = np.clip(np.diag(H) * sc ** 2, mn, mx) / (radius * sc ** 2)
    d = np.linalg.solve(H + np.diag(lam), -gf)
    assert np.allclose(dc, d[:6 * F], rtol=1e-7, atol=1e-12)
    assert np.allclose(P.keep["Xc"] - b["X"], d[6 * F:].reshape(M, 3), rtol=1e-7, atol=1e-12)
    assert np.isclose(out3[0], -(gf @ d + 0.5 * d @ H @ d), rtol=1e-8)
