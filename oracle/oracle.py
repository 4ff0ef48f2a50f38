"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/liboracle.so`` (the dependency-free C++ restatement of the reference's
hot path, see the headers in this directory for the reference file:line each function follows) and
of ``oracle/_ref/libref_math.so`` (the REAL ``/root/reference/base/Math.h`` compiled where it lies).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package ``panovlm_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_O0 = None
_REF = None

KIND_P2PLANE_METER = 0
KIND_P2PLANE_ANGLE = 1
KIND_P2LINE_METER = 2
KIND_P2LINE_ANGLE = 3
KIND_PLANE2PLANE_GLOBAL = 4
KIND_PLANE_IOU = 5
STRIDE = {0: 8, 1: 8, 2: 10, 3: 10, 4: 10, 5: 12}


def build(force=False):
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


class OrcScan(C.Structure):
    _fields_ = [
        ("id", C.c_int), ("valid", C.c_int),
        ("R_wl", C.POINTER(C.c_double)), ("t_wl", C.POINTER(C.c_double)),
        ("n_flat", C.c_int), ("flat_xyz", C.POINTER(C.c_float)), ("flat_tag", C.POINTER(C.c_float)),
        ("n_less", C.c_int), ("less_xyz", C.POINTER(C.c_float)), ("less_tag", C.POINTER(C.c_float)),
        ("n_corner", C.c_int), ("corner_xyz", C.POINTER(C.c_float)),
        ("p2s_offsets", C.POINTER(C.c_int)), ("p2s_ids", C.POINTER(C.c_int)),
        ("n_seg", C.c_int), ("seg_size", C.POINTER(C.c_int)), ("seg_coeffs", C.POINTER(C.c_double)),
        ("end_points", C.POINTER(C.c_double)),
    ]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_fast_atan2_f.restype = C.c_float
        _LIB.orc_fast_atan2_f.argtypes = [C.c_float, C.c_float]
        _LIB.orc_fast_atan2_d.restype = C.c_double
        _LIB.orc_fast_atan2_d.argtypes = [C.c_double, C.c_double]
    return _LIB


def ref_math():
    """The real reference Math.h build, or None when oracle/_ref is absent."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_math.so")
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
        _REF.ref_fast_atan2_f.restype = C.c_float
        _REF.ref_fast_atan2_f.argtypes = [C.c_float, C.c_float]
        _REF.ref_fast_atan2_d.restype = C.c_double
        _REF.ref_fast_atan2_d.argtypes = [C.c_double, C.c_double]
    return _REF


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def num_threads():
    return int(lib().orc_num_threads())


def evaluate(kind, rec, ref_id, nei_id, aa, t, normalize=False, jac=True, threads=0, extended=False):
    """AutoDiff-style evaluation of n residual blocks.  rec: n x STRIDE[kind]; returns (r, J|None).
    extended=True: the same statements in x87 extended precision (oracle/costfunction.hpp AutoDiffEvaluateExt), rounded to double."""
    rec = _f64(rec); aa = _f64(aa); t = _f64(t)
    ref_id = _i32(ref_id); nei_id = _i32(nei_id)
    n = rec.shape[0]
    assert rec.shape[1] == STRIDE[kind]
    r = np.empty(n, np.float64)
    J = np.empty((n, 12), np.float64) if jac else None
    rc = lib().orc_eval(C.c_int(kind), C.c_int((1 if normalize else 0) | (2 if extended else 0)), C.c_long(n), _p(rec, C.c_double),
                        C.c_int(rec.shape[1]), _p(ref_id, C.c_int), _p(nei_id, C.c_int), _p(aa, C.c_double),
                        _p(t, C.c_double), _p(r, C.c_double), _p(J, C.c_double), C.c_int(threads))
    assert rc == 0
    return r, J


def evaluate_relative(kind, rec, aa_cl, t_cl, jac=True):
    """The calibration-mode functors of CameraLidarOptimizer::Optimize(line_pairs, T_cl): kind 4 = Plane2Plane_Relative (residual in
    degrees), 5 = PlaneRelativeIOUResidual; one pose (aa_cl, t_cl); records as for evaluate().  Returns (r, J n x 6 | None)."""
    rec = _f64(rec); aa_cl = _f64(aa_cl); t_cl = _f64(t_cl)
    n = rec.shape[0]
    assert rec.shape[1] == STRIDE[kind]
    r = np.empty(n, np.float64)
    J = np.empty((n, 6), np.float64) if jac else None
    rc = lib().orc_eval_relative(C.c_int(kind), C.c_long(n), _p(rec, C.c_double), C.c_int(rec.shape[1]), _p(aa_cl, C.c_double), _p(t_cl, C.c_double),
                                 _p(r, C.c_double), _p(J, C.c_double))
    assert rc == 0
    return r, J


def branch_distance(kind, rec, ref_id, nei_id, aa, t):
    """The distance the *_Angle functors (kind 1, 3) test against 1e-3 first (CostFunction.h:680-684, :893-897), extended precision."""
    rec = _f64(rec); aa = _f64(aa); t = _f64(t); ref_id = _i32(ref_id); nei_id = _i32(nei_id)
    d = np.empty(rec.shape[0], np.float64)
    rc = lib().orc_branch_distance(C.c_int(kind), C.c_long(rec.shape[0]), _p(rec, C.c_double), C.c_int(rec.shape[1]), _p(ref_id, C.c_int),
                                   _p(nei_id, C.c_int), _p(aa, C.c_double), _p(t, C.c_double), _p(d, C.c_double))
    assert rc == 0
    return d


def evaluate_reproj(bearing, weight, cam_id, pt_id, aa, t, X, jac=True):
    """PanoramaReprojResidual_1Angle blocks: returns (r, J n x 9 [aa_cw | t_cw | X]); bearing n x 3 (any norm)."""
    bearing = _f64(bearing); aa = _f64(aa); t = _f64(t); X = _f64(X)
    cam_id = _i32(cam_id); pt_id = _i32(pt_id)
    n = bearing.shape[0]
    r = np.empty(n, np.float64)
    J = np.empty((n, 9), np.float64) if jac else None
    rc = lib().orc_eval_reproj(C.c_long(n), _p(bearing, C.c_double), C.c_double(weight), _p(cam_id, C.c_int), _p(pt_id, C.c_int),
                               _p(aa, C.c_double), _p(t, C.c_double), _p(X, C.c_double), _p(r, C.c_double), _p(J, C.c_double))
    assert rc == 0
    return r, J


def evaluate_unoptimised(kind, rec, ref_id, nei_id, aa, t, normalize=False, threads=1):
    """Same as evaluate (r + J), run by the -O0 build of the oracle (liboracle_O0.so): the reference's CMakeLists sets no
    optimisation flag.  Timing aid for bench.py's cpu_baseline only."""
    global _LIB_O0
    if _LIB_O0 is None:
        build()
        _LIB_O0 = C.CDLL(os.path.join(_HERE, "liboracle_O0.so"))
    rec = _f64(rec); aa = _f64(aa); t = _f64(t); ref_id = _i32(ref_id); nei_id = _i32(nei_id)
    n = rec.shape[0]
    r = np.empty(n, np.float64); J = np.empty((n, 12), np.float64)
    rc = _LIB_O0.orc_eval(C.c_int(kind), C.c_int(1 if normalize else 0), C.c_long(n), _p(rec, C.c_double), C.c_int(rec.shape[1]), _p(ref_id, C.c_int),
                          _p(nei_id, C.c_int), _p(aa, C.c_double), _p(t, C.c_double), _p(r, C.c_double), _p(J, C.c_double), C.c_int(threads))
    assert rc == 0
    return r, J


def huber(a, s):
    s = _f64(s)
    rho = np.empty((s.shape[0], 3), np.float64)
    lib().orc_huber(C.c_double(a), C.c_long(s.shape[0]), _p(s, C.c_double), _p(rho, C.c_double))
    return rho


def angle_axis_to_matrix(aa):
    aa = _f64(aa); R = np.empty(9, np.float64)
    lib().orc_angle_axis_to_matrix(_p(aa, C.c_double), _p(R, C.c_double))
    return R.reshape(3, 3).T.copy()  # column-major -> numpy row-major


def matrix_to_angle_axis(R):
    Rc = _f64(np.asarray(R).T.reshape(-1)); aa = np.empty(3, np.float64)
    lib().orc_matrix_to_angle_axis(_p(Rc, C.c_double), _p(aa, C.c_double))
    return aa


def form_plane_lsq(pts, tol):
    pts = _f64(pts); plane = np.empty(4, np.float64)
    ok = lib().orc_form_plane_lsq(_p(pts, C.c_double), C.c_int(pts.shape[0]), C.c_double(tol), _p(plane, C.c_double))
    return bool(ok), plane


def form_line_pca(pts, tol, dis_thr=0.0):
    pts = _f64(pts); line = np.empty(6, np.float64)
    ok = lib().orc_form_line_pca(_p(pts, C.c_double), C.c_int(pts.shape[0]), C.c_double(tol), C.c_double(dis_thr), _p(line, C.c_double))
    return bool(ok), line


def eig_sym3(S):
    S = _f64(S); w = np.empty(3, np.float64); V = np.empty((3, 3), np.float64)
    lib().orc_eig_sym3(_p(S, C.c_double), _p(w, C.c_double), _p(V, C.c_double))
    return w, V


def knn(tgt, q, k):
    tgt = _f32(tgt); q = _f32(q)
    idx = np.empty((q.shape[0], k), np.int32); sqd = np.empty((q.shape[0], k), np.float32)
    bad = lib().orc_knn(_p(tgt, C.c_float), C.c_int(tgt.shape[0]), _p(q, C.c_float), C.c_int(q.shape[0]), C.c_int(k),
                        _p(idx, C.c_int), _p(sqd, C.c_float))
    assert bad == 0
    return idx, sqd


class ScanArrays:
    """Keeps numpy buffers alive for an OrcScan."""

    def __init__(self, scan):
        """scan: dict with optional keys id, valid, R_wl(3x3), t_wl(3), flat_xyz, flat_tag, less_xyz,
        less_tag, corner_xyz, p2s (list of lists), seg_size, seg_coeffs(Sx6), end_points(Sx6)."""
        g = scan.get
        self.R = _f64(g("R_wl", np.eye(3))).reshape(-1)
        self.t = _f64(g("t_wl", np.zeros(3)))
        self.flat = _f32(g("flat_xyz", np.zeros((0, 3)))); self.flat_tag = _f32(g("flat_tag", np.ones(len(self.flat))))
        self.less = _f32(g("less_xyz", np.zeros((0, 3)))); self.less_tag = _f32(g("less_tag", np.ones(len(self.less))))
        self.corner = _f32(g("corner_xyz", np.zeros((0, 3))))
        p2s = g("p2s", None)
        if p2s is None:
            p2s = [[] for _ in range(len(self.corner))]
        off = np.zeros(len(p2s) + 1, np.int32)
        for i, l in enumerate(p2s):
            off[i + 1] = off[i] + len(l)
        self.p2s_off = off
        self.p2s_ids = _i32([v for l in p2s for v in l]) if off[-1] > 0 else np.zeros(1, np.int32)
        self.seg_size = _i32(g("seg_size", np.zeros(0)))
        self.seg_coeffs = _f64(g("seg_coeffs", np.zeros((0, 6))))
        self.end_points = _f64(g("end_points", np.zeros((len(self.seg_size), 6))))
        s = OrcScan()
        s.id = int(g("id", 0)); s.valid = int(g("valid", 1))
        s.R_wl = _p(self.R, C.c_double); s.t_wl = _p(self.t, C.c_double)
        s.n_flat = len(self.flat); s.flat_xyz = _p(self.flat, C.c_float); s.flat_tag = _p(self.flat_tag, C.c_float)
        s.n_less = len(self.less); s.less_xyz = _p(self.less, C.c_float); s.less_tag = _p(self.less_tag, C.c_float)
        s.n_corner = len(self.corner); s.corner_xyz = _p(self.corner, C.c_float)
        s.p2s_offsets = _p(self.p2s_off, C.c_int); s.p2s_ids = _p(self.p2s_ids, C.c_int)
        s.n_seg = len(self.seg_size); s.seg_size = _p(self.seg_size, C.c_int)
        s.seg_coeffs = _p(self.seg_coeffs, C.c_double); s.end_points = _p(self.end_points, C.c_double)
        self.c = s


def assoc_point2plane(ref, nei, tol, thr, want_knn=False):
    """ref/nei: scan dicts (world-frame float clouds).  Returns dict(point, plane, qidx, nn[, knn_all])."""
    r = ScanArrays(ref); n = ScanArrays(nei)
    cap = max(1, n.c.n_flat)
    pt = np.empty((cap, 3)); pl = np.empty((cap, 4)); qi = np.empty(cap, np.int32); nn = np.empty((cap, 10), np.int32)
    kall = np.empty((cap, 10), np.int32) if want_knn else None
    m = lib().orc_assoc_point2plane(C.byref(r.c), C.byref(n.c), C.c_double(tol), C.c_float(thr), _p(pt, C.c_double),
                                    _p(pl, C.c_double), _p(qi, C.c_int), _p(nn, C.c_int), _p(kall, C.c_int))
    out = dict(point=pt[:m].copy(), plane=pl[:m].copy(), qidx=qi[:m].copy(), nn=nn[:m].copy())
    if want_knn:
        out["knn_all"] = kall
    return out


def assoc_point2line(ref, nei, thr, mode="knn"):
    """mode "knn": AssociatePoint2Line, "segment_knn": AssociatePoint2LineSegmentKNN, "segment": AssociatePoint2LineSegment."""
    r = ScanArrays(ref); n = ScanArrays(nei)
    cap = max(1, n.c.n_corner * max(1, r.c.n_seg))
    pt = np.empty((cap, 3)); a = np.empty((cap, 3)); b = np.empty((cap, 3)); qi = np.empty(cap, np.int32)
    m = lib().orc_assoc_point2line(C.byref(r.c), C.byref(n.c), C.c_float(thr), C.c_int({"knn": 0, "segment_knn": 1, "segment": 2}[mode]),
                                   _p(pt, C.c_double), _p(a, C.c_double), _p(b, C.c_double), _p(qi, C.c_int))
    return dict(point=pt[:m].copy(), a=a[:m].copy(), b=b[:m].copy(), qidx=qi[:m].copy())


def assoc_line2line(ref, nei, thr, knn=False):
    """knn=False: AssociateLine2Line (distance votes); knn=True: AssociateLine2LineKNN (5-NN segment votes)."""
    r = ScanArrays(ref); n = ScanArrays(nei)
    ns, rs = max(1, n.c.n_seg), max(1, r.c.n_seg)
    ni = np.empty(ns, np.int32); ri = np.empty(ns, np.int32); p1 = np.empty((ns, 3)); p2 = np.empty((ns, 3))
    votes = np.zeros((ns, rs), np.int32)
    m = lib().orc_assoc_line2line(C.byref(r.c), C.byref(n.c), C.c_float(thr), C.c_int(1 if knn else 0), _p(ni, C.c_int), _p(ri, C.c_int),
                                  _p(p1, C.c_double), _p(p2, C.c_double), _p(votes, C.c_int))
    return dict(nei_idx=ni[:m].copy(), ref_idx=ri[:m].copy(), p1=p1[:m].copy(), p2=p2[:m].copy(),
                votes=votes[:n.c.n_seg, :r.c.n_seg].copy())


def find_neighbors(poses, valid, neighbor_size):
    """poses: F x 12 ([R_wl row-major | t_wl]).  Returns list of lists."""
    poses = _f64(poses); valid = _i32(valid)
    F = poses.shape[0]
    cap = F * (neighbor_size + 4) + F * 64 + 16
    off = np.empty(F + 1, np.int32); ids = np.empty(cap, np.int32)
    tot = lib().orc_find_neighbors(C.c_int(F), _p(poses, C.c_double), _p(valid, C.c_int), C.c_int(neighbor_size),
                                   _p(off, C.c_int), _p(ids, C.c_int), C.c_int(cap))
    assert tot >= 0
    return [ids[off[i]:off[i + 1]].tolist() for i in range(F)]


def fast_atan2_f(y, x):
    return float(lib().orc_fast_atan2_f(C.c_float(y), C.c_float(x)))


def fast_atan2_d(y, x):
    return float(lib().orc_fast_atan2_d(C.c_double(y), C.c_double(x)))


def cam_to_image(rows, cols, cam):
    cam = np.ascontiguousarray(cam)
    if cam.dtype == np.float32:
        px = np.empty((cam.shape[0], 2), np.float32)
        lib().orc_cam_to_image_f(C.c_int(rows), C.c_int(cols), C.c_long(cam.shape[0]), _p(cam, C.c_float), _p(px, C.c_float))
    else:
        cam = _f64(cam); px = np.empty((cam.shape[0], 2), np.float64)
        lib().orc_cam_to_image_d(C.c_int(rows), C.c_int(cols), C.c_long(cam.shape[0]), _p(cam, C.c_double), _p(px, C.c_double))
    return px


def project_lidar_depth(rows, cols, xyz, T_cl, size=3):
    """ProjectLidar2PanoramaDepth (util/Visualization.h:407-441): rows x cols uint16 image of depth * 256."""
    xyz = _f32(xyz).reshape(-1, 3); T = _f64(T_cl).reshape(16)
    out = np.zeros((rows, cols), np.uint16)
    lib().orc_project_lidar_depth(C.c_int(rows), C.c_int(cols), C.c_long(xyz.shape[0]), _p(xyz, C.c_float), _p(T, C.c_double), C.c_ulong(size),
                                  _p(out, C.c_ushort))
    return out


def mvs_init_conf_map(ref_gray, nei_grays, R_nr, t_nr, depth, normal, half_window=3, step=1, conf=None, nei_depths=None):
    """InitPatchMap + InitConfMap(use_geometry=False) (mvs/MVS.cpp:586-680, :774-923).  Returns (conf, depth, normal) copies:
    conf = score of every pixel with depth > 0 (-1 = invalid; depth / normal of those are zeroed), other pixels keep `conf`."""
    ref = np.ascontiguousarray(ref_gray, np.uint8); rows, cols = ref.shape
    neis = [np.ascontiguousarray(g, np.uint8) for g in nei_grays]
    ptrs = (C.POINTER(C.c_ubyte) * max(len(neis), 1))(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
    R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
    d = np.array(depth, np.float32, copy=True); nrm = np.array(normal, np.float32, copy=True)
    c = np.zeros((rows, cols), np.float32) if conf is None else np.array(conf, np.float32, copy=True)
    dptrs = None
    if nei_depths is not None:      # use_geometry = true: the neighbours' photometric depth maps (depth_filter)
        nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
        dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
    lib().orc_mvs_init_conf_map(C.c_int(rows), C.c_int(cols), C.c_int(half_window), C.c_int(step), _p(ref, C.c_ubyte), C.c_int(len(neis)), ptrs,
                                _p(R, C.c_float), _p(t, C.c_float), _p(d, C.c_float), _p(nrm, C.c_float), _p(c, C.c_float), dptrs)
    return c, d, nrm


def mvs_propagate(ref_gray, nei_grays, R_nr, t_nr, depth, normal, conf, half_window=3, step=1, nei_depths=None, depth_constant=None, min_depth=0.1,
                  max_depth=20.0, seed=1, max_iter=1, conf_threshold=-1.0, sequential=False):
    """EstimateDepthMapSingle(CHECKER_BOARD) (mvs/MVS.cpp:682-772, :1098-1129, :1254-1431), or with sequential=True
    EstimateDepthMapSingle(SEQUENTIAL) (PropagateSequential :1057-1097, single-threaded): returns (depth, normal, conf) copies."""
    ref = np.ascontiguousarray(ref_gray, np.uint8); rows, cols = ref.shape
    neis = [np.ascontiguousarray(g, np.uint8) for g in nei_grays]
    ptrs = (C.POINTER(C.c_ubyte) * max(len(neis), 1))(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in neis])
    R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
    d = np.array(depth, np.float32, copy=True); nrm = np.array(normal, np.float32, copy=True); c = np.array(conf, np.float32, copy=True)
    dptrs = None
    if nei_depths is not None:
        nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
        dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
    dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
    fn = lib().orc_mvs_propagate_sequential if sequential else lib().orc_mvs_propagate
    fn(C.c_int(rows), C.c_int(cols), C.c_int(half_window), C.c_int(step), _p(ref, C.c_ubyte), C.c_int(len(neis)), ptrs, _p(R, C.c_float),
       _p(t, C.c_float), _p(d, C.c_float), _p(nrm, C.c_float), _p(c, C.c_float), dptrs, _p(dc, C.c_ubyte), C.c_float(min_depth),
       C.c_float(max_depth), C.c_ulonglong(seed), C.c_int(max_iter), C.c_float(conf_threshold))
    return d, nrm, c


def mvs_depth_to_cloud(depth, bgr, T_wc, max_depth=20.0, filter_sky=True, normal=None):
    """MVS::DepthImageToCloud (mvs/MVS.cpp:2073-2107) / DepthNormalToCloud (:2109-2142, normal given, filter_sky False):
    returns (xyz n x 3 float32, rgb n x 3 uint8[, normal n x 3 float32]) in raster order."""
    d = np.ascontiguousarray(depth, np.float32); rows, cols = d.shape
    c = np.ascontiguousarray(bgr, np.uint8).reshape(rows, cols, 3)
    T = np.ascontiguousarray(np.asarray(T_wc, np.float64).reshape(-1)[:12])
    xyz = np.zeros((rows * cols, 3), np.float32); rgb = np.zeros((rows * cols, 3), np.uint8)
    nin = None if normal is None else np.ascontiguousarray(normal, np.float32).reshape(rows, cols, 3)
    nout = None if normal is None else np.zeros((rows * cols, 3), np.float32)
    lib().orc_mvs_depth_to_cloud.restype = C.c_longlong
    n = lib().orc_mvs_depth_to_cloud(C.c_int(rows), C.c_int(cols), _p(d, C.c_float), _p(c, C.c_ubyte), _p(T, C.c_double), C.c_float(max_depth), _p(xyz, C.c_float),
                                     _p(rgb, C.c_ubyte), C.c_int(1 if filter_sky else 0), None if nin is None else _p(nin, C.c_float),
                                     None if nout is None else _p(nout, C.c_float))
    if normal is None:
        return xyz[:n].copy(), rgb[:n].copy()
    return xyz[:n].copy(), rgb[:n].copy(), nout[:n].copy()


def mvs_fuse_depth_images(depth_filter, depth_saved, conf, bgr, T_wc, neighbors, max_depth=20.0, thr=0.01, frame_id=None):
    """MVS::FuseDepthImages (mvs/MVS.cpp:2168-2334).  depth_filter / depth_saved: lists of rows x cols float32 maps or None (empty Mat / no
    file); conf, bgr: lists of maps; T_wc: n x 4 x 4; neighbors[i] = list of (id, R_nr 3x3, t_nr 3).  Returns (xyz, rgb, depth maps after the
    call — None where released)."""
    n = len(conf); rows, cols = np.shape(conf[0])
    work = [None if d is None else np.array(d, np.float32, order="C", copy=True) for d in depth_filter]
    saved = [None if d is None else np.ascontiguousarray(d, np.float32) for d in depth_saved]
    cf = [np.ascontiguousarray(c, np.float32) for c in conf]; cl = [np.ascontiguousarray(c, np.uint8).reshape(rows, cols, 3) for c in bgr]
    fp = lambda arrs, t: (C.POINTER(t) * n)(*[None if a is None else a.ctypes.data_as(C.POINTER(t)) for a in arrs])
    T = _f64(T_wc).reshape(n, 16)
    off = np.zeros(n + 1, np.int32); ids = []; R = []; t = []
    for i, nb in enumerate(neighbors):
        for (j, Rn, tn) in nb:
            ids.append(j); R.append(np.asarray(Rn, np.float32).reshape(9)); t.append(np.asarray(tn, np.float32).reshape(3))
        off[i + 1] = len(ids)
    ids = np.array(ids if ids else [0], np.int32)
    R = np.ascontiguousarray(np.array(R, np.float32).reshape(-1)) if R else np.zeros(9, np.float32)
    t = np.ascontiguousarray(np.array(t, np.float32).reshape(-1)) if t else np.zeros(3, np.float32)
    fid = np.arange(n, dtype=np.int32) if frame_id is None else np.ascontiguousarray(frame_id, np.int32)
    cap = n * rows * cols
    xyz = np.zeros((cap, 3), np.float32); rgb = np.zeros((cap, 3), np.uint8)
    present = np.zeros(n, np.int32); after = np.zeros((n, rows, cols), np.float32)
    lib().orc_mvs_fuse_depth_images.restype = C.c_longlong
    m = lib().orc_mvs_fuse_depth_images(C.c_int(n), C.c_int(rows), C.c_int(cols), fp(work, C.c_float), fp(saved, C.c_float), fp(cf, C.c_float), fp(cl, C.c_ubyte),
                                        _p(T, C.c_double), _p(fid, C.c_int), _p(off, C.c_int), _p(ids, C.c_int), _p(R, C.c_float), _p(t, C.c_float),
                                        C.c_float(max_depth), C.c_float(thr), _p(xyz, C.c_float), _p(rgb, C.c_ubyte), C.c_longlong(cap), _p(present, C.c_int),
                                        _p(after, C.c_float))
    return xyz[:m].copy(), rgb[:m].copy(), [after[i].copy() if present[i] else None for i in range(n)]


def mvs_init_depth_normal(rows, cols, lidar_depth16=None, mask=None, min_depth=0.1, max_depth=20.0, keep_lidar_constant=True, seed=1):
    """MVS::InitDepthNormal (mvs/MVS.cpp:496-584): returns (depth, normal, depth_constant uint8)."""
    l16 = None if lidar_depth16 is None else np.ascontiguousarray(lidar_depth16, np.uint16)
    m = None if mask is None else np.ascontiguousarray(mask, np.float32)
    d = np.zeros((rows, cols), np.float32); n = np.zeros((rows, cols, 3), np.float32); c = np.zeros((rows, cols), np.uint8)
    lib().orc_mvs_init_depth_normal(C.c_int(rows), C.c_int(cols), _p(l16, C.c_ushort), _p(m, C.c_float), C.c_float(min_depth), C.c_float(max_depth),
                                    C.c_int(1 if keep_lidar_constant else 0), C.c_ulonglong(seed), _p(d, C.c_float), _p(n, C.c_float), _p(c, C.c_ubyte))
    return d, n, c


def mvs_remove_small_segments(depth, normal, conf, depth_diff_threshold=0.01, min_segment=100):
    """MVS::RemoveSmallSegments (mvs/MVS.cpp:1504-1577): returns (depth, normal, conf, removed)."""
    d = np.array(depth, np.float32, copy=True); n = np.array(normal, np.float32, copy=True); c = np.array(conf, np.float32, copy=True)
    rows, cols = d.shape
    k = lib().orc_mvs_remove_small_segments(C.c_int(rows), C.c_int(cols), C.c_float(depth_diff_threshold), C.c_int(min_segment), _p(d, C.c_float), _p(n, C.c_float),
                                            _p(c, C.c_float))
    return d, n, c, int(k)


def mvs_select_neighbors(valid, R_wc, t_wc, neighbor_size, sq_distance_threshold):
    """MVS::SelectNeighborKNN (mvs/MVS.cpp:334-382): (ids n x k with -1 padding, R_nr n x k x 9 float32, t_nr n x k x 3 float32)."""
    v = _i32(np.asarray(valid, np.int32)); n = len(v)
    R = _f64(R_wc).reshape(-1); t = _f64(t_wc).reshape(-1)
    ids = np.full((n, neighbor_size), -1, np.int32); oR = np.zeros((n, neighbor_size, 9), np.float32); ot = np.zeros((n, neighbor_size, 3), np.float32)
    lib().orc_mvs_select_neighbors(C.c_int(n), _p(v, C.c_int), _p(R, C.c_double), _p(t, C.c_double), C.c_int(neighbor_size), C.c_float(sq_distance_threshold),
                                   _p(ids, C.c_int), _p(oR, C.c_float), _p(ot, C.c_float))
    return ids, oR, ot


def mvs_filter_depth(nei_depths, R_nr, t_nr, depth, conf=None, depth_constant=None, thr=0.01):
    """FilterDepthImage (mvs/MVS.cpp:1735-1790) with ProjectDepthConfToRef (:2011-2070): returns (depth_filter, conf_filter)."""
    d = np.ascontiguousarray(depth, np.float32); rows, cols = d.shape
    nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
    dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
    R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
    cf = None if conf is None else np.ascontiguousarray(conf, np.float32)
    dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
    out_d = np.zeros((rows, cols), np.float32); out_c = np.zeros((rows, cols), np.float32)
    lib().orc_mvs_filter_depth(C.c_int(rows), C.c_int(cols), C.c_int(len(nd)), dptrs, _p(R, C.c_float), _p(t, C.c_float), _p(d, C.c_float),
                               _p(cf, C.c_float), _p(dc, C.c_ubyte), C.c_float(thr), _p(out_d, C.c_float), _p(out_c, C.c_float))
    return out_d, out_c


def mvs_filter_depth_refine(nei_depths, nei_confs, R_nr, t_nr, depth, conf, depth_constant=None, thr=0.01, min_depth=0.1, max_depth=20.0):
    """FilterDepthImageRefine (mvs/MVS.cpp:1794-1890): returns (depth_filter, conf_filter, conf) — conf is the reference
    frame's conf_map after the call (zeroed where depth <= 0)."""
    d = np.ascontiguousarray(depth, np.float32); rows, cols = d.shape
    nd = [np.ascontiguousarray(x, np.float32) for x in nei_depths]
    nc = [np.ascontiguousarray(x, np.float32) for x in nei_confs]
    dptrs = (C.POINTER(C.c_float) * max(len(nd), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nd])
    cptrs = (C.POINTER(C.c_float) * max(len(nc), 1))(*[x.ctypes.data_as(C.POINTER(C.c_float)) for x in nc])
    R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
    cf = np.array(conf, np.float32, copy=True, order="C")
    dc = None if depth_constant is None else np.ascontiguousarray(depth_constant, np.uint8)
    out_d = np.zeros((rows, cols), np.float32); out_c = np.zeros((rows, cols), np.float32)
    lib().orc_mvs_filter_depth_refine(C.c_int(rows), C.c_int(cols), C.c_int(len(nd)), dptrs, cptrs, _p(R, C.c_float), _p(t, C.c_float), _p(d, C.c_float),
                                      _p(cf, C.c_float), _p(dc, C.c_ubyte), C.c_float(thr), C.c_float(min_depth), C.c_float(max_depth),
                                      _p(out_d, C.c_float), _p(out_c, C.c_float))
    return out_d, out_c, cf


def mvs_project_depth_conf(nei_depth, nei_conf, R_nr, t_nr):
    """ProjectDepthConfToRef with depth + confidence (mvs/MVS.cpp:2011-2070): returns (depth_projected, conf_projected)."""
    d = np.ascontiguousarray(nei_depth, np.float32); c = np.ascontiguousarray(nei_conf, np.float32); rows, cols = d.shape
    R = _f32(R_nr).reshape(-1); t = _f32(t_nr).reshape(-1)
    od = np.zeros((rows, cols), np.float32); oc = np.zeros((rows, cols), np.float32)
    lib().orc_mvs_project_depth_conf(C.c_int(rows), C.c_int(cols), _p(d, C.c_float), _p(c, C.c_float), _p(R, C.c_float), _p(t, C.c_float),
                                     _p(od, C.c_float), _p(oc, C.c_float))
    return od, oc


def mvs_fill_patch(gray, px, py, half_window=3, step=1):
    g = np.ascontiguousarray(gray, np.uint8); rows, cols = g.shape
    w = 2 * half_window + 1; q = w // step + (1 if step > 1 else 0); n = q * q
    weight = np.zeros(n, np.float32); tex = np.zeros(n, np.float32)
    lib().orc_mvs_fill_patch.restype = C.c_float
    sq0 = lib().orc_mvs_fill_patch(C.c_int(rows), C.c_int(cols), C.c_int(half_window), C.c_int(step), _p(g, C.c_ubyte), C.c_int(px), C.c_int(py),
                                   _p(weight, C.c_float), _p(tex, C.c_float))
    return weight, tex, float(sq0)


def image_to_cam(rows, cols, px, r=1.0):
    px = np.ascontiguousarray(px)
    if px.dtype == np.float32:
        cam = np.empty((px.shape[0], 3), np.float32)
        lib().orc_image_to_cam_f(C.c_int(rows), C.c_int(cols), C.c_long(px.shape[0]), _p(px, C.c_float), C.c_float(r), _p(cam, C.c_float))
    else:
        px = _f64(px); cam = np.empty((px.shape[0], 3), np.float64)
        lib().orc_image_to_cam_d(C.c_int(rows), C.c_int(cols), C.c_long(px.shape[0]), _p(px, C.c_double), C.c_double(r), _p(cam, C.c_double))
    return cam


def break_to_segments(rows, cols, start, end, length):
    s = _f32(start); e = _f32(end)
    out = np.empty(4096, np.float32)
    n = lib().orc_break_to_segments(C.c_int(rows), C.c_int(cols), _p(s, C.c_float), _p(e, C.c_float), C.c_float(length),
                                    _p(out, C.c_float), C.c_int(4096))
    assert n >= 0
    return out[:2 * n].reshape(n, 2).copy()


def assoc_by_angle(rows, cols, lines, lidar_local, T_cl, multiple=True):
    lines = _f32(lines); T = _f64(T_cl).reshape(-1)
    l = ScanArrays(lidar_local)
    cap = max(16, lines.shape[0] * max(1, l.c.n_seg))
    ii = np.empty(cap, np.int32); li = np.empty(cap, np.int32); sc = np.empty(cap, np.float32)
    st = np.empty((cap, 3)); en = np.empty((cap, 3))
    votes = np.zeros((lines.shape[0], max(1, l.c.n_seg)), np.int32)
    m = lib().orc_assoc_by_angle(C.c_int(rows), C.c_int(cols), _p(lines, C.c_float), C.c_int(lines.shape[0]), C.byref(l.c),
                                 _p(T, C.c_double), C.c_int(1 if multiple else 0), C.c_int(cap), _p(ii, C.c_int),
                                 _p(li, C.c_int), _p(sc, C.c_float), _p(st, C.c_double), _p(en, C.c_double), _p(votes, C.c_int))
    assert m >= 0
    return dict(image_line_id=ii[:m].copy(), lidar_line_id=li[:m].copy(), score=sc[:m].copy(), start=st[:m].copy(),
                end=en[:m].copy(), votes=votes)


def _features_handle(cloud, n_scans=16, horizon=1800, max_curvature=1000.0, intersect_angle_threshold=5.0, segment=True, extract=True, edge_to_line=False):
    c = _f32(cloud).reshape(-1, 4)
    L = lib()
    L.orc_features_create.restype = C.c_void_p
    return C.c_void_p(L.orc_features_create(C.c_long(len(c)), _p(c, C.c_float), C.c_int(n_scans), C.c_int(horizon), C.c_float(max_curvature),
                                            C.c_float(intersect_angle_threshold), C.c_int(1 if segment else 0),
                                            C.c_int((1 if extract else 0) | (2 if (extract and edge_to_line) else 0))))


class ScanFeatures:
    """ReOrderVLP (+ ExtractFeatures, ADAPTIVE, planar branch) of oracle/features.hpp on one raw scan (n x 4 float32)."""

    CLOUDS = {"cloud_scan": 0, "cornerSharp": 1, "cornerLessSharp": 2, "surfFlat": 3, "surfLessFlat": 4}

    def __init__(self, cloud, n_scans=16, horizon=1800, max_curvature=1000.0, intersect_angle_threshold=5.0, segment=True, extract=True, edge_to_line=False):
        """edge_to_line: also run Velodyne::EdgeToLine (oracle/lines.hpp): cornerSharp / cornerLessSharp are then the filtered clouds and
        edge_segmented (list of n x 4 arrays), segment_coeffs (S x 6), end_points (S x 2 x 3), point_to_segment (list of lists),
        cornerBeforeFilter are filled."""
        L = lib()
        L.orc_features_cloud.restype = C.c_long
        h = _features_handle(cloud, n_scans, horizon, max_curvature, intersect_angle_threshold, segment, extract, edge_to_line)
        try:
            self.valid = bool(L.orc_features_valid(h))
            for name, which in self.CLOUDS.items():
                n = L.orc_features_cloud(h, C.c_int(which), None)
                out = np.zeros((n, 4), np.float32)
                L.orc_features_cloud(h, C.c_int(which), _p(out, C.c_float))
                setattr(self, name, out)
            n = len(self.cloud_scan)
            self.rc = np.zeros((n, 2), np.int32)
            have = extract and self.valid and n > 0
            self.curvature = np.zeros(n if have else 0, np.float32)
            self.state, self.sort_ind, self.left, self.right = (np.zeros(n if have else 0, np.int32) for _ in range(4))
            self.scan_start = np.zeros(n_scans, np.int32); self.scan_end = np.zeros(n_scans, np.int32)
            self.range_image = np.zeros((n_scans, horizon), np.float32)
            self.image_to_point_idx = np.zeros((n_scans, horizon), np.int32)
            L.orc_features_arrays(h, _p(self.rc, C.c_int), _p(self.curvature, C.c_float) if have else None, _p(self.state, C.c_int) if have else None,
                                  _p(self.sort_ind, C.c_int) if have else None, _p(self.left, C.c_int) if have else None,
                                  _p(self.right, C.c_int) if have else None, _p(self.scan_start, C.c_int), _p(self.scan_end, C.c_int),
                                  _p(self.range_image, C.c_float), _p(self.image_to_point_idx, C.c_int))
            if extract and edge_to_line:
                ns, npt, nid, nb = C.c_int(), C.c_int(), C.c_int(), C.c_int()
                L.orc_lines_sizes(h, C.byref(ns), C.byref(npt), C.byref(nid), C.byref(nb))
                so = np.zeros(ns.value + 1, np.int32); sp = np.zeros((max(npt.value, 1), 4), np.float32)
                co = np.zeros((max(ns.value, 1), 6)); ep = np.zeros((max(ns.value, 1), 2, 3))
                po = np.zeros(len(self.cornerLessSharp) + 1, np.int32); pi = np.zeros(max(nid.value, 1), np.int32)
                bf = np.zeros((max(nb.value, 1), 4), np.float32)
                L.orc_lines_get(h, _p(so, C.c_int), _p(sp, C.c_float), _p(co, C.c_double), _p(ep, C.c_double), _p(po, C.c_int), _p(pi, C.c_int), _p(bf, C.c_float))
                self.edge_segmented = [sp[so[k]:so[k + 1]].copy() for k in range(ns.value)]
                self.segment_coeffs = co[:ns.value]; self.end_points = ep[:ns.value]
                self.point_to_segment = [pi[po[k]:po[k + 1]].tolist() for k in range(len(self.cornerLessSharp))]
                self.cornerBeforeFilter = bf[:nb.value]
        finally:
            L.orc_features_free(h)


def line_consensus(cloud_xyzi, threshold=0.02):
    """The exhaustive 2-point maximum-consensus line that stands in for pcl::SACSegmentation in FuseLines (oracle/lines.hpp)."""
    c = _f32(cloud_xyzi).reshape(-1, 4)
    out = np.zeros(max(len(c), 1), np.int32)
    n = lib().orc_line_consensus(C.c_int(len(c)), _p(c, C.c_float), C.c_double(threshold), _p(out, C.c_int))
    return out[:n]


def voxel_grid(cloud, leaf):
    c = _f32(cloud).reshape(-1, 4)
    out = np.zeros((max(len(c), 1), 4), np.float32)
    lib().orc_voxel_grid.restype = C.c_long
    n = lib().orc_voxel_grid(C.c_long(len(c)), _p(c, C.c_float), C.c_float(leaf), _p(out, C.c_float))
    return out[:n]


def _pose12(R, t):
    return np.concatenate([_f64(R).reshape(9), _f64(t).reshape(3)])


def undistort_cloud(cloud, R_wl, t_wl, R_we, t_we, pose_valid=True):
    """Velodyne::UndistortCloud (oracle/undistort.hpp): returns (done, cloud n x 4 float32)."""
    c = _f32(cloud).reshape(-1, 4).copy()
    a = _pose12(R_wl, t_wl); b = _pose12(R_we, t_we)
    done = lib().orc_undistort_cloud(_p(c, C.c_float), C.c_long(len(c)), _p(a, C.c_double), C.c_int(1 if pose_valid else 0), _p(b, C.c_double))
    return bool(done), c


def slerp_pose(R1, t1, R2, t2, ratio):
    """SlerpPose(pose_w1, pose_w2, ratio) of base/Geometry.hpp:572-583: returns (R, t)."""
    out = np.zeros(12)
    a = _pose12(R1, t1); b = _pose12(R2, t2)
    lib().orc_slerp_pose(_p(a, C.c_double), _p(b, C.c_double), C.c_double(ratio), _p(out, C.c_double))
    return out[:9].reshape(3, 3), out[9:]


def sweep_end_pose(poses, pose_ok, ok, i, gap_time):
    """The pose LidarOdometry::UndistortLidars hands to scan i's UndistortCloud, or None when the scan is left as it is.  poses: list of (R, t)."""
    flat = np.concatenate([_pose12(R, t) for R, t in poses])
    po = np.asarray(pose_ok, np.int8); o = np.asarray(ok, np.int8)
    out = np.zeros(12)
    got = lib().orc_sweep_end_pose(C.c_int(len(poses)), _p(flat, C.c_double), _p(po, C.c_char), _p(o, C.c_char), C.c_int(i), C.c_float(gap_time), _p(out, C.c_double))
    return (out[:9].reshape(3, 3), out[9:]) if got else None
