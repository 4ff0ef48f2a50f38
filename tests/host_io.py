"""Binary scan files for the C++ host-mirror test driver (tests/cpp/pvlm_host_driver.cpp)."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def driver():
    from panovlm_amd import build
    build.build_host()
    return build.HOST_DRIVER


def _cloud(f, xyz, tag=None):
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    tag = np.ones(len(xyz), np.float32) if tag is None else np.asarray(tag, np.float32)
    f.write(struct.pack("<i", len(xyz)))
    f.write(np.concatenate([xyz, tag[:, None]], axis=1).astype(np.float32).tobytes())


def write_scans(path, scans, world=True):
    """scans: list of dicts with LOCAL-frame clouds: id, valid, R_wl, t_wl, flat_local(+flat_tag), less_local(+less_tag),
    corner_local, p2s, seg_points (list of index arrays into corner_local), seg_coeffs, end_points."""
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(scans)))
        for s in scans:
            f.write(struct.pack("<iii", int(s.get("id", 0)), int(s.get("valid", 1)), 1 if world else 0))
            f.write(np.asarray(s["R_wl"], np.float64).reshape(9).tobytes())
            f.write(np.asarray(s["t_wl"], np.float64).reshape(3).tobytes())
            _cloud(f, s.get("flat_local", np.zeros((0, 3))), s.get("flat_tag"))
            _cloud(f, s.get("less_local", np.zeros((0, 3))), s.get("less_tag"))
            corner = np.asarray(s.get("corner_local", np.zeros((0, 3))), np.float32).reshape(-1, 3)
            _cloud(f, corner)
            segs = s.get("seg_points", [])
            f.write(struct.pack("<i", len(segs)))
            for k, idx in enumerate(segs):
                _cloud(f, corner[np.asarray(idx, int)])
                f.write(np.asarray(s["seg_coeffs"][k], np.float64).tobytes())
                f.write(np.asarray(s["end_points"][k], np.float64).tobytes())
            p2s = s.get("p2s", [[] for _ in range(len(corner))])
            for l in p2s:
                f.write(struct.pack("<i", len(l)))
                for v in sorted(l):
                    f.write(struct.pack("<i", int(v)))


def run(*args, timeout=600):
    # PVLM_DRIVER_PREFIX: a command the driver is run under, e.g. "rocprofv3 --kernel-trace --stats -d DIR --" (profiling tools only)
    prefix = os.environ.get("PVLM_DRIVER_PREFIX", "").split()
    out = subprocess.run(prefix + [driver()] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    if out.returncode != 0:
        raise RuntimeError("driver failed (%d): %s" % (out.returncode, out.stderr[-2000:]))
    if os.environ.get("PVLM_HOST_EVAL_TRACE") or os.environ.get("PVLM_FEATURE_PROFILE"):
        import sys
        sys.stderr.write(out.stderr)
    return out.stdout.splitlines()


def write_frames(path, T_cl, frames):
    """frames: list of dict(id, rows, cols, valid, R_wc, t_wc, lines (n x 4 float32))."""
    with open(path, "wb") as f:
        f.write(np.asarray(T_cl, np.float64).reshape(16).tobytes())
        f.write(struct.pack("<i", len(frames)))
        for fr in frames:
            f.write(struct.pack("<iiii", int(fr["id"]), int(fr["rows"]), int(fr["cols"]), int(fr.get("valid", 1))))
            f.write(np.asarray(fr["R_wc"], np.float64).reshape(9).tobytes())
            f.write(np.asarray(fr["t_wc"], np.float64).reshape(3).tobytes())
            lines = np.asarray(fr["lines"], np.float32).reshape(-1, 4)
            f.write(struct.pack("<i", len(lines)))
            f.write(lines.tobytes())


def write_structure(path, frames, tracks):
    """frames: list of dict with "keypoints" (n x 2 float32 pixels); tracks: list of dict(point (3), obs [(frame, keypoint), ...])."""
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for fr in frames:
            kp = np.asarray(fr.get("keypoints", np.zeros((0, 2))), np.float32).reshape(-1, 2)
            f.write(struct.pack("<i", len(kp))); f.write(kp.tobytes())
        f.write(struct.pack("<i", len(tracks)))
        for t in tracks:
            f.write(np.asarray(t["point"], np.float64).reshape(3).tobytes())
            f.write(struct.pack("<i", len(t["obs"])))
            for fi, ki in t["obs"]:
                f.write(struct.pack("<II", int(fi), int(ki)))


def extract_features(raw, n_scans=16, horizon=1800, max_curvature=1000.0, angle_threshold=5.0, segment=True, extract=True, edge_to_line=False, on_gpu=False):
    """Velodyne::ReOrderVLP (+ ExtractFeatures, optionally with EdgeToLine) of the host mirror on one raw scan (n x 4 float32)
    through the test driver.  on_gpu: Velodyne::ExtractFeaturesBatch instead (range-image stages, picks and voxel grid on the GPU; PVLM_FEATURE_PICKS=host: picks on the host).
    Returns a dict with the same fields as oracle.ScanFeatures."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        dst = os.path.join(d, "out.bin")
        if isinstance(raw, str):                       # a .pcd file: the driver goes through Velodyne::LoadLidar
            src = raw
        else:
            raw = np.ascontiguousarray(raw, np.float32).reshape(-1, 4)
            src = os.path.join(d, "raw.bin")
            with open(src, "wb") as f:
                f.write(struct.pack("<i", len(raw))); f.write(raw.tobytes())
        log = run("features", src, dst, str(n_scans), str(horizon), repr(float(max_curvature)), repr(float(angle_threshold)), "1" if segment else "0",
                  "1" if extract else "0", "1" if edge_to_line else "0", "1" if on_gpu else "0")
        buf = open(dst, "rb").read()
    pos = [0]

    def block(dtype, width=1):
        n = struct.unpack_from("<i", buf, pos[0])[0]; pos[0] += 4
        a = np.frombuffer(buf, dtype, n * width, pos[0]).copy(); pos[0] += a.nbytes
        return a.reshape(n, width) if width > 1 else a

    out = dict(log=log, valid=bool(struct.unpack_from("<i", buf, 0)[0]))
    pos[0] = 4
    for name in ("cloud_scan", "cornerSharp", "cornerLessSharp", "surfFlat", "surfLessFlat"):
        out[name] = block(np.float32, 4)
    out["rc"] = block(np.int32, 2)
    out["scan_start"] = block(np.int32); out["scan_end"] = block(np.int32)
    ri = block(np.float32)                                    # empty when the ring count is unsupported (upstream returns before sizing it)
    out["range_image"] = (ri if len(ri) else np.zeros(n_scans * horizon, np.float32)).reshape(n_scans, horizon)
    out["image_to_point_idx"] = block(np.int32).reshape(n_scans, horizon)
    out["curvature"] = block(np.float32)
    for name in ("state", "sort_ind", "left", "right"):
        out[name] = block(np.int32)
    if edge_to_line and extract:
        out["cornerBeforeFilter"] = block(np.float32, 4)
        so = block(np.int32); pts = block(np.float32, 4)
        out["edge_segmented"] = [pts[so[k]:so[k + 1]] for k in range(len(so) - 1)]
        out["segment_coeffs"] = block(np.float64, 6)
        out["end_points"] = block(np.float64, 6).reshape(-1, 2, 3)
        po = block(np.int32); pid = block(np.int32)
        out["point_to_segment"] = [pid[po[k]:po[k + 1]].tolist() for k in range(len(po) - 1)]
    assert pos[0] == len(buf)
    return out


def write_raw_scans(path, scans):
    """scans: list of dicts id, R_wl, t_wl, raw (n x 4 float32, firing order) for the driver's `rawodometry` command."""
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(scans)))
        for s in scans:
            raw = np.ascontiguousarray(s["raw"], np.float32).reshape(-1, 4)
            f.write(struct.pack("<i", int(s["id"])))
            f.write(np.asarray(s["R_wl"], np.float64).reshape(9).tobytes()); f.write(np.asarray(s["t_wl"], np.float64).reshape(3).tobytes())
            f.write(struct.pack("<i", len(raw))); f.write(raw.tobytes())



def fuse_depth_images(depth_filter, depth_file, conf, bgr, T_wc, neighbors, max_depth=20.0, thr=0.01, frame_id=None):
    """Host mirror pvlm::FuseDepthImages through the driver (`fusedepth`): same arguments and return value as oracle.mvs_fuse_depth_images."""
    import tempfile
    n = len(conf); rows, cols = np.shape(conf[0])
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(src, "wb") as f:
            f.write(struct.pack("<iii", n, rows, cols))
            for i in range(n):
                f.write(struct.pack("<i", i if frame_id is None else int(frame_id[i])))
                for m in (depth_filter[i], depth_file[i]):
                    f.write(struct.pack("<i", 0 if m is None else 1))
                    if m is not None:
                        f.write(np.ascontiguousarray(m, np.float32).tobytes())
                f.write(np.ascontiguousarray(conf[i], np.float32).tobytes())
                f.write(np.ascontiguousarray(bgr[i], np.uint8).tobytes())
                f.write(np.asarray(T_wc[i], np.float64).reshape(16).tobytes())
                f.write(struct.pack("<i", len(neighbors[i])))
                for (j, R, t) in neighbors[i]:
                    f.write(struct.pack("<i", int(j))); f.write(np.asarray(R, np.float32).reshape(9).tobytes()); f.write(np.asarray(t, np.float32).reshape(3).tobytes())
        run("fusedepth", src, dst, repr(float(max_depth)), repr(float(thr)))
        raw = open(dst, "rb").read()
    m = struct.unpack_from("<q", raw, 0)[0]; at = 8
    xyz = np.frombuffer(raw, np.float32, 3 * m, at).reshape(m, 3).copy(); at += 12 * m
    rgb = np.frombuffer(raw, np.uint8, 3 * m, at).reshape(m, 3).copy(); at += 3 * m
    maps = []
    for i in range(n):
        present = struct.unpack_from("<i", raw, at)[0]; at += 4
        if present:
            maps.append(np.frombuffer(raw, np.float32, rows * cols, at).reshape(rows, cols).copy()); at += 4 * rows * cols
        else:
            maps.append(None)
    return xyz, rgb, maps
