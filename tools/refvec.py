#!/usr/bin/env python3
"""Re-pinning the oracle against a REAL PanoVLM build.

The reference cannot be compiled in the development image (no Eigen/PCL/Ceres/OpenCV), so the golden
fixtures under tests/golden/ come from the CPU oracle ("parity unpinned").  On a machine where PanoVLM
builds, three steps re-pin them:

    python tools/refvec.py export /tmp/pvv                       # fixtures' INPUTS -> flat .pvv files
    ./dump_reference_vectors /tmp/pvv                            # tools/dump_reference_vectors.cpp, linked to PanoVLM
    python tools/refvec.py compare /tmp/pvv                      # reference OUTPUTS vs the fixtures' expectations
    python tools/refvec.py regenerate /tmp/pvv [out_dir] [tag]   # rewrite the fixtures' expectations FROM the reference outputs (-> pinned)
    python tools/refvec.py table                                 # which fixture pins which row of SURVEY.md section 8

`make repin REFERENCE=/path/to/PanoVLM` at the repository root runs all of it (and the CPU suite against the regenerated fixtures).

.pvv container: b"PVV1", u32 count, then per array: u32 name_len, name, u8 dtype (0 f32, 1 f64, 2 i32, 3 i64),
u32 ndim, u64 dims[ndim], raw little-endian data (C order).
"""
import os
import struct
import sys

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CODES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2, np.dtype(np.int64): 3}
DTYPES = {v: k for k, v in CODES.items()}


def write_pvv(path, arrays):
    with open(path, "wb") as f:
        f.write(b"PVV1" + struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype not in CODES:
                a = a.astype(np.int32 if a.dtype.kind in "iu" and a.dtype.itemsize < 4 else np.float64)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<BI", CODES[a.dtype], a.ndim))
            f.write(struct.pack("<%dQ" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read_pvv(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"PVV1"
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (nl,) = struct.unpack("<I", f.read(4))
            name = f.read(nl).decode()
            code, ndim = struct.unpack("<BI", f.read(5))
            dims = struct.unpack("<%dQ" % ndim, f.read(8 * ndim))
            dt = DTYPES[code]
            n = int(np.prod(dims)) if ndim else 1
            out[name] = np.frombuffer(f.read(n * dt.itemsize), dt).reshape(dims)
    return out


# which arrays of each fixture are EXPECTED OUTPUTS (everything else is input)
def _is_output(fixture, key):
    if fixture == "functors":
        return key.endswith("_r") or key.endswith("_J")
    if fixture == "assoc_point2plane":
        return key.startswith("c") and key != "cases"
    if fixture == "equirect":
        return key.startswith(("px_", "cam_f64_", "seg_"))
    if fixture == "lines":
        return key.startswith(("t03_", "t04_", "m0_", "m1_")) or key == "c_votes"
    if fixture == "neighbors":
        return key in ("off", "ids")
    if fixture == "fast_atan2":
        return key.startswith("out_")
    if fixture == "reproj":
        return key in ("r", "J")
    if fixture == "depth":
        return key.startswith("depth_size")
    if fixture in ("features", "line_extraction"):
        return key not in ("raw", "horizon")
    if fixture == "undistort":
        return key.startswith("out") or key == "slerp"
    raise KeyError(fixture)


# intermediate results the reference API does not expose (k-NN table, query indices, vote matrices)
INTERNAL = ("_qidx", "_nn", "votes",
            # private working arrays of Velodyne (sensors/Velodyne.h:97-120); cornerSharp is re-filtered by EdgeToLine upstream
            "rc", "scan_start", "scan_end", "range_image", "image_to_point_idx", "curvature", "state", "sort_ind", "left", "right", "cornerSharp")


def is_internal(fixture, key):
    """Golden arrays the reference API does not expose.  In line_extraction cornerSharp is the public, re-filtered cloud."""
    return key.endswith(INTERNAL) and fixture != "line_extraction"


FIXTURES = ("functors", "assoc_point2plane", "equirect", "lines", "neighbors", "fast_atan2", "reproj", "depth", "features", "line_extraction", "undistort")   # mvs.npz: the
# reference entry point (MVS::InitConfMap) is a private member driven by the whole MVS object — not exported here


def export(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    for fx in FIXTURES:
        z = np.load(os.path.join(GOLDEN, fx + ".npz"))
        write_pvv(os.path.join(out_dir, fx + ".in.pvv"), {k: z[k] for k in z.files if not k.startswith("__") and not _is_output(fx, k)})
        print("wrote", fx + ".in.pvv")


def compare(out_dir):
    bad = 0
    for fx in FIXTURES:
        path = os.path.join(out_dir, fx + ".ref.pvv")
        if not os.path.exists(path):
            print("%-20s MISSING (dump_reference_vectors did not write it)" % fx)
            bad += 1
            continue
        ref = read_pvv(path)
        z = np.load(os.path.join(GOLDEN, fx + ".npz"))
        if "__pinned__" in z.files:
            print("%-20s (fixture already rewritten from %s)" % (fx, str(z["__pinned__"][0])))
        for k in z.files:
            if k.startswith("__") or not _is_output(fx, k):
                continue
            if k not in ref:
                if is_internal(fx, k):
                    print("%-20s %-18s n/a (internal to the reference function, pinned through its outputs)" % (fx, k))
                else:
                    print("%-20s %-18s not produced" % (fx, k)); bad += 1
                continue
            exp, got = z[k], ref[k]
            if fx == "line_extraction" and k in ("segment_coeffs", "end_points") and exp.shape == got.shape:
                # FuseLines fits with pcl's SAC_RANSAC (sensors/LidarLineExtraction.cpp:148-175): a line through two of the segment's points, then
                # inlier-refined; the oracle takes the exhaustive 2-point maximum consensus.  Same line up to the fit's freedom: compare the
                # direction (sign-free) and the end points to a centimetre instead of bit for bit.
                if k == "segment_coeffs":
                    d0 = exp[:, 3:] / np.linalg.norm(exp[:, 3:], axis=1, keepdims=True); d1 = got[:, 3:] / np.linalg.norm(got[:, 3:], axis=1, keepdims=True)
                    ang = np.degrees(np.arccos(np.clip(np.abs((d0 * d1).sum(1)), 0, 1)))
                    off = np.linalg.norm(np.cross(got[:, :3] - exp[:, :3], d0), axis=1)
                    ok = bool((ang < 1.0).all() and (off < 0.02).all()); worst = "max angle %.3g deg, max offset %.3g m" % (ang.max(initial=0), off.max(initial=0))
                else:
                    d = np.linalg.norm(exp - got, axis=2).max(initial=0)
                    ok = bool(d < 0.02); worst = "max |d| %.3g m" % d
                print("%-20s %-18s RANSAC-tolerant %s (%s)" % (fx, k, "ok" if ok else "MISMATCH", worst))
                bad += 0 if ok else 1
                continue
            if exp.shape != got.shape:
                print("%-20s %-18s SHAPE %s vs reference %s" % (fx, k, exp.shape, got.shape)); bad += 1
                continue
            if fx == "undistort" and exp.dtype == np.float32:
                # double sines of two libms and, for a real build, Eigen's own quaternion kernels: north_star's floating-point tolerance, not bit equality
                scale = np.maximum(np.abs(exp[:, :3]).max(axis=1, keepdims=True), 1.0)
                ok = bool(np.all(np.abs(exp[:, :3] - got[:, :3]) <= 1e-6 * scale) and np.array_equal(exp[:, 3], got[:, 3]))
                print("%-20s %-18s 1e-6 %s (%d of %d points bit-identical)" % (fx, k, "ok" if ok else "MISMATCH", int(np.sum(np.all(exp.view(np.uint32) == got.astype(np.float32).view(np.uint32), axis=1))), len(exp)))
                bad += 0 if ok else 1
                continue
            if exp.dtype.kind in "iu" and exp.dtype.itemsize == 2:      # .pvv has no 16-bit type: images travel as int32
                ok = np.array_equal(exp.astype(np.int32), got.astype(np.int32)); how = "bit-exact"
            elif exp.dtype.kind in "iu" or exp.dtype == np.float32:
                ok = np.array_equal(exp, got.astype(exp.dtype)); how = "bit-exact"
            else:
                # north_star tolerance: 1e-6 relative (values of O(1)); Jacobians of the angle functors are
                # ill-conditioned as r -> 0, the GPU tests use the same conditioning-aware bound
                ok = np.allclose(exp, got, rtol=1e-6, atol=1e-9); how = "1e-6"
            print("%-20s %-18s %s %s" % (fx, k, how, "ok" if ok else "MISMATCH (max |d| %.3g)" % np.max(np.abs(exp.astype(np.float64) - got))))
            bad += 0 if ok else 1
    print("re-pin: %s" % ("ALL OUTPUTS AGREE with the reference build" if bad == 0 else "%d disagreement(s)" % bad))
    return bad


# fixture -> (rows of SURVEY.md section 8 it pins, the reference entry points dump_reference_vectors.cpp drives for it, how compare() judges it)
PINS = {
    "functors": ("A3 A4 A6 A9", "Point2Plane_Meter/_Angle, Point2Line_Meter/_Angle, Plane2Plane_Global, PlaneIOUResidual ::Create(...)->Evaluate (base/CostFunction.h:294-934, ceres AutoDiff)", "r, J: 1e-6 relative"),
    "reproj": ("A11", "PanoramaReprojResidual_1Angle::Create(...)->Evaluate (base/CostFunction.h:218-247)", "r, J: 1e-6 relative"),
    "assoc_point2plane": ("A2 (+ FLANN k-NN order, Eigen QR / eigen solver)", "AssociatePoint2Plane (lidar_mapping/LidarFeatureAssociate.cpp:550-630)", "records 1e-6 relative, count and order exact"),
    "lines": ("A5 N1", "AssociateLine2Line, FindAssociations (LidarFeatureAssociate.cpp:442-476, 120-197), LidarLineMatch::GenerateTracks", "matches exact"),
    "neighbors": ("A1", "FindNeighbors (LidarFeatureAssociate.cpp:19-111)", "lists exact"),
    "equirect": ("A7", "Equirectangular::CamToImage / ImageToCam / BreakToSegments (sensors/Equirectangular.h:42-204)", "float32 bit-exact, float64 1e-6"),
    "fast_atan2": ("A7", "FastAtan2<float>, FastAtan2<double> (base/Math.h:15-29) — ALREADY pinned in the development image (oracle/_ref)", "bit-exact"),
    "depth": ("N4 (depth prior)", "ProjectLidar2PanoramaDepth (util/Visualization.h:407-441)", "uint16 image bit-exact"),
    "features": ("N3", "Velodyne::ReOrderVLP + ExtractFeatures, ADAPTIVE (sensors/Velodyne.cpp:371-1189)", "clouds, picks bit-exact"),
    "undistort": ("N5 (motion compensation; outside §8)", "Velodyne::UndistortCloud (sensors/Velodyne.cpp:1642-1674), SlerpPose (base/Geometry.hpp:572-583) — Eigen's Quaternion(Matrix3), "
                  "slerp, q * v, Matrix4d::inverse", "points and poses 1e-6 relative (bit-identical count printed)"),
    "line_extraction": ("N3 (EdgeToLine, FuseLines)", "Velodyne::EdgeToLine / LidarLineExtraction (sensors/LidarLineExtraction.cpp:113-389)",
                        "segment membership exact; FuseLines coefficients RANSAC-tolerant (1 deg, 2 cm): PCL's seeded RANSAC vs the oracle's exhaustive 2-point consensus"),
}
NOT_PINNED = {
    "mvs / mvs_cloud": "N4: MVS::InitConfMap / ScorePixel / ProcessPixel are private members driven by the whole MVS object — no public entry point to export",
    "A8": "CameraLidarLineAssociate::AssociateByAngle is driven inside `lines` (c_votes are internal); its accepted pairs are compared through `lines`",
    "A10 (LM trajectory)": "ceres::Solve itself: `ceres_adapter_check` (tools/repin) runs integration/pvlm_ceres.hpp under the real Ceres on a machine with a GPU and "
                           "prints final cost / poses next to the in-repo driver's",
}


# behaviours of THIRD-PARTY code the path depends on that no fixture of the reference's API isolates: what they are pinned against today and what `make repin`
# re-runs on the reference machine
THIRD_PARTY = {
    "libstdc++ `std::sort`: the order it leaves EQUAL keys in (curvatures of a sector — sensors/Velodyne.cpp:896, :1110; voxel indices of pcl::VoxelGrid)": (
        "N3 (picks, voxel grid)", "`csrc/pvlm_stdsort.h` (introsort restated, host + device) against the toolchain's own `std::sort`: `tests/cpp/stdsort_check.cpp`, "
        "`tests/test_stdsort_cpu.py`, the load-time self-check and `pvlm_ring_debug_sort` on the GPU",
        "pinned against THIS image's libstdc++ (%s); the reference's Dockerfile builds on Ubuntu 23 (GCC 12 / 13): `make repin` compiles `stdsort_check.cpp` with the "
        "reference machine's `$(CXX)` and runs `tests/test_stdsort_cpu.py` against it — permutations element for element"),
}


def _toolchain():
    import subprocess
    try:
        return subprocess.run([os.environ.get("CXX", "g++"), "--version"], capture_output=True, text=True).stdout.splitlines()[0].strip()
    except Exception:
        return "g++ not found"


def table():
    print("| fixture (tests/golden) | pins SURVEY §8 row | reference entry point | judged |")
    print("|---|---|---|---|")
    for fx in FIXTURES:
        rows, entry, how = PINS[fx]
        print("| `%s.npz` | %s | %s | %s |" % (fx, rows, entry, how))
    for k, why in NOT_PINNED.items():
        print("| — | %s | not pinned by a fixture | %s |" % (k, why))
    for k, (rows, what, how) in THIRD_PARTY.items():
        print("| — (third party) | %s | %s — %s | %s |" % (rows, k, what, how % _toolchain()))
    return 0


def regenerate(out_dir, golden_out=None, tag="reference build"):
    """Rewrites the EXPECTED OUTPUTS of every fixture from the reference's outputs (<fixture>.ref.pvv); inputs and the arrays the reference API does
    not expose are kept.  Every rewritten fixture gets a `__pinned__` entry naming the reference build, so that a pinned fixture can be told from
    one the oracle produced."""
    golden_out = golden_out or GOLDEN
    os.makedirs(golden_out, exist_ok=True)
    done = 0
    for fx in FIXTURES:
        path = os.path.join(out_dir, fx + ".ref.pvv")
        if not os.path.exists(path):
            print("%-20s MISSING: left as the oracle produced it" % fx)
            continue
        ref = read_pvv(path)
        z = np.load(os.path.join(GOLDEN, fx + ".npz"))
        arrays, replaced, kept = {}, [], []
        for k in z.files:
            if k == "__pinned__":
                continue
            if _is_output(fx, k) and k in ref:
                arrays[k] = ref[k].astype(z[k].dtype); replaced.append(k)
            else:
                arrays[k] = z[k]
                if _is_output(fx, k):
                    kept.append(k)
        arrays["__pinned__"] = np.array([tag, ",".join(replaced)])
        np.savez_compressed(os.path.join(golden_out, fx + ".npz"), **arrays)
        print("%-20s %d arrays from the reference, %d internal ones kept (%s)" % (fx, len(replaced), len(kept), ", ".join(kept) or "-"))
        done += 1
    print("regenerated %d of %d fixtures under %s" % (done, len(FIXTURES), golden_out))
    return 0


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "table" and len(sys.argv) == 2:
        sys.exit(table())
    if cmd == "regenerate" and 3 <= len(sys.argv) <= 5:
        sys.exit(regenerate(*sys.argv[2:]))
    if len(sys.argv) != 3 or cmd not in ("export", "compare"):
        sys.exit(__doc__)
    sys.exit(export(sys.argv[2]) if cmd == "export" else compare(sys.argv[2]))
