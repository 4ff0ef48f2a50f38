#!/usr/bin/env python3
"""AddressSanitizer + UBSan sweep of everything that runs on the CPU (no GPU needed): the host mirror's file readers and
feature extractor (through a sanitized build of the test driver), and the HIP kernels' device bodies that compile for the
host (tests/cpp/mvs_math_check.cpp) on odd-sized images.  Prints what the sanitizers flag; exits non-zero if anything.
Run from the repo root:  python tools/sanitize_check.py"""
import ctypes as C
import glob
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-O1", "-g", "-std=c++17"]


def main():
    from panovlm_amd import synthetic as sy
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if os.environ.get("PVLM_SANITIZE_STAGE") == "bodies":          # re-executed below with libasan preloaded
        return bodies()
    flagged = 0
    with tempfile.TemporaryDirectory() as d:
        drv = os.path.join(d, "driver_asan")
        subprocess.check_call(["g++"] + SAN + ["-ffp-contract=off", os.path.join(ROOT, "tests/cpp/pvlm_host_driver.cpp")] +
                              sorted(glob.glob(os.path.join(ROOT, "panovlm_amd/host/*.cpp"))) +
                              [                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests/cpp/ceres_double"),
                               "-o", drv, "-L" + os.path.join(ROOT, "panovlm_amd"), "-lpvlm", "-pthread",
                               "-Wl,-rpath," + os.path.join(ROOT, "panovlm_amd")])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")

        def run(*args):
            r = subprocess.run([drv] + [str(a) for a in args], capture_output=True, timeout=600, env=env)
            bad = [l for l in r.stderr.decode(errors="ignore").splitlines() if "runtime error" in l or "AddressSanitizer" in l]
            if r.returncode != 0 or bad:
                print("FLAGGED", args[:2], r.returncode, bad[:3])
            return 1 if (r.returncode != 0 or bad) else 0
        rng = np.random.default_rng(5)
        dense = sy.raw_vlp16_scan(4); dense[:, :3] *= 0.05
        n = 3000
        az = rng.uniform(0, 2 * np.pi, n); el = np.deg2rad(rng.uniform(-17, 17, n)); r = rng.uniform(0.0, 10, n)
        xyz = np.stack([r * np.cos(el) * np.sin(az), -r * np.sin(el), r * np.cos(el) * np.cos(az)], 1).astype(np.float32)
        xyz[::50] = np.nan; xyz[7::60] = np.inf; xyz[3::70] = 0
        clouds = {"vlp": sy.raw_vlp16_scan(3, clutter=40), "dropout": sy.raw_vlp16_scan(7, start_deg=180.0, dropout=0.9), "tiny": sy.raw_vlp16_scan(9, cols=180)[:40],
                  "dense": dense.astype(np.float32), "hostile": np.concatenate([xyz, np.zeros((n, 1), np.float32)], 1), "empty": np.zeros((0, 4), np.float32)}
        for name, c in clouds.items():
            src = os.path.join(d, name + ".bin")
            with open(src, "wb") as f:
                f.write(struct.pack("<i", len(c))); f.write(np.ascontiguousarray(c, np.float32).tobytes())
            for ns, hz, seg in ((16, 1800, 1), (16, 90, 0), (32, 360, 1), (64, 1800, 1)):
                flagged += run("features", src, os.path.join(d, "o.bin"), ns, hz, 1000.0, 5.0, seg, 1, 1)   # last 1: EdgeToLine too (the line branch)
        hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\n"
               "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n")
        pts = (rng.normal(size=(5000, 4)) * 3).astype(np.float32)
        good = (hdr % (5000, 5000, "binary")).encode() + pts.tobytes()
        files = {"good": good, "trunc": good[:len(good) // 2], "huge": (hdr % (5000, 2 ** 31 - 1, "binary")).encode() + pts.tobytes(),
                 "cgarb": (hdr % (5000, 5000, "binary_compressed")).encode() + struct.pack("<II", 100, 80000) + bytes(rng.integers(0, 256, 100, dtype=np.uint8)),
                 "ascii": (hdr % (5000, 5000, "ascii")).encode() + b"1 2\n3 4 5 6\nfoo bar baz qux\n", "rand": bytes(rng.integers(0, 256, 4096, dtype=np.uint8))}
        for name, data in files.items():
            p = os.path.join(d, name + ".pcd")
            open(p, "wb").write(data)
            flagged += run("loadpcd", p)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PVLM_SANITIZE_STAGE="bodies", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0"),
                       capture_output=True, timeout=1800)
    bad = [l for l in r.stderr.decode(errors="ignore").splitlines() if "runtime error" in l or "AddressSanitizer" in l]
    print(r.stdout.decode().strip())
    if r.returncode != 0 or bad:
        print("FLAGGED device bodies", r.returncode, bad[:5]); flagged += 1
    print("sanitizer sweep: %d finding(s)" % flagged)
    return 1 if flagged else 0


def bodies():
    from oracle import oracle
    from tests.test_mvs_cpu import sweep_scene
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "libmvs_check_asan.so")
        subprocess.check_call(["g++"] + SAN + ["-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests/cpp/mvs_math_check.cpp")], env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
        lib = C.CDLL(so)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        for rows, cols in ((96, 192), (33, 47)):
            S = sweep_scene(oracle, rows, cols)
            ptrs = (C.POINTER(C.c_ubyte) * len(S["neis"]))(*[np.ascontiguousarray(g).ctypes.data_as(C.POINTER(C.c_ubyte)) for g in S["neis"]])
            R = np.ascontiguousarray(S["Rn"], np.float32); t = np.ascontiguousarray(S["tn"], np.float32)
            nd = [np.ascontiguousarray(x, np.float32) for x in S["nd"]]
            dptrs = (C.POINTER(C.c_float) * len(nd))(*[fp(x) for x in nd])
            for geo in (False, True):
                dd = S["depth"].copy(); n = S["normal"].copy(); c = S["conf"].copy()
                lib.chk_mvs_propagate(C.c_int(rows), C.c_int(cols), C.c_int(3), C.c_int(1), S["gray"].ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(len(nd)), ptrs, fp(R), fp(t),
                                      fp(dd), fp(n), fp(c), dptrs if geo else None, S["const"].ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_float(0.1), C.c_float(20.0),
                                      C.c_ulonglong(3), C.c_int(1), C.c_float(0.5))
                want = oracle.mvs_propagate(S["gray"], S["neis"], S["Rn"], S["tn"], S["depth"], S["normal"], S["conf"], nei_depths=S["nd"] if geo else None,
                                            depth_constant=S["const"], seed=3, max_iter=1, conf_threshold=0.5)
                print("device bodies %dx%d geometric=%s equal to the oracle: %s" % (rows, cols, geo, np.array_equal(dd, want[0]) and np.array_equal(c, want[2])))
        # the range-image bodies (csrc/pvlm_ring_core.h) through their host-compiled driver: stress, scaled, tiny, empty and hostile clouds,
        # the one-lane loop and the workgroup form
        so = os.path.join(d, "libring_check_asan.so")
        subprocess.check_call(["g++"] + SAN + ["-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests/cpp/ring_core_check.cpp")],
                              env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
        lib = C.CDLL(so); lib.chk_ring.restype = C.c_int
        from panovlm_amd import synthetic as sy
        from tests.test_ring_core_cpu import run as ring_run
        from tests import ring_cases
        rng = np.random.default_rng(2)
        dense = sy.raw_vlp16_scan(4); dense[:, :3] *= 0.02
        wild = (rng.normal(size=(4000, 4)) * 5).astype(np.float32); wild[::40, :3] = 0
        ok = True
        for name, raw, rings, hz in (("vlp", sy.raw_vlp16_scan(3, clutter=40), 16, 1800), ("dense", dense, 16, 1800), ("wild", wild, 32, 360), ("wild64", wild, 64, 90),
                                     ("tiny", sy.raw_vlp16_scan(9, cols=180)[:40], 16, 180), ("one", sy.raw_vlp16_scan(9, cols=180)[:1], 16, 180)):
            for threads in (0, 64):
                for seg in (True, False):
                    g = ring_run(lib, raw, rings, hz, seg, threads=threads)
                    try:
                        ring_cases.assert_matches_oracle(oracle, raw, rings, hz, seg, g)
                    except AssertionError:
                        ok = False; print("ring bodies differ from the oracle:", name, threads, seg)
        print("range-image device bodies under the sanitizers equal to the oracle: %s" % ok)
        # the std::sort restatement (csrc/pvlm_stdsort.h, serial and level-by-level forms) on tie-heavy, structured and depth-limit-forcing keys
        so = os.path.join(d, "libstdsort_check_asan.so")
        subprocess.check_call(["g++"] + SAN + ["-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests/cpp/stdsort_check.cpp")], env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
        lib = C.CDLL(so)
        ok = True
        for trial in range(400):
            n = int(rng.integers(0, 3000))
            key = (rng.integers(0, max(2, n // int(rng.integers(1, 30)) + 1), n) if trial % 3 else np.arange(n) % 7).astype(np.float32)
            ok = ok and lib.chk_sort_by_float(fp(key), n, None) == 0 and lib.chk_sort_by_levels(fp(key), n) == 0
        killer = np.zeros(4000, np.int32)
        lib.chk_killer_keys(4000, killer.ctypes.data_as(C.POINTER(C.c_int)))
        for coarse in (1, 3):
            key = (killer // coarse).astype(np.float32)
            ok = ok and lib.chk_sort_by_float(fp(key), len(key), None) == 0 and lib.chk_sort_by_levels(fp(key), len(key)) == 0
        print("std::sort restatement under the sanitizers equal to std::sort: %s" % ok)
    return 0


if __name__ == "__main__":
    sys.exit(main())
