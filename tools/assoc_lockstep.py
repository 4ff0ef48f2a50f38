#!/usr/bin/env python3
"""Lane efficiency of the thread-per-query search K2 (CPU only): the per-query bodies of csrc/pvlm_assoc_core.h compiled for the host
(tests/cpp/assoc_core_check.cpp, chk_knn_lockstep) on the bench's scans; a wave of 64 consecutive queries walks the (r, dz, dy) loop in
lockstep and pays the longest run of any lane in every iteration.  Writes profiles/r5_assoc_lockstep.json:
    candidates_per_query            what a lane needs
    network_executions_per_query    what the wave executes (sum over iterations of the longest run)
    lane_efficiency                 the ratio."""
import ctypes, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panovlm_amd import synthetic as sy


def main():
    so = os.path.join(tempfile.mkdtemp(), "assoc_core_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas"] + [a for a in sys.argv[1:] if a.startswith("-D")] +
                          ["-o", so, os.path.join(ROOT, "tests", "cpp", "assoc_core_check.cpp")])
    lib = ctypes.CDLL(so)
    lib.chk_knn_lockstep.restype = ctypes.c_longlong
    P = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    out = {"tool": "tools/assoc_lockstep.py", "wave": 64, "scans": "panovlm_amd.synthetic.make_scan(5 / 6, cols=4096): queries = scan 6, targets = scan 5"}
    for kind, vox in (("voxel", 0.2), ("raw", 0.0)):
        t = sy.make_scan(5, cols=4096, downsample_targets=vox); s = sy.make_scan(6, cols=4096, downsample_targets=vox)
        tg = np.ascontiguousarray(t["less_xyz"], np.float32); q = np.ascontiguousarray(s["flat_xyz"][:16384], np.float32)
        cap = 40 * len(q)
        rec = np.zeros((cap, 3), np.int32); off = np.zeros(len(q) + 1, np.int64)
        n = lib.chk_knn_lockstep(P(tg, ctypes.c_float), len(tg), P(q, ctypes.c_float), len(q), ctypes.c_float(1.0), P(rec, ctypes.c_int), ctypes.c_longlong(cap), P(off, ctypes.c_longlong))
        assert n <= cap
        cand = exe = waves = flat = flat_rows = rows = 0
        for w in range(0, len(q), 64):
            waves += 1
            longest = {}
            lane_total = lane_rows = 0
            for i in range(w, min(w + 64, len(q))):
                r = rec[off[i]:off[i + 1]]
                cand += int(r[:, 1].sum())
                rows += len(r)
                lane_total = max(lane_total, int(r[:, 1].sum()))
                lane_rows = max(lane_rows, int(r[:, 1].sum()) + len(r))
                for it, ln in zip(r[:, 0].tolist(), r[:, 1].tolist()):
                    if ln > longest.get(it, 0):
                        longest[it] = ln
            exe += sum(longest.values())
            flat += lane_total; flat_rows += lane_rows
        out[kind] = {"targets": int(len(tg)), "queries": int(len(q)), "candidates_per_query": cand / len(q), "network_executions_per_query": exe / waves,
                     "lane_efficiency": (cand / len(q)) / max(exe / waves, 1e-9),
                     # a single loop in which every lane walks its OWN runs (row iterator per lane): the wave pays the busiest lane's total
                     "runs_per_query": rows / len(q), "flattened_executions_per_query": flat / waves, "flattened_plus_one_step_per_run": flat_rows / waves}
    if not any(a.startswith("-D") for a in sys.argv[1:]):
        json.dump(out, open(os.path.join(ROOT, "profiles", "r5_assoc_lockstep.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
