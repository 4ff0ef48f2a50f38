#!/usr/bin/env python3
"""Host timing of the LiDAR feature extraction in front of the hot path (planar branch): ReOrderVLP +
ExtractFeatures per raw VLP-16 scan (16 x 1800 firing-order returns), the host mirror (panovlm_amd/host/pvlm_features.cpp,
one thread) beside the CPU oracle (oracle/features.hpp, one thread).  No GPU involved — this stage is host code upstream and here."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--clutter", type=int, default=30, help="boxes standing in the room: more edges, more line segments")
    a = ap.parse_args()
    from oracle import oracle as orc     # the CPU leg of a measurement tool, not the product
    from panovlm_amd import synthetic as sy
    from tests import host_io
    scans = []
    for k in range(a.scans):
        R, t = sy.estimated_pose(k)
        scans.append(dict(id=k, R_wl=R, t_wl=t, raw=sy.raw_vlp16_scan(k, clutter=a.clutter)))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "raw.bin")
        host_io.write_raw_scans(path, scans)
        line = [l for l in host_io.run("featbench", path, a.reps, 1) if l.startswith("featbench")][0].split()
    host = {line[i]: float(line[i + 1]) for i in range(1, len(line), 2)}
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for s in scans:
            orc.lib().orc_features_free(orc._features_handle(s["raw"]))
    oracle_ms = 1e3 * (time.perf_counter() - t0) / (a.reps * a.scans)
    print(json.dumps(dict(scans=a.scans, reps=a.reps, points_per_scan=host["points_per_scan"], host_reorder_ms=host["reorder_ms"], host_extract_ms=host["extract_ms"],
                          host_lines_ms=host["lines_ms"], host_ms_per_scan=host["reorder_ms"] + host["extract_ms"] + host["lines_ms"], oracle_ms_per_scan=oracle_ms,
                          surf_flat=host["flat"], surf_less_flat=host["less_flat"], line_segments=host["segments"], corner_points=host["corner"],
                          room_454_scans_one_thread_s=454e-3 * (host["reorder_ms"] + host["extract_ms"] + host["lines_ms"]))))


if __name__ == "__main__":
    main()
