"""ms per batch of raw scans through pvlm_ring_extract_batch (ReOrderVLP + Segmentation + adaptive curvature on the GPU), next to the
host mirror's time per scan on one core (panovlm_amd/host/pvlm_features.cpp through the test driver is not timed here: the oracle's
statement-by-statement loop is, as the CPU figure).  python tools/ring_bench.py [--scans 454] [--cols 1800] [--json out]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=454)
    ap.add_argument("--cols", type=int, default=1800)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--json", default="")
    ap.add_argument("--cpu", type=int, default=4, help="scans timed through the CPU oracle")
    a = ap.parse_args()
    import panovlm_amd as pv
    from panovlm_amd import synthetic as sy
    base = [sy.raw_vlp16_scan(k, cols=a.cols, clutter=40) for k in range(a.distinct)]
    raws = [base[k % a.distinct] for k in range(a.scans)]
    points = sum(len(r) for r in raws)
    ctx = pv.Context(0)
    rec = dict(scans=a.scans, cols=a.cols, points=points, runs=[])
    for rep in range(a.reps):
        t0 = time.perf_counter()
        b = pv.RingBatch(ctx, raws, n_rings=16, horizon=a.cols, segment=True)
        wall = time.perf_counter() - t0
        tm = b.timing()
        res = [b.result(k) for k in (0, a.scans - 1)]
        rec["runs"].append(dict(wall_ms=wall * 1e3, stage_ms=tm, device_ms=sum(v for k, v in tm.items() if k not in ("upload", "download")),
                                kept=[r["n_kept"] for r in res], resolved_points=sum(b.result(k)["resolved_points"] for k in range(a.scans)),
                                replayed=sum(b.result(k)["replayed"] for k in range(a.scans))))
        b.close()
    if a.cpu > 0:
        from oracle import oracle as orc
        orc.build()
        t0 = time.perf_counter()
        for k in range(a.cpu):
            orc.ScanFeatures(raws[k], n_scans=16, horizon=a.cols, segment=True, extract=True)
        rec["cpu_oracle_ms_per_scan_full_extraction"] = (time.perf_counter() - t0) * 1e3 / a.cpu
        t0 = time.perf_counter()
        for k in range(a.cpu):
            orc.ScanFeatures(raws[k], n_scans=16, horizon=a.cols, extract=False)
        rec["cpu_oracle_ms_per_scan_reorder_only"] = (time.perf_counter() - t0) * 1e3 / a.cpu
    best = min(rec["runs"], key=lambda r: r["wall_ms"])
    rec["best_wall_ms_per_batch"] = best["wall_ms"]; rec["best_device_ms_per_batch"] = min(r["device_ms"] for r in rec["runs"])
    rec["ms_per_scan_wall"] = best["wall_ms"] / a.scans
    print(json.dumps(rec, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
