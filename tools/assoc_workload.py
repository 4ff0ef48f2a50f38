#!/usr/bin/env python3
"""Association only (pvlm_assoc_point2plane over the synthetic batch of bench.py), for rocprofv3 passes:
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python tools/assoc_workload.py --scans 256
Prints one JSON line with the query / target / accepted counts the summariser (tools/pmc_assoc.py) normalises by."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (scan generation helper)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=256)
    ap.add_argument("--neighbors", type=int, default=8)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--targets", choices=["voxel", "raw"], default="voxel")
    ap.add_argument("--tolerance", type=float, default=0.05)
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--exact", action="store_true", help="PVLM_FLAG_ASSOC_EXACT_FIT: the reference's QR for every query")
    a = ap.parse_args()
    from panovlm_amd import synthetic as sy
    ref, nei = sy.pair_list(a.scans, a.neighbors)
    scans = bench.generate_scans(range(a.scans), a.cols, 0.2 if a.targets == "voxel" else 0.0)
    import panovlm_amd as pv
    ctx = pv.Context(0)
    keys = sorted(scans)
    ds = dict(zip(keys, pv.Scan.upload_batch(ctx, [scans[k] for k in keys])))
    nq = int(sum(len(scans[int(n)]["flat_xyz"]) for n in nei)); nt = int(sum(len(scans[int(r)]["less_xyz"]) for r in ref))
    walls, acc = [], 0
    for _ in range(a.calls):
        ctx.synchronize()
        t0 = time.perf_counter()
        rs = ctx.assoc_point2plane([ds[int(r)] for r in ref], [ds[int(n)] for n in nei], a.tolerance, 1.0, kind=pv.POINT2PLANE_ANGLE,
                                   flags=pv.FLAG_NORMALIZE_DISTANCE | (pv.FLAG_ASSOC_EXACT_FIT if a.exact else 0))
        ctx.synchronize()
        walls.append(time.perf_counter() - t0)
        acc = rs.n
        exact_fits = rs.assoc_exact_fits(); stats = rs.assoc_stats()
        rs.close()
    print(json.dumps({"scans": a.scans, "pairs": int(len(ref)), "queries": nq, "targets": nt, "accepted": int(acc), "calls": a.calls, "targets_kind": a.targets, "exact_mode": bool(a.exact), "queries_refused_by_the_fast_fit": exact_fits, "batches": stats["batches"], "batches_on_the_exact_kernel": stats["exact_kernel_batches"],
                      "wall_s": walls}))


if __name__ == "__main__":
    main()
