#!/usr/bin/env python3
"""Groups the dispatches of a rocprofv3 --kernel-trace CSV by (kernel, grid size) — bench.py launches the fused kernel on
the full batch (the timed steps), on rank-0 shards (per_rank_projection) and on the Floor subset, and rocprofv3's own
--stats averages them together.  usage: trace_groups.py <kernel_trace.csv> <out.csv> [name filter ...]"""
import collections
import csv
import sys

src, out = sys.argv[1], sys.argv[2]
filt = sys.argv[3:] or ["k_eval_fused", "k_eval_materialise", "k_knn_pairs", "k_fit_pairs", "k_compact", "k_pair_epilogue", "k_neq_gather", "k_pair_table", "k_pose_table"]
g = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not any(f in n for f in filt):
        continue
    g[(n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Workgroups_X", "Grid_Y", "Calls", "AverageNs", "MinNs", "MaxNs", "AverageNs_last_half"])
    for (n, gx, gy), d in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        h = d[len(d) // 2:]
        w.writerow([n, gx, gy, len(d), "%.1f" % (sum(d) / len(d)), min(d), max(d), "%.1f" % (sum(h) / len(h))])
print(open(out).read()[:3000])
