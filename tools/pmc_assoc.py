#!/usr/bin/env python3
"""Summarises rocprofv3 passes over tools/assoc_workload.py into profiles/r2_pmc_assoc_*.json.
usage: pmc_assoc.py <workload_json_line_file> <out_json> <pass_dir> [<pass_dir> ...]
Every pass directory holds one rocprofv3 --pmc run (counter_collection.csv) and/or a --kernel-trace run.
Per kernel (k_knn_pairs, k_fit_pairs, k_compact): counters summed over the dispatches of ONE association call
(total / calls), normalised per query; derived figures:
  valu_issue_frac      = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of resident-wave cycles spent issuing VALU)
  wait_frac            = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  valu_insts_per_query = SQ_INSTS_VALU / (queries / 64)   (SQ_INSTS_VALU counts wave instructions; one lane = one query, so an
                         instruction issued by a wave is one instruction of each of its 64 queries)
  fetch bytes          = FETCH_SIZE KiB x 1024 (raw) and x 2 (the guide's correction for 16 B/lane streams; K2's
                         candidate loads are 16 B/lane but divergent, so the truth lies between the two)
  compulsory ratio     = fetch bytes / ((Nq + Nt) x 16 B)
  simd_valu_util_lower_bound = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz): the roof of these kernels."""
import collections
import csv
import glob
import json
import sys

wl = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
out = sys.argv[2]
calls = wl["calls"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for d in sys.argv[3:]:
    import os
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:      # one run per directory: the newest
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]] += 1
    for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1:]:
        if glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            continue        # durations under counter collection are not representative
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
comp = (wl["queries"] + wl["targets"]) * 16
res = {"workload": wl, "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python tools/assoc_workload.py --scans %d (one pass per counter set)" % wl["scans"],
       "queries": wl["queries"], "compulsory_bytes_per_call": comp, "kernels": {}}
for k in sorted(agg):
    if not any(x in k for x in ("k_knn_", "k_fit_pairs", "k_compact")):
        continue
    c = {n: v / calls for n, v in agg[k].items()}
    e = {"dispatches_per_call": {n: disp[k][n] / calls for n in disp[k]}, "per_call": c}
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        for name, key in (("valu_issue_frac", "SQ_ACTIVE_INST_VALU"), ("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"),
                          ("any_issue_frac", "SQ_ACTIVE_INST_ANY")):
            if key in c:
                e[name] = c[key] / c["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES", 0) > 0:
        # per 64 queries (one wave's worth): k_knn_pairs runs one wave per 64 queries, the persistent k_fit_pairs does not — normalise by the queries
        q64 = wl["queries"] / 64.0
        e["valu_insts_per_query"] = c["SQ_INSTS_VALU"] / q64
        e["vmem_rd_insts_per_query"] = c.get("SQ_INSTS_VMEM_RD", 0.0) / q64
        e["waves_launched_per_call"] = c["SQ_WAVES"]
        e["waves_per_call"] = c["SQ_WAVES"]
    for extra in ("SQ_INSTS_SMEM", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        if extra in c and c.get("SQ_WAVES", 0) > 0:
            e[extra.lower() + "_per_wave"] = c[extra] / c["SQ_WAVES"]
    if "FETCH_SIZE" in c:
        e["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024; e["fetch_bytes_x2"] = 2 * c["FETCH_SIZE"] * 1024
        e["fetch_over_compulsory_raw"] = e["fetch_bytes_raw"] / comp; e["fetch_over_compulsory_x2"] = e["fetch_bytes_x2"] / comp
        e["fetch_bytes_per_query_raw"] = e["fetch_bytes_raw"] / wl["queries"]
    if "WRITE_SIZE" in c:
        e["write_bytes"] = c["WRITE_SIZE"] * 1024; e["write_bytes_per_query"] = e["write_bytes"] / wl["queries"]
    if k in dur and dur[k]:
        e["kernel_trace_ms_per_call"] = sum(dur[k]) / calls
        if "SQ_INSTS_VALU" in c:
            # share of the chip's VALU issue capacity the kernel uses: a wave64 instruction occupies its SIMD16 for 4 cycles
            # (fp64 and transcendental ones longer, so this is a LOWER bound), 1024 SIMDs at 2.4 GHz
            e["simd_valu_util_lower_bound"] = c["SQ_INSTS_VALU"] * 4.0 / (1024 * e["kernel_trace_ms_per_call"] * 1e-3 * 2.4e9)
    res["kernels"][k] = e
for k in sorted(dur):
    if k not in res["kernels"] and any(x in k for x in ("k_knn_", "k_fit_pairs", "k_compact")):
        res["kernels"][k] = {"kernel_trace_ms_per_call": sum(dur[k]) / calls}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {n: v for n, v in e.items() if n not in ("per_call", "dispatches_per_call")} for k, e in res["kernels"].items()}, indent=1))
