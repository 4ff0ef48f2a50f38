#!/usr/bin/env python3
"""Summarises one rocprofv3 --pmc pass (SQ counters) + one --kernel-trace pass of any workload into per-kernel figures:
usage: pmc_kernels.py <out_json> <units_per_dispatch_json> <trace_dir> <pmc_dir> [kernel substrings ...]
  valu_issue_frac   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      share of a resident wave's cycles spent issuing VALU
  wait_frac         = SQ_WAIT_ANY / SQ_WAVE_CYCLES
  valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES
  simd_valu_util    = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz)   a LOWER bound of the VALU pipe's
                      utilisation: a wave64 instruction occupies a SIMD16 for 4 cycles, an fp64 / transcendental one longer
<units_per_dispatch_json>: {"kernel substring": units per dispatch} -> valu instructions per unit (pixel)."""
import collections, csv, glob, json, sys

out, units = sys.argv[1], json.loads(sys.argv[2])
trace_dir, pmc_dir = sys.argv[3], sys.argv[4]
want = sys.argv[5:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ndisp = collections.Counter(); dur = collections.defaultdict(list)
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = name(r)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); ndisp[k] += 1
for f in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
res = {}
for k in sorted(agg):
    if want and not any(w in k for w in want):
        continue
    c = agg[k]; n = max(ndisp[k], 1)
    e = {"dispatches": ndisp[k]}
    if c.get("SQ_WAVE_CYCLES", 0) > 0:
        e["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
        e["wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        e["any_issue_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_WAVES", 0) > 0:
        e["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        e["lds_insts_per_wave"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_WAVES"]
        e["vmem_rd_insts_per_wave"] = c.get("SQ_INSTS_VMEM_RD", 0.0) / c["SQ_WAVES"]
    if dur.get(k):
        ms = sum(dur[k]) / len(dur[k])
        e["kernel_trace_ms"] = ms
        e["simd_valu_util_lower_bound"] = (c["SQ_INSTS_VALU"] / n) * 4.0 / (1024 * ms * 1e-3 * 2.4e9)
    for sub, u in units.items():
        if sub in k:
            e["valu_wave_insts_per_unit"] = (c["SQ_INSTS_VALU"] / n) / u
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
