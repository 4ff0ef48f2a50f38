#!/bin/bash
# Association kernels on tools/assoc_workload.py under rocprofv3 --kernel-trace --stats: us per dispatch of k_knn_pairs and k_fit_pairs, accepted
# rows and the call's wall time; PVLM_LIB selects a variant library (python -m panovlm_amd.build --variant <tag> -D...).
#   TARGETS="voxel raw" tools/ab_k2.sh   ->  gpurun_out/ab_k2.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/ab_k2.txt; : > $out
for tg in ${TARGETS:-voxel raw}; do
  scans=256; [ $tg = raw ] && scans=32
  rm -rf /tmp/prof_ab
  PVLM_LIB=${PVLM_LIB:-$PWD/panovlm_amd/libpvlm.so} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/assoc_workload.py --scans $scans --targets $tg > /tmp/ab.log 2>&1
  f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
  python - "$f" $tg "$(grep '^{' /tmp/ab.log | tail -1)" >> $out <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pick = lambda n: next((float(r["AverageNs"]) / 1e3 for r in rows if n in r["Name"]), 0.0)
w = json.loads(sys.argv[3])
print("%-6s knn_us %8.1f fit_us %8.1f  accepted %d wall_ms %.2f" % (sys.argv[2], pick("k_knn_pairs"), pick("k_fit_pairs"), w["accepted"], 1e3 * min(w["wall_s"])))
PY
done
cat $out
