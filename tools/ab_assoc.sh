#!/bin/bash
# A/B of association-kernel variants (build/var/libpvlm_<tag>.so from `python -m panovlm_amd.build --variant <tag> -D...`, selected through PVLM_LIB;
# "base" = the in-tree library): per variant the rocprofv3 kernel-trace average of k_knn_pairs / k_fit_pairs / k_compact on tools/assoc_workload.py.
# VARIANTS="base batch1 ..." TARGETS="voxel raw" tools/ab_assoc.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/ab_assoc.txt; : > $out
for tg in ${TARGETS:-voxel raw}; do
  scans=256; [ $tg = raw ] && scans=32
  for v in ${VARIANTS:-base}; do
   for sc in ${SCALES:-1.0}; do
    lib=panovlm_amd/libpvlm.so; [ $v != base ] && lib=build/var/libpvlm_$v.so
    rm -rf /tmp/prof_ab
    PVLM_CELL_SCALE=$sc PVLM_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/assoc_workload.py --scans $scans --targets $tg > /tmp/ab.log 2>&1
    f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
    python - "$f" $tg $v@$sc >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pick = lambda n: next((float(r["AverageNs"]) / 1e3 for r in rows if n in r["Name"]), float("nan"))
print("%-6s %-14s knn_us %9.1f fit_us %9.1f compact_us %8.1f" % (sys.argv[2], sys.argv[3], pick("k_knn_pairs"), pick("k_fit_pairs"), pick("k_compact")))
PY
   done
  done
done
cat $out
