#!/bin/bash
# A/B of the k-NN stage on tools/assoc_workload.py: K2g (k_knn_groups: pilots searched by the whole wave, every lane a follower) against the
# thread-per-query search K2 (k_knn_pairs), same library, selected through PVLM_K2_GROUP (16 / 0; auto = the host's own choice);
# rocprofv3 --kernel-trace --stats, us per dispatch.   TARGETS="voxel raw" KGROUPS="auto 16 0" tools/ab_k2.sh   ->  gpurun_out/ab_k2.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/ab_k2.txt; : > $out
for tg in ${TARGETS:-voxel raw}; do
  scans=256; [ $tg = raw ] && scans=32
  for v in ${KGROUPS:-auto 16 0}; do
    rm -rf /tmp/prof_ab
    if [ $v = auto ]; then unset PVLM_K2_GROUP; else export PVLM_K2_GROUP=$v; fi
    PVLM_LIB=${PVLM_LIB:-$PWD/panovlm_amd/libpvlm.so} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python tools/assoc_workload.py --scans $scans --targets $tg > /tmp/ab.log 2>&1
    f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
    python - "$f" $tg $v "$(grep '^{' /tmp/ab.log | tail -1)" >> $out <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pick = lambda n: next((float(r["AverageNs"]) / 1e3 for r in rows if r["Name"].startswith(n)), 0.0)
w = json.loads(sys.argv[4])
k = {n: pick(n) for n in ("k_knn_pairs", "k_knn_groups", "k_fit_pairs", "k_compact")}
print("%-6s group %-5s knn_us %8.1f (K2 k_knn_pairs %.1f, K2g k_knn_groups %.1f) fit_us %8.1f  accepted %d wall_ms %.2f" % (
    sys.argv[2], sys.argv[3], k["k_knn_pairs"] + k["k_knn_groups"], k["k_knn_pairs"], k["k_knn_groups"], k["k_fit_pairs"], w["accepted"], 1e3 * min(w["wall_s"])))
PY
  done
done
cat $out
