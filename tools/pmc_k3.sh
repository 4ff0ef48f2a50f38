#!/bin/bash
# SQ counters + kernel trace of the association kernels on tools/assoc_workload.py for one or more library variants (build/var/libpvlm_<tag>.so, "base" = in-tree),
# summarised by tools/pmc_assoc.py into gpurun_out/pmc_k3/<tag>_<targets>.json.   VARIANTS="base w2" TARGETS="voxel raw" EXTRA="--exact" bash tools/pmc_k3.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_k3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD"
for TG in ${TARGETS:-voxel}; do
  SC=256; [ $TG = raw ] && SC=32
  for v in ${VARIANTS:-base}; do
    lib=$R/panovlm_amd/libpvlm.so; [ $v != base ] && lib=$R/build/var/libpvlm_$v.so
    tag=${v}${EXTRA:+_exact}_$TG
    W2="python $R/tools/assoc_workload.py --scans $SC --targets $TG $EXTRA"
    rm -rf $O/${tag}_trace $O/${tag}_sq
    PVLM_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- $W2 > $O/${tag}_trace.log 2>&1
    PVLM_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${tag}_sq -- $W2 > $O/${tag}_sq.log 2>&1
    grep '^{' $O/${tag}_trace.log | tail -1 > $O/${tag}_workload.json
    (cd $R && python tools/pmc_assoc.py $O/${tag}_workload.json $O/${tag}.json $O/${tag}_trace $O/${tag}_sq > /dev/null)
    cp $(find $O/${tag}_trace -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
    python - $O/${tag}.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d["kernels"].items():
    print("%-22s %-40s valu/query %7.1f  valu_issue %.3f  wait %.3f  simd_busy_lb %s  ms_per_call %s" % (sys.argv[2], k[:40], e.get("valu_insts_per_query", float("nan")), e.get("valu_issue_frac", float("nan")),
          e.get("wait_frac", float("nan")), e.get("simd_valu_util_lower_bound"), e.get("kernel_trace_ms_per_call")))
PY
    rm -rf $O/${tag}_trace $O/${tag}_sq
  done
done
