#!/bin/bash
# SQ counters of the association kernels on tools/assoc_workload.py (two passes of 8 SQ counters; no other trace domain beside --kernel-trace)
#   TARGETS="voxel raw" tools/pmc_k2.sh  ->  gpurun_out/pmc_k2_<targets>.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_k2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp

for TG in ${TARGETS:-voxel raw}; do
  SC=256; [ $TG = raw ] && SC=32
  W="python $R/tools/assoc_workload.py --scans $SC --targets $TG"
  rm -rf $O/${TG}_*
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TG}_trace -- $W > $O/${TG}_trace.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $O/${TG}_sq1 -- $W > $O/${TG}_sq1.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM --output-format csv -d $O/${TG}_sq2 -- $W > $O/${TG}_sq2.log 2>&1
  grep '^{' $O/${TG}_trace.log | tail -1 > $O/${TG}_workload.json
  (cd $R && python tools/pmc_assoc.py $O/${TG}_workload.json $R/gpurun_out/pmc_k2_${TG}.json $O/${TG}_trace $O/${TG}_sq1 $O/${TG}_sq2 | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items()})")
  find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
done
