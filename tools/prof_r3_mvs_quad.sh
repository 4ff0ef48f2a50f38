#!/bin/bash
# Round 3: SQ counters of the batched sequential sweep at 64 views per launch — four threads per pixel (default) against one wave per pixel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3quad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 16384 0; do
  export PVLM_MVS_QUAD_MIN=$M
  W="python $R/tools/mvs_batch_bench.py --views 64"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$M -- $W > $O/trace$M.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc$M -- $W > $O/pmc$M.log 2>&1
  (cd $R && python tools/pmc_kernels.py $O/r3_pmc_mvs_batch64_quadmin$M.json '{}' $O/trace$M $O/pmc$M k_mvs_propagate_diag | tail -45)
  grep '^{' $O/trace$M.log | tail -1
done
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
