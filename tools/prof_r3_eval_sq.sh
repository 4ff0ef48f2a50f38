#!/bin/bash
# Round 3: SQ counters of the fused evaluation kernels after the instruction diet (separate --pmc pass of the default bench command)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3evalsq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="python $R/bench.py --no-cpu-baseline --no-mvs --no-projection --steps 10"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $W > $O/trace.log 2>&1
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc -- $W > $O/pmc.log 2>&1
cd $R && python tools/pmc_kernels.py $O/r3_pmc_eval_sq.json '{}' $O/trace $O/pmc "k_eval_fused" | tail -40
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
