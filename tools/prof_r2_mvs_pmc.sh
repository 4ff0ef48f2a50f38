#!/bin/bash
# SQ counters of the MVS kernels and the equirect maps (separate passes: kernel trace / PMC)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="python $R/tools/mvs_bench.py"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/mvs_trace -- $W > $O/mvs_trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/mvs_pmc -- $W > $O/mvs_pmc.log 2>&1
cd $R && python tools/pmc_kernels.py $O/r2_pmc_mvs.json '{"k_mvs_conf": 1036800, "k_mvs_propagate": 518400}' $O/mvs_trace $O/mvs_pmc k_mvs_conf k_mvs_propagate k_mvs_refine k_mvs_project | tail -60
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
