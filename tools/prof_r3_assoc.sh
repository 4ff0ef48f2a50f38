# Round 3, association kernels (K2 min/max insertion network + row pruning, K3 certified collinearity test):
# GPU parity of the association, kernel trace + SQ counters of the 134 M-query workload, cell-size sweep.
#   gpurun --timeout 1500 -- 'bash tools/prof_r3_assoc.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assoc_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py tests/test_runtime_gpu.py -x -q -m gpu > $O/assoc_gpu_tests.txt 2>&1
tail -3 $O/assoc_gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
W="python $R/tools/assoc_workload.py --scans 256"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/assoc_trace -- $W > $O/assoc_trace.log 2>&1
tail -1 $O/assoc_trace.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $O/assoc_pmc_sq -- $W > $O/assoc_pmc_sq.log 2>&1
tail -1 $O/assoc_trace.log > $O/assoc_workload.json
cd $R && python tools/pmc_assoc.py $O/assoc_workload.json $O/r3_pmc_assoc_scans256.json $O/assoc_trace $O/assoc_pmc_sq
find $O/assoc_trace -name "*kernel_stats.csv" -exec cp {} $O/r3_assoc_kernel_stats_scans256.csv \;
# cell-size sweep around the heuristic (identical output by construction; wall time of the steady call)
for S in 0.4 0.5 0.65 0.8 1.0; do
  echo "PVLM_CELL_SCALE=$S" >> $O/r3_cell_scale_sweep.txt
  PVLM_CELL_SCALE=$S timeout 200 python $R/tools/assoc_workload.py --scans 128 --calls 4 2>&1 | tail -1 >> $O/r3_cell_scale_sweep.txt
done
cat $O/r3_cell_scale_sweep.txt
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
