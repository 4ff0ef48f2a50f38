# Round 3, association, second pass: x-refined cells + closed-form collinearity screen.  GPU parity, then A/B of the occupancy
# variants and a (cell scale x x-refinement) sweep on the 67 M-query workload (identical output by construction).
#   python -m panovlm_amd.build --variant k2w5 -DPVLM_K2_WAVES=5; python -m panovlm_amd.build --variant k3w3 -DPVLM_K3_WAVES=3
#   gpurun --timeout 1200 -- 'bash tools/prof_r3_assoc2.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_assoc_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > $O/assoc2_gpu_tests.txt 2>&1
tail -3 $O/assoc2_gpu_tests.txt
W="python $R/tools/assoc_workload.py --scans 128 --calls 5"
: > $O/r3_assoc_variants.txt
for V in base k2w5 k3w3; do
  L=$R/panovlm_amd/libpvlm.so; [ $V != base ] && L=$R/build/var/libpvlm_$V.so
  echo "variant $V" >> $O/r3_assoc_variants.txt
  PVLM_LIB=$L timeout 200 $W 2>&1 | grep '^{' | tail -1 >> $O/r3_assoc_variants.txt
done
for XF in 1 4 8; do for S in 0.5 0.65 0.8 1.0; do
  echo "PVLM_CELL_XF=$XF PVLM_CELL_SCALE=$S" >> $O/r3_assoc_variants.txt
  PVLM_CELL_XF=$XF PVLM_CELL_SCALE=$S timeout 200 $W 2>&1 | grep '^{' | tail -1 >> $O/r3_assoc_variants.txt
done; done
cat $O/r3_assoc_variants.txt
cd /tmp && export TMPDIR=/tmp
W2="python $R/tools/assoc_workload.py --scans 256"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/assoc2_trace -- $W2 > $O/assoc2_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $O/assoc2_pmc_sq -- $W2 > $O/assoc2_pmc_sq.log 2>&1
grep '^{' $O/assoc2_trace.log | tail -1 > $O/assoc2_workload.json
cd $R && python tools/pmc_assoc.py $O/assoc2_workload.json $O/r3_pmc_assoc2_scans256.json $O/assoc2_trace $O/assoc2_pmc_sq | head -60
find $O/assoc2_trace -name "*kernel_stats.csv" -exec cp {} $O/r3_assoc2_kernel_stats_scans256.csv \;
head -5 $O/r3_assoc2_kernel_stats_scans256.csv | cut -c1-160
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
