set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W="python $R/tools/assoc_workload.py --scans 256"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/assoc_trace -- $W > $O/assoc_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $O/assoc_pmc_sq -- $W > $O/assoc_pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/assoc_pmc_fetch -- $W > $O/assoc_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/assoc_pmc_write -- $W > $O/assoc_pmc_write.log 2>&1
tail -1 $O/assoc_trace.log > $O/assoc_workload.json
cd $R && python tools/pmc_assoc.py $O/assoc_workload.json $O/r2_pmc_assoc_scans256.json $O/assoc_trace $O/assoc_pmc_sq $O/assoc_pmc_fetch $O/assoc_pmc_write
# the default bench under the kernel trace (roofline cross-check)
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
tail -1 $O/bench_under_rocprof.log | head -c 400
find $O/bench_trace -name "*kernel_stats.csv" | head
# drop the big per-dispatch traces before the merge (64 MiB cap)
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
