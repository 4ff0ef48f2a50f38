#!/bin/bash
# end-of-round evidence (round 3): GPU suite, fused-kernel HBM traffic (FETCH_SIZE pass), association counters, default bench (+ the same
# command under rocprofv3 --kernel-trace --stats), Room- / Floor-scale runs.   PVLM_COMMIT=<short hash> gpurun ... 'bash tools/prof_r3_final.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/r3_gpu_tests.txt; cat $O/r3_gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
# HBM read traffic of the fused kernel: FETCH_SIZE in its own pass (the WRITE_SIZE pass hung on this pool in round 2 and is skipped)
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python $R/bench.py --no-cpu-baseline --no-mvs --no-projection --steps 10 > $O/bench_fetch.log 2>&1
grep '^{' $O/bench_fetch.log | tail -1 > $O/bench_fetch.json
cd $R && python tools/pmc_traffic.py $O/bench_fetch - $O/bench_fetch.json $O/r3_pmc_traffic_default.json > /dev/null && cp $O/r3_pmc_traffic_default.json $R/profiles/r3_pmc_traffic_default.json   # bench.py quotes it
# association counters, final configuration
cd /tmp
W2="python $R/tools/assoc_workload.py --scans 256"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/assoc_trace -- $W2 > $O/assoc_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $O/assoc_pmc_sq -- $W2 > $O/assoc_pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/assoc_pmc_fetch -- $W2 > $O/assoc_pmc_fetch.log 2>&1
grep '^{' $O/assoc_trace.log | tail -1 > $O/assoc_workload.json
cd $R && python tools/pmc_assoc.py $O/assoc_workload.json $O/r3_pmc_assoc_scans256.json $O/assoc_trace $O/assoc_pmc_sq $O/assoc_pmc_fetch > /dev/null && cp $O/r3_pmc_assoc_scans256.json $R/profiles/r3_pmc_assoc_scans256.json
cp $(find $O/assoc_trace -name "*kernel_stats.csv" | head -1) $O/r3_assoc_kernel_stats_scans256.csv
# the default bench, then the same command under the kernel trace
timeout 900 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err; tail -c 300 $O/r3_bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python $R/bench.py --no-cpu-baseline --no-mvs > $O/bench_under_rocprof.log 2>&1
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/r3_bench_under_rocprof.json
cd $R && python tools/trace_groups.py $(find $O/bench_trace -name "*kernel_trace.csv" | head -1) $O/r3_kernel_groups_default.csv > /dev/null
cp $(find $O/bench_trace -name "*kernel_stats.csv" | head -1) $O/r3_kernel_stats_default.csv
# MVS: the bench tool + SQ counters of the image-space kernels (thread-per-pixel colour pass / scoring pass, wave-per-pixel diagonals)
timeout 600 python tools/mvs_bench.py 2> $O/mvs_bench.err | tail -1 > $O/r3_mvs_bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/mvs_trace -- python $R/tools/mvs_bench.py > $O/mvs_trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/mvs_pmc -- python $R/tools/mvs_bench.py > $O/mvs_pmc.log 2>&1
cd $R && python tools/pmc_kernels.py $O/r3_pmc_mvs.json '{"k_mvs_conf": 1036800, "k_mvs_propagate_lane": 518400, "k_mvs_propagate<": 518400}' $O/mvs_trace $O/mvs_pmc k_mvs_conf k_mvs_propagate k_mvs_refine k_mvs_project > /dev/null
cp $O/r3_pmc_mvs.json $R/profiles/r3_pmc_mvs.json
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 --repeat 2 > $O/r3_room_like_lines454.txt 2>&1
python tools/room_like_joint.py --frames 454 --points 150000 > $O/r3_room_like_joint454.txt 2>&1
python tools/floor_like_odometry.py --scans 1593 --ranks 2,8 --iters 2 > $O/r3_floor_like_1593.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
du -sh $O
head -c 600 $O/r3_bench_default.json
