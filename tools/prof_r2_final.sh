#!/bin/bash
# end-of-round evidence: GPU suite, default bench (+ the same command under rocprofv3), MVS counters, Room-scale timings
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $O/gpu_tests.txt; cat $O/gpu_tests.txt
# counters first: bench.py quotes them
cd /tmp && export TMPDIR=/tmp
# HBM read traffic of the fused kernel (FETCH_SIZE in its own pass; the WRITE_SIZE pass hung on this pool in round 2 and is skipped)
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python $R/bench.py --no-cpu-baseline --no-mvs --no-projection --steps 10 > $O/bench_fetch.log 2>&1
grep '^{' $O/bench_fetch.log | tail -1 > $O/bench_fetch.json
cd $R && python tools/pmc_traffic.py $O/bench_fetch - $O/bench_fetch.json $O/r2_pmc_traffic_default.json > /dev/null && cp $O/r2_pmc_traffic_default.json $R/profiles/r2_pmc_traffic_default.json   # bench.py quotes it
cd /tmp
W="python $R/tools/mvs_bench.py"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/mvs_trace -- $W > $O/mvs_trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/mvs_pmc -- $W > $O/mvs_pmc.log 2>&1
cd $R && python tools/pmc_kernels.py $O/r2_pmc_mvs.json '{"k_mvs_conf": 1036800, "k_mvs_propagate": 518400}' $O/mvs_trace $O/mvs_pmc k_mvs_conf k_mvs_propagate > /dev/null && cp $O/r2_pmc_mvs.json $R/profiles/r2_pmc_mvs.json
grep '^{' $O/mvs_trace.log | tail -1 > $O/mvs_bench.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python $R/bench.py --no-cpu-baseline --no-mvs > $O/bench_under_rocprof.log 2>&1
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
cd $R && python tools/trace_groups.py $(find $O/bench_trace -name "*kernel_trace.csv" | head -1) $O/kernel_groups_default.csv > /dev/null
cp $(find $O/bench_trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_default.csv
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > /dev/null 2>&1
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > $O/room_like_lines454.txt 2>&1
python tools/room_like_joint.py --frames 454 --points 150000 > $O/room_like_joint454.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
du -sh $O
