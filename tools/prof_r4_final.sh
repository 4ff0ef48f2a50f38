#!/bin/bash
# end-of-round evidence (round 4): GPU suite, fused-kernel HBM traffic (FETCH_SIZE pass), association counters (voxel AND raw targets), the
# range-image kernels of N3 (trace + SQ counters), default bench (+ the same command under rocprofv3 --kernel-trace --stats), MVS launch
# forms, Room- / Floor-scale runs, the Floor-shaped pose solve.   PVLM_COMMIT=<short hash> gpurun ... 'bash tools/prof_r4_final.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/r4_gpu_tests.txt; cat $O/r4_gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD"
# HBM read traffic of the fused kernel: FETCH_SIZE in its own pass (the WRITE_SIZE pass hung on this pool in round 2 and is skipped)
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python $R/bench.py --no-cpu-baseline --no-mvs --no-projection --steps 10 > $O/bench_fetch.log 2>&1
grep '^{' $O/bench_fetch.log | tail -1 > $O/bench_fetch.json
cd $R && python tools/pmc_traffic.py $O/bench_fetch - $O/bench_fetch.json $O/r4_pmc_traffic_default.json > /dev/null && cp $O/r4_pmc_traffic_default.json $R/profiles/r4_pmc_traffic_default.json   # bench.py quotes it
# association counters: voxel targets (256 scans) and the literal raw targets (32 scans)
for TG in voxel raw; do
  SC=256; [ $TG = raw ] && SC=32
  W2="python $R/tools/assoc_workload.py --scans $SC --targets $TG"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/assoc_${TG}_trace -- $W2 > $O/assoc_${TG}_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/assoc_${TG}_sq -- $W2 > $O/assoc_${TG}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/assoc_${TG}_fetch -- $W2 > $O/assoc_${TG}_fetch.log 2>&1
  grep '^{' $O/assoc_${TG}_trace.log | tail -1 > $O/assoc_${TG}_workload.json
  cd $R && python tools/pmc_assoc.py $O/assoc_${TG}_workload.json $O/r4_pmc_assoc_${TG}_scans$SC.json $O/assoc_${TG}_trace $O/assoc_${TG}_sq $O/assoc_${TG}_fetch > /dev/null
  cp $(find $O/assoc_${TG}_trace -name "*kernel_stats.csv" | head -1) $O/r4_assoc_kernel_stats_${TG}_scans$SC.csv
done
# N3: the range-image kernels of a Room-sized batch
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ring_trace -- python $R/tools/ring_bench.py --cpu 0 --reps 3 > $O/ring_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/ring_pmc -- python $R/tools/ring_bench.py --cpu 0 --reps 3 > $O/ring_pmc.log 2>&1
cp $(find $O/ring_trace -name "*kernel_stats.csv" | head -1) $O/r4_ring_kernel_stats_454x1800.csv
cd $R && python tools/pmc_kernels.py $O/r4_pmc_ring.json '{"k_ring_classify": 12943748, "k_ring_columns": 12943748, "k_seg_": 13075200, "k_curvature": 12943748}' $O/ring_trace $O/ring_pmc k_ring k_seg k_curvature > /dev/null
python tools/ring_bench.py --reps 4 --json $O/r4_ring_bench_454x1800.json > /dev/null 2>&1
python tools/ring_bench.py --scans 256 --cols 4096 --cpu 2 --json $O/r4_ring_bench_256x4096.json > /dev/null 2>&1
python tools/feature_batch_bench.py 454 > $O/r4_feature_batch_454.txt 2>&1
# the default bench, then the same command under the kernel trace
timeout 900 python bench.py > $O/r4_bench_default.json 2> $O/r4_bench_default.err; tail -c 300 $O/r4_bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python $R/bench.py --no-cpu-baseline --no-mvs > $O/bench_under_rocprof.log 2>&1
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/r4_bench_under_rocprof.json
cd $R && python tools/trace_groups.py $(find $O/bench_trace -name "*kernel_trace.csv" | head -1) $O/r4_kernel_groups_default.csv > /dev/null
cp $(find $O/bench_trace -name "*kernel_stats.csv" | head -1) $O/r4_kernel_stats_default.csv
# MVS: the bench tool, the launch forms of the sequential sweep, SQ counters of the image-space kernels incl. the data-flow sweep
timeout 600 python tools/mvs_bench.py 2> $O/mvs_bench.err | tail -1 > $O/r4_mvs_bench.json
for M in 2 1 0; do PVLM_MVS_FLOW=$M timeout 200 python tools/mvs_seq_bench.py; done > $O/r4_mvs_seq_forms_1440.jsonl 2>/dev/null
for M in 1 0; do PVLM_MVS_FLOW=$M timeout 400 python tools/mvs_seq_bench.py --rows 2880 --cols 5760 --iters 1; done > $O/r4_mvs_seq_forms_5760.jsonl 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/mvs_trace -- python $R/tools/mvs_seq_bench.py > $O/mvs_trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/mvs_pmc -- python $R/tools/mvs_seq_bench.py > $O/mvs_pmc.log 2>&1
cd $R && python tools/pmc_kernels.py $O/r4_pmc_mvs.json '{"k_mvs_propagate_flow": 1036800, "k_mvs_conf": 1036800}' $O/mvs_trace $O/mvs_pmc k_mvs_propagate k_mvs_conf > /dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
# scale runs
python tools/spd_floor_bench.py > $O/r4_spd_floor.txt 2>&1; python tools/spd_floor_bench.py --scans 454 >> $O/r4_spd_floor.txt 2>&1
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 --repeat 2 > $O/r4_room_like_lines454.txt 2>&1
python tools/room_like_joint.py --frames 454 --points 150000 > $O/r4_room_like_joint454.txt 2>&1
python tools/floor_like_odometry.py --scans 1593 --ranks 2,8 --iters 2 --repeat 5 > $O/r4_floor_like_1593.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
du -sh $O
head -c 400 $O/r4_bench_default.json
