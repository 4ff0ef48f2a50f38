#!/bin/bash
# end-of-round evidence (round 6), in two parts so that each gpurun call stays bounded:
#   PART=a  GPU suite, fused-kernel HBM traffic (FETCH_SIZE pass), SQ counters of the association (voxel AND raw targets), default bench (+ the same command under
#           rocprofv3 --kernel-trace --stats), bench.py as 2 and as 8 ranks sharing this GPU (functional check of the N > 1 branch)
#   PART=b  K8 / MVS counters, the level-scheduled pose solve (kernel trace), Room- / Floor-scale runs, the feature batch
#   PVLM_COMMIT=<short hash> PART=a gpurun ... 'bash tools/prof_r6_final.sh'     results under gpurun_out/r6f (copy the r6_* files to profiles/)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD"
if [ "${PART:-a}" = a ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/r6_gpu_tests.txt; cat $O/r6_gpu_tests.txt
  cd /tmp && export TMPDIR=/tmp
  # HBM read traffic of the fused kernel: FETCH_SIZE in its own pass (the WRITE_SIZE pass hung on this pool in round 2 and is skipped)
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python $R/bench.py --no-cpu-baseline --no-mvs --no-projection --steps 10 > $O/bench_fetch.log 2>&1
  grep '^{' $O/bench_fetch.log | tail -1 > $O/bench_fetch.json
  cd $R && python tools/pmc_traffic.py $O/bench_fetch - $O/bench_fetch.json $O/r6_pmc_traffic_default.json > /dev/null && cp $O/r6_pmc_traffic_default.json $R/profiles/   # bench.py quotes it
  # association counters: voxel targets (256 scans) and the literal raw targets (32 scans)
  for TG in voxel raw; do
    SC=256; [ $TG = raw ] && SC=32
    W2="python $R/tools/assoc_workload.py --scans $SC --targets $TG"
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/assoc_${TG}_trace -- $W2 > $O/assoc_${TG}_trace.log 2>&1
    timeout 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/assoc_${TG}_sq -- $W2 > $O/assoc_${TG}_sq.log 2>&1
    grep '^{' $O/assoc_${TG}_trace.log | tail -1 > $O/assoc_${TG}_workload.json
    cd $R && python tools/pmc_assoc.py $O/assoc_${TG}_workload.json $O/r6_pmc_assoc_${TG}_scans$SC.json $O/assoc_${TG}_trace $O/assoc_${TG}_sq > /dev/null
    cp $O/r6_pmc_assoc_${TG}_scans$SC.json $R/profiles/
    cp $(find $O/assoc_${TG}_trace -name "*kernel_stats.csv" | head -1) $O/r6_assoc_kernel_stats_${TG}_scans$SC.csv
  done
  find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
  # the default bench, then the same command under the kernel trace
  cd $R
  timeout 900 python bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err; tail -c 300 $O/r6_bench_default.err
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python $R/bench.py --no-cpu-baseline --no-mvs > $O/bench_under_rocprof.log 2>&1
  grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/r6_bench_under_rocprof.json
  cd $R && python tools/trace_groups.py $(find $O/bench_trace -name "*kernel_trace.csv" | head -1) $O/r6_kernel_groups_default.csv > /dev/null
  cp $(find $O/bench_trace -name "*kernel_stats.csv" | head -1) $O/r6_kernel_stats_default.csv
  # the N > 1 branch of bench.py as ranks sharing this GPU (gloo; a functional check, not a measurement)
  for N in 2 8; do
    PVLM_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus $N --scans 256 --steps 5 --warmup 2 --no-cpu-baseline --no-mvs --no-projection 2> $O/bench_shared_$N.err | grep '^{' | tail -1 > $O/r6_bench_shared_gpu_${N}ranks.json
    tail -c 200 $O/bench_shared_$N.err
  done
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
else
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/k8_trace -- python $R/tools/k8_workload.py > $O/k8_trace.log 2>&1
  timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $O/k8_pmc -- python $R/tools/k8_workload.py > $O/k8_pmc.log 2>&1
  cd $R && python tools/pmc_kernels.py $O/k8_kernels.json '{"k_cam_lidar_votes_points": 420040800}' $O/k8_trace $O/k8_pmc k_cam_lidar_votes_points > /dev/null
  python - $O/k8_kernels.json $O/r6_pmc_k8.json <<'PY'
import json, sys
k = [v for n, v in json.load(open(sys.argv[1])).items() if "k_cam_lidar_votes_points" in n][0]
out = {"kernel": "k_cam_lidar_votes_points", "command": "rocprofv3 --pmc <SQ set> -- python tools/k8_workload.py (1 362 pairs + an 8-pair warm-up launch)",
       "valu_insts_per_wave": k["valu_insts_per_wave"], "valu_wave_insts_per_test": k["valu_insts_per_wave"] / (64.0 * 200.0),   # a lane = a point walking the pair's 200 lines
       "tests_per_wave": 12800,
       "valu_issue_frac": k.get("valu_issue_frac"), "wait_frac": k.get("wait_frac"), "vmem_rd_insts_per_wave": k.get("vmem_rd_insts_per_wave")}
json.dump(out, open(sys.argv[2], "w"), indent=1); print(out)
PY
  cp $O/r6_pmc_k8.json $R/profiles/
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/mvs_trace -- python $R/tools/mvs_block_bench.py --small > $O/mvs_trace.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/mvs_pmc -- python $R/tools/mvs_block_bench.py --small > $O/mvs_pmc.log 2>&1
  cd $R && python tools/pmc_kernels.py $O/r6_pmc_mvs.json '{"k_mvs_propagate_flow": 1036800, "k_mvs_propagate_lane": 518400, "k_mvs_conf": 518400}' $O/mvs_trace $O/mvs_pmc k_mvs_propagate k_mvs_conf > /dev/null
  cp $O/r6_pmc_mvs.json $R/profiles/
  # the level-scheduled pose solve on the Floor-shaped system: kernel statistics of one run, wall per solve beside the round-5 plan
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/spd_trace -- python $R/tools/spd_levels_bench.py --reps 5 > $O/spd_under_rocprof.log 2>&1
  cp $(find $O/spd_trace -name "*kernel_stats.csv" | head -1) $O/r6_spd_levels_kernel_stats.csv
  cd $R
  # the one-launch factorisation (default), the level launches with the dense tail, the level launches alone, the column-by-column plan of round 5
  { python tools/spd_levels_bench.py --reps 9 | tail -1; PVLM_SPD_FLOW=0 python tools/spd_levels_bench.py --reps 9 | tail -1; PVLM_SPD_FLOW=0 PVLM_SPD_TAIL=0 python tools/spd_levels_bench.py --reps 9 | tail -1;
    PVLM_SPD_LEVELS=0 python tools/spd_levels_bench.py --reps 7 | tail -1; } > $O/r6_spd_levels.txt 2>&1
  PVLM_SPD_TAIL_CLOCK=1 python tools/spd_levels_bench.py --reps 1 2> $O/r6_spd_flow_clocks.txt > /dev/null
  cd /tmp
  PVLM_SPD_FLOW=0 PVLM_SPD_TAIL=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/spd_trace_levels -- python $R/tools/spd_levels_bench.py --reps 5 > $O/spd_levels_under_rocprof.log 2>&1
  cp $(find $O/spd_trace_levels -name "*kernel_stats.csv" | head -1) $O/r6_spd_level_launches_kernel_stats.csv
  cd $R
  timeout 600 python tools/mvs_block_bench.py 2> /dev/null | tail -1 > $O/r6_mvs_block.json
  timeout 300 python tools/k8_workload.py 2> /dev/null | tail -1 > $O/r6_k8_block.json
  # scale runs
  python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 --repeat 2 > $O/r6_room_like_lines454.txt 2>&1
  python tools/room_like_joint.py --frames 454 --points 150000 > $O/r6_room_like_joint454.txt 2>&1
  python tools/floor_like_odometry.py --scans 1593 --ranks 2,8 --iters 2 --repeat 5 > $O/r6_floor_like_1593.txt 2>&1
  PVLM_HOST_REUPLOAD=1 python tools/floor_like_odometry.py --scans 1593 --ranks 1 --iters 2 --repeat 3 > $O/r6_floor_like_1593_reupload.txt 2>&1
  PVLM_SPD_LEVELS=0 python tools/floor_like_odometry.py --scans 1593 --ranks 1 --iters 2 --repeat 3 > $O/r6_floor_like_1593_column_by_column.txt 2>&1
  PVLM_SPD_FLOW=0 PVLM_SPD_TAIL=0 python tools/floor_like_odometry.py --scans 1593 --ranks 1 --iters 2 --repeat 5 > $O/r6_floor_like_1593_level_launches.txt 2>&1
  PVLM_NO_PLAN_PREFETCH=1 python tools/floor_like_odometry.py --scans 1593 --ranks 1 --iters 2 --repeat 5 > $O/r6_floor_like_1593_no_plan_prefetch.txt 2>&1
  python tools/feature_batch_bench.py 454 32 --ab 2>&1 | cut -c1-330 > $O/r6_feature_batch_454.txt
  # K27 (line growth of the feature batch): kernel statistics of one bench run
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k27_trace -- python $R/tools/feature_batch_bench.py 454 32 > $O/k27_under_rocprof.log 2>&1
  cp $(find $O/k27_trace -name "*kernel_stats.csv" | head -1) $O/r6_feature_batch_kernel_stats.csv
  cd $R
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +4M -delete
fi
du -sh $O
