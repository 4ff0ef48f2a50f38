# Round 3: host mirror after the one-submission evaluation (flat block table, pvlm_neq_accumulate_async): host GPU tests, Room-scale
# EstimatePose / JointOptimize stage times.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_host_gpu.py tests/test_runtime_gpu.py tests/test_room_scale_gpu.py -x -q -m gpu > $O/r3_host_gpu_tests.txt 2>&1
tail -4 $O/r3_host_gpu_tests.txt
timeout 600 python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 --repeat 2 > $O/r3_room_like_lines454.txt 2>&1
cat $O/r3_room_like_lines454.txt
timeout 900 python tools/room_like_joint.py --frames 454 --points 150000 --iters 2 > $O/r3_room_like_joint454.txt 2>&1
tail -30 $O/r3_room_like_joint454.txt
