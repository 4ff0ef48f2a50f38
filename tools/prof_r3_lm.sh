# Round 3: host mirror, one pose table + one packed buffer per linearisation: host GPU tests, Room / Floor-scale stage times and per-call traces
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_host_gpu.py tests/test_runtime_gpu.py tests/test_room_scale_gpu.py tests/test_floor_scale_gpu.py tests/test_reproj_gpu.py tests/test_abi.py -x -q -m gpu > $O/r3_host_gpu_tests.txt 2>&1
tail -4 $O/r3_host_gpu_tests.txt
timeout 600 python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 --repeat 2 > $O/r3_room_like_lines454.txt 2>&1
grep -E "reproducible|iter|call|linearisation|solve" $O/r3_room_like_lines454.txt
PVLM_HOST_EVAL_TRACE=1 python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 2>&1 | grep -E "^\[eval" > $O/r3_room_like_lines454_eval_trace.txt; cat $O/r3_room_like_lines454_eval_trace.txt
PVLM_HOST_EVAL_TRACE=1 python tools/floor_like_odometry.py --scans 1593 --ranks "" --iters 2 2>&1 | grep -E "^\[eval [3-6]\]|EstimatePose call|linearisation|solve"
timeout 900 python tools/room_like_joint.py --frames 454 --points 150000 --iters 2 > $O/r3_room_like_joint454.txt 2>&1
grep -E "wall|iter|solve" $O/r3_room_like_joint454.txt
