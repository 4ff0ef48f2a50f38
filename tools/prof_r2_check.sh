#!/bin/bash
python -m pytest tests/test_linalg_gpu.py tests/test_host_gpu.py tests/test_reproj_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | head -5
python tools/chol_bench.py 2>&1 | tail -1
PVLM_CHOL_LOOKAHEAD=1 python tools/chol_bench.py 2>&1 | tail -1
python tools/chol_bench.py --poses 455 2>&1 | tail -1
PVLM_CHOL_LOOKAHEAD=1 python tools/chol_bench.py --poses 455 2>&1 | tail -1
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > /dev/null 2>&1
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 2>&1 | grep -E "iter|call|Cholesky"
for k in 1 2; do python tools/room_like_joint.py --frames 454 --points 150000 2>&1 | grep -E "JointOptimize|iter|GPU Cholesky|reprojection blocks$"; done
