#!/bin/bash
python -m pytest tests/test_reproj_gpu.py tests/test_host_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | head -5
python tools/bundle_bench.py 2>&1 | tail -1
for k in 1 2 3; do python tools/room_like_joint.py --frames 454 --points 150000 2>&1 | grep -E "JointOptimize|iter"; done
