#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > /dev/null 2>&1
rm -f gpurun_out/r2/trace454.txt
PVLM_TRACE=$PWD/gpurun_out/r2/trace454.txt python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > gpurun_out/r2/room_like_lines454_batch.txt 2>&1
cat gpurun_out/r2/room_like_lines454_batch.txt
cat gpurun_out/r2/trace454.txt
