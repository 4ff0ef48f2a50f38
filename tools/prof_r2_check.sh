#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_host_gpu.py tests/test_assoc_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | head -5
PVLM_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --scans 64 > gpurun_out/r2/bench_2rank_shared.json 2> gpurun_out/r2/bench_2rank_shared.err
echo "rc=$?"; tail -c 600 gpurun_out/r2/bench_2rank_shared.err; head -c 700 gpurun_out/r2/bench_2rank_shared.json
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > /dev/null 2>&1
python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 2>&1 | head -24
