#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_mvs_gpu.py tests/test_mvs_5p7k_gpu.py -x -q -m gpu 2>&1 | tail -5
python tools/mvs_bench.py > gpurun_out/r2/mvs_bench_cols.json 2> gpurun_out/r2/mvs_bench_cols.err; tail -3 gpurun_out/r2/mvs_bench_cols.err; cat gpurun_out/r2/mvs_bench_cols.json
