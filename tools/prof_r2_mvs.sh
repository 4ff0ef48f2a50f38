#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_mvs_gpu.py tests/test_mvs_5p7k_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed"
python tools/mvs_bench.py > gpurun_out/r2/mvs_bench_default.json 2> gpurun_out/r2/mvs_bench_default.err
python - <<P
import json
d=json.load(open('gpurun_out/r2/mvs_bench_default.json'))
print("K11 ms", d["kernel_ms"], "maxdiff", d["max_abs_diff_vs_oracle"], "K13 ms/colour", d["sweep"]["kernel_ms_per_colour_pass"], "agree", d["sweep"]["agree_with_oracle"], "geo", d["sweep_geometric"])
P
