#!/bin/bash
python -m pytest tests/test_mvs_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -8
python tools/mvs_bench.py 2>gpurun_out/mvsb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K13 checkerboard ms/colour', d['sweep']['kernel_ms_per_colour_pass'], 'seq', d['sweep_sequential'])"; tail -3 gpurun_out/mvsb.err
