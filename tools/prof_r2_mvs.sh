#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_equirect_gpu.py tests/test_golden_gpu.py tests/test_torch_interop_gpu.py tests/test_edge_cases_gpu.py tests/test_mvs_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed"
python bench.py --scans 64 --no-projection --no-cpu-baseline --no-mvs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['panorama'])[:520])"
