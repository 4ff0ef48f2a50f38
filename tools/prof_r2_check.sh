#!/bin/bash
ulimit -c 0
for k in 1 2 3 4 5 6; do
  python -m pytest tests/test_host_gpu.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tr '\n' ' '; echo
done
