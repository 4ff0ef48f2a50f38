set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_eval_gpu.py tests/test_runtime_gpu.py tests/test_host_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -4
for W in 0 1; do
  echo "PVLM_WAVE_UNITS=$W" >> $O/r3_wave_units_ab.txt
  PVLM_WAVE_UNITS=$W PVLM_HOST_EVAL_TRACE=1 python tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 2>&1 | grep -E "^\[eval [3-8]\]" | head -4 >> $O/r3_wave_units_ab.txt
  PVLM_WAVE_UNITS=$W PVLM_HOST_EVAL_TRACE=1 python tools/floor_like_odometry.py --scans 1593 --ranks "" --iters 2 2>&1 | grep -E "^\[eval [3-8]\]|EstimatePose call|FindNeighbors" | head -6 >> $O/r3_wave_units_ab.txt
done
cat $O/r3_wave_units_ab.txt
