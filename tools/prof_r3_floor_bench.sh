# Round 3: whole GPU suite, the Floor-scale sharded EstimatePose (2 and 8 ranks on the one GPU), the default bench.
#   gpurun --timeout 2400 -- 'bash tools/prof_r3_floor_bench.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 900 python tools/floor_like_odometry.py --scans 1593 --ranks 2,8 --iters 2 > $O/r3_floor_like_1593.txt 2>&1
tail -40 $O/r3_floor_like_1593.txt
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_floor_scale_gpu.py > $O/r3_gpu_tests.txt 2>&1
tail -5 $O/r3_gpu_tests.txt
timeout 900 python bench.py > $O/r3_bench_default.log 2>$O/r3_bench_default.err
tail -1 $O/r3_bench_default.log > $O/r3_bench_default.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3/r3_bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:d["association"][k] for k in ("kernel_ms","wall_s")})
PY
